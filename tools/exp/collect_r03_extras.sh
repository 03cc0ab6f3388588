#!/bin/bash
# Round-3 artifacts beyond tools/collect_profiles.sh: the wide path's microbenchmarks, the 2-D nets and C4 as parity cases
# with a timing, per-kernel tables of the hybrid net's captured step.  usage (GPU box, repo root): bash tools/exp/collect_r03_extras.sh
O=gpurun_out/final
mkdir -p $O
python tools/microbench_generic.py 256 bf16 wide > $O/microbench_wide_bf16.txt 2>/dev/null
python tools/microbench_generic.py 256 f32 wide > $O/microbench_wide_f32.txt 2>/dev/null
python tools/microbench_generic.py 256 bf16 > $O/microbench_generic_bf16.txt 2>/dev/null
bash tools/exp/bench_2d_nets.sh > $O/bench_2d_nets.txt 2>&1
python bench.py --config config/psd_c4_deep_fp16.json --samples 512 --dtype f16 --cpu-steps 4 > $O/c4_f16.json 2> $O/c4_f16.log
python tools/bench_eval.py > $O/bench_eval.json 2> $O/bench_eval.log
for dt in bf16 f32; do
  bash tools/exp/prof_c5.sh final_$dt $dt
  python tools/kernel_table.py gpurun_out/prof_c5_final_$dt/s_kernel_stats.csv 50 > $O/c5_${dt}_kernel_table.txt
  cp gpurun_out/prof_c5_final_$dt.json $O/c5_${dt}_step.json
done
tail -3 $O/c5_bf16_kernel_table.txt
