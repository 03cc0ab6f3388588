#!/bin/bash
# Round-4 artifacts beyond tools/collect_profiles.sh: the strided builds alone (256 and 2048 events, 512- and 1024-thread
# workgroups), the build's knock-outs, the products at 2048 events, the other configs as parity cases with a timing, the
# per-kernel tables of the hybrid net's captured step, the from-files soak.
# usage (GPU box, repo root; `make -C waveformml_amd/csrc knock_ec` in the container first): bash tools/exp/collect_r04_extras.sh
O=gpurun_out/final
mkdir -p $O
{
  for ev in 256 2048; do
    echo "## python tools/microbench_strided_build.py 50 $ev   (default: 512-thread workgroups)"
    python tools/microbench_strided_build.py 50 $ev 2>/dev/null
    echo "## WFS_EC_THREADS=1024 python tools/microbench_strided_build.py 50 $ev"
    WFS_EC_THREADS=1024 python tools/microbench_strided_build.py 50 $ev 2>/dev/null | grep -E "event-local"
  done
} > $O/microbench_strided_build.txt
bash tools/exp/knock_ec.sh 256 > $O/event_local_conv_build_knockouts.txt 2>&1
python tools/microbench_conv.py 30 bf16 2048 > $O/microbench_conv_bf16_batch2048.txt 2>/dev/null
bash tools/exp/bench_2d_nets.sh > $O/bench_2d_nets.txt 2>&1
python bench.py --config config/psd_c4_deep_fp16.json --samples 512 --dtype f16 --cpu-steps 4 > $O/c4_f16.json 2> $O/c4_f16.log
python tools/bench_eval.py > $O/bench_eval.json 2> $O/bench_eval.log
for dt in bf16 f32; do
  bash tools/exp/prof_c5.sh final_$dt $dt
  python tools/kernel_table.py gpurun_out/prof_c5_final_$dt/s_kernel_stats.csv 50 > $O/c5_${dt}_kernel_table.txt
done
WFH5_THREADS=1 timeout -k 10 420 python tools/soak_from_files.py 600 85 8 16 > $O/soak_from_files.json 2> $O/soak_from_files.log
tail -3 $O/microbench_strided_build.txt
cut -c1-300 $O/soak_from_files.json
