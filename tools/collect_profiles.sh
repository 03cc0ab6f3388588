#!/bin/bash
# Round artifacts on the GPU box: GPU tests, the default bench line, rocprofv3 kernel stats of the captured bf16 step,
# and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) -> gpurun_out/final/ (copied into profiles/ afterwards).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -20 $O/pytest_gpu.log; exit 1; }
tail -n 2 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.log || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --cpu-steps 0 --no-roofline --steps 100 > $O/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --cpu-steps 0 --no-roofline --steps 20 > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --cpu-steps 0 --no-roofline --steps 20 > $O/pmc_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f32 -- python3 $R/bench.py --dtype f32 --cpu-steps 0 --no-roofline --steps 100 > $O/stats_f32.log 2>&1 || exit 1
cd $R
cp $(ls $O/stats/*/*kernel_stats.csv | tail -n 1) $O/kernel_stats.csv
python tools/trace_step.py $O/stats -v > $O/step_timeline_bf16.txt
python tools/trace_step.py $O/stats_f32 -v > $O/step_timeline_f32.txt
cp $(ls $O/stats_f32/*/*kernel_stats.csv | tail -n 1) $O/kernel_stats_f32.csv
python tools/pmc_summary.py $(ls $O/pmc_fetch/*/*counter_collection.csv | tail -n 1) $(ls $O/pmc_write/*/*counter_collection.csv | tail -n 1) $O/pmc_hbm_traffic_bf16.json
rm -rf $O/stats $O/stats_f32 $O/pmc_fetch $O/pmc_write
cut -c1-400 $O/bench_default.json
