#!/bin/bash
# The one-rank RCCL test (eager + in-graph exchange) as separate processes, one after the other, until one fails: the
# full output of a failing run lands in gpurun_out/nccl1_fail.log.  usage (GPU box): bash tools/exp/repeat_nccl_world_one.sh [runs]
n=${1:-10}
for i in $(seq 1 $n); do
  python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k nccl_backend_world_one > gpurun_out/nccl1_run.log 2>&1
  rc=$?
  echo "run $i rc $rc $(grep -c 'first attempt died' gpurun_out/nccl1_run.log) early deaths"
  if [ $rc -ne 0 ] || grep -q 'first attempt died' gpurun_out/nccl1_run.log; then cp gpurun_out/nccl1_run.log gpurun_out/nccl1_fail.log; break; fi
done
