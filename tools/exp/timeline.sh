# one steady-state replay of the captured step, kernel by kernel (tools/trace_step.py over a rocprofv3 kernel trace)
# usage (on the GPU box): bash tools/exp/timeline.sh <tag> [env assignments...]   -> gpurun_out/tl_<tag>.txt
tag=$1; shift
for kv in "$@"; do export "$kv"; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$tag -- python3 $R/bench.py --cpu-steps 0 --no-roofline --steps 100 $BENCH_ARGS > $R/gpurun_out/tl_$tag.json 2> $R/gpurun_out/tl_$tag.log || exit 1
python3 $R/tools/trace_step.py $R/gpurun_out/tl_$tag -v > $R/gpurun_out/tl_$tag.txt
rm -rf $R/gpurun_out/tl_$tag
