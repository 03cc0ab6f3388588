# per-kernel times of the captured bf16 step: rocprofv3 --kernel-trace --stats of bench.py (no CPU leg, no roofline pass)
# usage (on the GPU box): bash tools/exp/prof_step.sh <tag> [env assignments...]
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_step_$tag -o s -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-roofline --steps 50 --repeats 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_step_$tag.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_step_$tag.log
