# per-kernel times of the captured fp32 step: rocprofv3 --kernel-trace --stats of bench.py --dtype f32
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_step_$tag -o s -- python3 $GRAFT_REPO_ROOT/bench.py --dtype f32 --cpu-steps 0 --no-roofline --steps 50 --repeats 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_step_$tag.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_step_$tag.log
