# per-kernel times of the C5 hybrid step (tools/bench_gep.py 256 20 1024 3 0.2 <dtype>) under rocprofv3
# usage (on the GPU box): bash tools/exp/prof_c5.sh <tag> [dtype]
tag=$1; dt=${2:-bf16}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c5_$tag -o s -- python3 $GRAFT_REPO_ROOT/tools/bench_gep.py 256 20 1024 3 0.2 $dt > $GRAFT_REPO_ROOT/gpurun_out/prof_c5_$tag.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_c5_$tag.log
