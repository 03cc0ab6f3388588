# per-kernel times of the GEP / C1 step (tools/bench_gep.py 256 30 150 0 0.2 <dtype>) under rocprofv3
# usage (GPU box): bash tools/exp/prof_gep.sh <tag> [dtype]
tag=$1; dt=${2:-bf16}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_gep_$tag -o s -- python3 $GRAFT_REPO_ROOT/tools/bench_gep.py 256 30 150 0 0.2 $dt > $GRAFT_REPO_ROOT/gpurun_out/prof_gep_$tag.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_gep_$tag.log
