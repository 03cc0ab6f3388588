# per-kernel times of tools/microbench_generic.py <events> <dtype> wide under rocprofv3
# usage (on the GPU box): bash tools/exp/prof_mb_wide.sh <tag> [dtype]
tag=$1; dt=${2:-bf16}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mbw_$tag -o s -- python3 $GRAFT_REPO_ROOT/tools/microbench_generic.py 256 $dt wide > $GRAFT_REPO_ROOT/gpurun_out/prof_mbw_$tag.txt 2> $GRAFT_REPO_ROOT/gpurun_out/prof_mbw_$tag.log
