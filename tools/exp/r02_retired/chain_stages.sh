cd /tmp && export TMPDIR=/tmp
for d in 1 2 3 4 5 0; do
  export WFS_CHAIN_DEBUG=$d
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dbg$d -o rb -- python3 $GRAFT_REPO_ROOT/tools/microbench_rulebook.py 256 256 5 > /dev/null 2>&1
done
