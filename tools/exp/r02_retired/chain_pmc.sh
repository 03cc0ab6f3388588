cd /tmp && export TMPDIR=/tmp
for d in 4 5; do
  export WFS_CHAIN_DEBUG=$d
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pmc$d -o rb -- python3 $GRAFT_REPO_ROOT/tools/microbench_rulebook.py 256 256 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pmcb$d -o rb -- python3 $GRAFT_REPO_ROOT/tools/microbench_rulebook.py 256 256 3 > /dev/null 2>&1
done
