#!/bin/bash
# counters of the 32 -> 32 conv kernels on the PSD batch (separate --pmc passes, kernel-trace only)
# usage (GPU box): bash tools/exp/conv_pmc.sh [fwd|dx|dw] [f32|bf16] [kernel name filter]
which=${1:-fwd}; dt=${2:-f32}; filt=${3:-k_g}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  rm -rf /tmp/cpmc
  rocprofv3 --kernel-trace --pmc $set -d /tmp/cpmc -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/exp/conv_pmc.py $which $dt > /tmp/cpmc.log 2>&1
  f=$(find /tmp/cpmc -name "*counter_collection.csv" | head -1)
  echo "== $set"
  python3 - "$f" "$filt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no counters:", e); rows = []
for row in rows:
    k = row["Kernel_Name"]
    if sys.argv[2] in k:
        acc[k[:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: (round(sum(v) / len(v)), len(v)) for c, v in d.items()})
PY
done
