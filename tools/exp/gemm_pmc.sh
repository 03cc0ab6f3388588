#!/bin/bash
# counters of k_gemm16 on the hybrid net's 1697 -> 1021 layer (separate --pmc passes, kernel-trace only)
# usage (GPU box): bash tools/exp/gemm_pmc.sh [fwd|dx|dw] [mode]
which=${1:-fwd}; mode=${2:-1}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  rm -rf /tmp/gpmc
  rocprofv3 --kernel-trace --pmc $set -d /tmp/gpmc -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/exp/gemm_pmc.py $which $mode > /tmp/gpmc.log 2>&1
  f=$(find /tmp/gpmc -name "*counter_collection.csv" | head -1)
  echo "== $set"
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no counters:", e); rows = []
for row in rows:
    k = row["Kernel_Name"]
    if "k_gemm16" in k:
        acc["k_gemm16"][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: (round(sum(v) / len(v)), len(v)) for c, v in d.items()})
PY
done
