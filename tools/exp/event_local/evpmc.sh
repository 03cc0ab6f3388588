#!/bin/bash
# LDS / wave counters of k_evconv32 on the bench layers (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  rm -rf /tmp/evpmc
  EV_ONLY=1 rocprofv3 --kernel-trace --pmc $set -d /tmp/evpmc -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/exp/event_local/microbench_evconv.py 3 > /tmp/evpmc.log 2>&1
  f=$(find /tmp/evpmc -name "*counter_collection.csv" | head -1)
  echo "== $set ($f)"
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "k_evconv32" in k:
        acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: (sum(v) / len(v), len(v)) for c, v in d.items()})
PY
done
