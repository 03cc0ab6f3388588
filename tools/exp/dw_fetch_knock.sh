#!/bin/bash
# Is the bf16 dW kernel's excess fetch (counter traffic 2.1x its algorithmic bytes) time?  k_gdw32_bf16 with its gathers
# reading the tile's OWN rows (WFS_KNOCK=64: `make -C waveformml_amd/csrc knock KNOCKS=64`) against the product: time
# inside a replayed graph, FETCH_SIZE / WRITE_SIZE per launch.   usage (GPU box, repo root): bash tools/exp/dw_fetch_knock.sh
R=${GRAFT_REPO_ROOT:-$PWD}
for lib in "" "$R/tools/exp/k64/libwfsparse.so"; do
  if [ -z "$lib" ]; then unset WFS_LIB; echo "== product"; else export WFS_LIB=$lib; echo "== WFS_KNOCK=64 (gathers read the tile's own rows)"; fi
  python $R/tools/microbench_conv.py 30 bf16 2>/dev/null | grep -E "dW 32x32|s4 dW"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/dwk; ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc $c -d /tmp/dwk -o x --output-format csv -- python3 $R/tools/exp/conv_pmc.py dw bf16 > /tmp/dwk.log 2>&1 )
    python3 - "$(find /tmp/dwk -name '*counter_collection.csv' | head -1)" $c <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "k_gdw32_bf16" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
print("  %s per launch: %.1f KB (%d launches)%s" % (sys.argv[2], sum(v) / max(len(v), 1), len(v), "  [x2 for gfx950 fetch]" if sys.argv[2] == "FETCH_SIZE" else ""))
PY
  done
done
