#!/bin/bash
# per-phase knock-outs of the register-resident BatchNorm kernels (`make -C waveformml_amd/csrc knock_bn` builds bn.hip
# with -DWFS_BN_KNOCK=bits: 1 no fold of the partials, 2 no row loads (every thread reads row 0), 4 no row stores,
# 8 no LDS reduction / partial store in the reduce kernel), timed by tools/microbench_conv.py inside a replayed graph
cd "$(dirname "$0")/../.."
for kn in base 1 2 4 8 3 7 15; do
  if [ $kn = base ]; then unset WFS_LIB; else export WFS_LIB=$PWD/tools/exp/bnk$kn/libwfsparse.so; fi
  echo "knock $kn: $(python tools/microbench_conv.py 50 bf16 2>/dev/null | grep -E '^bn\+relu' | tr '\n' ' ')"
done
