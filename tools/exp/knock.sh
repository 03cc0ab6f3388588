#!/bin/bash
# per-phase knock-outs of k_gconv32_bf16 (`make -C waveformml_amd/csrc knock` builds conv_mfma.hip with -DWFS_KNOCK=bits:
# 1 no filter staging, 2 no table reads, 4 no gathers / MFMA, 8 no stores), timed by tools/microbench_conv.py inside a
# replayed graph
# EVENTS / SAMPLES (default 256 / 256): the batch the kernels are timed on
cd "$(dirname "$0")/../.."
for kn in ${KNOCK_LIST:-base 1 2 4 8 6 14 15 16}; do
  if [ $kn = base ]; then unset WFS_LIB; else export WFS_LIB=$PWD/tools/exp/k$kn/libwfsparse.so; fi
  echo "knock $kn: $(python tools/microbench_conv.py ${ITERS:-50} bf16 ${EVENTS:-256} ${SAMPLES:-256} 2>/dev/null | grep -E '^subm fwd 32->32  |^conv s4 fwd' | tr '\n' ' ')"
done
