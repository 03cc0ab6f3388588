#!/bin/bash
# per-phase cost of the event-local rulebook kernels (make -C waveformml_amd/csrc erknock)
for k in ${ER_KNOCKS:-0 16 32 96 224}; do
  if [ $k = 0 ]; then lib=waveformml_amd/lib/libwfsparse.so; else lib=tools/exp/erk$k/libwfsparse.so; fi
  echo "== ER_KNOCK=$k"
  WFS_LIB=$PWD/$lib ER_ONLY=1 python tools/microbench_evrulebook.py 30 2>&1 | grep "event-local"
done
