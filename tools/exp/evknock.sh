#!/bin/bash
# per-phase cost of k_evconv32 inside a replayed graph: one library per knock-out (make -C waveformml_amd/csrc evknock)
for k in ${EV_KNOCKS:-0 1 32 64 96}; do
  if [ $k = 0 ]; then lib=waveformml_amd/lib/libwfsparse.so; else lib=tools/exp/evk$k/libwfsparse.so; fi
  echo "== EV_KNOCK=$k"
  WFS_LIB=$PWD/$lib EV_ONLY=1 python tools/microbench_evconv.py 30 2>&1 | grep "event-local"
done
