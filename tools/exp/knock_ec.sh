#!/bin/bash
# timing knock-outs of the event-local strided build (csrc/evconv.hip, EC_KNOCK; `make -C waveformml_amd/csrc knock_ec` first)
# usage (GPU box, repo root): bash tools/exp/knock_ec.sh [events]
EV=${1:-256}
echo "== EC_KNOCK=0"
python tools/microbench_strided_build.py 30 $EV 2>&1 | grep -E "event-local|^events"
for k in 1 2 4 5 12 13; do
  echo "== EC_KNOCK=$k"
  WFS_LIB=tools/exp/eck$k/libwfsparse.so python tools/microbench_strided_build.py 30 $EV 2>&1 | grep "event-local"
done
exit 0
