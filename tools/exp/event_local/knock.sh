#!/bin/bash
# per-phase cost of the experimental kernels inside replayed graphs: one library per knock-out
#   usage (GPU box, repo root): tools/exp/event_local/knock.sh conv "0 1 32 64 128 256"   (EV_KNOCK of evconv.hip)
#                               tools/exp/event_local/knock.sh rulebook "0 16 32 96 224"   (ER_KNOCK of evconv_rulebook.hip)
what=$1; ks=$2; d=tools/exp/event_local
for k in $ks; do
  if [ $what = conv ]; then make -s -C $d OUT=libwfs_evexp_k$k.so EV_KNOCK=$k; s=microbench_evconv.py; v=EV_ONLY; else make -s -C $d OUT=libwfs_evexp_k$k.so ER_KNOCK=$k; s=microbench_evrulebook.py; v=ER_ONLY; fi
  echo "== knock $k"
  env $v=1 WFS_EVEXP_LIB=$PWD/$d/libwfs_evexp_k$k.so python $d/$s 30 2>&1 | grep "event-local"
done
