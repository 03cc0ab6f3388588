#!/bin/bash
# per-phase cost of k_gemm16 (csrc/wide.hip WFS_GEMM_KNOCK; build: make -C waveformml_amd/csrc knock_gemm):
# the three products of the hybrid net's 1697 -> 1021 layer, one library per knock-out, selected with WFS_LIB.
#   usage (GPU box, repo root): bash tools/exp/knock_gemm.sh [mode]      mode: wfs_wide_enable bits (7 = 2 register stages)
mode=${1:-1}
for k in 0 1 2 4 5 7; do
  lib=$PWD/tools/exp/gk$k/libwfsparse.so
  [ $k = 0 ] && lib=$PWD/waveformml_amd/lib/libwfsparse.so
  echo "== knock $k"
  WFS_LIB=$lib WFS_WIDE_MODE=$mode python tools/exp/gemm_time.py
done
