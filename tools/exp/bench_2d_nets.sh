#!/bin/bash
# the 2-D nets (parity cases with a timing): GEP (C1 at batch 256), the segment classifier, the hybrid net (C5), fp32 and bf16
#   usage (GPU box, repo root): bash tools/exp/bench_2d_nets.sh
for dt in f32 bf16; do
  python tools/bench_gep.py 256 30 150 0 0.2 $dt 2>/dev/null
  python tools/bench_ioni.py 256 30 $dt 2>/dev/null
  python tools/bench_gep.py 256 20 1024 3 0.2 $dt 2>/dev/null
done
