#!/bin/bash
# A/B/C... of environment settings on one box: tools/exp/ab_multi.sh "VAR=a" "VAR=b" "X=1 Y=2" ...   (2 rounds x 200 captured steps each)
cd "$(dirname "$0")/../.."
for round in 1 2; do
  for setting in "$@"; do
    env $setting python bench.py --cpu-steps 0 --no-roofline --steps 200 --warmup 10 $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$setting', round(r['ms_per_step'], 4), 'ms/step')
" || exit 1
  done
done
