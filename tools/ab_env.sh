#!/bin/bash
# A/B of an environment switch on one box: tools/ab_env.sh VAR=a VAR=b [bench.py args]  (3 rounds x 200 captured steps)
cd "$(dirname "$0")/.."
A=$1; B=$2; shift 2
for round in 1 2 3; do
  for setting in "$A" "$B"; do
    env "$setting" python bench.py --cpu-steps 0 --no-roofline --steps 200 --warmup 10 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$setting', round(r['ms_per_step'], 4), 'ms/step')
" || exit 1
  done
done
