#!/bin/bash
# A/B timing of two builds of libwfsparse.so on the same box: alternates the captured bf16 training step between the
# in-tree library and tools/exp/base/libwfsparse.so (a build of another commit, see below), 3 rounds of 200 steps each.
#   git stash; make -C waveformml_amd/csrc; cp waveformml_amd/lib/libwfsparse.so tools/exp/base/; git stash pop; make ...
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export WFS_LIB=$PWD/tools/exp/base/libwfsparse.so; else unset WFS_LIB; fi
    python bench.py --cpu-steps 0 --no-roofline --steps 200 --warmup 10 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$which', round(r['ms_per_step'], 4), 'ms/step', 'f32', round(r.get('f32_path', {}).get('ms_per_step', 0), 4))
" || exit 1
  done
done
