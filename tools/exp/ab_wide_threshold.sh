#!/bin/bash
# GEP (300 -> 252 -> 158 -> 64) and the segment classifier (130 ... 154 channels) in bf16 with the wide path from 256 /
# 128 / 64 channels on and off, beside fp32.   usage (GPU box, repo root): bash tools/exp/ab_wide_threshold.sh
for w in 256 128 64 0; do
  echo "== bf16, WFS_WIDE_MIN_CHANNELS=$w"
  WFS_WIDE_MIN_CHANNELS=$w python tools/bench_gep.py 256 30 150 0 0.2 bf16 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("GEP  eager %.3f graph %.3f ms  rel loss diff %.1e" % (d["gpu_eager_ms_per_step"], d["gpu_graph_ms_per_step"], d["rel_loss_diff_first_step"]))'
  WFS_WIDE_MIN_CHANNELS=$w python tools/bench_ioni.py 256 30 bf16 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("Ioni eager %.3f graph %.3f ms  rel loss diff %.1e" % (d["gpu_eager_ms_per_step"], d["gpu_graph_ms_per_step"], d["rel_loss_diff_first_step"]))'
done
echo "== f32"
python tools/bench_gep.py 256 30 150 0 0.2 f32 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("GEP  eager %.3f graph %.3f ms" % (d["gpu_eager_ms_per_step"], d["gpu_graph_ms_per_step"]))'
python tools/bench_ioni.py 256 30 f32 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("Ioni eager %.3f graph %.3f ms" % (d["gpu_eager_ms_per_step"], d["gpu_graph_ms_per_step"]))'
