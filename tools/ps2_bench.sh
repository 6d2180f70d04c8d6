#!/bin/bash
# developer tool: the 640x448 geometry (more regions than CUs: no resident tower), bf16, block-fused vs per-conv
for m in layers convs; do
  echo "== JU_TOWER=$m"; JU_TOWER=$m python bench.py --preset ps2-quality --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d['roofline']['launches_per_frame'])"
done
