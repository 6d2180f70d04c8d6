#!/bin/bash
# developer tool, GPU box: do the flow net's fused 128-filter blocks (JU_FLOW_WIDE, flow_kernels.hip) show in whole frames?
# Interleaved bench lines, both presets whose frame the flow net is a large part of.
for r in 1 2 3; do
  for preset in "psp-quality bf16" "psp-fast fp16"; do
    set -- $preset
    for wide in 0 2; do
      JU_FLOW_WIDE=$wide python3 bench.py --preset $1 --dtype $2 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d['config'].get('sclk_mhz_during_preroll') or {}
print('$1 $2 JU_FLOW_WIDE=$wide  %8.1f frames/s  %.4f ms/frame  kernel in frame %.1f us  sclk %s' % (d['value'], d['ms_per_step'], d['roofline']['launch_ms'] * 1e3, c.get('median')))"
    done
  done
done
