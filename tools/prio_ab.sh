#!/bin/bash
# GPU box: A/B of the static wave-priority schemes (JU_WAVE_PRIO, kernel_common.h) on the kernels that
# run two waves per SIMD: res_block_fp8_kernel (ps2-quality fp8), flow_block_kernel / conv_splitk_kernel (flow net).
R=$GRAFT_REPO_ROOT
cd $R
for m in 0 1 2 3 4 0 1; do
  echo "== ps2-quality fp8, JU_WAVE_PRIO=$m"
  JU_WAVE_PRIO=$m python3 bench.py --preset ps2-quality --dtype fp8 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   fps %.1f  block kernel %.2f us' % (d['value'], d['roofline']['launch_ms']*1e3))"
done
for m in 0 1 2 3 0 1; do
  echo "== psp-quality bf16, JU_WAVE_PRIO=$m"
  JU_WAVE_PRIO=$m python3 tools/flow_layers.py 2>/dev/null | grep -E "flow#|stage flow|stage ALL"
done
