#!/bin/bash
# developer tool: the 8-bit tower, block-fused vs per-conv, at both geometries, next to bf16
run() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   %.1f fps  %.3f ms/frame  dominant %.1f us x %s  frac %.3f' % (d['value'], d['ms_per_step'], r['launch_ms']*1e3, r['launches_per_frame'], r['frac']))"; }
for preset in psp-quality ps2-quality; do
  echo "== $preset bf16"; run --preset $preset --dtype bf16
  echo "== $preset fp8 (block-fused)"; run --preset $preset --dtype fp8
  echo "== $preset fp8 (per conv)"; JU_TOWER=convs run --preset $preset --dtype fp8
done
