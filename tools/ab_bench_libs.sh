#!/bin/bash
# developer tool, GPU box: interleaved A/B of library builds through bench.py itself -- frames/s of whole frames,
# the dominant kernel inside them (HIP events) and the clock the box held, one line per (round, library).
# usage: tools/ab_bench_libs.sh "<bench args>" <lib.so>...     (developer builds carry the hooks: tools/dev_tower_lib.sh)
ARGS="$1"; shift
for r in 1 2 3; do
  for L in "$@"; do
    JU_LIBRARY=$PWD/$L python3 bench.py $ARGS --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']; c = d['config'].get('sclk_mhz_during_preroll') or {}
print('%-22s %8.1f frames/s   kernel in frame %7.1f us   back to back %7.1f us   sclk %s MHz' % (os.path.basename('$L'), d['value'], r['launch_ms'] * 1e3, (r.get('launch_ms_back_to_back') or 0) * 1e3, c.get('median')))"
  done
done
