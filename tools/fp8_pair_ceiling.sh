#!/bin/bash
# developer tool, GPU box: what "two residual blocks per launch" could buy the 8-bit per-block tower (BASELINE config 5,
# 640x448) at best -- a bytes-only ablation on the probe build (`make ablate`): the first block of every pair stores
# neither the fp16 stream nor the e4m3 copy (JU_FB_SKIP=16), the second stages no e4m3 tile and fetches no skip records
# (JU_FB_SKIP_ALT=9).  Wrong frames by design; frames/s of the whole frame loop, interleaved with the unablated build.
export JU_LIBRARY=$PWD/build/ablate/libJoshUpscale_test.so
for r in 1 2 3; do
  for mode in "0 -1" "16 9" "16 -1" "9 -1" "25 25"; do
    set -- $mode
    JU_FB_SKIP=$1 JU_FB_SKIP_ALT=$2 python3 bench.py --preset ps2-quality --dtype fp8 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('skip %3s / alt %3s  %7.1f frames/s   block kernel %6.2f us in frame' % ('$1', '$2', d['value'], r['launch_ms'] * 1e3))"
  done
done
