#!/bin/bash
# developer tool, GPU box: the host-frame path (JU_LOC_CPU: PCIe-inclusive, never `value`) with and without page-locking
# recycled caller buffers (JU_PIN_HOST), interleaved, beside the device-frame line of the same box.
# (Historical: the switch was removed at the end of round 5 -- csrc/engine.h says why; the "host 1" lines now equal "host 0".)
for r in 1 2 3; do
  for mode in "device 1" "host 0" "host 1"; do
    set -- $mode
    JU_PIN_HOST=$2 python3 bench.py --location $1 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-6s JU_PIN_HOST=$2  %8.1f frames/s  %.3f ms/frame  p50 %.3f p99 %.3f ms' % ('$1', d['value'], d['ms_per_step'], d['config']['latency_ms']['p50'], d['config']['latency_ms']['p99']))"
  done
done
