#!/bin/bash
# GPU box: does a HIP runtime knob change the synchronous frame rate?  (The frame is 13 graph nodes; the host-side cost of
# a replay and the kernels' start latency are ~12-14 us of every 490 us frame.)
R=$GRAFT_REPO_ROOT
cd $R
run() {
  echo -n "$1: "
  env $1 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.1f frames/s, p50 %.4f ms' % (d['value'], d['config']['latency_ms']['p50']))"
}
for r in 1 2; do
run "JU_NOP=1"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "GPU_MAX_HW_QUEUES=1"
run "HSA_ENABLE_INTERRUPT=0"
done
