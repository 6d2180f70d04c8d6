#!/bin/bash
# GPU box: per-launch times of the flow net's block launches by tile height (JU_FLOW_TILE forces one where it fits;
# unset = the launcher's choice), at both benchmark geometries -> the basis of fbLaunchCost's fixed part.
for preset in ps2-quality psp-quality; do
  for th in "" 20 18 10 6; do
    if [ -z "$th" ]; then unset JU_FLOW_TILE; else export JU_FLOW_TILE=$th; fi
    echo "== $preset JU_FLOW_TILE=${th:-default}"
    python3 tools/flow_layers.py $preset | grep -E "flow# ?(0|1|9|10):|back-to-back"
  done
done
unset JU_FLOW_TILE
for p in "ps2-quality fp8" "ps2-quality bf16" "psp-quality bf16"; do
  set -- $p
  for lib in build/base/libJoshUpscale.so joshupscale_amd/lib/libJoshUpscale.so; do
    JU_LIBRARY=$lib python3 bench.py --preset $1 --dtype $2 --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 $lib', round(d['value'],1), 'fps')"
  done
done
