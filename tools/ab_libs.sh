#!/bin/bash
# GPU box: interleaved A/B of two builds of libJoshUpscale.so (stage timings).
# usage: bash tools/ab_libs.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    echo "== $L"
    JU_LIBRARY=$L timeout 100 python3 tools/flow_layers.py | tail -6 | grep -E "flow|tower|ALL"
  done
done
