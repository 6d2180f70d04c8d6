#!/bin/bash
# GPU box: interleaved A/B of two builds of libJoshUpscale.so (stage timings).
# usage: bash tools/ab_libs.sh <libA.so> <libB.so> [rounds] [grep pattern]
A=$1; B=$2; N=${3:-3}; PAT=${4:-flow|tower|ALL}
for i in $(seq $N); do
  for L in $A $B; do
    echo "== $L"
    JU_LIBRARY=$L timeout 100 python3 tools/flow_layers.py | tail -6 | grep -E "$PAT"
  done
done
