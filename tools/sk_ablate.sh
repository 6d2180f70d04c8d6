#!/bin/bash
# developer tool: timing ablation of conv_splitk_kernel (JU_FB_SKIP bits: 1 weights, 2 staging, 4 K loops, 8 reduction, 16 stores)
# needs the probe build: `make ablate` (the product library ignores JU_FB_SKIP)
export JU_LIBRARY=${JU_LIBRARY:-$PWD/build/ablate/libJoshUpscale_test.so}
for s in 0 1 2 4 8 16 12 14 15 31; do
  echo "== JU_FB_SKIP=$s"; JU_FB_SKIP=$s python tools/flow_layers.py 2>&1 | grep -E "flow# 3|flow# 4|flow# 5|flow# 7|flow# 8" | tr '\n' ' '; echo
done
