#!/bin/bash
# developer tool: timing ablation of flow_block_kernel (JU_FB_SKIP bits: 1 staging, 2 conv A, 4 conv B, 8 stores)
# needs the probe build: `make ablate` (the product library ignores JU_FB_SKIP)
export JU_LIBRARY=${JU_LIBRARY:-$PWD/build/ablate/libJoshUpscale_test.so}
for s in 0 1 2 4 8 6 7 15; do
  echo "== JU_FB_SKIP=$s"; JU_FB_SKIP=$s python tools/flow_layers.py 2>&1 | grep -E "flow# 0|flow# 1:|flow# 9|flow#10"
done
