#!/bin/bash
# developer tool, GPU box: the flow net's 128-filter blocks as launches of their own (JU_FLOW_WIDE=0), the encoder block fused
# (1), both fused (2 = default), interleaved; per-launch times of tools/flow_layers.py
for r in 1 2; do
  for wide in 0 1 2; do echo "== JU_FLOW_WIDE=$wide"; JU_FLOW_WIDE=$wide python3 tools/flow_layers.py | head -14; done
done
