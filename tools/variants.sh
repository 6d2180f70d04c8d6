#!/bin/bash
# GPU box: the bench line of the non-default configurations quoted in DESIGN.md
# (host frames = PCIe-inclusive, fp16, psp-fast fp16, ps2-quality, flow-resnet).
R=$GRAFT_REPO_ROOT
cd $R
for v in "--location host" "--dtype fp16" "--preset psp-fast --dtype fp16" "--preset ps2-quality" "--preset psp-quality-flowres"; do
  echo "== $v"
  timeout 200 python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 $v 2>/dev/null | tail -1 | cut -c1-420
done
