#!/bin/bash
# GPU box: the bench line of the non-default configurations quoted in DESIGN.md
# (host frames = PCIe-inclusive, fp16, psp-fast fp16, ps2-quality 16-bit and 8-bit, flow-resnet), condensed.
R=$GRAFT_REPO_ROOT
cd $R
for v in "--location host" "--dtype fp16" "--preset psp-quality-lrelu" "--preset psp-quality-lrelu --dtype fp8" "--preset psp-fast --dtype fp16" "--preset ps2-quality" "--preset ps2-quality --dtype fp8" "--preset psp-quality --dtype fp8" "--preset psp-quality-flowres"; do
  timeout 200 python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 $v 2>/dev/null | tail -1 | V="$v" python3 -c '
import json, os, sys
d = json.loads(sys.stdin.read())
l = d["config"]["latency_ms"]
print(json.dumps({"args": os.environ["V"], "metric": d["metric"], "frames_per_s": round(d["value"], 1),
                  "ms_per_frame": round(d["ms_per_step"], 4), "dtype": d["dtype"],
                  "latency_ms": {k: round(x, 4) for k, x in l.items()},
                  "roofline_frac": round(d["roofline"]["frac"], 4), "roofline_kernel": d["roofline"]["kernel"][:40]}))'
done
