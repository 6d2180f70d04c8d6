#!/bin/bash
# GPU box: 8-bit tower against the 16-bit per-layer and resident towers
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  env "${@:2}" timeout 300 python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 $1 2>/dev/null | tail -1 | python3 -c '
import json, sys
d = json.loads(sys.stdin.read())
r = d["roofline"]
print(json.dumps({"fps": round(d["value"], 1), "ms": round(d["ms_per_step"], 4), "dtype": d["dtype"],
  "kernel": r["kernel"][:32], "launch_ms": round(r["launch_ms"], 4), "achieved": round(r["achieved"], 1), "unit": r["unit"],
  "frac": round(r["frac"], 4), "first_conv": r.get("first_conv", {}).get("launch_ms")}))'
}
run "--preset ps2-quality --dtype fp8" X=1
run "--preset ps2-quality --dtype fp16" X=1
run "--preset psp-quality --dtype fp8" X=1
run "--preset psp-quality --dtype fp16" JU_TOWER=layers
run "--preset psp-quality --dtype fp16" X=1
