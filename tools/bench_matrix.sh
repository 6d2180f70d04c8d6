#!/bin/bash
# GPU box: bench.py over the configurations DESIGN.md section 5 tabulates; one JSON line per configuration into
# gpurun_out/<dir>/bench_<preset>_<dtype>.json and a one-line summary each.  usage: tools/bench_matrix.sh <dir> [extra bench args]
D=gpurun_out/$1; shift
mkdir -p $D
for cfg in "psp-quality bf16" "psp-fast fp16" "ps2-quality fp8" "psp-quality fp8" "ps2-quality bf16" "psp-quality-lrelu bf16" "psp-quality fp16" "psp-quality-flowres bf16"; do
  set -- $cfg
  python3 bench.py --preset $1 --dtype $2 --no-cpu-baseline "${@:3}" > $D/bench_$1_$2.json 2>> $D/bench.err
  python3 - "$D/bench_$1_$2.json" "$1 $2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]; c = d["config"].get("sclk_mhz_during_preroll") or {}
    g = lambda k: ("%7.1f" % d[k]) if d.get(k) else "   n/a "
    print("%-24s value %7.1f  fbf %s  lookahead %s  host %s  host-la %s  | %s %.1f us frac %.3f | sclk %s" % (
        sys.argv[2], d["value"], g("frame_by_frame_value"), g("lookahead_value"), g("host_frames_value"), g("host_frames_lookahead_value"),
        r["kernel"].split(":")[0], r["launch_ms"] * 1e3, r["frac"], c.get("median")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
