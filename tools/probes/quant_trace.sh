#!/bin/bash
# per-dispatch durations by position inside the frame: conv_1 fused with the e4m3 copy against the separate quantize launch
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in split fused split fused; do
  if [ $m = split ]; then export JU_QUANT=split; else unset JU_QUANT; fi
  rm -rf /tmp/qt_$m
  ( cd $R && rocprofv3 --kernel-trace --output-format csv -d /tmp/qt_$m -- python3 bench.py --preset ps2-quality --dtype fp8 --steps 200 --warmup 30 --no-cpu-baseline --roofline-iters 1 > /dev/null 2>&1 )
  f=$(find /tmp/qt_$m -name "*kernel_trace.csv" | head -1)
  echo "== $m"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# frames of the timed loop: the last 200 occurrences of warp_pack_kernel start a generator phase
idx = [i for i, r in enumerate(rows) if "warp_pack" in r["Kernel_Name"]]
idx = idx[-150:-1]
pos = collections.defaultdict(list)
gap = collections.defaultdict(list)
for a in idx:
    k = 0
    for i in range(a, min(a + 30, len(rows))):
        n = rows[i]["Kernel_Name"]
        short = "warp" if "warp_pack" in n else "conv1" if "conv_tower_kernel" in n else "quant" if "quantize" in n else \
                "block" if "res_block_fp8" in n else "tail" if "tail_fused" in n else None
        if short is None:
            break
        d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
        g = (int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3
        key = f"{k:02d} {short}"
        pos[key].append(d); gap[key].append(g)
        k += 1
        if short == "tail":
            break
tot = 0
for key in sorted(pos):
    v = sorted(pos[key]); g = sorted(gap[key])
    tot += v[len(v)//2] + g[len(g)//2]
    print(f"  {key:10s} median {v[len(v)//2]:7.2f} us  gap before {g[len(g)//2]:6.2f} us  (n {len(v)})")
print(f"  generator phase, medians summed: {tot:.1f} us")
PY
done
