#!/bin/bash
# GPU box: the resident tower's real durations (rocprofv3 kernel trace) across tools/probes/two_runtimes.py -- does the
# kernel run faster while a second runtime is alive, or do only the HIP events of ju_time_steps read shorter?
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/two_rt_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/two_rt_trace -- python3 $R/tools/probes/two_runtimes.py > $R/gpurun_out/r05/two_rt.txt 2>&1
cd $R
tail -10 gpurun_out/r05/two_rt.txt
python3 - <<'PY' | tee -a gpurun_out/r05/two_rt.txt
import csv, glob
f = glob.glob("/tmp/two_rt_trace/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "tower_resident" in r["Kernel_Name"]]
d = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id")) for r in rows)
print(len(d), "tower launches in the trace; average duration per 100 consecutive launches (queue ids):")
for i in range(0, len(d), 100):
    c = d[i:i + 100]
    print("%5d  %.1f us  %s" % (i, sum(x[1] for x in c) / len(c), sorted(set(x[2] for x in c))))
PY
