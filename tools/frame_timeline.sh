#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# GPU box: per-dispatch timeline of one frame (rocprofv3 kernel trace).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-iters 1 > $R/gpurun_out/tl.log 2>&1
python3 - <<'PY'
import csv, glob, os
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/tl/*/*_kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'pack_frames' in r['Kernel_Name']]
s,e=idx[-3],idx[-2]
t0=int(rows[s]['Start_Timestamp']); prev=None
for r in rows[s:e]:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(st-prev)/1e3 if prev else 0; prev=en
    n=r['Kernel_Name'].split('(')[0]
    n=n.replace('_ZN2ju12_GLOBAL__N_1','').replace('void ju::(anonymous namespace)::','')[:60]
    print(f"{(st-t0)/1e3:8.1f} gap {gap:5.1f} dur {(en-st)/1e3:7.1f} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):>5}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} {n}")
print("frame total us:", (int(rows[e]['Start_Timestamp'])-t0)/1e3)
PY
