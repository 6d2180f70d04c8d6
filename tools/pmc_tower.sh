#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# GPU box: PMC counters of the tower kernel (separate passes, --kernel-trace only).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc$i -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --roofline-iters 1 > $R/gpurun_out/pmc$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/pmc[0-9]/*/*_counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        k=r['Kernel_Name']
        key='tower' if ('tower_resident' in k or 'conv_tower' in k) else None
        if key: agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, 'n=', len(list(v.values())[0]))
tr=glob.glob(R+'/gpurun_out/pmc1/*/*_kernel_trace.csv')[0]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(tr)) if 'tower_resident' in r['Kernel_Name']]
print('tower durations us (under PMC):', [round(x) for x in d])
PY
