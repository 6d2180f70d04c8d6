#!/bin/bash
# GPU box: where does a kernel's time go?  Issue / wait / LDS / VMEM counters of the kernels whose name matches a
# pattern, one counter group per rocprofv3 pass (--kernel-trace --pmc only), averaged per launch.
# usage: bash tools/pmc_stall.sh <kernel name pattern> <tag> [bench.py arguments]
PAT=$1; TAG=$2; shift; shift
R=$GRAFT_REPO_ROOT
EXTRA="$*"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/ps_$TAG$i
  timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/ps_$TAG$i -- python3 $R/bench.py $EXTRA --steps 6 --warmup 2 --no-cpu-baseline --roofline-iters 1 > $R/gpurun_out/ps_$TAG$i.log 2>&1
done
PAT="$PAT" TAG="$TAG" python3 - <<'PY'
import csv, glob, os, collections, json
R=os.environ['GRAFT_REPO_ROOT']; pat=os.environ['PAT']; tag=os.environ['TAG']
agg=collections.defaultdict(list); dur=[]
for d in sorted(glob.glob(R+f'/gpurun_out/ps_{tag}[0-9]/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(d)):
        if pat in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for d in sorted(glob.glob(R+f'/gpurun_out/ps_{tag}1/*/*_kernel_trace.csv')):
    for r in csv.DictReader(open(d)):
        if pat in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
out={k: sum(v)/len(v) for k,v in agg.items()}
out['us_per_launch_under_pmc']=sum(dur)/max(len(dur),1); out['launches']=len(dur)
json.dump(out, open(R+f'/gpurun_out/pmc_stall_{tag}.json','w'), indent=1, sort_keys=True)
cyc=out.get('GRBM_GUI_ACTIVE',0)/8.0
print(f"{pat} [{tag}]: {out['launches']} launches, {out['us_per_launch_under_pmc']:.1f} us under PMC, {cyc:.0f} cycles per launch")
for k in sorted(out):
    if k.startswith('SQ_') or k.startswith('GRBM'):
        print(f"  {k:34s} {out[k]:16.0f}" + (f"   per SIMD-cycle {out[k]/(1024*cyc):.3f}" if cyc else ""))
PY
