#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# GPU box: where does a kernel's time go?  Issue / wait / LDS / VMEM counters of the kernels whose name matches a
# pattern, one counter group per rocprofv3 pass (--kernel-trace --pmc only), averaged per launch.
# usage: bash tools/pmc_stall.sh <kernel name pattern[,pattern...]> <tag> [bench.py arguments]
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
  timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/ps_$TAG$i -- python3 $R/bench.py $EXTRA --lookahead 1 --steps 6 --warmup 2 --no-cpu-baseline --roofline-iters 1 > $R/gpurun_out/ps_$TAG$i.log 2>&1
done
PAT="$PAT" TAG="$TAG" python3 - <<'PY'
import csv, glob, os, collections, json, re
R=os.environ['GRAFT_REPO_ROOT']; pats=os.environ['PAT'].split(','); tag=os.environ['TAG']
def key(name):
    for p in pats:
        if p in name: return p
    return None
agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for d in sorted(glob.glob(R+f'/gpurun_out/ps_{tag}[0-9]/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(d)):
        k=key(r['Kernel_Name'])
        if k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for d in sorted(glob.glob(R+f'/gpurun_out/ps_{tag}1/*/*_kernel_trace.csv')):
    for r in csv.DictReader(open(d)):
        k=key(r['Kernel_Name'])
        if k: dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
res={}
for k in pats:
    out={c: sum(v)/len(v) for c,v in agg[k].items()}
    out['us_per_launch_under_pmc']=sum(dur[k])/max(len(dur[k]),1); out['launches']=len(dur[k])
    res[k]=out
    cyc=out.get('GRBM_GUI_ACTIVE',0)/8.0
    print(f"{k} [{tag}]: {out['launches']} launches, {out['us_per_launch_under_pmc']:.1f} us under PMC, {cyc:.0f} cycles per launch")
    if not cyc: continue
    q=lambda c: 4.0*out.get(c,0)/(1024*cyc)      # SQ_*: quad-cycles summed over waves / SIMDs -> share of SIMD cycles
    print(f"   waves per SIMD {q('SQ_WAVE_CYCLES'):.2f}; of a SIMD's cycles: VALU issue {q('SQ_ACTIVE_INST_VALU'):.3f}, LDS issue {q('SQ_ACTIVE_INST_LDS'):.3f}, "
          f"VMEM issue {q('SQ_ACTIVE_INST_VMEM'):.3f}, scalar {q('SQ_ACTIVE_INST_SCA'):.3f}; MFMA busy {out.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*cyc):.3f}")
    print(f"   waiting on s_waitcnt {q('SQ_WAIT_INST_ANY'):.3f} (LDS {q('SQ_WAIT_INST_LDS'):.3f}), waiting on anything {q('SQ_WAIT_ANY'):.3f} of SIMD cycles x waves")
    lds=out.get('SQ_LDS_IDX_ACTIVE',0)
    print(f"   LDS busy {lds/(256*cyc):.3f} of CU cycles, bank conflicts {out.get('SQ_LDS_BANK_CONFLICT',0)/(256*cyc):.3f} ({100*out.get('SQ_LDS_BANK_CONFLICT',0)/max(lds,1):.0f} % of LDS cycles); "
          f"per launch: {out.get('SQ_INSTS_VALU',0):.0f} VALU, {out.get('SQ_INSTS_MFMA',0):.0f} MFMA, {out.get('SQ_INSTS_LDS',0):.0f} LDS, {out.get('SQ_INSTS_VMEM',0):.0f} VMEM, {out.get('SQ_INSTS_SALU',0):.0f} SALU instructions")
json.dump(res, open(R+f'/gpurun_out/pmc_stall_{tag}.json','w'), indent=1, sort_keys=True)
PY
