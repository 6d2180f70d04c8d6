#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# GPU box: per-kernel evidence table for DESIGN.md -- duration (kernel trace), HBM bytes
# (FETCH_SIZE x2 + WRITE_SIZE, separate passes, gfx950 correction of MI355X_MICROARCH.md)
# and chip-level MFMA busy share: SQ_VALU_MFMA_BUSY_CYCLES (summed over SIMDs) /
# (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs).
# --pmc runs use --kernel-trace only.  Output: gpurun_out/pmc_per_kernel.json
# usage: bash tools/pmc_all.sh [extra bench.py arguments, e.g. --preset ps2-quality --dtype fp8]
R=$GRAFT_REPO_ROOT
EXTRA="$*"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pa$i
  timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pa$i -- python3 $R/bench.py $EXTRA --extra-frames 0 --steps 8 --warmup 2 --no-cpu-baseline --roofline-iters 1 > $R/gpurun_out/pa$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections, json, re
R=os.environ['GRAFT_REPO_ROOT']
def short(k):
    k=k.split('(')[0]
    m=re.search(r'(\d+)(tower8_resident_kernel|tower_resident_kernel|res_block_fp8_kernel|res_block_pipe_kernel|res_block_kernel|flow_block_kernel|conv_splitk_kernel|conv_tower_fp8_kernel|conv_tower_kernel|quantize_tower_kernel|conv_mfma_kernel|tail_fused_kernel|warp_pack_kernel|pack_frames_kernel|upsample2_kernel|maxpool2_kernel)(.*)', k)
    if not m: return None
    name=m.group(2)
    if name=='conv_tower_fp8_kernel':
        name+='<stream>' if 'Lb1E' in m.group(3) else '<first>'
    if name=='flow_block_kernel':
        p=re.search(r'Li(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELi(\d)', m.group(3))
        if p: name+=f'<cin{p.group(1)},cmid{p.group(2)},th{p.group(3)},ups{p.group(4)},pool{p.group(5)},out{p.group(6)}>'
    if name=='conv_splitk_kernel':
        p=re.search(r'Li(\d+)ELi(\d+)ELb(\d)', m.group(3))
        if p: name+=f'<cin{p.group(1)},cb{p.group(2)},pool{p.group(3)}>'
    if name=='conv_mfma_kernel':
        p=re.search(r'Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)', m.group(3))
        if p: name+=f'<taps{p.group(1)},ck{p.group(2)},nb{p.group(3)},rw{p.group(4)},dbuf{p.group(5)}>'
    return name
out=collections.defaultdict(dict)
for d in sorted(glob.glob(R+'/gpurun_out/pa[0-9]/*/*_counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        k=short(r['Kernel_Name'])
        if k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        for c,x in v.items(): out[k][c]=sum(x)/len(x)
tr=glob.glob(R+'/gpurun_out/pa1/*/*_kernel_trace.csv')[0]
dur=collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    k=short(r['Kernel_Name'])
    if k: dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in dur.items():
    out[k]['us_per_launch_under_pmc']=sum(v)/len(v); out[k]['launches']=len(v)
for k,v in out.items():
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        v['hbm_bytes_per_launch']=(2*v['FETCH_SIZE']+v['WRITE_SIZE'])*1024
        v['hbm_GBps']=v['hbm_bytes_per_launch']/v['us_per_launch_under_pmc']/1e3
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and v.get('GRBM_GUI_ACTIVE'):
        v['mfma_busy_frac']=v['SQ_VALU_MFMA_BUSY_CYCLES']/(1024.0*v['GRBM_GUI_ACTIVE']/8.0)
import sys
sys.path.insert(0, R)
from joshupscale_amd.provenance import kernel_source_digest
for k,v in out.items():
    v['source_sha256']=kernel_source_digest(k)   # bench.py quotes a byte count only for the source it was counted on
json.dump(out, open(R+'/gpurun_out/pmc_per_kernel.json','w'), indent=1, sort_keys=True)
for k,v in sorted(out.items(), key=lambda kv:-kv[1].get('us_per_launch_under_pmc',0)*kv[1].get('launches',0)):
    print(f"{k:55s} {v.get('launches',0):4d} x {v.get('us_per_launch_under_pmc',0):7.1f} us  hbm {v.get('hbm_bytes_per_launch',0)/1e6:7.1f} MB {v.get('hbm_GBps',0):7.0f} GB/s  mfma busy {100*v.get('mfma_busy_frac',0):5.1f}%")
PY
