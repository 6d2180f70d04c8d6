#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# GPU box: HBM traffic of the tower kernel (FETCH_SIZE / WRITE_SIZE in separate passes).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --roofline-iters 1 > $R/gpurun_out/pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections, json
R=os.environ['GRAFT_REPO_ROOT']
out={}
for c in ['FETCH_SIZE','WRITE_SIZE']:
    f=glob.glob(R+f'/gpurun_out/pmc_{c}/*/*_counter_collection.csv')[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']==c:
            k=r['Kernel_Name'].split('(')[0]
            k='tower_resident_kernel' if 'tower_resident' in k else ('conv_mfma_kernel' if 'conv_mfma' in k else k[-40:])
            agg[k].append(float(r['Counter_Value']))
    out[c]={k: sum(v)/len(v) for k,v in agg.items()}
print(json.dumps(out, indent=1))
json.dump(out, open(R+'/gpurun_out/traffic_raw.json','w'), indent=1)
PY
