#!/bin/bash
# GPU box: what the shader clock does while the frame loop runs (rocm-smi / amd-smi sampled beside a long bench run).
OUT=gpurun_out/clock_under_load.txt
{
echo "== idle"; rocm-smi --showclocks 2>&1 | grep -iE "sclk|mclk|fclk" | head -4
rocm-smi --showpower --showmaxpower 2>&1 | grep -iE "power|watt" | head -4
python3 bench.py --steps 60000 --warmup 30 --no-cpu-baseline > gpurun_out/clock_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do
  echo "== under load, sample $i"
  rocm-smi --showclocks 2>&1 | grep -iE "sclk|mclk" | head -2
  rocm-smi --showpower 2>&1 | grep -iE "power|watt" | head -2
  sleep 2
done
amd-smi metric --clock --power 2>&1 | head -60
wait $BP
python3 -c "import json; d=json.loads(open('gpurun_out/clock_bench.json').read().strip().splitlines()[-1]); print('bench', round(d['value'],1), 'fps', d['roofline']['launch_ms'])"
} > $OUT 2>&1
cat $OUT
