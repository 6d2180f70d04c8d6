#!/bin/bash
# developer tool: headline bench lines of both engines (bf16, fp8) on this box, short form
for dt in bf16 fp8; do
  python bench.py --dtype $dt --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt', round(d['value'],1), 'fps', round(d['ms_per_step'],4), 'ms; tower', round(d['roofline']['launch_ms']*1e3,1), 'us frac', round(d['roofline']['frac'],4))"
done
