for v in 0 2000 0 2000; do
  echo "spin=$v"
  JU_SYNC_SPIN_US=$v python bench.py --steps 3000 --warmup 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('latency_ms'))"
done
