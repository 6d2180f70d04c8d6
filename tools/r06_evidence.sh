#!/bin/bash
# GPU box: the evidence set of round 6 (the driver's command line, bench lines + rocprofv3 kernel stats, PMC tables per
# preset / dtype -- bench.py's roofline.traffic quotes those, and only while their source digest matches --, the
# flow net per launch, variants, soak).  Everything lands in gpurun_out/r06/; copy what is to be judged into profiles/.
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r06
mkdir -p $O
# PMC tables, one per (preset, dtype)
for pd in "psp-quality bf16" "psp-quality fp8" "ps2-quality fp8" "ps2-quality bf16" "psp-fast fp16"; do
  set -- $pd
  bash tools/pmc_all.sh --preset $1 --dtype $2 > $O/pmc_table_$1_$2.txt 2>&1
  cp gpurun_out/pmc_per_kernel.json $O/pmc_per_kernel_$1_$2.json
  # (bench.py quotes roofline.traffic from profiles/: the table of THIS source, before the bench lines below are taken)
  cp gpurun_out/pmc_per_kernel.json profiles/r06_pmc_per_kernel_$1_$2.json
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_bench.json 2> $O/driver_cmd.err
for spec in "final" "fp8_psp --dtype fp8 --no-cpu-baseline" "fp8_ps2 --preset ps2-quality --dtype fp8 --no-cpu-baseline" \
            "ps2 --preset ps2-quality --no-cpu-baseline" "fast --preset psp-fast --dtype fp16 --no-cpu-baseline" \
            "lrelu --preset psp-quality-lrelu --no-cpu-baseline"; do
  set -- $spec
  tag=$1; shift
  bash tools/profile_bench.sh r06_$tag "$@" > /dev/null 2>&1
  cp gpurun_out/bench_r06_$tag.json $O/${tag}_bench.json
  cp $(ls gpurun_out/prof_r06_$tag/*/*_kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
done
bash tools/bench_matrix.sh r06/matrix > $O/bench_matrix.txt 2>&1
python3 tools/flow_layers.py > $O/flow_layers.txt 2>&1
python3 tools/tower_phases.py > $O/tower_phases.txt 2>&1
timeout 600 python3 tests/soak_determinism.py 3000 psp-quality bf16 8 > $O/soak.txt 2>&1
timeout 600 python3 tests/soak_determinism.py 3000 psp-quality fp8 8 >> $O/soak.txt 2>&1
python3 tools/tower_phase_map.py > $O/tower_phase_map.txt 2>&1
python3 -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1
cp gpurun_out/parity_stats.json $O/parity_stats.json
tail -3 $O/soak.txt; tail -12 $O/pmc_table_psp-quality_bf16.txt; cat $O/flow_layers.txt | tail -8; cat $O/bench_matrix.txt; grep -E "passed|failed" $O/gputests.txt
