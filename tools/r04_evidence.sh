#!/bin/bash
# GPU box: the evidence set of round 4 (bench lines + rocprofv3 kernel stats + PMC tables per preset / dtype + quality
# + soak).  The driver's own command line first.
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_driver_cmd_bench.json 2> gpurun_out/r04_driver_cmd.err
bash tools/profile_bench.sh r04_final
bash tools/profile_bench.sh r04_lrelu --preset psp-quality-lrelu --no-cpu-baseline
bash tools/profile_bench.sh r04_fp8_psp --dtype fp8 --no-cpu-baseline
bash tools/profile_bench.sh r04_fp8_ps2 --preset ps2-quality --dtype fp8 --no-cpu-baseline
bash tools/profile_bench.sh r04_ps2 --preset ps2-quality --no-cpu-baseline
bash tools/profile_bench.sh r04_fast --preset psp-fast --dtype fp16 --no-cpu-baseline
# PMC tables, one per (preset, dtype): bench.py's roofline.traffic quotes these files
for pd in "psp-quality bf16" "psp-quality fp8" "ps2-quality fp8" "ps2-quality bf16" "psp-fast fp16"; do
  set -- $pd
  bash tools/pmc_all.sh --preset $1 --dtype $2 > gpurun_out/r04_pmc_table_$1_$2.txt 2>&1
  cp gpurun_out/pmc_per_kernel.json gpurun_out/r04_pmc_per_kernel_$1_$2.json
done
bash tools/variants.sh > gpurun_out/r04_variants.txt 2>&1
python3 tools/flow_layers.py > gpurun_out/r04_flow_layers.txt 2>&1
python3 tools/tower_phases.py > gpurun_out/r04_tower_phases.txt 2>&1
python3 tests/quality_report.py --frames 6 --preset psp-quality > gpurun_out/r04_quality_psp.json 2>/dev/null
python3 tests/quality_report.py --frames 4 --preset ps2-quality > gpurun_out/r04_quality_ps2.json 2>/dev/null
timeout 600 python3 tests/soak_determinism.py 3000 psp-quality bf16 > gpurun_out/r04_soak.txt 2>&1
timeout 600 python3 tests/soak_determinism.py 3000 psp-quality fp8 >> gpurun_out/r04_soak.txt 2>&1
timeout 600 python3 tests/soak_determinism.py 3000 psp-quality-lrelu bf16 >> gpurun_out/r04_soak.txt 2>&1
tail -3 gpurun_out/r04_soak.txt; tail -12 gpurun_out/r04_pmc_table_psp-quality_bf16.txt
