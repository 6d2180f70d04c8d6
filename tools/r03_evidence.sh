#!/bin/bash
# GPU box: the evidence set of round 3 (bench lines + rocprofv3 kernel stats + PMC tables + quality + soak).
# The driver's own command line first: `bench.py --steps 20 --warmup 5` must read what the 300-step run reads.
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_driver_cmd_bench.json 2> gpurun_out/r03_driver_cmd.err
bash tools/profile_bench.sh r03_final
bash tools/profile_bench.sh r03_lrelu --preset psp-quality-lrelu --no-cpu-baseline
bash tools/profile_bench.sh r03_fp8_psp --dtype fp8 --no-cpu-baseline
bash tools/profile_bench.sh r03_fp8_ps2 --preset ps2-quality --dtype fp8 --no-cpu-baseline
bash tools/profile_bench.sh r03_ps2 --preset ps2-quality --no-cpu-baseline
bash tools/profile_bench.sh r03_fast --preset psp-fast --dtype fp16 --no-cpu-baseline
bash tools/pmc_all.sh > gpurun_out/r03_pmc_table.txt 2>&1; cp gpurun_out/pmc_per_kernel.json gpurun_out/r03_pmc_per_kernel.json
bash tools/pmc_all.sh --dtype fp8 > gpurun_out/r03_pmc_table_fp8.txt 2>&1; cp gpurun_out/pmc_per_kernel.json gpurun_out/r03_pmc_per_kernel_fp8.json
bash tools/variants.sh > gpurun_out/r03_variants.txt 2>&1
python3 tools/flow_layers.py > gpurun_out/r03_flow_layers.txt 2>&1
python3 tests/quality_report.py --frames 6 --preset psp-quality > gpurun_out/r03_quality_psp.json 2>/dev/null
python3 tests/quality_report.py --frames 4 --preset ps2-quality > gpurun_out/r03_quality_ps2.json 2>/dev/null
timeout 600 python3 tests/soak_determinism.py 3000 psp-quality bf16 > gpurun_out/r03_soak.txt 2>&1
timeout 600 python3 tests/soak_determinism.py 3000 psp-quality fp8 >> gpurun_out/r03_soak.txt 2>&1
timeout 600 python3 tests/soak_determinism.py 3000 psp-quality-lrelu bf16 >> gpurun_out/r03_soak.txt 2>&1
tail -3 gpurun_out/r03_soak.txt; tail -12 gpurun_out/r03_pmc_table.txt
