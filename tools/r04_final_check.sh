#!/bin/bash
# GPU box: what the driver runs at round end (GPU suite, smoke, its bench command line), the default bench line, the
# headline profile once more (boxes differ by +-3 %), and the long soaks.
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error" | tail -4 > gpurun_out/r04_final_gputests.log
cat gpurun_out/r04_final_gputests.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_driver_cmd_bench_b.json 2> gpurun_out/r04_driver_cmd_b.err
tail -c 600 gpurun_out/r04_driver_cmd_bench_b.json; echo
bash tools/profile_bench.sh r04_final_b > /dev/null 2>&1
timeout 600 python3 tests/soak_determinism.py 20000 psp-quality bf16 > gpurun_out/r04_soak_b.txt 2>&1
timeout 600 python3 tests/soak_determinism.py 3000 psp-quality fp8 >> gpurun_out/r04_soak_b.txt 2>&1
timeout 600 python3 tests/soak_determinism.py 2000 ps2-quality fp8 >> gpurun_out/r04_soak_b.txt 2>&1
timeout 600 python3 tests/soak_determinism.py 3000 psp-quality-lrelu bf16 >> gpurun_out/r04_soak_b.txt 2>&1
cat gpurun_out/r04_soak_b.txt
