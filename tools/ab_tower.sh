#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# GPU box: interleaved A/B of two builds of libJoshUpscale.so on the tower stage.
# usage: bash tools/ab_tower.sh <libA.so> <libB.so> [dtype: bf16|fp16|fp8] [preset] [rounds]
A=$1; B=$2; DT=${3:-bf16}; PRESET=${4:-psp-quality}; N=${5:-3}
for i in $(seq $N); do
  for L in $A $B; do
    JU_LIBRARY=$L DT=$DT PRESET=$PRESET python3 - <<PY
import os, sys
sys.path.insert(0, ".")
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS[os.environ["PRESET"]]
dt = {"bf16": R.DTYPE_BF16, "fp16": R.DTYPE_F16, "fp8": R.DTYPE_FP8}[os.environ["DT"]]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, dt)
ms, n, fl = rt.time_steps("tower", 20)
print("%-20s tower %8.1f us (%d launches)" % (os.path.basename(os.environ["JU_LIBRARY"]), ms * n * 1e3, n))
PY
  done
done
