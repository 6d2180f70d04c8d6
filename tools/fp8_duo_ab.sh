#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# GPU box: res_block_fp8_kernel as one 8-wave workgroup per CU (solo) against two 4-wave workgroups per CU (duo)
for r in 1 2 3; do
for f in solo duo; do
  JU_FP8_BLOCK=$f python3 - <<PY
import os, sys
sys.path.insert(0, ".")
from joshupscale_amd import model_file as M, runtime as R
for preset in ("ps2-quality",):
    cfg = M.PRESETS[preset]
    rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_FP8)
    ms = min(rt.time_steps("tower#3", 10)[0] for _ in range(3))
    fr = min(rt.time_steps("", 10)[0] * rt.time_steps("", 1)[1] for _ in range(3))
    print(preset, os.environ["JU_FP8_BLOCK"], "%.2f us per block, %.1f us per frame (eager, back to back)" % (ms * 1e3, fr * 1e3))
PY
done; done
