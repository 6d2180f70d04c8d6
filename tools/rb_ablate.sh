#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# developer tool: timing ablation of res_block_kernel at 640x448 (JU_FB_SKIP bits: 1 staging, 2 conv A MFMA,
# 4 conv B MFMA, 8 skip loads, 16 stores, 32 conv A epilogue, 64 conv B epilogue)
# needs the probe build: `make ablate` (the product library ignores JU_FB_SKIP)
export JU_LIBRARY=${JU_LIBRARY:-$PWD/build/ablate/libJoshUpscale_test.so}
for s in 0 1 2 4 8 16 32 64 6 102 127; do
  JU_FB_SKIP=$s JU_TOWER=layers python - <<PY
import os, sys
sys.path.insert(0, ".")
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["ps2-quality"]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_BF16)
ms, n, fl = rt.time_steps("tower", 5)
print("skip", os.environ["JU_FB_SKIP"], "%.2f us per block" % (ms * 1e3))
PY
done
