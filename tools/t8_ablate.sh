#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# developer tool: timing ablation of tower8_resident_kernel (JU_FB_SKIP bits: 1 halo exchange, 2 K loop, 4 epilogue)
# needs the probe build: `make ablate` (the product library ignores JU_FB_SKIP)
export JU_LIBRARY=${JU_LIBRARY:-$PWD/build/ablate/libJoshUpscale_test.so}
for s in 0 1 2 4 3 6 7; do
  JU_FB_SKIP=$s python3 - <<PY
import os, sys
sys.path.insert(0, ".")
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["psp-quality"]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_FP8)
ms = min(rt.time_steps("tower", 10)[0] for _ in range(3))
print("skip", os.environ["JU_FB_SKIP"], "%.1f us per tower, %.2f us per layer" % (ms * 1e3, ms * 1e3 / 48))
PY
done
