#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# GPU box: does res_block_fp8_kernel stall on the ACKNOWLEDGEMENT of its own output stores?  (gfx950's vmcnt
# counts stores; the wait for a pair's skip records is vmcnt(0).)  JU_FB_SKIP bits: 16 = no stores, 128 = no wait.
# needs `make ablate`
export JU_LIBRARY=$PWD/build/ablate/libJoshUpscale_test.so
for s in 0 16 128 144 0 128; do
  JU_FB_SKIP=$s python3 - <<PY
import os, sys
sys.path.insert(0, ".")
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["ps2-quality"]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_FP8)
ms = min(rt.time_steps("tower#3", 10)[0] for _ in range(3))
print("ps2-quality fp8 block, skip", os.environ["JU_FB_SKIP"], "%.2f us per block" % (ms * 1e3))
PY
done
unset JU_LIBRARY
for m in 0 1 2 3 0 1; do
  JU_WAVE_PRIO=$m python3 - <<PY
import os, sys
sys.path.insert(0, ".")
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["ps2-quality"]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_FP8)
ms = min(rt.time_steps("tower#3", 10)[0] for _ in range(3))
print("product lib, JU_WAVE_PRIO", os.environ["JU_WAVE_PRIO"], "%.2f us per block" % (ms * 1e3))
PY
done
