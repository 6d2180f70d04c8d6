#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# developer tool, GPU box: interleaved timing of the resident tower in several builds of the library
# usage: tools/ab_tower_libs.sh <lib.so>... ; prints us per tower launch (min of 3 x 20 launches) per round
LIBS="$@"
for r in 1 2 3; do
  for L in $LIBS; do
    JU_LIBRARY=$L python3 - <<PY
import os, sys
sys.path.insert(0, ".")
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["psp-quality"]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_BF16)
ms = min(rt.time_steps("tower", 20)[0] for _ in range(3))
print("%-28s %.1f us" % (os.path.basename(os.environ["JU_LIBRARY"]), ms * 1e3))
PY
  done
done
