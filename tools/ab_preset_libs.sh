#!/bin/bash
export JU_TEST_HOOKS=1
# developer tool, GPU box: interleaved timing of the "tower" stage of any preset / dtype in several builds of the library
# usage: tools/ab_preset_libs.sh <preset> <bf16|fp16|fp8> <lib.so>...
P=$1; D=$2; shift 2
for r in 1 2 3; do
  for L in "$@"; do
    JU_LIBRARY=$L python3 - "$P" "$D" <<PY
import os, sys
sys.path.insert(0, ".")
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS[sys.argv[1]]
dt = {"bf16": R.DTYPE_BF16, "fp16": R.DTYPE_F16, "fp8": R.DTYPE_FP8}[sys.argv[2]]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, dt)
ms = min(rt.time_steps("tower@frame", 20)[0] for _ in range(3))
print("%-20s %-22s %-5s tower in frame %.1f us" % (os.path.basename(os.environ["JU_LIBRARY"]), sys.argv[1], sys.argv[2], ms * 1e3))
PY
  done
done
