export JU_TEST_HOOKS=1
export JU_LIBRARY=$PWD/build/ablate/libJoshUpscale_test.so
for s in 0 1 2 4 6 8 16 32 64 96 102 127 25; do
  JU_FB_SKIP=$s python - <<PY
import os, sys
sys.path.insert(0, ".")
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["ps2-quality"]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_FP8)
ms = min(rt.time_steps("tower#3", 10)[0] for _ in range(3))
print("ps2-quality skip %4s  %.2f us per block" % (os.environ["JU_FB_SKIP"], ms * 1e3))
PY
done
