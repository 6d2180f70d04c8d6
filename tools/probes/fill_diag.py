import os, sys
import numpy as np
sys.path.insert(0, ".")
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["psp-quality"]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_BF16)
lib = R.load_library()
lib.ju_debug_set(b"tower_variant", 4)
ms, n, fl = rt.time_steps("tower", 3)
raw = rt.read_tensor("tower_profile").view(np.uint64).reshape(256, 4, 8)[:255].astype(np.float64)
names = ["fill", "wstream", "compute", "fill-barrier", "publish", "prerun", "wait-after-prerun", "passes*1000"]
for w in range(4):
    med = np.median(raw[:, w, :], axis=0) / 48
    mx = np.max(raw[:, w, :], axis=0) / 48
    print(f" wave {w}: " + ", ".join(f"{n} {v:6.0f}" for n, v in zip(names, med)), "| max passes", mx[7], "max fill", mx[0])
