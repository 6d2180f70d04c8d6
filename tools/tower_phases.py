#!/usr/bin/env python3
"""In-kernel phase profile of the resident tower (diagnostic variant 4, s_memtime
stamps; shares only, never quote its run time).  Needs a GPU."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["psp-quality"]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_BF16)
lib = R.load_library()
lib.ju_debug_set(b"tower_variant", 4)
ms, n, fl = rt.time_steps("tower", 3)
raw = rt.read_tensor("tower_profile").view(np.uint64).reshape(256, 4, 8)[:255, :, :8].astype(np.float64)
lib.ju_debug_set(b"tower_variant", 0)
names = ["halo fill", "weight issue", "compute", "barrier", "publish", "(K loops)", "(epilogues)", "(pre-run K)"]
print(f"diagnostic launch {ms*1e3:.0f} us; cycles per layer (49 layers), median over regions")
for w in range(4):
    med = np.median(raw[:, w, :], axis=0) / 49
    print(f" wave {w} (ch {w&1}, rp {w>>1}): " + ", ".join(f"{n} {v:6.0f}" for n, v in zip(names, med)) + f" | sum {med[:5].sum():6.0f}")
