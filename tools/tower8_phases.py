#!/usr/bin/env python3
"""In-kernel phase profile of the 8-bit resident tower (developer build -DJU_T8_PROF of tower8_kernels.hip, s_memtime stamps;
shares only).  usage: JU_LIBRARY=build/ab/lib_<x>.so python tools/tower8_phases.py   (needs a GPU)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("JU_TEST_HOOKS", "1")
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["psp-quality"]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_FP8)
ms, n, fl = rt.time_steps("tower", 3)
raw = rt.read_tensor("tower_profile").view(np.uint64).reshape(256, 4, 8)[:255].astype(np.float64) / 48
names = ["sweep", "weights", "units", "barrier", "bias+publish", "(K loops)", "(writes+last epi)", "-"]
print(f"profiled launch {ms*1e3:.0f} us; cycles per layer (48 layers)")
start, xcd = 0, np.zeros(255, int)
for x in range(8):
    c = (255 - x + 7) // 8
    xcd[start:start + c] = x
    start += c
for w in range(4):
    med = np.median(raw[:, w, :], axis=0)
    print(f" wave {w}: " + ", ".join(f"{nm} {v:6.0f}" for nm, v in zip(names[:7], med[:7])) + f" | sum {med[:5].sum():6.0f}")
for x in range(8):
    med = np.median(raw[xcd == x, 0, :], axis=0)
    print(f" xcd {x} wave 0: " + ", ".join(f"{nm} {v:6.0f}" for nm, v in zip(names[:7], med[:7])) + f" | sum {med[:5].sum():6.0f}")
