#!/usr/bin/env python3
"""Fixed cost of one resident-tower launch (developer tool, needs a GPU): the launch timed inside
whole frames for generators of 1 .. 24 residual blocks, and the line a + b * layers through it --
a = prologue (LDS clear, first-layer input, layer-0 weights, slot descriptors) + last-layer store +
launch and drain, b = one layer."""
import dataclasses
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M, runtime as R  # noqa: E402

dt = {"bf16": R.DTYPE_BF16, "fp16": R.DTYPE_F16, "fp8": R.DTYPE_FP8}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
xs, ys = [], []
for blocks in (1, 2, 4, 8, 16, 24):
    cfg = dataclasses.replace(M.PRESETS["psp-quality"], gen_blocks=blocks)
    rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, dt)
    for _ in range(3):
        ms = min(rt.time_steps("tower@frame", 20)[0] for _ in range(3))
    layers = 2 * blocks + (0 if dt == R.DTYPE_FP8 else 1)
    xs.append(layers)
    ys.append(ms * 1e3)
    print(f"{blocks:2d} blocks = {layers:2d} layers: {ms * 1e3:7.1f} us per launch")
    rt.close()
b, a = np.polyfit(xs, ys, 1)
print(f"fit: {a:.1f} us fixed + {b:.2f} us per layer")
