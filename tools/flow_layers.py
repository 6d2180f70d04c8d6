#!/usr/bin/env python3
"""Per-launch timing of the flow net (developer tool, needs a GPU): each of the
"flow" steps timed alone with HIP events on the runtime's stream."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M  # noqa: E402
from joshupscale_amd import runtime as R  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "psp-quality"
cfg = M.PRESETS[preset]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_BF16)
_, n, _ = rt.time_steps("flow", 1)
total = 0.0
for k in range(n):
    best = min(rt.time_steps(f"flow#{k}", 20)[0] for _ in range(3)) * 1e3
    fl = rt.time_steps(f"flow#{k}", 1)[2]
    total += best
    print(f"flow#{k:2d}: {best:7.2f} us  {fl / 1e9:6.2f} GFLOP  {fl / best / 1e6 if best else 0:7.1f} TFLOP/s")
ms, n, fl = rt.time_steps("flow", 20)
print(f"sum of isolated launches {total:.1f} us; back-to-back {ms * n * 1e3:.1f} us per frame")
# the same launches over the frames of a look-ahead pass (JU_LOOKAHEAD, default 8), timed inside whole passes
look = int(rt.stat("lookahead_max"))
if look > 1:
    try:
        _, n, _ = rt.time_steps("flow@pass", 1)
        total = 0.0
        for k in range(n):
            best = min(rt.time_steps(f"flow#{k}@pass", 10)[0] for _ in range(3)) * 1e3
            fl = rt.time_steps(f"flow#{k}@pass", 1)[2]
            total += best
            print(f"pass of {look} flow#{k:2d}: {best:7.2f} us = {best / look:6.2f} us per frame  {fl / 1e9:6.2f} GFLOP  "
                  f"{fl / best / 1e6 if best else 0:7.1f} TFLOP/s")
        print(f"pass of {look}: flow launches {total:.1f} us = {total / look:.1f} us per frame")
    except R.JoshUpscaleError as e:
        print("no look-ahead passes:", e)
for tag in ["pack", "flow", "warp", "tower", "tail", ""]:
    ms, n, fl = rt.time_steps(tag, 10)
    print(f"stage {tag or 'ALL':9s}: {n:3d} launches, {ms * n * 1e3:8.1f} us per frame")
