#!/usr/bin/env python3
"""Stage timings of one configuration (developer tool, needs a GPU):
tools/stage_times.py [preset] [bf16|fp16|fp8] [steps of one tag to list, e.g. tower]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M  # noqa: E402
from joshupscale_amd import runtime as R  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "psp-quality"
dtype = {"bf16": R.DTYPE_BF16, "fp16": R.DTYPE_F16, "fp8": R.DTYPE_FP8}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
cfg = M.PRESETS[preset]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, dtype)
if len(sys.argv) > 3:
    tag = sys.argv[3]
    _, n, _ = rt.time_steps(tag, 1)
    for k in range(min(n, 8)):
        best = min(rt.time_steps(f"{tag}#{k}", 20)[0] for _ in range(3)) * 1e3
        fl = rt.time_steps(f"{tag}#{k}", 1)[2]
        print(f"{tag}#{k:2d}: {best:7.2f} us  {fl / 1e9:6.2f} GFLOP  {fl / best / 1e6 if best else 0:7.1f} TFLOP/s")
for tag in ["pack", "flow", "warp", "gen_head", "tower", "tail", ""]:
    ms, n, fl = rt.time_steps(tag, 10)
    print(f"stage {tag or 'ALL':9s}: {n:3d} launches, {ms * n * 1e3:8.1f} us per frame")
