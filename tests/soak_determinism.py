#!/usr/bin/env python3
"""Soak: the resident tower's halo exchange is timing-dependent (retries, neighbours
running ahead) but its RESULT must not be.  Runs the same long clip twice on the full
benchmark model and compares every output frame's checksum.  Needs a GPU.
The 8-bit per-layer tower has no exchange, but a missing DMA wait once produced rare
stale tiles there: same check, and with a preset / dtype argument the
second run uses one workgroup per CU (JU_FP8_GRID=256), which must not change a byte.
usage: python tests/soak_determinism.py [frames] [preset] [bf16|fp16|fp8]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M, runtime as R  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
preset = sys.argv[2] if len(sys.argv) > 2 else "psp-quality"
dtype = {"bf16": R.DTYPE_BF16, "fp16": R.DTYPE_F16, "fp8": R.DTYPE_FP8}[sys.argv[3] if len(sys.argv) > 3 else "bf16"]
cfg = M.PRESETS[preset]
blob = M.serialize(cfg, M.make_seeded_weights(cfg))
clip = M.synthetic_frames(16, cfg.frame_height, cfg.frame_width, seed=7, kind="smooth")


def run(grid=None):
    if grid:
        os.environ["JU_FP8_GRID"] = str(grid)
        os.environ["JU_NO_GRAPH"] = "1"   # the override acts on new launches only
    rt = R.Runtime(blob, 0, dtype)
    h = hashlib.sha256()
    t0 = time.perf_counter()
    for i in range(n):
        h.update(rt.process_image(clip[i % 16]).tobytes()[:: 4099])   # sparse sample of every frame
    dt = time.perf_counter() - t0
    rt.close()
    return h.hexdigest(), dt


a, ta = run()
b, tb = run(256 if dtype == R.DTYPE_FP8 else None)
print(f"{n} frames twice: {ta:.1f} s / {tb:.1f} s (host frames), digests {'EQUAL' if a == b else 'DIFFER'}: {a[:16]} {b[:16]}")
sys.exit(0 if a == b else 1)
