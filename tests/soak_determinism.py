#!/usr/bin/env python3
"""Soak: the resident tower's halo exchange is timing-dependent (retries, neighbours
running ahead) but its RESULT must not be.  Runs the same long clip twice on the full
benchmark model and compares every output frame's checksum.  Needs a GPU.
usage: python tests/soak_determinism.py [frames]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from joshupscale_amd import model_file as M, runtime as R  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cfg = M.PRESETS["psp-quality"]
blob = M.serialize(cfg, M.make_seeded_weights(cfg))
clip = M.synthetic_frames(16, cfg.frame_height, cfg.frame_width, seed=7, kind="smooth")


def run():
    rt = R.Runtime(blob, 0, R.DTYPE_BF16)
    h = hashlib.sha256()
    t0 = time.perf_counter()
    for i in range(n):
        h.update(rt.process_image(clip[i % 16]).tobytes()[:: 4099])   # sparse sample of every frame
    dt = time.perf_counter() - t0
    rt.close()
    return h.hexdigest(), dt


a, ta = run()
b, tb = run()
print(f"{n} frames twice: {ta:.1f} s / {tb:.1f} s (host frames), digests {'EQUAL' if a == b else 'DIFFER'}: {a[:16]} {b[:16]}")
sys.exit(0 if a == b else 1)
