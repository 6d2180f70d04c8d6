"""Seconds per frame of the C restatement (oracle/ju_oracle_c.c) at the benchmark geometry, for
tools/cpu_scaling.sh (thread count from OMP_NUM_THREADS)."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from joshupscale_amd import model_file as M  # noqa: E402
from oracle.c_binding import CSession  # noqa: E402

cfg = M.PRESETS[sys.argv[1] if len(sys.argv) > 1 else "psp-quality"]
blob = M.serialize(cfg, M.make_seeded_weights(cfg, seed=42))
sess = CSession(blob, cfg.frame_height, cfg.frame_width)
frames = M.synthetic_frames(3, cfg.frame_height, cfg.frame_width, seed=1, kind="noise")
for i in range(3):
    t = time.perf_counter()
    out = sess.run(frames[i])
    print(f"  frame {i}: {time.perf_counter() - t:.3f} s  threads {sess.threads}  {sess.vector_bits}-bit  {hashlib.sha256(out).hexdigest()[:12]}")
