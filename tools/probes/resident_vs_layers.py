#!/usr/bin/env python3
"""Developer probe (GPU box): the resident tower against the per-layer tower kernels on the same
model and frames, for a list of frame geometries (HxW).  The two paths share no exchange code and
accumulate in a different order, so the outputs may differ by 1 LSB, not more.

usage: python tools/probes/resident_vs_layers.py 30x48 32x32 270x480 ...
"""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def run(h, w, blocks, mode):
    env = dict(os.environ)
    if mode:
        env["JU_TOWER"] = mode
    code = (
        "import sys, numpy as np\n"
        "import os; os.environ.setdefault('JU_TEST_HOOKS', '1')\n"
        "from joshupscale_amd import model_file as M, runtime as R\n"
        f"cfg = M.ModelConfig(frame_height={h}, frame_width={w}, gen_blocks={blocks})\n"
        "rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_BF16)\n"
        f"fr = M.synthetic_frames(3, {h}, {w}, seed=5, kind='smooth')\n"
        "outs = [rt.process_image(f) for f in fr]\n"
        "sys.stdout.buffer.write(np.stack(outs).tobytes())\n"
    )
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, check=True).stdout
    return np.frombuffer(out, np.uint8).reshape(3, 4 * h, 4 * w, 4)


for arg in sys.argv[1:]:
    h, w = (int(v) for v in arg.split("x"))
    blocks = 3 if h * w < 100000 else 24
    a = run(h, w, blocks, "")
    b = run(h, w, blocks, "layers")
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    bad = np.argwhere(d[..., :3].max(axis=-1) > 1)
    rows = sorted(set((bad[:, 1] // 4).tolist()))[:24]
    cols = sorted(set((bad[:, 2] // 4).tolist()))[:24]
    print(f"{arg}: max diff {d.max()}, pixels off by more than 1: {len(bad)}; LR rows {rows} LR cols {cols}")
