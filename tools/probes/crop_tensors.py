"""Developer probe beside crop_consistency.py: WHICH tensor of a large frame stops agreeing with the crop's.
usage: crop_tensors.py H W [dtype]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import os
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M, runtime as R  # noqa: E402

H, W = int(sys.argv[1]), int(sys.argv[2])
dt = {"bf16": R.DTYPE_BF16, "fp16": R.DTYPE_F16, "fp8": R.DTYPE_FP8}[sys.argv[3] if len(sys.argv) > 3 else "bf16"]
CROP, MARGIN = 384, 128
kw = dict(gen_blocks=2)
big = M.ModelConfig(frame_height=H, frame_width=W, **kw)
wts = M.make_seeded_weights(big, seed=42)
rng = np.random.default_rng(3)
base = rng.integers(0, 256, size=(1, H, W, 4), dtype=np.uint8)
y0, x0 = (H - CROP) // 2 // 8 * 8, (W - CROP) // 3 // 8 * 8
small = M.ModelConfig(frame_height=CROP, frame_width=CROP, **kw)
rs = R.Runtime(M.serialize(small, wts), 0, dt)
rs.process_image(np.ascontiguousarray(base[0, y0:y0 + CROP, x0:x0 + CROP]))
ref = {}
for name, ch, scale in (("flow", 32, 1), ("trunk", 64, 1), ("state", 4, 4)):
    t = rs.read_tensor(name).reshape(scale * CROP, scale * CROP, ch)
    ref[name] = t[scale * MARGIN:scale * (CROP - MARGIN), scale * MARGIN:scale * (CROP - MARGIN)].copy()
rs.close()
rt = R.Runtime(M.serialize(big, wts), 0, dt)
out = rt.process_image(base[0])
for name, ch, scale in (("flow", 32, 1), ("trunk", 64, 1), ("state", 4, 4)):
    t = rt.read_tensor(name).reshape(scale * H, scale * W, ch)
    sub = t[scale * (y0 + MARGIN):scale * (y0 + CROP - MARGIN), scale * (x0 + MARGIN):scale * (x0 + CROP - MARGIN)]
    d = np.abs(sub - ref[name])
    print(f"{name:6s}: crop region max |diff| {d.max():.4g} (values up to {np.abs(ref[name]).max():.3g}), differing {np.mean(d > 0):.3f}")
    # where in the whole tensor do things look wrong?  rows that are all zero / NaN
    rows = np.abs(t).reshape(scale * H, -1).max(axis=1)
    zero_rows = np.flatnonzero(rows == 0)
    print(f"        all-zero rows: {len(zero_rows)}" + (f" (first {zero_rows[0]}, last {zero_rows[-1]})" if len(zero_rows) else ""),
          "nan:", bool(np.isnan(rows).any()))
    del t
rt.close()
