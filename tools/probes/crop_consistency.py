"""Developer probe: a large frame against crops of itself.  Far enough from a crop's inner borders an output pixel
depends on nothing outside the crop (receptive field of the flow auto-encoder + warp + a short generator), so the big
frame's output there must equal the small frame's: a size-independent property for geometries no CPU oracle finishes.
Crops at the top-left corner, the centre and the bottom-right corner (the highest addresses of every tensor).
usage: crop_consistency.py H W [dtype] [key=value model fields ...]      (H, W multiples of 8)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import os
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M, runtime as R  # noqa: E402

H, W = int(sys.argv[1]), int(sys.argv[2])
dt = {"bf16": R.DTYPE_BF16, "fp16": R.DTYPE_F16, "fp8": R.DTYPE_FP8}[sys.argv[3] if len(sys.argv) > 3 else "bf16"]
kw = dict(gen_blocks=2)
for a in sys.argv[4:]:
    k, v = a.split("=")
    try:
        kw[k] = int(v)
    except ValueError:
        try:
            kw[k] = float(v)
        except ValueError:
            kw[k] = {"True": True, "False": False}.get(v, v)
CROP, MARGIN = 384, 128
big = M.ModelConfig(frame_height=H, frame_width=W, **kw)
wts = M.make_seeded_weights(big, seed=42)
rng = np.random.default_rng(3)
base = rng.integers(0, 256, size=(2, H, W, 4), dtype=np.uint8)
spots = {"top-left": (0, 0), "centre": ((H - CROP) // 2 // 8 * 8, (W - CROP) // 3 // 8 * 8), "bottom-right": (H - CROP, W - CROP)}


def window(y0, x0):
    """rows / columns of the crop (LR) that are farther than MARGIN from every border the crop does not share with the frame"""
    ya, yb = (0 if y0 == 0 else MARGIN), (CROP if y0 + CROP == H else CROP - MARGIN)
    xa, xb = (0 if x0 == 0 else MARGIN), (CROP if x0 + CROP == W else CROP - MARGIN)
    return ya, yb, xa, xb


t0 = time.time()
rt = R.Runtime(M.serialize(big, wts), 0, dt)
print(f"{H}x{W} {kw}: engine built in {time.time() - t0:.1f} s, resident tower {rt.stat('resident_tower')}")
keep = {k: [] for k in spots}
for f in base:
    t0 = time.time()
    out = rt.process_image(f)
    for k, (y0, x0) in spots.items():
        ya, yb, xa, xb = window(y0, x0)
        keep[k].append(out[4 * (y0 + ya):4 * (y0 + yb), 4 * (x0 + xa):4 * (x0 + xb)].copy())
    print(f"  frame in {time.time() - t0:.2f} s")
    del out
rt.close()
small = M.ModelConfig(frame_height=CROP, frame_width=CROP, **kw)
worst = 0
for k, (y0, x0) in spots.items():
    rs = R.Runtime(M.serialize(small, wts), 0, dt)
    ya, yb, xa, xb = window(y0, x0)
    for t, f in enumerate(base):
        o = rs.process_image(np.ascontiguousarray(f[y0:y0 + CROP, x0:x0 + CROP]))[4 * ya:4 * yb, 4 * xa:4 * xb]
        d = np.abs(o[..., :3].astype(int) - keep[k][t][..., :3].astype(int))
        worst = max(worst, int(d.max()))
        print(f"  {k:12s} frame {t}: max |big - crop| = {d.max()} LSB, differing {np.mean(d > 0):.2e}, mean value {o[..., :3].mean():.1f}")
    rs.close()
print("WORST", worst)
