#!/usr/bin/env python3
"""Per-REGION view of the resident tower's phase profile (diagnostic variant 4): who waits for whom.  The sweep's wait is
where faster regions absorb the pace of slower neighbours, so the regions with the SHORTEST halo fill are the ones that
set the pace of the whole grid.  Prints, per XCD (the kernel deals each XCD a contiguous run of regions) and per
region row / column, the median cycles per layer of K loops, pre-run, fill and publish.  Needs a GPU."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("JU_TEST_HOOKS", "1")
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["psp-quality"]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_BF16)
lib = R.load_library()
lib.ju_debug_set(b"tower_variant", 4)
ms, n, fl = rt.time_steps("tower", 3)
raw = rt.read_tensor("tower_profile").view(np.uint64).reshape(256, 4, 8)[:255].astype(np.float64) / 49
lib.ju_debug_set(b"tower_variant", 0)
GX, GY = 15, 17
names = ["fill", "wissue", "compute", "barrier", "publish", "Kloops", "epi", "preK"]
w0 = raw[:, 0, :]
tot = w0[:, :5].sum(axis=1)
print(f"diagnostic launch {ms*1e3:.0f} us; per-layer cycles, wave 0; all regions: total {np.median(tot):.0f} (min {tot.min():.0f} max {tot.max():.0f})")
def line(tag, idx):
    v = w0[idx]
    print(f"{tag:>10s} n={len(idx):3d}  " + "  ".join(f"{nm} {np.median(v[:, k]):6.0f}" for k, nm in enumerate(names)) + f"   fill min {v[:,0].min():6.0f} max {v[:,0].max():6.0f}")
# XCD of a region: XCD x holds count_x = ceil((255 - x) / 8) consecutive regions
start, xcd = 0, np.zeros(255, int)
for x in range(8):
    c = (255 - x + 7) // 8
    xcd[start:start + c] = x
    start += c
for x in range(8):
    line(f"xcd {x}", np.where(xcd == x)[0])
for gy in range(GY):
    line(f"row {gy}", np.arange(gy * GX, (gy + 1) * GX))
for gx in range(GX):
    line(f"col {gx}", np.arange(gx, 255, GX))
order = np.argsort(w0[:, 0])
print("regions with the shortest fill (pace setters):", [(int(r), int(r) // GX, int(r) % GX, int(xcd[r]), int(w0[r, 0]), int(w0[r, 5] + w0[r, 7])) for r in order[:12]])
print("regions with the longest fill:", [(int(r), int(r) // GX, int(r) % GX, int(xcd[r]), int(w0[r, 0]), int(w0[r, 5] + w0[r, 7])) for r in order[-8:]])
