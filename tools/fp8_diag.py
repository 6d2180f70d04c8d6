#!/usr/bin/env python3
"""8-bit tower against the fp16 engine on the same frames (developer tool, needs a GPU):
where the trunk differs most, by tile / row / column / channel.  Found the stale-tile race
described in DESIGN.md section 4b.   usage: tools/fp8_diag.py H W blocks [frames]"""
import sys, os
os.environ.setdefault("JU_TAIL", "fused")  # the 16-bit engine's trunk tensor is read below: keep the tail a launch of its own
sys.path.insert(0, os.getcwd())
import numpy as np
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M, runtime as R
h, w = int(sys.argv[1]), int(sys.argv[2]); blocks = int(sys.argv[3]); nfr = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cfg = M.ModelConfig(frame_height=h, frame_width=w, gen_blocks=blocks)
blob = M.serialize(cfg, M.make_seeded_weights(cfg))
frames = M.synthetic_frames(nfr, h, w, seed=1234, kind="noise")
r8 = R.Runtime(blob, 0, R.DTYPE_FP8); r16 = R.Runtime(blob, 0, R.DTYPE_F16)
for f in frames:
    a = r8.process_image(f).astype(np.int32); b = r16.process_image(f).astype(np.int32)
    t8 = r8.read_tensor("trunk").reshape(h, w, 64); t16 = r16.read_tensor("trunk").reshape(h, w, 64)
    d = np.abs(a - b)[..., :3].max(axis=2)
    dt = np.abs(t8 - t16).max(axis=2)
    print("u8 max", d.max(), "psnr", 10*np.log10(255**2/np.mean((a-b)[..., :3].astype(float)**2)), "trunk max diff", dt.max(), "trunk absmax", np.abs(t16).max())
    ys, xs = np.nonzero(dt > 0.5 * dt.max())
    print("  worst trunk px (y,x) tile(ty,tx):", [(int(y), int(x), int(y)//8, int(x)//32) for y, x in list(zip(ys, xs))[:12]], "count", len(ys))
    big = np.nonzero(dt > 1.0)
    print("  image rows with diff > 1:", sorted(set(big[0].tolist()))[:40])
    print("  px with trunk diff > 1.0:", len(big[0]), "rows", sorted(set((big[0]//8).tolist()))[:20], "cols", sorted(set((big[1]//32).tolist()))[:20])
    d3 = np.abs(t8 - t16)
    ys, xs, cs = np.nonzero(d3 > 0.8)
    import collections
    print("  by (y%8): ", sorted(collections.Counter((ys % 8).tolist()).items()))
    print("  by (x%32):", sorted(collections.Counter((xs % 32).tolist()).items()))
    print("  by chan//8:", sorted(collections.Counter((cs // 8).tolist()).items()))
    print("  by chan%8:", sorted(collections.Counter((cs % 8).tolist()).items()))
    print("  first 10:", list(zip(ys.tolist(), xs.tolist(), cs.tolist()))[:10])
