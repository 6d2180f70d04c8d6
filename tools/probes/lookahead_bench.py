#!/usr/bin/env python3
"""Frames/s of ju_process_batch by pass length against ju_process (developer probe, needs a GPU).
usage: lookahead_bench.py [preset] [dtype] [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from joshupscale_amd import model_file as M  # noqa: E402
from joshupscale_amd import runtime as R  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "psp-quality"
dtype = {"bf16": R.DTYPE_BF16, "fp16": R.DTYPE_F16, "fp8": R.DTYPE_FP8}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
total = int(sys.argv[3]) if len(sys.argv) > 3 else 960
cfg = M.PRESETS[preset]
h, w = cfg.frame_height, cfg.frame_width
blob = M.serialize(cfg, M.make_seeded_weights(cfg))
frames = M.synthetic_frames(16, h, w, seed=1, kind="noise")
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(frames).to(dev)
d_out = torch.empty((4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
rt = R.Runtime(blob, 0, dtype, hooks=False)
ins = [rt.device_image(d_in[t].data_ptr(), w, h) for t in range(16)]
out = rt.device_image(d_out.data_ptr(), 4 * w, 4 * h)
for i in ins:
    rt.prepare_frames(i, out)


def run(n, count):
    t = 0
    while t < count:
        if n == 1:
            rt.process(ins[t % 16], out)
            t += 1
        else:
            rt.process_batch([ins[(t + i) % 16] for i in range(n)], [out] * n)
            t += n


digests = {}
for rnd in range(2):
    for n in (1, 2, 4, 8, 1, 4, 8, 16):
        rt.reset()
        run(n, 96)            # eager sightings + captures
        t0 = time.perf_counter()
        run(n, total)
        dt = time.perf_counter() - t0
        digests.setdefault(n, set()).add(hash(d_out.cpu().numpy().tobytes()))
        print(f"round {rnd} pass of {n:2d}: {total / dt:8.1f} frames/s  ({dt / total * 1e6:6.1f} us/frame)", flush=True)
print("digest per pass length equal:", len({frozenset(v) for v in digests.values()}) == 1 and all(len(v) == 1 for v in digests.values()))
