#!/usr/bin/env python3
"""Where a synchronous frame's time goes beyond its kernels (developer tool, needs a GPU): CPU cost
of one submission (ju_enqueue returns after hipGraphLaunch), frame time through the synchronous
boundary (ju_process), and the back-to-back rate of the same graphs with the stream kept full
(ju_enqueue x N, one ju_synchronize) -- the GPU-side floor of a frame."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M, runtime as R  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "psp-quality"
cfg = M.PRESETS[preset]
h, w = cfg.frame_height, cfg.frame_width
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_BF16)
dev = torch.device("cuda", 0)
clip = torch.from_numpy(M.synthetic_frames(16, h, w, seed=1234, kind="noise")).to(dev)
out = torch.empty((4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
ins = [rt.device_image(clip[i].data_ptr(), w, h) for i in range(16)]
o = rt.device_image(out.data_ptr(), 4 * w, 4 * h)
for i in ins:
    rt.prepare_frames(i, o)
for i in range(400):
    rt.process(ins[i % 16], o)
N = 600
t0 = time.perf_counter()
for i in range(N):
    rt.process(ins[i % 16], o)
sync_us = (time.perf_counter() - t0) / N * 1e6
sub = []
t0 = time.perf_counter()
for i in range(N):
    t1 = time.perf_counter()
    rt.enqueue(ins[i % 16], o)
    sub.append(time.perf_counter() - t1)
rt.synchronize()
async_us = (time.perf_counter() - t0) / N * 1e6
sub.sort()
print(f"{preset}: synchronous ju_process {sync_us:.1f} us/frame ({1e6 / sync_us:.0f} fps); stream kept full (ju_enqueue) "
      f"{async_us:.1f} us/frame ({1e6 / async_us:.0f} fps); CPU cost of one submission p50 {sub[N // 2] * 1e6:.1f} us, "
      f"p90 {sub[int(N * .9)] * 1e6:.1f} us; gap per synchronous frame {sync_us - async_us:.1f} us")
