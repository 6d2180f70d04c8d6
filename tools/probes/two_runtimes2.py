"""Why does ju_time_steps("tower@frame") read ~8 % shorter while a second runtime is alive (two_runtimes.py, confirmed by
a kernel trace: the kernel itself runs 320 instead of 350 us)?  Variants: a dummy HIP stream beside ONE runtime; eager
frame loops (JU_NO_GRAPH=1) with one and with two runtimes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["psp-quality"]
blob = M.serialize(cfg, M.make_seeded_weights(cfg, seed=42))
dev = torch.device("cuda", 0)
h, w = cfg.frame_height, cfg.frame_width
clip = M.synthetic_frames(16, h, w, seed=1234, kind="noise")
d_in = torch.from_numpy(clip).to(dev)
d_out = torch.empty((4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
def loop(rt, n, prepare=True):
    ins = [rt.device_image(d_in[i].data_ptr(), w, h) for i in range(16)]
    out = rt.device_image(d_out.data_ptr(), 4 * w, 4 * h)
    if prepare:
        for i in range(16): rt.prepare_frames(ins[i], out)
    for i in range(300): rt.process(ins[i % 16], out)
    t0 = time.perf_counter()
    for i in range(n): rt.process(ins[i % 16], out)
    return (time.perf_counter() - t0) / n * 1e6
def tf(rt):
    rt.time_steps("tower@frame", 200)
    return rt.time_steps("tower@frame", 20)[0] * 1e3
eager = os.environ.get("JU_NO_GRAPH") == "1"
print("JU_NO_GRAPH =", os.environ.get("JU_NO_GRAPH"))
a = R.Runtime(blob, 0, R.DTYPE_BF16, hooks=True)
print("A alone: frame %.1f us (%s), tower@frame %.1f us" % (loop(a, 300), "eager" if eager else "graph", tf(a)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    x = torch.zeros(1024, device=dev) + 1
torch.cuda.synchronize()
print("A + a dummy torch stream: frame %.1f us, tower@frame %.1f us" % (loop(a, 300, False), tf(a)))
b = R.Runtime(blob, 0, R.DTYPE_BF16, hooks=True)
print("A with B alive: frame %.1f us, tower@frame %.1f us" % (loop(a, 300, False), tf(a)))
print("B with A alive: frame %.1f us, tower@frame %.1f us" % (loop(b, 300), tf(b)))
a.close()
print("B alone: frame %.1f us, tower@frame %.1f us" % (loop(b, 300, False), tf(b)))
