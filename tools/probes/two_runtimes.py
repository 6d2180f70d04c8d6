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
def loop(rt, n):
    ins = [rt.device_image(d_in[i].data_ptr(), w, h) for i in range(16)]
    out = rt.device_image(d_out.data_ptr(), 4 * w, 4 * h)
    for i in range(16): rt.prepare_frames(ins[i], out)
    for i in range(300): rt.process(ins[i % 16], out)
    t0 = time.perf_counter()
    for i in range(n): rt.process(ins[i % 16], out)
    return (time.perf_counter() - t0) / n * 1e6
a = R.Runtime(blob, 0, R.DTYPE_BF16, hooks=True)
print("A (test lib): frame %.1f us" % loop(a, 300))
print("A tower@frame %.1f us, back-to-back %.1f" % (a.time_steps("tower@frame", 20)[0] * 1e3, a.time_steps("tower", 20)[0] * 1e3))
b = R.Runtime(blob, 0, R.DTYPE_BF16, hooks=True)
b.time_steps("tower@frame", 300)
print("B (second runtime, same lib, A alive) tower@frame %.1f us" % (b.time_steps("tower@frame", 20)[0] * 1e3))
print("A again tower@frame %.1f us" % (a.time_steps("tower@frame", 20)[0] * 1e3))
a.close()
print("B after A closed tower@frame %.1f us" % (b.time_steps("tower@frame", 20)[0] * 1e3))
print("B frame loop %.1f us" % loop(b, 300))
print("B tower@frame %.1f us" % (b.time_steps("tower@frame", 20)[0] * 1e3))
c = R.Runtime(blob, 0, R.DTYPE_BF16, hooks=False)
print("C (product lib, B alive) frame loop %.1f us" % loop(c, 300))
print("B tower@frame after C's loop %.1f us" % (b.time_steps("tower@frame", 20)[0] * 1e3))
