import os, sys
import numpy as np
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.init()
    torch.zeros(1, device="cuda").cpu()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from helpers import M, O, oracle_config, small_config
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import runtime as R
a = M.synthetic_frames(1, 30, 48, seed=21, kind="smooth")
b = M.synthetic_frames(1, 30, 48, seed=22, kind="noise")
frames = np.concatenate([a, a, a, b, b, b])
mode = dict(temporal_window=16, temporal_gain=40.0) if len(sys.argv) < 2 else dict(temporal_window=16)
READ = len(sys.argv) > 2
base = dict(temporal_strength=0.5, **mode)
wts = M.make_seeded_weights(small_config(**base))
probe = O.Session(wts, oracle_config(small_config(temporal_threshold=1.0, **base)))
stats = []
for f in frames:
    tr = {}
    probe.run(f, trace=tr)
    stats.append(np.asarray(tr["temporal_mean"], np.float64).ravel())
allm = np.sort(np.concatenate(stats))
lo, hi = int(0.2 * len(allm)), max(int(0.8 * len(allm)), int(0.2 * len(allm)) + 2)
gaps = allm[lo + 1:hi] - allm[lo:hi - 1]
k = lo + int(np.argmax(gaps))
thr = float(np.float32(0.5 * (allm[k] + allm[k + 1])))
print("thr", thr)
cfg = small_config(temporal_threshold=thr, **base)
rt = R.Runtime(M.serialize(cfg, wts), 0, R.DTYPE_F16)
sess = O.Session(wts, oracle_config(cfg))
for t, f in enumerate(frames):
    tr = {}
    ref = sess.run(f, tr)
    out = rt.process_image(f)
    d = np.abs(out[..., :3].astype(int) - ref[..., :3].astype(int)).max(-1)
    print("frame", t, "max", d.max())
    if READ:
        rt.read_tensor("state")
    if t == 3:
        m = tr["temporal_mean"]
        print(np.array2string(m, precision=3, max_line_width=200))
        print(d[:112].reshape(7, 16, 12, 16).max(axis=(1, 3)))
        st = rt.read_tensor("state").reshape(120, 192, 4)[..., :3]
        pw = rt.read_tensor("pre_warp").reshape(120, 192, 4)[..., :3]
        print("state err", np.abs(st - sess.state.pre_gen).max(), "pre_warp err", np.abs(pw - sess.last.pre_warp).max())
        e = np.abs(pw - sess.last.pre_warp).max(-1)
        print(e[:112].reshape(7, 16, 12, 16).max(axis=(1, 3)).round(3))
