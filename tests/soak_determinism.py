#!/usr/bin/env python3
"""Soak: the resident tower's halo exchange is timing-dependent (retries, neighbours
running ahead) but its RESULT must not be.  Runs the same long clip twice on the full
benchmark model and compares every output frame's checksum.  Needs a GPU.
The 8-bit per-layer tower has no exchange, but a missing DMA wait once produced rare
stale tiles there: same check, and with a preset / dtype argument the
second run uses one workgroup per CU (JU_FP8_GRID=256), which must not change a byte.
With a fourth argument (frames per pass) a third run sends the same clip through ju_process_batch as device-resident
look-ahead passes: the digest of its frames must be the frame-by-frame one.
usage: python tests/soak_determinism.py [frames] [preset] [bf16|fp16|fp8] [frames per look-ahead pass]"""
import hashlib
import os
import sys
import time

look = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if look > 1:
    import torch  # (torch's HIP runtime first: the library then shares it)
    torch.zeros(1, device="cuda:0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M, runtime as R  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
preset = sys.argv[2] if len(sys.argv) > 2 else "psp-quality"
dtype = {"bf16": R.DTYPE_BF16, "fp16": R.DTYPE_F16, "fp8": R.DTYPE_FP8}[sys.argv[3] if len(sys.argv) > 3 else "bf16"]
cfg = M.PRESETS[preset]
blob = M.serialize(cfg, M.make_seeded_weights(cfg))
clip = M.synthetic_frames(16, cfg.frame_height, cfg.frame_width, seed=7, kind="smooth")


def run(grid=None):
    if grid:
        os.environ["JU_FP8_GRID"] = str(grid)
        os.environ["JU_NO_GRAPH"] = "1"   # the override acts on new launches only
    rt = R.Runtime(blob, 0, dtype)
    h = hashlib.sha256()
    t0 = time.perf_counter()
    for i in range(n):
        h.update(rt.process_image(clip[i % 16]).tobytes()[:: 4099])   # sparse sample of every frame
    dt = time.perf_counter() - t0
    rt.close()
    return h.hexdigest(), dt


def run_passes():
    hh, ww = cfg.frame_height, cfg.frame_width
    d_in = torch.from_numpy(clip).to("cuda:0")
    d_out = torch.zeros((look, 4 * hh, 4 * ww, 4), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    rt = R.Runtime(blob, 0, dtype)
    ins = [rt.device_image(d_in[k].data_ptr(), ww, hh) for k in range(16)]
    outs = [rt.device_image(d_out[k].data_ptr(), 4 * ww, 4 * hh) for k in range(look)]
    h = hashlib.sha256()
    t0 = time.perf_counter()
    i = 0
    while i < n:
        k = min(look, n - i)
        rt.process_batch([ins[(i + j) % 16] for j in range(k)], outs[:k])
        got = d_out[:k].cpu().numpy()
        for j in range(k):
            h.update(got[j].tobytes()[:: 4099])
        i += k
    dt = time.perf_counter() - t0
    took = rt.stat("lookahead_frames")
    rt.close()
    return h.hexdigest(), dt, took


a, ta = run()
b, tb = run(256 if dtype == R.DTYPE_FP8 else None)
print(f"{n} frames twice: {ta:.1f} s / {tb:.1f} s (host frames), digests {'EQUAL' if a == b else 'DIFFER'}: {a[:16]} {b[:16]}")
ok = a == b
if look > 1:
    os.environ.pop("JU_FP8_GRID", None)
    os.environ.pop("JU_NO_GRAPH", None)
    c, tc, took = run_passes()
    print(f"{n} frames as look-ahead passes of {look} (device frames, {int(took)} of them in passes): {tc:.1f} s, digest "
          f"{'EQUAL' if c == a else 'DIFFERS'}: {c[:16]}")
    ok = ok and c == a and took >= n - look
sys.exit(0 if ok else 1)
