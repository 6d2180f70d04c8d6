#!/usr/bin/env python3
"""Layer-by-layer comparison of the HIP engine with the float64 oracle on a small
model (developer tool; needs a GPU).  Prints one line per tensor.  Lives under
tests/ because it uses the oracle, which is test infrastructure."""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import (M, O, err, gen_in_to_reference, oracle_config, small_config,  # noqa: E402
                     tail_y_to_reference, u8_stats)
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import runtime as R  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="autoencoder")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--h", type=int, default=30)
    ap.add_argument("--w", type=int, default=48)
    ap.add_argument("--blocks", type=int, default=3)
    a = ap.parse_args()
    pad = 8 if a.arch == "autoencoder" else 0
    cfg = small_config(frame_height=a.h, frame_width=a.w, gen_blocks=a.blocks,
                       flow_arch=a.arch, flow_pad_factor=pad, flow_res_blocks=2)
    wts = M.make_seeded_weights(cfg)
    blob = M.serialize(cfg, wts)
    dt = R.DTYPE_F16 if a.dtype == "fp16" else R.DTYPE_BF16
    rt = R.Runtime(blob, 0, dt)
    ocfg = oracle_config(cfg)
    sess = O.Session(wts, ocfg)
    frames = M.synthetic_frames(a.frames, a.h, a.w, kind="smooth")
    h, w, ph, pw = a.h, a.w, ocfg.padded_height, ocfg.padded_width
    for t in range(a.frames):
        trace = {}
        ref = sess.run(frames[t], trace)
        out = rt.process_image(frames[t])
        print(f"--- frame {t}: u8 {u8_stats(out, ref)}  X==0: {bool((out[..., 3] == 0).all())}")
        names = ["flow_in"] + [k for k in trace if k.startswith("flow/")] + ["flow"]
        for n in names:
            g = rt.read_tensor(n)
            r = trace[n]
            if n == "flow_in":
                g = g.reshape(ph, pw, 16)[..., :r.shape[2]]
            print(f"{n:28s} {err(g.reshape(r.shape), r)}")
        g = gen_in_to_reference(rt.read_tensor("gen_in"), h, w)
        print(f"{'gen_in':28s} {err(g, trace['gen_in_ref'])}")
        print(f"{'trunk':28s} {err(rt.read_tensor('trunk').reshape(h, w, -1), trace['trunk'])}")
        print(f"{'tail_y':28s} {err(tail_y_to_reference(rt.read_tensor('tail_y'), h, w), trace['tail_mid'])}")
        st = rt.read_tensor("state").reshape(4 * h, 4 * w, 4)
        print(f"{'state':28s} {err(st[..., :3], sess.last.output_raw)}  pad=={float(np.abs(st[..., 3]).max())}")
    ms, n, fl = rt.time_steps("", 5)
    print(f"all steps: {n} launches, {ms * n:.3f} ms/frame eager")
    return 0


if __name__ == "__main__":
    sys.exit(main())
