"""Shared pieces of the GPU parity tests (test infrastructure).

Tolerances, stated here as north_star asks.  The engine computes convolutions with fp16
or bf16 MFMA operands and fp32 accumulation and stores activations in that 16-bit type;
the recurrent HR state is fp16.  Against the float64 oracle (or its float32 C
restatement) on the u8 output (B,G,R bytes; X must be 0):

    fp16:  PSNR >= 61 dB, max |diff| <= 1 LSB
    bf16:  PSNR >= 53 dB, max |diff| <= 3 LSB, <= 0.01 % of bytes off by more than 1
           on the smooth clips; on the benchmark's NOISE clip (uniform random bytes: every LR
           pixel is an edge, the tower's activations are larger and the 8-bit mantissa of its
           16-bit stream costs more) <= 0.035 % of bytes off by more than 1 -- measured at
           full size 0.0231 % (psp-quality) / 0.0223 % (ps2-quality), profiles/r02_quality_*.json;
           the bound is 1.4 x the worst measured, like every other one here

Large flows have a bound of their own (TOL_LARGE_FLOW; test_large_flows_reach_across_and_beyond_the_frame: flow heads
scaled to 14 and 68 HR pixels on small frames).  The f16 flow head resolves 2^-6 / 2^-5 px at |flow| in [16, 32) / [32, 64),
and a sample fetched that far away is an unrelated pixel, so the head's storage rounding becomes interpolation-weight
error on EVERY output pixel instead of on the few that move: fp16 PSNR >= 59 dB, bf16 >= 52 dB, max <= 3 LSB, <= 0.08 %
of bytes off by more than 1 (measured, profiles/r05_parity_stats.json: fp16 62.9-73.5 dB, bf16 55.5-64.4 dB, 2 LSB,
0.055 % -- the suite's worst frac_gt1, which round 5 left unstated; the bound is 1.4 x it).

and on internal tensors (max abs error): flow head 0.003 / 0.02 HR pixels (the head is
stored as f16: up to 2^-9 px of storage rounding at |flow| in [2, 4)), output_raw and the
generator input 0.001 / 0.007 (fp16 / bf16).

Measured on MI355X over the 216 comparisons of this suite (profiles/r02_parity_stats.json;
round 1: profiles/r01_g_quality.json): worst case fp16 63.9 dB, max 1 LSB; bf16 56.0 dB,
max 2 LSB, 0.0014 % of bytes off by more than 1 (every full-size preset: bf16 56.0-59.9
dB, fp16 64.9-68.8 dB; small models 60.4-72.1 dB); flow 0.0018 / 0.0128, output_raw
0.0007 / 0.0049.  The bounds sit 3 dB (a factor 1.4) from the worst measured case, so a
3 dB regression fails.  (The truncating float->u8 cast of
the reference, cuda_convert.cc.cu:76-81, turns any sub-LSB difference at an integer
boundary into 1 LSB.)  Byte-level paths (staging, strides, X byte, state reset, graph
replay, device-direct frames) are bit-exact.
"""

import json
import os

import numpy as np

from helpers import M, ROOT, u8_stats
from joshupscale_amd import runtime as R

TOL = {
    R.DTYPE_F16: dict(psnr=61.0, max=1, frac=0.0, flow=0.003, raw=0.001),
    R.DTYPE_BF16: dict(psnr=53.0, max=3, frac=0.0001, frac_noise=0.00035, flow=0.02, raw=0.007),
}
# flows of tens of HR pixels (see the header): the only case with a looser u8 bound than TOL
TOL_LARGE_FLOW = {
    R.DTYPE_F16: dict(psnr=59.0, max=3, frac=0.0008),
    R.DTYPE_BF16: dict(psnr=52.0, max=3, frac=0.0008),
}
GOLD = os.path.join(ROOT, "tests", "golden")

# every comparison's measured numbers, written to gpurun_out/parity_stats.json at the end
# of the session (conftest.py): the evidence the tolerances above are set against
STATS = []


def record(what, dtype, st):
    STATS.append({"what": [str(x) for x in (what if isinstance(what, (tuple, list)) else (what,))],
                  "dtype": R.DTYPE_NAMES.get(dtype, str(dtype)),
                  **{k: (float(v) if not isinstance(v, (int, str)) else v) for k, v in st.items()}})


def dump_stats():
    if not STATS:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    worst = {}
    for s in STATS:
        w = worst.setdefault(s["dtype"], {"psnr_min": 1e9, "max": 0, "frac_gt1_max": 0.0, "n": 0})
        if "psnr" in s:
            w["psnr_min"] = min(w["psnr_min"], s["psnr"])
            w["max"] = max(w["max"], s["max"])
            w["frac_gt1_max"] = max(w["frac_gt1_max"], s["frac_gt1"])
            w["n"] += 1
    with open(os.path.join(out, "parity_stats.json"), "w") as f:
        json.dump({"worst": worst, "comparisons": STATS}, f, indent=1)


def check_u8(out, ref, dtype, what="", clip="smooth"):
    """clip="noise": the benchmark's uniform-random clip (its own bf16 bound on the share of
    bytes off by more than 1, see the header)."""
    st = u8_stats(out, ref)
    tol = TOL[dtype]
    frac = tol.get("frac_noise", tol["frac"]) if clip == "noise" else tol["frac"]
    record(what, dtype, st)
    assert (out[..., 3] == 0).all(), "X byte must be written as 0"
    assert st["psnr"] >= tol["psnr"] and st["max"] <= tol["max"] and st["frac_gt1"] <= frac, \
        (what, st)
    return st


def make(cfg, dtype, seed=42):
    wts = M.make_seeded_weights(cfg, seed=seed)
    blob = M.serialize(cfg, wts)
    return wts, blob, R.Runtime(blob, 0, dtype)
