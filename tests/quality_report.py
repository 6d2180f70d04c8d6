#!/usr/bin/env python3
"""Quality of the engine at the FULL benchmark size (PSNR over u8 BGR,
max abs error, share of bytes off by more than 1) against the CPU restatement of the
reference path, on the same synthetic clips the benchmark uses.  Needs a GPU; lives
under tests/ because it uses the oracle.  The fp32 C restatement (oracle/ju_oracle_c.c,
OpenMP) is the comparison here: it agrees with the float64 numpy oracle to < 1e-6 on
output_raw (tests/test_oracle_cross.py) and finishes a 480x270 frame in seconds.

usage: python tests/quality_report.py [--frames 8] [--preset psp-quality] > profiles/<name>.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import M, u8_stats  # noqa: E402
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import runtime as R  # noqa: E402
from oracle.c_binding import CSession  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--preset", default="psp-quality")
    args = ap.parse_args()
    cfg = M.PRESETS[args.preset]
    blob = M.serialize(cfg, M.make_seeded_weights(cfg, seed=42))
    h, w = cfg.frame_height, cfg.frame_width
    report = {"preset": args.preset, "frames": args.frames, "size": f"{w}x{h} -> {4 * w}x{4 * h}",
              "reference": "oracle/ju_oracle_c.c (fp32 CPU restatement of the reference graph)",
              "weights": "seeded random-init (seed 42)", "clips": {}}
    for kind in ("noise", "smooth"):
        clip = M.synthetic_frames(args.frames, h, w, seed=1234, kind=kind)
        ref = CSession(blob, h, w)
        refs = [ref.run(f) for f in clip]
        outs = {}
        for name, dt in (("bf16", R.DTYPE_BF16), ("fp16", R.DTYPE_F16), ("fp8", R.DTYPE_FP8)):
            rt = R.Runtime(blob, 0, dt)
            outs[name] = [rt.process_image(f).copy() for f in clip]
            stats = [u8_stats(o, r) for o, r in zip(outs[name], refs)]
            rt.close()
            report["clips"][f"{kind}/{name}"] = {
                "psnr_db_min": round(min(s["psnr"] for s in stats), 2),
                "psnr_db_mean": round(float(np.mean([s["psnr"] for s in stats])), 2),
                "psnr_db_last_frame": round(stats[-1]["psnr"], 2),
                "max_abs_u8": max(s["max"] for s in stats),
                "bytes_off_by_more_than_1_percent": round(100 * max(s["frac_gt1"] for s in stats), 4),
            }
        # BASELINE.json config 5: the 8-bit tower against the bf16 engine, frame by frame
        vs = [u8_stats(a, b) for a, b in zip(outs["fp8"], outs["bf16"])]
        report["clips"][f"{kind}/fp8_vs_bf16_engine"] = {
            "psnr_db_min": round(min(s["psnr"] for s in vs), 2),
            "psnr_db_mean": round(float(np.mean([s["psnr"] for s in vs])), 2),
            "max_abs_u8": max(s["max"] for s in vs),
            "bytes_off_by_more_than_1_percent": round(100 * max(s["frac_gt1"] for s in vs), 4),
        }
    print(json.dumps(report, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
