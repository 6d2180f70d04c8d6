#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/.

The reference (itmo153277/JoshUpscale) holds no golden vectors and cannot be run
here, so these are outputs of the float64 restatement oracle/ju_oracle.py on
seeded inputs ("parity unpinned").  They pin the oracle against
accidental change, give the C restatement and the HIP engine fixed targets, and
travel to the GPU box as plain data.

    python tests/golden/make_golden.py            # small fixtures (seconds)
    python tests/golden/make_golden.py --full     # + 270x480 fixture (~2 min)
"""

import argparse
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from helpers import M, O, oracle_config, small_config  # noqa: E402

CROPS = [(0, 0), (0, 1856), (1016, 0), (1016, 1856), (508, 928), (300, 1200)]  # 64x64 HR windows


def blob_sha(cfg, wts) -> str:
    return hashlib.sha256(M.serialize(cfg, wts)).hexdigest()


def small(name: str, cfg, n_frames: int, kind: str, seed: int, fp8_tower: bool = False) -> None:
    wts = M.make_seeded_weights(cfg, seed=42)
    sess = O.Session(wts, oracle_config(cfg, fp8_tower=fp8_tower))
    frames = M.synthetic_frames(n_frames, cfg.frame_height, cfg.frame_width, seed=seed, kind=kind)
    outs, raws, flows = [], [], []
    for t in range(n_frames):
        outs.append(sess.run(frames[t]))
        raws.append(sess.last.output_raw.astype(np.float32))
        flows.append(sess.last.flow.astype(np.float32))
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), frames=frames, outputs=np.stack(outs),
        output_raw_last=raws[-1], flow_last=flows[-1], model_sha256=blob_sha(cfg, wts))
    print(name, "ok", os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")


def full() -> None:
    cfg = M.PRESETS["psp-quality"]
    wts = M.make_seeded_weights(cfg, seed=42)
    sess = O.Session(wts, oracle_config(cfg))
    n = 3
    frames = M.synthetic_frames(n, 270, 480, seed=777, kind="smooth")
    crops_u8 = np.zeros((n, len(CROPS), 64, 64, 3), np.uint8)
    crops_raw = np.zeros((n, len(CROPS), 64, 64, 3), np.float32)
    means = np.zeros((n, 3))
    for t in range(n):
        out = sess.run(frames[t])
        for k, (y, x) in enumerate(CROPS):
            crops_u8[t, k] = out[y:y + 64, x:x + 64, :3]
            crops_raw[t, k] = sess.last.output_raw[y:y + 64, x:x + 64]
        means[t] = out[..., :3].reshape(-1, 3).mean(0)
        print("full frame", t, means[t], flush=True)
    np.savez_compressed(
        os.path.join(HERE, "full_psp_quality.npz"), crops=np.array(CROPS), crops_u8=crops_u8,
        crops_raw=crops_raw, means=means, frames_sha256=hashlib.sha256(frames.tobytes()).hexdigest(),
        model_sha256=blob_sha(cfg, wts), seed=777, n_frames=n)
    print("full ok")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    a = ap.parse_args()
    small("small_autoencoder", small_config(), 4, "smooth", 11)
    small("small_resnet", small_config(flow_arch="resnet", flow_pad_factor=0, flow_res_blocks=2,
                                       frame_height=34, frame_width=50), 4, "smooth", 12)
    small("small_noise", small_config(gen_blocks=2), 3, "noise", 13)
    # the 8-bit tower's restatement (BASELINE.json config 5; scheme of csrc/fp8.h)
    small("small_fp8", small_config(gen_blocks=4), 3, "smooth", 14, fp8_tower=True)
    if a.full:
        full()
