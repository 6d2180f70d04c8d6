#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/.

The reference (itmo153277/JoshUpscale) holds no golden vectors and cannot be run
here, so these are outputs of the float64 restatement oracle/ju_oracle.py on
seeded inputs ("parity unpinned").  They pin the oracle against
accidental change, give the C restatement and the HIP engine fixed targets, and
travel to the GPU box as plain data.

    python tests/golden/make_golden.py            # small fixtures (seconds)
    python tests/golden/make_golden.py --full     # + the four full-size presets (~15 min)
    python tests/golden/make_golden.py --preset psp-fast   # one full-size preset only
    python tests/golden/make_golden.py --preset psp-quality --fp8   # its 8-bit-tower fixture
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


FULL_PRESETS = {          # preset -> (golden file, frames)
    "psp-quality": ("full_psp_quality", 3),
    "psp-fast": ("full_psp_fast", 3),
    "psp-quality-flowres": ("full_psp_quality_flowres", 3),
    "ps2-quality": ("full_ps2_quality", 2),
}


def crops_for(h: int, w: int):
    """Six 64x64 HR windows: the four corners, the centre and one off-centre spot."""
    H, W = 4 * h, 4 * w
    return [(0, 0), (0, W - 64), (H - 64, 0), (H - 64, W - 64), (H // 2 - 32, W // 2 - 32),
            (H * 5 // 18 // 4 * 4, W * 5 // 8 // 4 * 4)]


# the 8-bit tower (BASELINE.json config 5) at the sizes it is quoted on: same clip, same crops
FULL_FP8 = {"psp-quality": ("full_psp_quality_fp8", 3), "ps2-quality": ("full_ps2_quality_fp8", 2)}  # (frame counts of the float fixtures: the smooth clip depends on it)


def full(preset: str = "psp-quality", fp8_tower: bool = False) -> None:
    """Full-size fixture of one preset, generated TWICE: by the numpy float64 oracle and
    by the independent PyTorch restatement (tests/torch_restatement.py).  Both whole-frame
    SHA-256 digests are stored (the only independent anchor available: the reference ships
    no vectors and cannot run here); the crops and means come from the numpy oracle.
    fp8_tower: the restatements of the 8-bit tower scheme (csrc/fp8.h) instead."""
    from torch_restatement import TorchSession
    name, n = (FULL_FP8 if fp8_tower else FULL_PRESETS)[preset]
    cfg = M.PRESETS[preset]
    h, w = cfg.frame_height, cfg.frame_width
    wts = M.make_seeded_weights(cfg, seed=42)
    sess = O.Session(wts, oracle_config(cfg, fp8_tower=fp8_tower))
    tsess = TorchSession(wts, oracle_config(cfg, fp8_tower=fp8_tower))
    frames = M.synthetic_frames(n, h, w, seed=777, kind="smooth")
    crops = crops_for(h, w)
    assert preset != "psp-quality" or crops == CROPS
    crops_u8 = np.zeros((n, len(crops), 64, 64, 3), np.uint8)
    crops_raw = np.zeros((n, len(crops), 64, 64, 3), np.float32)
    means = np.zeros((n, 3))
    sha_np, sha_torch, n_diff, raw_diff = [], [], [], []
    for t in range(n):
        out = sess.run(frames[t])
        tout = tsess.run(frames[t])
        traw = tsess.output_raw[0].permute(1, 2, 0).numpy()
        sha_np.append(hashlib.sha256(out.tobytes()).hexdigest())
        sha_torch.append(hashlib.sha256(tout.tobytes()).hexdigest())
        d = np.abs(out.astype(np.int32) - tout.astype(np.int32))
        assert d.max() <= 1, "the two restatements disagree by more than a truncation boundary"
        n_diff.append(int((d > 0).sum()))
        raw_diff.append(float(np.abs(traw - sess.last.output_raw).max()))
        for k, (y, x) in enumerate(crops):
            crops_u8[t, k] = out[y:y + 64, x:x + 64, :3]
            crops_raw[t, k] = sess.last.output_raw[y:y + 64, x:x + 64]
        means[t] = out[..., :3].reshape(-1, 3).mean(0)
        print(preset, "frame", t, means[t], "bytes differing numpy/torch:", n_diff[-1],
              "max |output_raw| diff:", raw_diff[-1], flush=True)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), crops=np.array(crops), crops_u8=crops_u8,
        crops_raw=crops_raw, means=means, frames_sha256=hashlib.sha256(frames.tobytes()).hexdigest(),
        model_sha256=blob_sha(cfg, wts), seed=777, n_frames=n,
        out_sha256_numpy=np.array(sha_np), out_sha256_torch=np.array(sha_torch),
        bytes_differing=np.array(n_diff), raw_max_diff=np.array(raw_diff))
    print(preset, "ok")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="every full-size preset (~15 min)")
    ap.add_argument("--preset", action="append", choices=sorted(FULL_PRESETS),
                    help="only this full-size preset (repeatable); skips the small fixtures")
    ap.add_argument("--only", action="append", help="only this small fixture (repeatable)")
    ap.add_argument("--fp8", action="store_true", help="with --preset / --full: the 8-bit tower fixtures")
    a = ap.parse_args()
    if a.preset:
        for p in a.preset:
            full(p, fp8_tower=a.fp8)
        sys.exit(0)
    smalls = {
        "small_autoencoder": (small_config(), 4, "smooth", 11, False),
        "small_resnet": (small_config(flow_arch="resnet", flow_pad_factor=0, flow_res_blocks=2,
                                      frame_height=34, frame_width=50), 4, "smooth", 12, False),
        "small_noise": (small_config(gen_blocks=2), 3, "noise", 13, False),
        # the 8-bit tower's restatement (BASELINE.json config 5; scheme of csrc/fp8.h)
        "small_fp8": (small_config(gen_blocks=4), 3, "smooth", 14, True),
        # `activation: lrelu` models (reference models.py:24-27): both sub-models leaky
        "small_lrelu": (small_config(flow_activation="lrelu", gen_activation="lrelu",
                                     gen_negative_slope=0.2), 4, "smooth", 15, False),
        # the 8-bit tower of a LeakyReLU generator (round 3: e4m3 clamped on both sides)
        "small_fp8_lrelu": (small_config(gen_blocks=3, gen_activation="lrelu", gen_negative_slope=0.2), 3, "smooth", 16, True),
    }
    for name, (cfg, n, kind, seed, fp8) in smalls.items():
        if a.only is None or name in a.only:
            small(name, cfg, n, kind, seed, fp8_tower=fp8)
    if a.full:
        for p in FULL_PRESETS:
            full(p)
        for p in FULL_FP8:
            full(p, fp8_tower=True)
