"""Shared helpers of the parity tests (test infrastructure)."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from joshupscale_amd import model_file as M  # noqa: E402
from oracle import ju_oracle as O  # noqa: E402


def oracle_config(cfg: M.ModelConfig, fp8_tower: bool = False) -> O.ModelConfig:
    """Oracle configuration equal to a container configuration (bn_eps goes
    through float32 exactly as it does in the file header)."""
    return O.ModelConfig(
        frame_height=cfg.frame_height, frame_width=cfg.frame_width,
        num_flow_inputs=cfg.num_flow_inputs, flow_arch=cfg.flow_arch,
        flow_filters=tuple(cfg.flow_filters), flow_res_filters=cfg.flow_res_filters,
        flow_res_blocks=cfg.flow_res_blocks, flow_pad_factor=cfg.flow_pad_factor,
        gen_filters=cfg.gen_filters, gen_blocks=cfg.gen_blocks,
        normalize_brightness=cfg.normalize_brightness,
        bn_eps=float(np.float32(cfg.bn_eps)),
        temporal_strength=float(np.float32(cfg.temporal_strength)),
        temporal_threshold=float(np.float32(cfg.temporal_threshold)),
        fp8_tower=fp8_tower,
        temporal_window=cfg.temporal_window, temporal_gain=float(np.float32(cfg.temporal_gain)),
        temporal_norm=cfg.temporal_norm, temporal_limit=cfg.temporal_limit,
        temporal_luma=cfg.temporal_luma,
        flow_activation=cfg.flow_activation, gen_activation=cfg.gen_activation,
        flow_negative_slope=float(np.float32(cfg.flow_negative_slope)),
        gen_negative_slope=float(np.float32(cfg.gen_negative_slope)))


def small_config(**kw) -> M.ModelConfig:
    """A geometry the float64 oracle steps in well under a second: 30x48 pads to
    32x48 like 270x480 pads to 272x480, and exercises partial MFMA tiles."""
    base = dict(frame_height=30, frame_width=48, gen_blocks=3)
    base.update(kw)
    return M.ModelConfig(**base)


def psnr_u8(a: np.ndarray, b: np.ndarray) -> float:
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    mse = np.mean(d * d)
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def u8_stats(a: np.ndarray, b: np.ndarray) -> dict:
    d = np.abs(a[..., :3].astype(np.int32) - b[..., :3].astype(np.int32))
    return {"psnr": psnr_u8(a, b), "max": int(d.max()),
            "frac_gt1": float(np.mean(d > 1))}


def gen_in_to_reference(packed: np.ndarray, h: int, w: int) -> np.ndarray:
    """Engine generator-input record [H, W, 64] -> reference order [H, W, 51]
    (LR frame, then space_to_depth(pre_warp))."""
    p = packed.reshape(h, w, 64)
    out = np.empty((h, w, 51), p.dtype)
    out[..., 0:3] = p[..., 12:15]
    for i in range(4):
        for j in range(4):
            out[..., 3 + (i * 4 + j) * 3:3 + (i * 4 + j) * 3 + 3] = \
                p[..., i * 16 + j * 3:i * 16 + j * 3 + 3]
    return out


def tail_y_to_reference(y: np.ndarray, h: int, w: int) -> np.ndarray:
    """Engine tail_y [H, W, (a*2+b)*32+o] -> reference [2H, 2W, 32]."""
    return y.reshape(h, w, 2, 2, 32).transpose(0, 2, 1, 3, 4).reshape(2 * h, 2 * w, 32)


def err(a: np.ndarray, b: np.ndarray) -> dict:
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    d = np.abs(a - b)
    scale = max(np.abs(b).max(), 1e-12)
    return {"max_abs": float(d.max()), "rel_to_max": float(d.max() / scale),
            "rms": float(np.sqrt(np.mean(d * d))), "ref_absmax": float(scale)}
