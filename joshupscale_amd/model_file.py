"""The engine's model container (``.jupw``) and the seeded synthetic models.

The reference deploys a serialized TensorRT engine that bakes geometry, dtype
and weights together (core/src/tensorrt_backend.cc:117-148; the optional
reindex trailer of scripts/inference/tensorrt/build_engine.py:295-305 is
TensorRT-specific).  This engine has no TensorRT, so it defines its own file:
a fixed little-endian header with the hyper-parameters of
``get_inference_model`` / ``get_generator_resnet`` / ``get_flow_autoencoder`` /
``get_flow_resnet`` (scripts/training/models.py:257-263, 334-339, 484-491,
680-688) followed by the *unfolded* Keras variables in their Keras layouts and
under their Keras layer names, all float32.  BatchNorm folding and the
kernel-ready layout transforms are done by the C++ loader
(csrc/model.cpp), so a ``.weights.h5`` importer only has to copy tensors.

Layout (all little endian):

    0   char[8]  magic  "JUPWGT\\x00\\x01"
    8   u32      version (1)
    12  u32      header_bytes (offset of the tensor table)
    16  u32      frame_height, frame_width, scale(=4), num_flow_inputs
    32  u32      flow_arch (0 autoencoder, 1 resnet), flow_pad_factor,
                 normalize_brightness, gen_filters
    48  u32      gen_blocks, flow_res_filters, flow_res_blocks, n_flow_filters
    64  u32[8]   flow_filters
    96  f32      bn_eps
    100 u32      compute_dtype hint (0 fp16, 1 bf16, 2 fp8 = e4m3 block convolutions over fp16)
    104 u32      n_tensors
    108 f32      temporal_strength (0 = filter off), temporal_threshold
    116 u32      activations: bits 0-7 flow net, bits 8-15 generator (0 relu, 1 lrelu)
    120 f32      flow negative_slope, generator negative_slope (read only for lrelu)
                                                    -> header_bytes = 128
    -- only when the temporal filter uses a non-default mode, header_bytes = 160: --
    128 u32      temporal_window (HR pixels, 0 = global gate)
    132 f32      temporal_gain (0 = sign gate, else tanh(gain * (m - t)))
    136 u32      temporal flags: bit 0 L2 norm, bit 1 limit pre_warp, bit 2 luma weighting
    140 u32      reserved[5]
    table: n_tensors x { char name[92]; u32 ndim; u32 dims[4]; u64 offset;
                         u64 count }                (128 bytes each)
    data:  float32, each tensor 64-byte aligned, offsets from file start
"""

from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, Tuple

import numpy as np

MAGIC = b"JUPWGT\x00\x01"
VERSION = 1
HEADER_BYTES = 128
ENTRY_BYTES = 128
DTYPE_F16 = 0
DTYPE_BF16 = 1
DTYPE_FP8 = 2  # e4m3 block convolutions over fp16 (csrc/fp8.h)

FLOW_ARCH = {"autoencoder": 0, "resnet": 1}
FLOW_ARCH_INV = {v: k for k, v in FLOW_ARCH.items()}
# ACTIVATIONS of scripts/training/models.py:24-27
ACTIVATION = {"relu": 0, "lrelu": 1}
ACTIVATION_INV = {v: k for k, v in ACTIVATION.items()}
# keras.layers.LeakyReLU() default (Keras 3, tensorflow-cpu==2.18.0 of the reference's
# requirements); a config may override it: {"name": "lrelu", "negative_slope": 0.2}
DEFAULT_NEGATIVE_SLOPE = 0.3


@dataclass
class ModelConfig:
    """Mirror of the hyper-parameters in the container header."""
    frame_height: int = 270
    frame_width: int = 480
    num_flow_inputs: int = 4
    flow_arch: str = "autoencoder"
    flow_filters: Tuple[int, ...] = (32, 64, 128, 256, 128, 64, 32)
    flow_res_filters: int = 64
    flow_res_blocks: int = 10
    flow_pad_factor: int = 8
    gen_filters: int = 64
    gen_blocks: int = 24
    normalize_brightness: bool = False
    bn_eps: float = 1e-3
    compute_dtype: int = DTYPE_BF16
    # Temporal moving-average output filter with a global scene-cut gate, the default
    # mode (window 0, sign gate, L1) of scripts/inference/onnx/frame_moving_avg.py:
    # out = pre_warp*(s/2)(1-c) + gen*(1 - s/2 + c*s/2), c = sign(mean|gen-pre_warp| - t).
    # strength 0 = off (the plain model).
    temporal_strength: float = 0.0
    temporal_threshold: float = 0.1
    # `activation` of get_flow_* / get_generator_resnet (models.py:261, 337, 489):
    # "relu" (the default) or "lrelu" with its negative_slope
    flow_activation: str = "relu"
    gen_activation: str = "relu"
    flow_negative_slope: float = DEFAULT_NEGATIVE_SLOPE
    gen_negative_slope: float = DEFAULT_NEGATIVE_SLOPE
    # the other switches of frame_moving_avg.py (:99-110): --window, --gain, --norm,
    # --limit, --luma-normalize (defaults = that script's defaults)
    temporal_window: int = 0
    temporal_gain: float = 0.0
    temporal_norm: str = "L1"
    temporal_limit: bool = False
    temporal_luma: bool = False

    @property
    def temporal_extended(self) -> bool:
        return self.temporal_strength > 0 and (
            self.temporal_window != 0 or self.temporal_gain != 0.0 or self.temporal_norm != "L1"
            or self.temporal_limit or self.temporal_luma)

    @property
    def padded_height(self) -> int:
        f = self.flow_pad_factor
        return self.frame_height if not f else (self.frame_height + f - 1) // f * f

    @property
    def padded_width(self) -> int:
        f = self.flow_pad_factor
        return self.frame_width if not f else (self.frame_width + f - 1) // f * f


# The named configurations of BASELINE.json / SURVEY.md section 8 (assumptions:
# the reference ships no model configs, so "quality" := constructor defaults,
# "fast" := fewer generator blocks, "PS2" := another geometry).
PRESETS: Dict[str, ModelConfig] = {
    "psp-quality": ModelConfig(),
    "psp-fast": ModelConfig(gen_blocks=8, compute_dtype=DTYPE_F16),
    "ps2-quality": ModelConfig(frame_height=448, frame_width=640),  # BASELINE config 5 runs it with dtype fp8
    "psp-quality-flowres": ModelConfig(flow_arch="resnet", flow_pad_factor=0),
    # `activation: lrelu` in both sub-models (reference models.py:24-27; the INT8 quantiser of the
    # deployed graphs lists LeakyRelu, quantize_int8.py:177-178), keras' default slope
    "psp-quality-lrelu": ModelConfig(flow_activation="lrelu", gen_activation="lrelu"),
}


def _bn(rng: np.random.Generator, c: int) -> Dict[str, np.ndarray]:
    return {
        "gamma": rng.uniform(0.8, 1.2, c),
        "beta": rng.normal(0.0, 0.05, c),
        "moving_mean": rng.normal(0.0, 0.05, c),
        "moving_variance": rng.uniform(0.8, 1.2, c),
    }


def make_seeded_weights(cfg: ModelConfig, seed: int = 42) -> Dict[str, np.ndarray]:
    """Random-initialised weights of the given architecture.

    There are no trained weights to be had (the reference ships none), so the
    benchmarks and parity tests run on seeded synthetic ones: conv kernels
    ``N(0, 2/fan_in)``, the second conv of every residual block scaled by 0.15
    so 24 blocks stay O(1) in 16-bit, BatchNorm statistics near identity, the
    flow head scaled so that |flow| stays within a few HR pixels.
    Names follow the Keras layer names of scripts/training/models.py.
    """
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}

    def conv(name, k, cin, cout, scale=1.0, bias=False):
        std = np.sqrt(2.0 / (k * k * cin)) * scale
        w[name + "/kernel"] = rng.normal(0.0, std, (k, k, cin, cout))
        if bias:
            w[name + "/bias"] = rng.normal(0.0, 0.02, cout)

    def bn(name, c):
        for k, v in _bn(rng, c).items():
            w[name + "/" + k] = v

    def res_block(name, c):
        conv(name + "/conv_1", 3, c, c)
        bn(name + "/bn_1", c)
        conv(name + "/conv_2", 3, c, c, scale=0.15)
        bn(name + "/bn_2", c)

    # --- flow model -------------------------------------------------------
    cin = 3 * cfg.num_flow_inputs
    if cfg.flow_arch == "autoencoder":
        f = cfg.flow_filters
        nb = len(f) // 2
        for i in range(2 * nb):
            n = f"flow/block_{i + 1}"
            conv(n + "/conv_1", 3, cin, f[i])
            bn(n + "/bn_1", f[i])
            conv(n + "/conv_2", 3, f[i], f[i])
            bn(n + "/bn_2", f[i])
            cin = f[i]
        if len(f) % 2:
            conv("flow/conv_1", 3, cin, f[-1])
            bn("flow/bn_1", f[-1])
            cin = f[-1]
        conv("flow/conv_2", 3, cin, 32, scale=0.6, bias=True)
    elif cfg.flow_arch == "resnet":
        n = cfg.flow_res_filters
        conv("flow/conv_1", 3, cin, n)
        bn("flow/bn_1", n)
        for i in range(cfg.flow_res_blocks):
            res_block(f"flow/block_{i + 1}", n)
        conv("flow/conv_2", 1, n, 32, scale=0.6, bias=True)
    else:
        raise ValueError(cfg.flow_arch)
    w["flow/conv_2/bias"] = rng.normal(0.0, 0.5, 32)

    # --- generator --------------------------------------------------------
    nf = cfg.gen_filters
    conv("generator/conv_1", 3, 3 + 48, nf)
    bn("generator/bn_1", nf)
    for i in range(cfg.gen_blocks):
        res_block(f"generator/block_{i + 1}", nf)
    # Conv2DTranspose kernels are [kh, kw, cout, cin] (keras layout).
    w["generator/conv_trans_1/kernel"] = rng.normal(
        0.0, np.sqrt(2.0 / nf), (2, 2, 32, nf))
    bn("generator/bn_2", 32)
    w["generator/conv_trans_2/kernel"] = rng.normal(
        0.0, 0.25 * np.sqrt(1.0 / 32), (2, 2, 3, 32))
    w["generator/conv_trans_2/bias"] = rng.normal(0.0, 0.02, 3)
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}


def validate_config(cfg: ModelConfig) -> None:
    """The width / depth limits of the C++ loader (csrc/model.cpp validateConfig), same messages:
    the reference constructors take any integer (models.py:257-263, 334-339, 484-491); the
    engine admits what its GPU parity tests run against the oracle."""
    def bad(what: str):
        raise ValueError("Invalid model: " + what)
    if not 1 <= cfg.num_flow_inputs <= 5:
        bad("1..5 flow inputs supported")
    if cfg.gen_filters <= 0 or cfg.gen_filters > 256 or cfg.gen_filters % 32:
        bad("gen_filters must be a multiple of 32 (at most 256)")
    if not 0 <= cfg.gen_blocks <= 256:
        bad("gen_blocks must be in 0..256")
    if cfg.flow_arch == "autoencoder":
        nb = len(cfg.flow_filters) // 2
        if nb < 1 or cfg.padded_height % (1 << nb) or cfg.padded_width % (1 << nb):
            bad("padded frame size must be divisible by 2^(flow depth)")
        if any(f <= 0 or f > 512 or f % 32 for f in cfg.flow_filters):
            bad("flow filters must be multiples of 32 (at most 512)")
    else:
        if cfg.flow_res_filters <= 0 or cfg.flow_res_filters > 256 or cfg.flow_res_filters % 32:
            bad("flow_res_filters must be a multiple of 32 (at most 256)")
        if not 0 <= cfg.flow_res_blocks <= 256:
            bad("flow_res_blocks must be in 0..256")
    # no activation tensor may reach 4 GiB (32-bit byte offsets in several kernels; csrc/model.cpp states the rule)
    widest = max(64, cfg.gen_filters)
    if cfg.flow_arch == "autoencoder":
        nb = len(cfg.flow_filters) // 2
        for i, f in enumerate(cfg.flow_filters):
            level = i if i < nb else max(0, 2 * nb - 1 - i)
            widest = max(widest, f >> (2 * min(level, 4)))
    else:
        widest = max(widest, cfg.flow_res_filters)
    rows = (cfg.padded_height + 7) // 8 * 8 + 2
    pitch = (cfg.padded_width + 31) // 32 * 32 + 2
    if rows * pitch * 2 * widest > 0xFFC00000:
        bad(f"frame too large for this model: an activation tensor would reach 4 GiB ({widest} channels x "
            f"{rows * pitch} pixels x 2 bytes)")


def serialize(cfg: ModelConfig, weights: Dict[str, np.ndarray], validate: bool = True) -> bytes:
    """Build the container bytes.  ``validate=False`` writes a header the loader will reject
    (the loader tests need such files)."""
    if validate:
        validate_config(cfg)
    names = list(weights.keys())
    ff = list(cfg.flow_filters) + [0] * (8 - len(cfg.flow_filters))
    if len(cfg.flow_filters) > 8:
        raise ValueError("at most 8 flow filters")
    acts = ACTIVATION[cfg.flow_activation] | ACTIVATION[cfg.gen_activation] << 8
    ext = b""
    if cfg.temporal_extended:
        flags = ({"L1": 0, "L2": 1}[cfg.temporal_norm] | int(cfg.temporal_limit) << 1
                 | int(cfg.temporal_luma) << 2)
        ext = struct.pack("<IfI5I", cfg.temporal_window, cfg.temporal_gain, flags, 0, 0, 0, 0, 0)
    header_bytes = HEADER_BYTES + len(ext)
    hdr = MAGIC + struct.pack(
        "<2I4I4I4I8IfIIffIff", VERSION, header_bytes,
        cfg.frame_height, cfg.frame_width, 4, cfg.num_flow_inputs,
        FLOW_ARCH[cfg.flow_arch], cfg.flow_pad_factor,
        int(cfg.normalize_brightness), cfg.gen_filters,
        cfg.gen_blocks, cfg.flow_res_filters, cfg.flow_res_blocks,
        len(cfg.flow_filters), *ff, cfg.bn_eps, cfg.compute_dtype, len(names),
        # (filter off: both words stay zero, the file is byte-identical to one written
        # before the filter existed)
        cfg.temporal_strength, cfg.temporal_threshold if cfg.temporal_strength > 0 else 0.0,
        # (all-ReLU models: the three words stay zero, as before the field existed)
        acts, cfg.flow_negative_slope if cfg.flow_activation == "lrelu" else 0.0,
        cfg.gen_negative_slope if cfg.gen_activation == "lrelu" else 0.0)
    assert len(hdr) == HEADER_BYTES, len(hdr)
    hdr += ext
    off = header_bytes + ENTRY_BYTES * len(names)
    off = (off + 63) // 64 * 64
    table = b""
    blobs = []
    for n in names:
        a = np.ascontiguousarray(weights[n], dtype="<f4")
        if a.ndim > 4 or len(n.encode()) >= 92:
            raise ValueError(n)
        dims = list(a.shape) + [1] * (4 - a.ndim)
        table += struct.pack("<92sI4IQQ", n.encode(), a.ndim, *dims, off, a.size)
        blobs.append((off, a.tobytes()))
        off = (off + a.nbytes + 63) // 64 * 64
    out = bytearray(off)
    out[:header_bytes] = hdr
    out[header_bytes:header_bytes + len(table)] = table
    for o, b in blobs:
        out[o:o + len(b)] = b
    return bytes(out)


def deserialize(blob: bytes) -> Tuple[ModelConfig, Dict[str, np.ndarray]]:
    """Parse container bytes (the Python twin of csrc/model.cpp)."""
    if len(blob) < HEADER_BYTES or blob[:8] != MAGIC:
        raise ValueError("not a JoshUpscale-AMD model container")
    vals = struct.unpack_from("<2I4I4I4I8IfIIffIff", blob, 8)
    version, header_bytes = vals[0], vals[1]
    if version != VERSION:
        raise ValueError(f"unsupported container version {version}")
    (fh, fw, scale, nfi, arch, pad, nb, gf, gb, frf, frb, nff) = vals[2:14]
    ff = vals[14:22]
    eps, cdt, nt, ts, tt, acts, fslope, gslope = vals[22:30]
    if scale != 4:
        raise ValueError("scale must be 4")
    # (as csrc/model.cpp: unknown codes and non-zero upper bits are errors, not KeyErrors)
    if acts >> 16:
        raise ValueError("Invalid model: unknown activation field")
    if (acts & 0xff) not in ACTIVATION_INV:
        raise ValueError(f"Invalid model: unknown flow activation {acts & 0xff}")
    if ((acts >> 8) & 0xff) not in ACTIVATION_INV:
        raise ValueError(f"Invalid model: unknown generator activation {(acts >> 8) & 0xff}")
    fact, gact = ACTIVATION_INV[acts & 0xff], ACTIVATION_INV[(acts >> 8) & 0xff]
    cfg = ModelConfig(fh, fw, nfi, FLOW_ARCH_INV[arch], tuple(ff[:nff]), frf,
                      frb, pad, gf, gb, bool(nb), eps, cdt, ts,
                      tt if ts > 0 else ModelConfig.temporal_threshold, fact, gact,
                      fslope if fact == "lrelu" else DEFAULT_NEGATIVE_SLOPE,
                      gslope if gact == "lrelu" else DEFAULT_NEGATIVE_SLOPE)
    if header_bytes >= 160:
        win, gain, flags = struct.unpack_from("<IfI", blob, 128)
        if flags >> 3:
            raise ValueError("Invalid model: unknown temporal filter flags")
        cfg.temporal_window, cfg.temporal_gain = win, gain
        cfg.temporal_norm = "L2" if flags & 1 else "L1"
        cfg.temporal_limit, cfg.temporal_luma = bool(flags & 2), bool(flags & 4)
    w = {}
    for i in range(nt):
        name, ndim, d0, d1, d2, d3, off, cnt = struct.unpack_from(
            "<92sI4IQQ", blob, header_bytes + i * ENTRY_BYTES)
        name = name.split(b"\0", 1)[0].decode()
        shape = (d0, d1, d2, d3)[:ndim]
        w[name] = np.frombuffer(blob, dtype="<f4", count=cnt,
                                offset=off).reshape(shape).copy()
    return cfg, w


def save(path: str, cfg: ModelConfig, weights: Dict[str, np.ndarray]) -> None:
    with open(path, "wb") as f:
        f.write(serialize(cfg, weights))


def load(path: str) -> Tuple[ModelConfig, Dict[str, np.ndarray]]:
    with open(path, "rb") as f:
        return deserialize(f.read())


def synthetic_frames(n: int, h: int, w: int, seed: int = 1234,
                     kind: str = "noise") -> np.ndarray:
    """Synthetic BGRX clips: ``noise`` = uniform random bytes;
    ``smooth`` = low-pass noise translating 1 px/frame so the flow/warp path
    sees coherent motion.  X is forced to 255 on input (it must be ignored)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if kind == "noise":
        f = rng.integers(0, 256, size=(n, h, w, 4), dtype=np.uint8)
    elif kind == "smooth":
        base = rng.random((h + 16, w + n + 16, 3))
        for _ in range(3):
            base = (base + np.roll(base, 1, 0) + np.roll(base, -1, 0)
                    + np.roll(base, 1, 1) + np.roll(base, -1, 1)
                    + np.roll(base, (2, 2), (0, 1))
                    + np.roll(base, (-2, -2), (0, 1))) / 7.0
        base = (base - base.min()) / (base.max() - base.min())
        f = np.empty((n, h, w, 4), dtype=np.uint8)
        for t in range(n):
            f[t, ..., :3] = (base[8:8 + h, 8 + t:8 + t + w] * 255).astype(np.uint8)
    else:
        raise ValueError(kind)
    f[..., 3] = 255
    return f
