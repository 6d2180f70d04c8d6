"""CPU oracle for the JoshUpscale per-frame recurrent SR step (numpy, float64).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``joshupscale_amd`` -> libJoshUpscale.so -> HIP kernels)
never calls into this module and fails loudly when the HIP library is missing.

PARITY UNPINNED.  The reference (itmo153277/JoshUpscale) ships no tests, no
golden vectors, no weights and no model files, and neither its native path
(CUDA + TensorRT) nor its Python path (TensorFlow/Keras, onnxruntime) can be
built or imported in this environment.  This file is therefore a from-source
restatement of the reference's arithmetic.  It is cross-checked by a second,
independent restatement built on third-party PyTorch CPU operators
(``tests/torch_restatement.py``) and by per-primitive known-answer tests, but
no output of the reference itself anchors it.

Every function cites the reference file:line (relative to the reference root)
whose behaviour it follows.

Tensor conventions (reference scripts/training/keras_layers.py:208,
scripts/training/dataset.py:270-289): activations are NHWC without the batch
dimension here, i.e. ``[H, W, C]``, channel order B,G,R, values in
[-0.5, 0.5].  Weights are kept in Keras layouts:
``Conv2D.kernel [kh, kw, cin, cout]``, ``Conv2DTranspose.kernel
[kh, kw, cout, cin]``.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

SCALE = 4  # reference core/src/tensorrt_backend.cc:27 (kScale)

# reference scripts/training/utils.py:151 (BGR_LUMA)
BGR_LUMA = np.array([0.114, 0.587, 0.2989], dtype=np.float64)


# --------------------------------------------------------------------------
# model configuration (what the reference bakes into the Keras graph)
# --------------------------------------------------------------------------
@dataclass
class ModelConfig:
    """Hyper-parameters of ``get_inference_model`` and its two sub-models.

    Defaults are the constructor defaults of the reference
    (scripts/training/models.py:257-263, 334-339, 365, 484-491, 680-688).
    """

    frame_height: int = 270
    frame_width: int = 480
    num_flow_inputs: int = 4
    flow_arch: str = "autoencoder"          # or "resnet"
    flow_filters: Tuple[int, ...] = (32, 64, 128, 256, 128, 64, 32)
    flow_res_filters: int = 64
    flow_res_blocks: int = 10
    flow_pad_factor: int = 8                # 0 = no padding
    gen_filters: int = 64
    gen_blocks: int = 24
    normalize_brightness: bool = False
    bn_eps: float = 1e-3                    # keras BatchNormalization default
    # output filter of scripts/inference/onnx/frame_moving_avg.py (defaults of that
    # script: window 0, gain 0, L1, no limit, no luma weighting); strength 0 = absent
    temporal_strength: float = 0.0
    temporal_threshold: float = 0.1
    # the script's other switches (frame_moving_avg.py:99-110): --window (scene detection
    # window in HR pixels, 0 = one global mean), --gain (0 = hard sign gate, else
    # tanh(gain * (m - t))), --norm L1|L2, --limit (clip pre_warp to +-0.5), --luma-normalize
    temporal_window: int = 0
    temporal_gain: float = 0.0
    temporal_norm: str = "L1"
    temporal_limit: bool = False
    temporal_luma: bool = False
    # BASELINE.json config 5: the 64->64 block convolutions of the generator on 8-bit
    # (OCP e4m3) operands.  Not a reference feature as such -- the reference's 8-bit
    # deployment is TensorRT INT8 (scripts/inference/tensorrt/quantize_int8.py) -- so
    # this restates the BUILD's scheme (joshupscale_amd/csrc/fp8.h) for parity checks.
    fp8_tower: bool = False
    # `activation` of the sub-model constructors (models.py:24-27, 36-60, 261, 337, 489):
    # "relu" -> keras.layers.ReLU(), "lrelu" -> keras.layers.LeakyReLU(negative_slope).
    # 0.3 is Keras 3's LeakyReLU default (the reference pins tensorflow-cpu==2.18.0;
    # Keras itself is not in the reference tree: stated assumption).
    flow_activation: str = "relu"
    gen_activation: str = "relu"
    flow_negative_slope: float = 0.3
    gen_negative_slope: float = 0.3

    def act(self, part: str):
        """Activation callable of the flow net (``part="flow"``) or the generator."""
        name = self.flow_activation if part == "flow" else self.gen_activation
        slope = self.flow_negative_slope if part == "flow" else self.gen_negative_slope
        if name == "relu":
            return relu
        if name == "lrelu":
            return lambda x: leaky_relu(x, slope)
        raise ValueError(f"Unknown activation: {name}")   # models.py:57-58

    @property
    def padded_height(self) -> int:
        # models.py:735-744
        f = self.flow_pad_factor
        if not f:
            return self.frame_height
        return (self.frame_height + f - 1) // f * f

    @property
    def padded_width(self) -> int:
        f = self.flow_pad_factor
        if not f:
            return self.frame_width
        return (self.frame_width + f - 1) // f * f


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def conv2d_same(x: np.ndarray, kernel: np.ndarray,
                bias: Optional[np.ndarray] = None) -> np.ndarray:
    """``layers.Conv2D(strides=1, padding="same")`` for odd square kernels.

    Reference call sites: scripts/training/models.py:218-225, 300-306,
    378-385, 469-475, 531-537.  TF "SAME" padding for stride 1 and kernel k
    pads (k-1)//2 zeros on every side; the op is a cross-correlation:
    ``y[h,w,o] = sum_{a,b,c} x[h+a-p, w+b-p, c] * K[a,b,c,o]``.
    """
    kh, kw, cin, cout = kernel.shape
    assert kh == kw and kh % 2 == 1 and x.shape[2] == cin
    p = (kh - 1) // 2
    h, w, _ = x.shape
    xp = np.zeros((h + 2 * p, w + 2 * p, cin), dtype=x.dtype)
    xp[p:p + h, p:p + w] = x
    y = np.zeros((h, w, cout), dtype=x.dtype)
    for a in range(kh):
        for b in range(kw):
            y += xp[a:a + h, b:b + w].reshape(-1, cin).dot(
                kernel[a, b]).reshape(h, w, cout)
    if bias is not None:
        y += bias
    return y


def batch_norm(x: np.ndarray, gamma, beta, mean, var, eps: float) -> np.ndarray:
    """``layers.BatchNormalization`` at inference (models.py:226-228).

    ``y = gamma * (x - mean) / sqrt(var + eps) + beta`` per channel.
    """
    return (x - mean) / np.sqrt(var + eps) * gamma + beta


def relu(x: np.ndarray) -> np.ndarray:
    """Default activation of every inference-path model (models.py:261, 337, 489)."""
    return np.maximum(x, 0)


def leaky_relu(x: np.ndarray, negative_slope: float) -> np.ndarray:
    """``layers.LeakyReLU(negative_slope)`` (models.py:24-27 "lrelu"):
    ``x if x >= 0 else negative_slope * x``."""
    return np.where(x >= 0, x, x * x.dtype.type(negative_slope))


def max_pool_2x2(x: np.ndarray) -> np.ndarray:
    """``layers.MaxPool2D(pool_size=2)`` (models.py:406-409): stride 2, VALID."""
    h, w, c = x.shape
    h2, w2 = h // 2, w // 2
    return x[:h2 * 2, :w2 * 2].reshape(h2, 2, w2, 2, c).max(axis=(1, 3))


def resize_bilinear_tf1(x: np.ndarray, scale: int) -> np.ndarray:
    """``UpscaleLayer`` (keras_layers.py:46-52).

    ``tf.compat.v1.image.resize_bilinear(align_corners=False,
    half_pixel_centers=False)``: source coordinate ``s = dst / scale``
    (asymmetric), ``lo = floor(s)``, ``hi = min(lo + 1, size - 1)``,
    ``frac = s - lo``; lerp in x inside lerp in y.
    """
    h, w, _ = x.shape

    def axis(n_in: int):
        dst = np.arange(n_in * scale, dtype=np.float64)
        src = dst / scale
        lo = np.floor(src).astype(np.int64)
        hi = np.minimum(lo + 1, n_in - 1)
        return lo, hi, (src - lo).astype(x.dtype)

    y0, y1, fy = axis(h)
    x0, x1, fx = axis(w)
    fx = fx[None, :, None]
    fy = fy[:, None, None]
    top = x[y0][:, x0] + (x[y0][:, x1] - x[y0][:, x0]) * fx
    bot = x[y1][:, x0] + (x[y1][:, x1] - x[y1][:, x0]) * fx
    return top + (bot - top) * fy


def space_to_depth(x: np.ndarray, bs: int) -> np.ndarray:
    """``tf.nn.space_to_depth`` NHWC (keras_layers.py:129):
    ``out[h, w, (i*bs + j)*C + c] = x[h*bs + i, w*bs + j, c]``."""
    h, w, c = x.shape
    return x.reshape(h // bs, bs, w // bs, bs, c).transpose(0, 2, 1, 3, 4) \
        .reshape(h // bs, w // bs, bs * bs * c)


def depth_to_space(x: np.ndarray, bs: int) -> np.ndarray:
    """``tf.nn.depth_to_space`` NHWC, "DCR" order (keras_layers.py:175):
    ``out[h*bs + i, w*bs + j, c] = x[h, w, (i*bs + j)*C' + c]``."""
    h, w, c = x.shape
    c2 = c // (bs * bs)
    return x.reshape(h, w, bs, bs, c2).transpose(0, 2, 1, 3, 4) \
        .reshape(h * bs, w * bs, c2)


def conv2d_transpose_k2s2(x: np.ndarray, kernel: np.ndarray,
                          bias: Optional[np.ndarray] = None) -> np.ndarray:
    """``layers.Conv2DTranspose(kernel_size=2, strides=2, padding="same")``
    (models.py:559-566, 573-579).  Kernel layout ``[kh, kw, cout, cin]``.
    With k == s there is no overlap:
    ``y[2h+a, 2w+b, o] = sum_c x[h,w,c] * K[a,b,o,c]``.
    """
    kh, kw, cout, cin = kernel.shape
    assert kh == 2 and kw == 2 and x.shape[2] == cin
    h, w, _ = x.shape
    y = np.zeros((h, 2, w, 2, cout), dtype=x.dtype)
    flat = x.reshape(-1, cin)
    for a in range(2):
        for b in range(2):
            y[:, a, :, b, :] = flat.dot(kernel[a, b].T).reshape(h, w, cout)
    y = y.reshape(2 * h, 2 * w, cout)
    if bias is not None:
        y = y + bias
    return y


def dense_image_warp(image: np.ndarray, flow: np.ndarray) -> np.ndarray:
    """``DenseWarpLayer`` -> ``dense_image_warp`` -> ``_interpolate_bilinear_impl``
    (keras_layers.py:79-97; tfa/dense_image_warp.py:182-245, 87-173).

    ``query = (y, x) - flow[y, x, (0, 1)]``; per axis
    ``floor = min(max(0, floor(q)), size - 2)``, ``alpha = clip(q - floor, 0, 1)``;
    gather 4 corners; ``top = TL + ax*(TR - TL)``, ``bot = BL + ax*(BR - BL)``,
    ``out = top + ay*(bot - top)`` (tfa/dense_image_warp.py:169-171).
    """
    h, w, _ = image.shape
    gy, gx = np.meshgrid(np.arange(h, dtype=flow.dtype),
                         np.arange(w, dtype=flow.dtype), indexing="ij")
    qy = gy - flow[..., 0]
    qx = gx - flow[..., 1]

    def axis(q, size):
        fl = np.minimum(np.maximum(0.0, np.floor(q)), size - 2)
        alpha = np.clip(q - fl, 0.0, 1.0)
        return fl.astype(np.int64), alpha

    y0, ay = axis(qy, h)
    x0, ax = axis(qx, w)
    y1 = y0 + 1
    x1 = x0 + 1
    tl = image[y0, x0]
    tr = image[y0, x1]
    bl = image[y1, x0]
    br = image[y1, x1]
    ax = ax[..., None].astype(image.dtype)
    ay = ay[..., None].astype(image.dtype)
    top = ax * (tr - tl) + tl
    bot = ax * (br - bl) + bl
    return ay * (bot - top) + top


def preprocess(frame_u8: np.ndarray, dtype=np.float64) -> np.ndarray:
    """``PreprocessLayer`` (keras_layers.py:192-208): ``x / 255 - 0.5``."""
    return frame_u8.astype(dtype) / dtype(255) - dtype(0.5)


def postprocess(x: np.ndarray) -> np.ndarray:
    """``PostprocessLayer`` (keras_layers.py:211-230): ``(x + 0.5) * 255`` then a
    truncating cast to uint8 (``ops.cast``; the deployed graph keeps the value
    in float and core/src/cuda_convert.cc.cu:76-81 truncates instead).  The
    input is already clipped to [-0.5, 0.5], so the value is in [0, 255]."""
    return np.trunc((x + 0.5) * 255).astype(np.uint8)


# --------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------
Weights = Dict[str, np.ndarray]


def _rec(trace: Optional[dict], name: str, value: np.ndarray) -> None:
    """Record an intermediate tensor for layer-by-layer parity diagnostics."""
    if trace is not None:
        trace[name] = value


def _conv_bn_act(x, wts: Weights, conv: str, bn: str, eps: float, act=relu) -> np.ndarray:
    y = conv2d_same(x, wts[conv + "/kernel"])
    y = batch_norm(y, wts[bn + "/gamma"], wts[bn + "/beta"],
                   wts[bn + "/moving_mean"], wts[bn + "/moving_variance"], eps)
    return act(y)


def res_block(x, wts: Weights, name: str, eps: float, act=relu, amax: Optional[list] = None) -> np.ndarray:
    """``res_block`` (models.py:193-254):
    ``act(BN2(conv2(act(BN1(conv1(x))))) + x)``; FadeIn is the identity at
    inference once its counter has saturated (keras_layers.py:321-323).
    ``amax``: receives max |.| of the two post-activation tensors (what a symmetric
    per-tensor activation calibration records, generate_calibration.py:93-234)."""
    y = _conv_bn_act(x, wts, name + "/conv_1", name + "/bn_1", eps, act)
    if amax is not None:
        amax.append(float(np.abs(y).max()))
    out = _res_block_tail(x, y, wts, name, eps, act)
    if amax is not None:
        amax.append(float(np.abs(out).max()))
    return out


def _res_block_tail(x, y, wts: Weights, name: str, eps: float, act) -> np.ndarray:
    y = conv2d_same(y, wts[name + "/conv_2/kernel"])
    b = name + "/bn_2"
    y = batch_norm(y, wts[b + "/gamma"], wts[b + "/beta"],
                   wts[b + "/moving_mean"], wts[b + "/moving_variance"], eps)
    return act(y + x)


# --------------------------------------------------------------------------
# 8-bit tower (scheme of joshupscale_amd/csrc/fp8.h, restated)
# --------------------------------------------------------------------------
FP8_DEFAULT_AMAX = 7.0


def e4m3_round(x: np.ndarray) -> np.ndarray:
    """Nearest OCP e4m3fn value, ties to even, saturating at +-448: 3 mantissa bits,
    normal exponents -6..8, subnormal step 2^-9."""
    x = np.asarray(x, np.float64)
    a = np.minimum(np.abs(x), 448.0)
    _, e = np.frexp(a)                       # a = m * 2^e, m in [0.5, 1)
    step = np.exp2(np.maximum(e - 1, -6) - 3.0)
    q = np.minimum(np.round(a / step) * step, 448.0)   # np.round: half to even
    return np.copysign(q, x)


def fp8_activation_exponent(amax: float) -> int:
    """Power-of-two scale with one bit of headroom: amax * 2^e in (112, 224]."""
    if not (amax > 0 and np.isfinite(amax)):
        return 0
    return int(np.clip(np.floor(np.log2(224.0 / amax)), -16, 16))


def _fold_bn(wts: Weights, conv: str, bn: str, eps: float):
    """BN folded into the convolution the way the engine's loader does it
    (csrc/model.cpp bnScaleShift / foldConv): scale and shift in float64 from the float32
    variables, the products rounded ONCE to float32.  Every operation is a correctly rounded
    IEEE one, so any implementation of this recipe gives the same float32 kernel -- which
    matters here because the e4m3 quantiser that follows is discontinuous."""
    f8 = np.float64
    k = np.asarray(wts[conv + "/kernel"], np.float32).astype(f8)
    g = np.asarray(wts[bn + "/gamma"], np.float32).astype(f8)
    var = np.asarray(wts[bn + "/moving_variance"], np.float32).astype(f8)
    scale = g / np.sqrt(var + f8(np.float32(eps)))
    bias = np.asarray(wts[bn + "/beta"], np.float32).astype(f8) - \
        np.asarray(wts[bn + "/moving_mean"], np.float32).astype(f8) * scale
    return (k * scale).astype(np.float32), bias.astype(np.float32)


def fp8_quantize_weights(k: np.ndarray) -> np.ndarray:
    """Per OUTPUT channel: 2^ew with max|w| * 2^ew in (224, 448], e4m3, scaled back."""
    amax = np.abs(k).reshape(-1, k.shape[-1]).max(axis=0).astype(np.float64)
    ew = np.where(amax > 0, np.floor(np.log2(448.0 / np.maximum(amax, 1e-300))), 0.0)
    ew = np.clip(ew, -32, 32)
    return e4m3_round(k.astype(np.float64) * np.exp2(ew)) * np.exp2(-ew)


def fp8_quantize_activation(x: np.ndarray, exponent: int) -> np.ndarray:
    """e4m3(clamp(x * 2^e, +-448)) * 2^-e.  Post-ReLU tensors only use the upper clamp; a
    LeakyReLU tensor (`activation: lrelu`) has negative values too."""
    return e4m3_round(np.clip(x * 2.0 ** exponent, -448.0, 448.0)) * 2.0 ** -exponent


def res_block_fp8(x, wts: Weights, name: str, eps: float, ex: int, et: int, act=relu) -> np.ndarray:
    """res_block with both convolutions on e4m3 operands; bias, activation and the skip
    connection in full precision (the stream ``x`` itself is never quantised)."""
    k1, b1 = _fold_bn(wts, name + "/conv_1", name + "/bn_1", eps)
    k2, b2 = _fold_bn(wts, name + "/conv_2", name + "/bn_2", eps)
    t = act(conv2d_same(fp8_quantize_activation(x, ex), fp8_quantize_weights(k1)) + b1)
    y = conv2d_same(fp8_quantize_activation(t, et), fp8_quantize_weights(k2)) + b2
    return act(y + x)


def flow_autoencoder(frames: Sequence[np.ndarray], wts: Weights,
                     cfg: ModelConfig, trace: Optional[dict] = None) -> np.ndarray:
    """``get_flow_autoencoder`` (models.py:334-481).  Returns the flow field
    ``[4*PH, 4*PW, 2]`` (channel 0 = dy, 1 = dx, in HR pixels)."""
    eps = cfg.bn_eps
    act = cfg.act("flow")
    x = np.concatenate(list(frames), axis=2)  # models.py:373-375
    filters = cfg.flow_filters
    nblk = len(filters) // 2
    for i in range(nblk):  # down blocks, models.py:377-410, 450-451
        n = f"flow/block_{i + 1}"
        x = _conv_bn_act(x, wts, n + "/conv_1", n + "/bn_1", eps, act)
        _rec(trace, n + "/a_1", x)
        x = _conv_bn_act(x, wts, n + "/conv_2", n + "/bn_2", eps, act)
        _rec(trace, n + "/a_2", x)
        x = max_pool_2x2(x)
        _rec(trace, n + "/resample", x)
    for i in range(nblk, 2 * nblk):  # up blocks, models.py:412-447, 452-453
        n = f"flow/block_{i + 1}"
        x = _conv_bn_act(x, wts, n + "/conv_1", n + "/bn_1", eps, act)
        _rec(trace, n + "/a_1", x)
        x = _conv_bn_act(x, wts, n + "/conv_2", n + "/bn_2", eps, act)
        _rec(trace, n + "/a_2", x)
        x = resize_bilinear_tf1(x, 2)
        _rec(trace, n + "/resample", x)
    if len(filters) % 2:  # models.py:454-468
        x = _conv_bn_act(x, wts, "flow/conv_1", "flow/bn_1", eps, act)
        _rec(trace, "flow/a_1", x)
    x = conv2d_same(x, wts["flow/conv_2/kernel"], wts["flow/conv_2/bias"])
    _rec(trace, "flow", x)  # head before depth-to-space: [PH, PW, 32]
    return depth_to_space(x, 4)  # models.py:476-479


def flow_resnet(frames: Sequence[np.ndarray], wts: Weights,
                cfg: ModelConfig, trace: Optional[dict] = None) -> np.ndarray:
    """``get_flow_resnet`` (models.py:257-331)."""
    eps = cfg.bn_eps
    act = cfg.act("flow")
    x = np.concatenate(list(frames), axis=2)
    x = _conv_bn_act(x, wts, "flow/conv_1", "flow/bn_1", eps, act)
    for i in range(cfg.flow_res_blocks):
        x = res_block(x, wts, f"flow/block_{i + 1}", eps, act)
    x = conv2d_same(x, wts["flow/conv_2/kernel"], wts["flow/conv_2/bias"])
    _rec(trace, "flow", x)
    return depth_to_space(x, 4)


def generator(images: np.ndarray, pre_warp: np.ndarray, wts: Weights,
              cfg: ModelConfig, trace: Optional[dict] = None) -> np.ndarray:
    """``get_generator_resnet`` (models.py:484-595)."""
    eps = cfg.bn_eps
    act = cfg.act("generator")
    x = np.concatenate([images, space_to_depth(pre_warp, 4)], axis=2)  # :523-530
    _rec(trace, "gen_in_ref", x)  # reference channel order, 51 channels
    x = _conv_bn_act(x, wts, "generator/conv_1", "generator/bn_1", eps, act)
    _rec(trace, "gen_head", x)
    # max |output| of generator/conv_1 and of the two activations of every residual block, in
    # execution order: the tensors an activation calibration ranges over
    layer_amax = [float(np.abs(x).max())] if trace is not None and not cfg.fp8_tower else None
    if cfg.fp8_tower:
        amax = wts.get("generator/fp8_amax")
        exps = [fp8_activation_exponent(FP8_DEFAULT_AMAX if amax is None else float(np.float32(amax[j])))
                for j in range(2 * cfg.gen_blocks)]
    for i in range(cfg.gen_blocks):
        if cfg.fp8_tower:
            x = res_block_fp8(x, wts, f"generator/block_{i + 1}", eps, exps[2 * i], exps[2 * i + 1], act)
        else:
            x = res_block(x, wts, f"generator/block_{i + 1}", eps, act, layer_amax)
    if layer_amax is not None:
        trace["tower_amax"] = np.asarray(layer_amax)
    _rec(trace, "trunk", x)
    x = conv2d_transpose_k2s2(x, wts["generator/conv_trans_1/kernel"])
    b = "generator/bn_2"
    x = act(batch_norm(x, wts[b + "/gamma"], wts[b + "/beta"],
                       wts[b + "/moving_mean"], wts[b + "/moving_variance"],
                       eps))                        # models.py:567-572
    _rec(trace, "tail_mid", x)  # [2H, 2W, 32]
    x = conv2d_transpose_k2s2(x, wts["generator/conv_trans_2/kernel"],
                              wts["generator/conv_trans_2/bias"])
    x = np.tanh(x)                                  # models.py:580-583
    x = resize_bilinear_tf1(images, 4) + x          # models.py:584-590
    return np.clip(x, -0.5, 0.5)                    # ClipLayer, :591-593


# --------------------------------------------------------------------------
# the recurrent step and the driver loop
# --------------------------------------------------------------------------
@dataclass
class State:
    """Recurrent state: previous HR output and the LR frame shift register.
    Zero-initialised (core/include/JoshUpscale/core/cuda.h:69-72,
    scripts/inference/onnx/inference.py:67-70)."""
    pre_gen: np.ndarray
    last_frames: List[np.ndarray]

    @staticmethod
    def zeros(cfg: ModelConfig, dtype=np.float64) -> "State":
        return State(
            np.zeros((cfg.frame_height * SCALE, cfg.frame_width * SCALE, 3),
                     dtype=dtype),
            [np.zeros((cfg.padded_height, cfg.padded_width, 3), dtype=dtype)
             for _ in range(cfg.num_flow_inputs - 1)])


@dataclass
class StepOutputs:
    output: np.ndarray        # uint8 [4H, 4W, 3]
    output_raw: np.ndarray
    pre_warp: np.ndarray
    flow: np.ndarray          # cropped flow [4H, 4W, 2]
    state: State = field(repr=False, default=None)


def temporal_filter(gen: np.ndarray, pre_warp: np.ndarray, strength: float,
                    threshold: float, window: int = 0, gain: float = 0.0, norm: str = "L1",
                    limit: bool = False, luma: bool = False,
                    trace: Optional[dict] = None) -> np.ndarray:
    """Moving-average output filter with a scene-cut gate: the graph that
    scripts/inference/onnx/frame_moving_avg.py:146-302 splices in place of the
    generator's clip output (its consumers -- postprocess AND the fed-back
    ``output_raw`` -- see the filtered tensor).  ``gen`` / ``pre_warp`` are HR ``[H, W, 3]``.

        p     = clip(pre_warp, -0.5, 0.5) if limit else pre_warp            (:157-166)
        d     = |gen - p| (L1) or (gen - p)^2 (L2)                           (:170-181)
        d    *= BGR_LUMA*3 (once for L1, twice for L2) if luma               (:95-96, 184-187, 219-222)
        window == 0:  m = mean(d) over every element, one scalar            (:183-205)
        window  > 0:  m = per window x window block mean over the 3 channels of the
                      zero-padded frame (padding split (x-y)//2 leading; divisor always
                      3*window^2), a [ceil(H/window), ceil(W/window)] grid  (:207-228)
        c     = sign(m - threshold) if gain == 0 else tanh(gain * (m - threshold))   (:229-238)
        window > 0:   c is resized back by `window` with asymmetric linear interpolation
                      (ONNX Resize linear/asymmetric == TF1 bilinear) and un-padded  (:239-270)
        out   = p * (s/2 - c*s/2) + gen * (c*s/2 + 1 - s/2)                 (:272-296)

    so a still scene blends ``s`` of the warped previous output into the new frame
    and a scene cut passes the generator output through."""
    p = np.clip(pre_warp, -0.5, 0.5) if limit else pre_warp
    d = gen - p
    d = np.abs(d) if norm == "L1" else d * d
    if norm not in ("L1", "L2"):
        raise ValueError(f"Unknown norm type {norm}")
    if luma:
        k = BGR_LUMA.astype(gen.dtype) * 3
        d = d * (k if norm == "L1" else k * k)
    gain_coef = 1.0 if gain == 0 else gain
    if window == 0:
        m = np.mean(d)
        _rec(trace, "temporal_mean", m)
        z = gain_coef * (m - threshold)
        c = np.sign(z) if gain == 0 else np.tanh(z)
    else:
        h, w, _ = d.shape
        oh, ow = (h + window - 1) // window * window, (w + window - 1) // window * window
        py, px = (oh - h) // 2, (ow - w) // 2
        padded = np.zeros((oh, ow, 3), d.dtype)
        padded[py:py + h, px:px + w] = d
        m = padded.reshape(oh // window, window, ow // window, window, 3).sum(axis=(1, 3, 4)) \
            / (3.0 * window * window)
        _rec(trace, "temporal_mean", m)
        z = gain_coef * (m - threshold)
        cg = np.sign(z) if gain == 0 else np.tanh(z)
        c = resize_bilinear_tf1(cg[..., None], window)[py:py + h, px:px + w]
    half = strength / 2
    return p * (half - c * half) + gen * (c * half + 1 - half)


def inference_step(cur_frame_u8: np.ndarray, state: State, wts: Weights,
                   cfg: ModelConfig, dtype=np.float64,
                   trace: Optional[dict] = None) -> StepOutputs:
    """One execution of ``get_inference_model`` (models.py:680-829) with
    ``skip_processing=False``.  ``cur_frame_u8`` is ``[H, W, 3]`` uint8 BGR."""
    h, w = cfg.frame_height, cfg.frame_width
    assert cur_frame_u8.shape == (h, w, 3) and cur_frame_u8.dtype == np.uint8
    cur = preprocess(cur_frame_u8, dtype)                         # :768-770
    cur_pad = cur
    brightness = None
    if cfg.normalize_brightness:                                  # :772-779
        brightness = np.mean(cur * BGR_LUMA.astype(dtype) * 3)
        cur_pad = cur_pad - brightness
    ph, pw = cfg.padded_height, cfg.padded_width
    if (ph, pw) != (h, w):                                        # :780-789
        pad_h, pad_w = ph - h, pw - w
        padded = np.zeros((ph, pw, 3), dtype=dtype)
        padded[pad_h // 2:pad_h // 2 + h, pad_w // 2:pad_w // 2 + w] = cur_pad
        cur_pad = padded
    frames = [cur_pad] + list(state.last_frames)
    if cfg.flow_arch == "autoencoder":
        flow = flow_autoencoder(frames, wts, cfg, trace)          # :790
    elif cfg.flow_arch == "resnet":
        flow = flow_resnet(frames, wts, cfg, trace)
    else:
        raise ValueError(cfg.flow_arch)
    if (ph, pw) != (h, w):                                        # :791-798
        oy = ((ph - h) // 2) * 4
        ox = ((pw - w) // 2) * 4
        flow = flow[oy:oy + h * 4, ox:ox + w * 4]
    pre_warp = dense_image_warp(state.pre_gen, flow)              # :799-801
    if cfg.normalize_brightness:
        pre_warp = pre_warp + brightness                          # :802-803
    _rec(trace, "flow_in", np.concatenate(frames, axis=2))
    output_raw = generator(cur, pre_warp, wts, cfg, trace)        # :804
    if cfg.temporal_strength > 0:
        output_raw = temporal_filter(output_raw, pre_warp, cfg.temporal_strength,
                                     cfg.temporal_threshold, cfg.temporal_window,
                                     cfg.temporal_gain, cfg.temporal_norm, cfg.temporal_limit,
                                     cfg.temporal_luma, trace)
    output = postprocess(output_raw)                              # :805-807
    if cfg.normalize_brightness:
        output_raw = output_raw - brightness                      # :809-810
    # :821-823 `[cur_frame_pad] + last_frames[:-1]`; with a single flow input the reference's graph
    # has no last-frame inputs and emits one state output nothing consumes: keep n-1 frames
    new_state = State(output_raw,
                      ([cur_pad] + list(state.last_frames[:-1]))[:cfg.num_flow_inputs - 1])
    return StepOutputs(output, output_raw, pre_warp, flow, new_state)


def bgrx_to_bgr(frame_bgrx: np.ndarray) -> np.ndarray:
    """Stage-in of the plugin boundary: 4 bytes per pixel, B,G,R,X; X ignored
    (core/include/JoshUpscale/core/tensor.h:18-20, 41-43;
    core/src/cuda_convert.cc.cu:95-108 copies lanes x,y,z only)."""
    return np.ascontiguousarray(frame_bgrx[..., :3])


def bgr_to_bgrx(frame_bgr: np.ndarray) -> np.ndarray:
    """Stage-out: the 4th byte is written as 0 (cuda_convert.cc.cu:39-45)."""
    h, w, _ = frame_bgr.shape
    out = np.zeros((h, w, 4), dtype=np.uint8)
    out[..., :3] = frame_bgr
    return out


class Session:
    """Recurrent driver, mirroring ``Session`` of
    scripts/inference/onnx/inference.py:46-94 (zero state, feed outputs[1:]
    back into inputs[1:]) on BGRX frames as ``Runtime::processImage`` sees them
    (core/src/tensorrt_backend.cc:270-278)."""

    def __init__(self, wts: Weights, cfg: ModelConfig, dtype=np.float64):
        self.wts = {k: np.asarray(v, dtype=dtype) for k, v in wts.items()}
        self.cfg = cfg
        self.dtype = dtype
        self.reset()

    def reset(self) -> None:
        self.state = State.zeros(self.cfg, self.dtype)
        self.last: Optional[StepOutputs] = None

    def run(self, frame_bgrx: np.ndarray, trace: Optional[dict] = None) -> np.ndarray:
        out = inference_step(bgrx_to_bgr(frame_bgrx), self.state, self.wts,
                             self.cfg, self.dtype, trace)
        self.state = out.state
        self.last = out
        return bgr_to_bgrx(out.output)


# --------------------------------------------------------------------------
# algorithmic work (used by bench.py / DESIGN.md for the roofline figures)
# --------------------------------------------------------------------------
def macs_per_frame(cfg: ModelConfig) -> Dict[str, int]:
    """Multiply-accumulates of one frame, by part."""
    h, w = cfg.frame_height, cfg.frame_width
    ph, pw = cfg.padded_height, cfg.padded_width
    nf = cfg.gen_filters
    gen = h * w * (9 * (3 + 48) * nf + cfg.gen_blocks * 2 * 9 * nf * nf
                   + 4 * nf * 32) + (2 * h) * (2 * w) * 4 * 32 * 3
    cin = 3 * cfg.num_flow_inputs
    flow = 0
    if cfg.flow_arch == "autoencoder":
        f = cfg.flow_filters
        nb = len(f) // 2
        px = ph * pw
        for i in range(nb):
            flow += px * 9 * (cin * f[i] + f[i] * f[i])
            cin = f[i]
            px //= 4
        for i in range(nb, 2 * nb):
            flow += px * 9 * (cin * f[i] + f[i] * f[i])
            cin = f[i]
            px *= 4
        if len(f) % 2:
            flow += px * 9 * cin * f[-1]
            cin = f[-1]
        flow += px * 9 * cin * 32
    else:
        n = cfg.flow_res_filters
        flow = ph * pw * (9 * cin * n + cfg.flow_res_blocks * 2 * 9 * n * n
                          + 1 * n * 32)  # conv_2 is 1x1 (models.py:320-325)
    return {"generator": gen, "flow": flow, "total": gen + flow}
