"""Keras variables -> ``.jupw`` container.

The reference trains with Keras and saves ``*.weights.h5``
(scripts/training/train_local.py:119-128, 187-188).  Keras 3 addresses the
variables in such a file by layer CLASS and creation order, not by layer name, so
a faithful reader needs the reference's own model constructors to rebuild the
layer list; that needs TensorFlow/Keras, which this engine's build and GPU boxes
do not have.  The import is therefore split:

* :func:`container_weights` (pure numpy, tested here): maps ``{keras layer name:
  [variables in Keras order]}`` of the flow and generator sub-models to the
  container's tensor names, checks every shape against the architecture, and
  derives the :class:`ModelConfig` fields that the shapes determine.
* ``tools/export_jupw_from_keras.py`` (runs where the reference's Python deps are
  installed): builds the reference models with ``create_models``
  (scripts/training/models.py:1138-1194), loads the weights file, and hands the
  two layer dictionaries to :func:`container_weights`.

Layer names are the reference's (scripts/training/models.py): inside the
generator model ``conv_1, bn_1, block_<i>_conv_1, block_<i>_bn_1,
block_<i>_conv_2, block_<i>_bn_2, conv_trans_1, bn_2, conv_trans_2``
(:224-252, :536-578); inside the flow model ``block_<i>_conv_1 ...`` and
``conv_1, bn_1, conv_2`` (:297-330, :384-474).  Variable order per layer is
Keras': Conv2D / Conv2DTranspose ``[kernel, bias?]``, BatchNormalization
``[gamma, beta, moving_mean, moving_variance]``.  The container keeps the Keras
layouts (Conv2D ``[kh, kw, cin, cout]``, Conv2DTranspose ``[kh, kw, cout, cin]``)
and the unfolded BatchNorm statistics: folding happens in the C++ loader.
"""

from __future__ import annotations

import re
from dataclasses import replace
from typing import Dict, List, Mapping, Sequence, Tuple

import numpy as np

from .model_file import ACTIVATION, DEFAULT_NEGATIVE_SLOPE, ModelConfig

BN_VARS = ("gamma", "beta", "moving_mean", "moving_variance")
_BLOCK = re.compile(r"^block_(\d+)_(conv_1|bn_1|conv_2|bn_2)$")

Layers = Mapping[str, Sequence[np.ndarray]]


def activation_fields(activation) -> Tuple[str, float]:
    """The reference's ``Activation`` spec (scripts/training/models.py:20, 36-60): a name
    or ``{"name": ..., **layer kwargs}`` -> ``(container activation, negative_slope)``.
    ``relu`` / ``lrelu`` are the reference's whole table (models.py:24-27); of the layer
    kwargs only LeakyReLU's ``negative_slope`` (Keras 2 spelling: ``alpha``) is supported."""
    if isinstance(activation, str):
        name, args = activation, {}
    elif isinstance(activation, dict):
        name = activation["name"]
        args = {k: v for k, v in activation.items() if k != "name"}
    else:
        raise TypeError("Unknown type")                       # models.py:55-56
    if name not in ACTIVATION:
        raise ValueError(f"Unknown activation: {name}")      # models.py:57-58
    slope = DEFAULT_NEGATIVE_SLOPE
    for key in ("negative_slope", "alpha"):
        if key in args:
            slope = float(args.pop(key))
    if args or (name == "relu" and slope != DEFAULT_NEGATIVE_SLOPE):
        raise ValueError(f"activation {name!r}: unsupported layer arguments {sorted(args) or ['negative_slope']}")
    return name, slope


def container_name(model: str, layer: str) -> str:
    """``("generator", "block_3_conv_1") -> "generator/block_3/conv_1"``."""
    m = _BLOCK.match(layer)
    if m:
        return f"{model}/block_{m.group(1)}/{m.group(2)}"
    return f"{model}/{layer}"


def keras_layer_name(container: str) -> Tuple[str, str]:
    """Inverse of :func:`container_name` for a layer prefix (no variable part)."""
    parts = container.split("/")
    return parts[0], "_".join(parts[1:])


def _put_conv(out: Dict[str, np.ndarray], name: str, vars_: Sequence[np.ndarray],
              bias: bool) -> np.ndarray:
    want = 2 if bias else 1
    if len(vars_) != want:
        raise ValueError(f"{name}: expected {want} variable(s) "
                         f"({'kernel, bias' if bias else 'kernel, use_bias=False'}), got {len(vars_)}")
    k = np.asarray(vars_[0], dtype=np.float32)
    if k.ndim != 4:
        raise ValueError(f"{name}: kernel must be 4-D, got shape {k.shape}")
    out[name + "/kernel"] = k
    if bias:
        b = np.asarray(vars_[1], dtype=np.float32)
        out[name + "/bias"] = b
    return k


def _put_bn(out: Dict[str, np.ndarray], name: str, vars_: Sequence[np.ndarray], c: int) -> None:
    if len(vars_) != 4:
        raise ValueError(f"{name}: BatchNormalization needs gamma, beta, moving_mean, "
                         f"moving_variance (got {len(vars_)} variables; scale/center disabled?)")
    for v, a in zip(BN_VARS, vars_):
        a = np.asarray(a, dtype=np.float32)
        if a.shape != (c,):
            raise ValueError(f"{name}/{v}: shape {a.shape}, expected ({c},)")
        out[f"{name}/{v}"] = a


def _need(layers: Layers, model: str, layer: str) -> Sequence[np.ndarray]:
    if layer not in layers:
        raise KeyError(f"{model} model has no layer {layer!r} "
                       f"(has: {', '.join(sorted(layers)[:8])}, ...)")
    return layers[layer]


def _count_blocks(layers: Layers) -> int:
    n = 0
    while f"block_{n + 1}_conv_1" in layers:
        n += 1
    return n


def container_weights(generator: Layers, flow: Layers, base: ModelConfig
                      ) -> Tuple[ModelConfig, Dict[str, np.ndarray]]:
    """Map the two sub-models' layers to container tensors.

    ``base`` supplies what the weights cannot tell (frame size, flow padding,
    brightness flag, BatchNorm epsilon, compute dtype, the two activations -- see
    :func:`activation_fields`); filters, block counts,
    the flow architecture and the number of flow inputs are read off the shapes
    and override ``base``.
    """
    w: Dict[str, np.ndarray] = {}

    # ---- generator (models.py:521-593) ----
    k = _put_conv(w, "generator/conv_1", _need(generator, "generator", "conv_1"), bias=False)
    if k.shape[:3] != (3, 3, 51):
        raise ValueError(f"generator/conv_1/kernel: shape {k.shape}, expected (3, 3, 51, F)")
    nf = int(k.shape[3])
    _put_bn(w, "generator/bn_1", _need(generator, "generator", "bn_1"), nf)
    gen_blocks = _count_blocks(generator)
    for i in range(1, gen_blocks + 1):
        for j in (1, 2):
            kk = _put_conv(w, f"generator/block_{i}/conv_{j}",
                           _need(generator, "generator", f"block_{i}_conv_{j}"), bias=False)
            if kk.shape != (3, 3, nf, nf):
                raise ValueError(f"generator/block_{i}/conv_{j}/kernel: shape {kk.shape}")
            _put_bn(w, f"generator/block_{i}/bn_{j}",
                    _need(generator, "generator", f"block_{i}_bn_{j}"), nf)
    kt1 = _put_conv(w, "generator/conv_trans_1", _need(generator, "generator", "conv_trans_1"),
                    bias=False)
    if kt1.shape != (2, 2, 32, nf):  # Conv2DTranspose: [kh, kw, cout, cin]
        raise ValueError(f"generator/conv_trans_1/kernel: shape {kt1.shape}, expected (2, 2, 32, {nf})")
    _put_bn(w, "generator/bn_2", _need(generator, "generator", "bn_2"), 32)
    kt2 = _put_conv(w, "generator/conv_trans_2", _need(generator, "generator", "conv_trans_2"),
                    bias=True)
    if kt2.shape != (2, 2, 3, 32) or w["generator/conv_trans_2/bias"].shape != (3,):
        raise ValueError(f"generator/conv_trans_2: kernel {kt2.shape}, expected (2, 2, 3, 32) + bias (3,)")

    # ---- flow (models.py:257-331 resnet, 334-481 autoencoder) ----
    head = np.asarray(_need(flow, "flow", "conv_2")[0])
    arch = "resnet" if head.shape[:2] == (1, 1) else "autoencoder"
    flow_filters: List[int] = []
    flow_res_filters, flow_res_blocks = base.flow_res_filters, base.flow_res_blocks
    if arch == "resnet":
        k = _put_conv(w, "flow/conv_1", _need(flow, "flow", "conv_1"), bias=False)
        cin, n = int(k.shape[2]), int(k.shape[3])
        _put_bn(w, "flow/bn_1", _need(flow, "flow", "bn_1"), n)
        flow_res_filters, flow_res_blocks = n, _count_blocks(flow)
        for i in range(1, flow_res_blocks + 1):
            for j in (1, 2):
                kk = _put_conv(w, f"flow/block_{i}/conv_{j}", _need(flow, "flow", f"block_{i}_conv_{j}"),
                               bias=False)
                if kk.shape != (3, 3, n, n):
                    raise ValueError(f"flow/block_{i}/conv_{j}/kernel: shape {kk.shape}")
                _put_bn(w, f"flow/block_{i}/bn_{j}", _need(flow, "flow", f"block_{i}_bn_{j}"), n)
        last = n
        flow_filters = list(base.flow_filters)
    else:
        nblk = _count_blocks(flow)
        if nblk < 2 or nblk % 2:
            raise ValueError(f"flow auto-encoder: {nblk} blocks, expected an even number >= 2")
        cin = int(np.asarray(_need(flow, "flow", "block_1_conv_1")[0]).shape[2])
        prev = cin
        for i in range(1, nblk + 1):
            k1 = _put_conv(w, f"flow/block_{i}/conv_1", _need(flow, "flow", f"block_{i}_conv_1"),
                           bias=False)
            f = int(k1.shape[3])
            if k1.shape != (3, 3, prev, f):
                raise ValueError(f"flow/block_{i}/conv_1/kernel: shape {k1.shape}, expected (3, 3, {prev}, F)")
            _put_bn(w, f"flow/block_{i}/bn_1", _need(flow, "flow", f"block_{i}_bn_1"), f)
            k2 = _put_conv(w, f"flow/block_{i}/conv_2", _need(flow, "flow", f"block_{i}_conv_2"),
                           bias=False)
            if k2.shape != (3, 3, f, f):
                raise ValueError(f"flow/block_{i}/conv_2/kernel: shape {k2.shape}")
            _put_bn(w, f"flow/block_{i}/bn_2", _need(flow, "flow", f"block_{i}_bn_2"), f)
            flow_filters.append(f)
            prev = f
        if "conv_1" in flow:  # odd filter list: one more conv-BN-act (models.py:454-468)
            k = _put_conv(w, "flow/conv_1", flow["conv_1"], bias=False)
            if k.shape[:3] != (3, 3, prev):
                raise ValueError(f"flow/conv_1/kernel: shape {k.shape}")
            prev = int(k.shape[3])
            _put_bn(w, "flow/bn_1", _need(flow, "flow", "bn_1"), prev)
            flow_filters.append(prev)
        last = prev
    kh = _put_conv(w, "flow/conv_2", _need(flow, "flow", "conv_2"), bias=True)
    ks = 1 if arch == "resnet" else 3
    if kh.shape != (ks, ks, last, 32) or w["flow/conv_2/bias"].shape != (32,):
        raise ValueError(f"flow/conv_2: kernel {kh.shape}, expected ({ks}, {ks}, {last}, 32) + bias (32,)")
    if cin % 3:
        raise ValueError(f"flow input has {cin} channels, expected 3 per frame")

    cfg = replace(base, num_flow_inputs=cin // 3, flow_arch=arch,
                  flow_filters=tuple(flow_filters), flow_res_filters=flow_res_filters,
                  flow_res_blocks=flow_res_blocks, gen_filters=nf, gen_blocks=gen_blocks)
    return cfg, w


def layers_from_container(weights: Mapping[str, np.ndarray]) -> Tuple[Dict[str, list], Dict[str, list]]:
    """Inverse direction (tests, and exporting seeded models to a Keras checkout):
    container tensors -> ``{keras layer name: [variables in Keras order]}`` for the
    generator and the flow model."""
    order = {"kernel": 0, "bias": 1, **{v: i for i, v in enumerate(BN_VARS)}}
    out: Dict[str, Dict[str, Dict[int, np.ndarray]]] = {"generator": {}, "flow": {}}
    for name, arr in weights.items():
        prefix, var = name.rsplit("/", 1)
        model, layer = keras_layer_name(prefix)
        out[model].setdefault(layer, {})[order[var]] = arr
    conv = lambda d: {k: [v[i] for i in sorted(v)] for k, v in d.items()}  # noqa: E731
    return conv(out["generator"]), conv(out["flow"])
