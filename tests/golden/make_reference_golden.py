#!/usr/bin/env python3
"""Reference-anchored golden vectors: outputs of the reference's OWN Keras graph.

THIS SCRIPT HAS NEVER RUN IN THE BUILD CONTAINER: it needs TensorFlow / Keras
(scripts/training/requirements.txt: tensorflow-cpu==2.18.0), which neither the build
container nor the GPU boxes have.  Until somebody runs it where those are installed,
the oracle stays "parity unpinned" (DESIGN.md section 2).  What it does, the day it runs:

  1. imports the reference's `models` module from a checkout (--reference, default
     /root/reference) -- nothing of it is copied, nothing of it travels;
  2. for every case below builds the flow model and the generator with the reference's own
     constructors (models.py:257-331 get_flow_resnet, :334-481 get_flow_autoencoder,
     :484-595 get_generator_resnet) and wires them with get_inference_model
     (:680-829, skip_processing=False: u8 frame in, u8 frame out);
  3. loads THIS repository's seeded weights (model_file.make_seeded_weights) into the Keras
     layers by layer name -- the inverse of keras_import.container_weights
     (`layers_from_container` + `layer.set_weights`), which also exercises the Keras-facing
     half of the importer: the script re-reads the layers with `layer.get_weights()`, maps
     them back through `container_weights` and requires the round trip to be exact;
  4. steps the recurrent loop exactly as the reference's own drivers do
     (keras_models.py:58-73; scripts/inference/onnx/inference.py:72-94): zero state, feed
     [cur_frame, last_output] + last_frames, take `output_raw` as the next `pre_gen` and
     `last_frames` as the next history;
  5. writes tests/golden/ref_<case>.npz: plain arrays only -- the case's configuration (JSON),
     the SHA-256 of the container the weights serialise to, the input frames, and per frame the
     reference's `output` (u8), `output_raw` and `pre_warp` (float32; crops for the large case).

tests/test_reference_golden.py compares oracle/ju_oracle.py with every ref_*.npz it finds
and SKIPS LOUDLY when there is none.

usage (on a machine with the reference's requirements installed):
    python tests/golden/make_reference_golden.py [--reference /path/to/JoshUpscale] [--only NAME]
"""

import argparse
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from joshupscale_amd import keras_import  # noqa: E402
from joshupscale_amd import model_file as M  # noqa: E402

LRELU = dict(flow_activation="lrelu", gen_activation="lrelu", gen_negative_slope=0.2)
# name -> (ModelConfig keyword arguments, frames, clip kind).  The small cases are the
# geometries of tests/test_gpu_parity.py; "full" is the benchmark geometry with a short tower
# (a CPU Keras run of 24 blocks at 480x270 takes minutes per frame and pins nothing more).
CASES = {
    "small_autoencoder": (dict(frame_height=30, frame_width=48, gen_blocks=3), 4, "smooth"),
    "small_resnet": (dict(frame_height=34, frame_width=50, gen_blocks=3, flow_arch="resnet",
                          flow_pad_factor=0, flow_res_blocks=2), 4, "smooth"),
    "small_ragged": (dict(frame_height=17, frame_width=33, gen_blocks=2), 4, "smooth"),
    "small_noise": (dict(frame_height=30, frame_width=48, gen_blocks=2), 3, "noise"),
    "small_lrelu": (dict(frame_height=30, frame_width=48, gen_blocks=3, **LRELU), 3, "smooth"),
    "small_brightness": (dict(frame_height=30, frame_width=48, gen_blocks=2, normalize_brightness=True), 3, "smooth"),
    "small_gen32_ae5_in2": (dict(frame_height=30, frame_width=48, gen_blocks=2, gen_filters=32,
                                 flow_filters=(32, 64, 128, 64, 32), num_flow_inputs=2), 3, "smooth"),
    "small_res128_in5": (dict(frame_height=34, frame_width=50, gen_blocks=2, flow_arch="resnet", flow_pad_factor=0,
                              flow_res_filters=128, flow_res_blocks=2, num_flow_inputs=5), 3, "smooth"),
    "full_psp_4blocks": (dict(frame_height=270, frame_width=480, gen_blocks=4), 3, "smooth"),
}
CROP = 96  # HR crop edge stored for cases whose float tensors would not be "small fixtures"


def activation_arg(name: str, slope: float):
    """ModelConfig fields -> the reference's `Activation` spec (models.py:20, 36-60)."""
    return "relu" if name == "relu" else {"name": "lrelu", "negative_slope": float(slope)}


def build_reference_model(ref_models, cfg: M.ModelConfig, weights):
    if cfg.flow_arch == "autoencoder":
        flow = ref_models.get_flow_autoencoder(
            num_inputs=cfg.num_flow_inputs, filters=list(cfg.flow_filters),
            activation=activation_arg(cfg.flow_activation, cfg.flow_negative_slope), name="flow")
    else:
        flow = ref_models.get_flow_resnet(
            num_inputs=cfg.num_flow_inputs, num_filters=cfg.flow_res_filters,
            num_res_blocks=cfg.flow_res_blocks,
            activation=activation_arg(cfg.flow_activation, cfg.flow_negative_slope), name="flow")
    gen = ref_models.get_generator_resnet(
        num_filters=cfg.gen_filters, num_res_blocks=cfg.gen_blocks,
        activation=activation_arg(cfg.gen_activation, cfg.gen_negative_slope), name="generator")
    gen_layers, flow_layers = keras_import.layers_from_container(weights)
    for model, layers in ((gen, gen_layers), (flow, flow_layers)):
        weighted = {layer.name for layer in model.layers if layer.weights}
        if weighted != set(layers):
            raise SystemExit(f"{model.name}: Keras layers with variables {sorted(weighted ^ set(layers))} "
                             "do not match the container's tensor names")
        for name, variables in layers.items():
            layer = model.get_layer(name)
            eps = getattr(layer, "epsilon", None)
            if eps is not None and abs(float(eps) - cfg.bn_eps) > 1e-12:
                raise SystemExit(f"{model.name}/{name}: BatchNormalization epsilon {eps}, the container says {cfg.bn_eps}")
            layer.set_weights([np.asarray(v, np.float32) for v in variables])
    # the Keras-facing half of the importer: layers -> container, exact round trip
    read = lambda m: {l.name: l.get_weights() for l in m.layers if l.weights}  # noqa: E731
    cfg2, w2 = keras_import.container_weights(read(gen), read(flow), cfg)
    if cfg2 != cfg or set(w2) != set(weights) or any(not np.array_equal(w2[k], weights[k]) for k in weights):
        raise SystemExit("keras_import.container_weights(layers of the Keras models) != the seeded container")
    inference = ref_models.get_inference_model(
        gen, flow, skip_processing=False, frame_height=cfg.frame_height, frame_width=cfg.frame_width,
        flow_pad_factor=cfg.flow_pad_factor or None, normalize_brightness=cfg.normalize_brightness)
    return inference


def run_case(ref_models, name: str, out_dir: str = HERE) -> str:
    import tensorflow as tf

    kw, n_frames, kind = CASES[name]
    cfg = M.ModelConfig(**kw)
    weights = M.make_seeded_weights(cfg, seed=42)
    blob = M.serialize(cfg, weights)
    model = build_reference_model(ref_models, cfg, weights)
    h, w = cfg.frame_height, cfg.frame_width
    ph, pw = cfg.padded_height, cfg.padded_width
    frames = M.synthetic_frames(n_frames, h, w, seed=1234, kind=kind)
    # zero state (keras_models.py:58-61; core/include/JoshUpscale/core/cuda.h:69-72)
    last_frames = [tf.zeros((1, ph, pw, 3))] * (cfg.num_flow_inputs - 1)
    last_output = tf.zeros((1, 4 * h, 4 * w, 3))
    outs, raws, warps = [], [], []
    for t in range(n_frames):
        cur = tf.constant(frames[t][None, ..., :3])  # B,G,R bytes; the X byte never enters the graph
        o = model([cur, last_output] + list(last_frames), training=False)
        last_output = o["output_raw"]
        last_frames = o["last_frames"]
        outs.append(np.asarray(o["output"][0]))
        raws.append(np.asarray(o["output_raw"][0], np.float32))
        warps.append(np.asarray(o["pre_warp"][0], np.float32))
    out = np.stack(outs)
    raw = np.stack(raws)
    warp = np.stack(warps)
    y0 = x0 = 0
    if raw[0].size > 4 * CROP * CROP * 3:  # big case: whole u8 frames' digests + float crops
        y0, x0 = (4 * h - CROP) // 2, (4 * w - CROP) // 3
        raw = raw[:, y0:y0 + CROP, x0:x0 + CROP]
        warp = warp[:, y0:y0 + CROP, x0:x0 + CROP]
    path = os.path.join(out_dir, f"ref_{name}.npz")
    np.savez_compressed(
        path,
        config=json.dumps(kw, sort_keys=True), clip=json.dumps({"kind": kind, "seed": 1234, "frames": n_frames}),
        model_sha256=hashlib.sha256(blob).hexdigest(),
        frames=frames, output=out if y0 == 0 else out[:, y0:y0 + CROP, x0:x0 + CROP],
        output_sha256=np.array([hashlib.sha256(np.ascontiguousarray(f)).hexdigest() for f in out]),
        output_raw=raw, pre_warp=warp, crop=np.array([y0, x0], np.int64),
        generator=f"tensorflow {tf.__version__}; reference models.get_inference_model, seeded weights (seed 42)")
    return path


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    ap.add_argument("--reference", default="/root/reference", help="checkout of itmo153277/JoshUpscale")
    ap.add_argument("--only", action="append", choices=sorted(CASES), help="generate only these cases")
    args = ap.parse_args()
    training = os.path.join(args.reference, "scripts", "training")
    if not os.path.isfile(os.path.join(training, "models.py")):
        raise SystemExit(f"{training}/models.py not found: point --reference at a checkout of the reference")
    try:
        import tensorflow  # noqa: F401
        import keras  # noqa: F401
    except ImportError as e:
        raise SystemExit(f"this script needs the reference's Python dependencies ({e}); "
                         "run it where scripts/training/requirements.txt is installed")
    sys.path.insert(0, training)
    import models as ref_models  # the reference's own module, imported where it lies

    for name in args.only or sorted(CASES):
        print(run_case(ref_models, name))
    return 0


if __name__ == "__main__":
    sys.exit(main())
