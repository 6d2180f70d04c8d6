#!/usr/bin/env python3
"""Keras weights of the reference -> .jupw container for the MI355X runtime.

Runs where the reference's Python dependencies are installed (TensorFlow/Keras,
scripts/training/requirements.txt) -- NOT on the engine's build or GPU boxes.
It imports the reference's own ``models.create_models``
(scripts/training/models.py:1138-1194) from a checkout you point it at, so the
layer list and the variable order are the reference's by construction, then maps
layer names to container tensors with joshupscale_amd.keras_import.

usage:
  export_jupw_from_keras.py --reference /path/to/JoshUpscale \\
      --config train_config.yaml --generator generator --flow flow \\
      --frame-size 270x480 [--flow-pad 8] [--normalize-brightness] [--fp16] out.jupw

``--config`` is a reference training/export config whose entries name the models
and their ``weights:`` files (create_models loads them); ``--generator`` /
``--flow`` are the entry names of the two sub-models.
"""

import argparse
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

from joshupscale_amd import keras_import, model_file as M  # noqa: E402


def keras_layers(model) -> dict:
    """{layer name: [numpy variables in Keras order]} of every weighted layer."""
    return {layer.name: layer.get_weights() for layer in model.layers if layer.weights}


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    ap.add_argument("output")
    ap.add_argument("--reference", required=True, help="checkout of the reference repository")
    ap.add_argument("--config", required=True, help="reference model config (YAML)")
    ap.add_argument("--generator", default="generator")
    ap.add_argument("--flow", default="flow")
    ap.add_argument("--frame-size", default="270x480", help="HxW of the low-resolution frame")
    ap.add_argument("--flow-pad", type=int, default=8, help="flow_pad_factor (models.py:735-744)")
    ap.add_argument("--normalize-brightness", action="store_true")
    ap.add_argument("--bn-eps", type=float, default=1e-3, help="BatchNormalization epsilon (Keras default)")
    ap.add_argument("--fp16", action="store_true", help="compute dtype hint fp16 (default bf16)")
    args = ap.parse_args()

    sys.path.insert(0, os.path.join(args.reference, "scripts", "training"))
    import yaml  # noqa: E402
    import models as ref_models  # the reference's module  # noqa: E402

    with open(args.config, "rt", encoding="utf-8") as f:
        config = yaml.safe_load(f)
    built = ref_models.create_models(config.get("models", config))
    gen, flow = built[args.generator], built[args.flow]
    h, w = (int(x) for x in args.frame_size.lower().split("x"))
    # the `activation` argument of each sub-model's config entry (models.py:261, 337, 489)
    entries = config.get("models", config)

    def activation_of(entry_name):
        entry = entries.get(entry_name, {}) if isinstance(entries, dict) else {}
        return keras_import.activation_fields(entry.get("activation", "relu"))

    (gact, gslope), (fact, fslope) = activation_of(args.generator), activation_of(args.flow)
    base = M.ModelConfig(frame_height=h, frame_width=w, flow_pad_factor=args.flow_pad,
                         normalize_brightness=args.normalize_brightness, bn_eps=args.bn_eps,
                         compute_dtype=M.DTYPE_F16 if args.fp16 else M.DTYPE_BF16,
                         flow_activation=fact, gen_activation=gact,
                         flow_negative_slope=fslope, gen_negative_slope=gslope)
    cfg, weights = keras_import.container_weights(keras_layers(gen), keras_layers(flow), base)
    M.save(args.output, cfg, weights)
    n = sum(v.size for v in weights.values())
    print(f"{args.output}: {len(weights)} tensors, {n / 1e6:.2f} M parameters, "
          f"flow={cfg.flow_arch} {cfg.flow_filters if cfg.flow_arch == 'autoencoder' else cfg.flow_res_blocks}, "
          f"generator {cfg.gen_filters} x {cfg.gen_blocks}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
