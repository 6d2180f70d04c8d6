#!/usr/bin/env python3
"""Activation ranges of a model over a clip: the input of an 8-bit (int8 / fp8 e4m3)
build of the convolutions (the reference's TensorRT flow is
scripts/inference/tensorrt/generate_calibration.py + quantize_int8.py: symmetric
per-tensor activation scales, per-channel weight scales).

Tower: the engine's calibration mode (JU_CALIBRATE=1: one launch per convolution and an
abs-max reduction of every layer's output -- any geometry, ReLU and LeakyReLU models; 640x448
included, where the resident tower does not fit).  --resident uses the resident kernel's
in-kernel maxima instead (ju_debug_set("tower_variant", 5), bf16); the GPU suite holds both
to the oracle's per-layer maxima.  Flow net and generator input: max |x| of the
materialised tensors (ju_read_tensor).  Needs a GPU.

With --write-fp8 OUT.jupw the ranges of the 48 block-convolution inputs are stored in the
container as "generator/fp8_amax" (what the 8-bit tower reads, csrc/fp8.h) and its
compute_dtype hint is set to fp8; --model calibrates an existing container instead of a
seeded preset.

usage: tools/calibrate.py [--preset psp-quality | --model in.jupw] [--frames 16] [--kind smooth]
                          [--write-fp8 out.jupw] > ranges.json
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M, runtime as R  # noqa: E402


def tower_ranges(cfg, weights, clip, resident=False, dtype=R.DTYPE_BF16):
    """max |output| of generator/conv_1 and of every residual-block activation over `clip`
    (1 + 2 * gen_blocks values, execution order) and of some other tensors.  The runtime is
    created with the calibration switches in the environment and they are restored after."""
    env = {"JU_NO_GRAPH": "1"} if resident else {"JU_CALIBRATE": "1"}   # (the variant switch acts on new launches)
    keep = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    lib = R.load_library()
    n_layers = 1 + 2 * cfg.gen_blocks
    tower = np.zeros(n_layers, np.float64)
    others = {}
    names = ["gen_in", "flow"] + [f"flow/block_{i + 1}/a_1" for i in range(len(cfg.flow_filters) - 1)]
    try:
        rt = R.Runtime(M.serialize(cfg, weights), 0, dtype)
        if resident:
            if rt.stat("resident_tower") != 1:
                raise SystemExit("--resident: this geometry does not run the resident tower")
            if lib.ju_debug_set(b"tower_variant", 5) != 0:
                raise SystemExit(lib.ju_last_error().decode())
        try:
            for f in clip:
                rt.process_image(f)
                m = rt.read_tensor("tower_profile")[:n_layers].view(np.uint32).view(np.float32).astype(np.float64)
                tower = np.maximum(tower, m)
                for name in names:
                    try:
                        v = float(np.abs(rt.read_tensor(name)).max())
                    except R.JoshUpscaleError:
                        continue
                    others[name] = max(others.get(name, 0.0), v)
        finally:
            if resident:
                lib.ju_debug_set(b"tower_variant", 0)
        rt.close()
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return tower, others


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="psp-quality")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--kind", default="smooth", choices=["smooth", "noise"])
    ap.add_argument("--model", help="calibrate this container instead of a seeded preset")
    ap.add_argument("--write-fp8", metavar="OUT", help="write the container + generator/fp8_amax")
    ap.add_argument("--resident", action="store_true",
                    help="in-kernel maxima of the resident tower (tower_variant 5) instead of the calibration mode")
    args = ap.parse_args()
    if args.model:
        cfg, weights = M.load(args.model)
        weights.pop("generator/fp8_amax", None)
    else:
        cfg = M.PRESETS[args.preset]
        weights = M.make_seeded_weights(cfg)
    clip = M.synthetic_frames(args.frames, cfg.frame_height, cfg.frame_width, seed=99, kind=args.kind)
    tower, others = tower_ranges(cfg, weights, clip, resident=args.resident)

    def scales(amax):
        return {"amax": round(amax, 6), "int8_scale": round(127.0 / amax, 4) if amax > 0 else None,
                "e4m3_scale": round(448.0 / amax, 4) if amax > 0 else None}

    layer_names = ["generator/conv_1"] + [f"generator/block_{i // 2 + 1}/conv_{i % 2 + 1}"
                                          for i in range(2 * cfg.gen_blocks)]
    w_amax = {}
    for name in layer_names:
        k = weights[name + "/kernel"]
        w_amax[name] = float(np.abs(k).max())   # (per-output-channel scales come from the folded kernels)
    report = {
        "preset": args.preset, "clip": f"{args.frames} frames, {args.kind}, seed 99",
        "note": "activation = largest |post-activation output| of the layer over the clip; "
                "scale = full-range / amax, symmetric per tensor",
        "how": "resident kernel, in-kernel maxima" if args.resident else "calibration mode (JU_CALIBRATE=1)",
        "tower_layers": {n: scales(float(a)) for n, a in zip(layer_names, tower)},
        "other_tensors": {n: scales(a) for n, a in others.items()},
        "weight_amax_unfolded": {n: round(a, 6) for n, a in w_amax.items()},
    }
    if args.write_fp8:
        import dataclasses
        # input of block i's conv_1 = output of tower layer 2i (layer 0 = generator/conv_1),
        # input of its conv_2 = output of layer 2i + 1
        weights["generator/fp8_amax"] = np.asarray(tower[:2 * cfg.gen_blocks], np.float32)
        M.save(args.write_fp8, dataclasses.replace(cfg, compute_dtype=M.DTYPE_FP8), weights)
        report["written"] = args.write_fp8
    print(json.dumps(report, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
