"""Keras-variable -> container mapping (joshupscale_amd/keras_import.py): pure numpy,
no Keras needed.  The Keras-facing shim (tools/export_jupw_from_keras.py) only
collects ``layer.get_weights()`` by layer name and calls this mapping."""
import os

import numpy as np
import pytest

from joshupscale_amd import keras_import as K
from joshupscale_amd import model_file as M


def small(flow_arch="autoencoder"):
    if flow_arch == "autoencoder":
        return M.ModelConfig(frame_height=30, frame_width=48, flow_filters=(32, 64, 32),
                             gen_blocks=2, flow_pad_factor=2)
    return M.ModelConfig(frame_height=30, frame_width=48, flow_arch="resnet", flow_res_blocks=2,
                         flow_res_filters=32, gen_blocks=3, flow_pad_factor=0)


def test_name_map_follows_the_reference_layer_names():
    assert K.container_name("generator", "block_12_conv_2") == "generator/block_12/conv_2"
    assert K.container_name("generator", "conv_trans_1") == "generator/conv_trans_1"
    assert K.container_name("flow", "block_3_bn_1") == "flow/block_3/bn_1"
    assert K.container_name("flow", "conv_2") == "flow/conv_2"
    assert K.keras_layer_name("generator/block_7/bn_2") == ("generator", "block_7_bn_2")


@pytest.mark.parametrize("arch", ["autoencoder", "resnet"])
def test_round_trip_through_keras_layer_dicts(arch):
    cfg = small(arch)
    w = M.make_seeded_weights(cfg, seed=5)
    gen, flow = K.layers_from_container(w)
    # Keras order: Conv2D [kernel(, bias)], BatchNormalization [gamma, beta, mean, variance]
    assert [a.shape for a in gen["block_1_bn_1"]] == [(64,)] * 4
    assert np.array_equal(gen["block_1_bn_1"][2], w["generator/block_1/bn_1/moving_mean"])
    assert len(gen["conv_1"]) == 1 and len(gen["conv_trans_2"]) == 2 and len(flow["conv_2"]) == 2
    base = M.ModelConfig(frame_height=30, frame_width=48, flow_pad_factor=cfg.flow_pad_factor)
    cfg2, w2 = K.container_weights(gen, flow, base)
    assert cfg2 == cfg                       # architecture recovered from the shapes
    assert set(w2) == set(w)
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    # and the container built from it is byte-identical
    assert M.serialize(cfg2, {k: w2[k] for k in w}) == M.serialize(cfg, w)


def test_default_architecture_is_recognised():
    cfg = M.PRESETS["psp-quality"]
    w = M.make_seeded_weights(cfg)
    gen, flow = K.layers_from_container(w)
    cfg2, _ = K.container_weights(gen, flow, M.ModelConfig())
    assert cfg2.flow_filters == (32, 64, 128, 256, 128, 64, 32) and cfg2.gen_blocks == 24
    assert cfg2.num_flow_inputs == 4 and cfg2.flow_arch == "autoencoder"


def test_errors_name_the_offending_layer():
    cfg = small()
    gen, flow = K.layers_from_container(M.make_seeded_weights(cfg))
    bad = dict(gen)
    bad["block_2_bn_1"] = bad["block_2_bn_1"][:2]            # BN without moving statistics
    with pytest.raises(ValueError, match="block_2/bn_1"):
        K.container_weights(bad, flow, cfg)
    bad = dict(gen)
    bad["conv_trans_1"] = [np.transpose(gen["conv_trans_1"][0], (0, 1, 3, 2))]   # [kh,kw,cin,cout]
    with pytest.raises(ValueError, match="conv_trans_1"):
        K.container_weights(bad, flow, cfg)
    bad = dict(flow)
    del bad["block_2_conv_2"]
    with pytest.raises(KeyError, match="block_2_conv_2"):
        K.container_weights(gen, bad, cfg)
    bad = dict(gen)
    bad["conv_1"] = list(gen["conv_1"]) + [np.zeros(64, np.float32)]            # use_bias=True
    with pytest.raises(ValueError, match="generator/conv_1"):
        K.container_weights(bad, flow, cfg)


def test_activation_spec_of_the_reference_configs():
    """models.py:20, 36-60: a name, or {"name": ..., **LeakyReLU kwargs}."""
    assert K.activation_fields("relu") == ("relu", M.DEFAULT_NEGATIVE_SLOPE)
    assert K.activation_fields("lrelu") == ("lrelu", 0.3)
    assert K.activation_fields({"name": "lrelu", "negative_slope": 0.2}) == ("lrelu", 0.2)
    assert K.activation_fields({"name": "lrelu", "alpha": 0.1}) == ("lrelu", 0.1)
    with pytest.raises(ValueError, match="Unknown activation"):
        K.activation_fields("gelu")
    with pytest.raises(TypeError):
        K.activation_fields(3)
    with pytest.raises(ValueError, match="unsupported"):
        K.activation_fields({"name": "relu", "max_value": 6.0})
    # the activation travels in `base` (the weights cannot tell it)
    cfg = M.ModelConfig(frame_height=30, frame_width=48, flow_filters=(32, 64, 32), gen_blocks=2,
                        flow_pad_factor=2, gen_activation="lrelu", gen_negative_slope=0.2)
    gen, flow = K.layers_from_container(M.make_seeded_weights(cfg, seed=5))
    cfg2, _ = K.container_weights(gen, flow, M.ModelConfig(
        frame_height=30, frame_width=48, flow_pad_factor=2, gen_activation="lrelu", gen_negative_slope=0.2))
    assert cfg2 == cfg


def test_export_tool_end_to_end_against_a_stand_in_checkout(tmp_path):
    """tools/export_jupw_from_keras.py had never executed (it imports the reference's `models` under TensorFlow).
    Here it runs as the user would run it -- its own argument parsing, YAML config, `create_models`, layer walk,
    `container_weights`, `M.save` -- against a stand-in checkout whose scripts/training/models.py is
    tests/fake_reference.py (layer lists and shapes of the reference constructors, no TensorFlow, no arithmetic):
    the container it writes must be byte for byte the one the seeded weights serialise to."""
    import subprocess
    import sys

    import yaml

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    training = tmp_path / "ref" / "scripts" / "training"
    training.mkdir(parents=True)
    (training / "models.py").write_text(
        "import sys\nsys.path.insert(0, %r)\nsys.path.insert(0, %r)\nfrom fake_reference import *  # noqa\n"
        % (os.path.join(root, "tests"), root))
    cfg = M.ModelConfig(frame_height=34, frame_width=50, flow_pad_factor=0, gen_blocks=2, gen_filters=32,
                        flow_arch="resnet", flow_res_filters=96, flow_res_blocks=1, num_flow_inputs=3,
                        gen_activation="lrelu", gen_negative_slope=0.2, compute_dtype=M.DTYPE_F16)
    wts = M.make_seeded_weights(cfg, seed=5)
    gen_layers, flow_layers = K.layers_from_container(wts)
    for name, layers in (("gen", gen_layers), ("flow", flow_layers)):
        np.savez(tmp_path / f"{name}.npz", **{f"{k}/{i}": v for k, vs in layers.items() for i, v in enumerate(vs)})
    config = {"models": {
        "generator": {"name": "generator-resnet", "num_filters": 32, "num_res_blocks": 2,
                      "activation": {"name": "lrelu", "negative_slope": 0.2}, "weights": str(tmp_path / "gen.npz")},
        "flow": {"name": "flow-resnet", "num_inputs": 3, "num_filters": 96, "num_res_blocks": 1,
                 "weights": str(tmp_path / "flow.npz")}}}
    (tmp_path / "config.yaml").write_text(yaml.safe_dump(config))
    out = tmp_path / "model.jupw"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "export_jupw_from_keras.py"), str(out),
                        "--reference", str(tmp_path / "ref"), "--config", str(tmp_path / "config.yaml"),
                        "--frame-size", "34x50", "--flow-pad", "0", "--fp16"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "flow=resnet" in r.stdout and "generator 32 x 2" in r.stdout
    cfg2, wts2 = M.load(str(out))
    # (bn_eps and the slope come back as float32: the bytes are what must agree)
    assert (cfg2.gen_filters, cfg2.flow_res_filters, cfg2.num_flow_inputs, cfg2.gen_activation) == (32, 96, 3, "lrelu")
    assert open(out, "rb").read() == M.serialize(cfg, {k: wts[k] for k in wts2})
    # a missing argument is a usage error, not a traceback from somewhere inside
    bad = subprocess.run([sys.executable, os.path.join(root, "tools", "export_jupw_from_keras.py"), str(out)],
                         capture_output=True, text=True)
    assert bad.returncode == 2 and "--reference" in bad.stderr
