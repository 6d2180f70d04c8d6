"""The oracle against outputs of the reference's own Keras graph -- when somebody has produced them.

tests/golden/make_reference_golden.py imports /root/reference/scripts/training/models.py under real
TensorFlow and writes tests/golden/ref_*.npz.  TensorFlow exists neither in the build container nor on
the GPU boxes, so those files do not exist yet and the comparison SKIPS, loudly: the oracle is
"parity unpinned"  until this test has run against them."""

import glob
import hashlib
import importlib.util
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest

from helpers import M, O, ROOT, oracle_config
from joshupscale_amd import keras_import

GOLD = os.path.join(ROOT, "tests", "golden")
SCRIPT = os.path.join(GOLD, "make_reference_golden.py")
REF_FILES = sorted(glob.glob(os.path.join(GOLD, "ref_*.npz")))


def _script_module():
    spec = importlib.util.spec_from_file_location("make_reference_golden", SCRIPT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(bool(REF_FILES), reason="reference fixtures present: compared below")
def test_reference_fixtures_are_absent_and_parity_is_unpinned():
    pytest.skip("PARITY UNPINNED: no tests/golden/ref_*.npz -- nobody has run tests/golden/make_reference_golden.py "
                "where TensorFlow is installed; the oracle is checked only against the build's own second restatement")


def compare_fixture(path):
    """oracle/ju_oracle.py against one ref_*.npz: float32 TensorFlow against the float64 oracle -- 1e-4 on `output_raw`
    (values in [-0.5, 0.5]); the truncating u8 cast turns a sub-LSB difference at an integer boundary into 1 LSB, on a
    handful of bytes."""
    g = np.load(path)
    kw = json.loads(str(g["config"]))
    if "flow_filters" in kw:
        kw["flow_filters"] = tuple(kw["flow_filters"])
    cfg = M.ModelConfig(**kw)
    wts = M.make_seeded_weights(cfg, seed=42)
    assert hashlib.sha256(M.serialize(cfg, wts)).hexdigest() == str(g["model_sha256"]), \
        "the fixture was generated from other weights than make_seeded_weights(seed=42) gives today"
    sess = O.Session(wts, oracle_config(cfg))
    y0, x0 = (int(v) for v in g["crop"])
    worst = 0.0
    for t, frame in enumerate(g["frames"]):
        out = sess.run(frame)
        raw = sess.last.output_raw
        ref_out, ref_raw = g["output"][t], g["output_raw"][t]
        if y0 or x0 or ref_raw.shape[0] != raw.shape[0]:
            ch, cw = ref_raw.shape[:2]
            raw = raw[y0:y0 + ch, x0:x0 + cw]
            crop = out[y0:y0 + ch, x0:x0 + cw, :3]
            assert str(g["output_sha256"][t])  # (whole-frame digest of the reference's u8 frame: informational)
        else:
            crop = out[..., :3]
            assert hashlib.sha256(np.ascontiguousarray(ref_out)).hexdigest() == str(g["output_sha256"][t])
        worst = max(worst, float(np.abs(raw - ref_raw).max()))
        assert np.abs(raw - ref_raw).max() <= 1e-4, (path, t)
        d = np.abs(crop.astype(int) - ref_out.astype(int))
        assert d.max() <= 1 and np.mean(d > 0) <= 1e-3, (path, t, d.max(), np.mean(d > 0))
    return worst


@pytest.mark.parametrize("path", REF_FILES or [None], ids=lambda p: os.path.basename(p) if p else "none")
def test_oracle_matches_the_reference_keras_graph(path):
    if path is None:
        pytest.skip("no reference fixtures (see test_reference_fixtures_are_absent_and_parity_is_unpinned)")
    compare_fixture(path)


def test_fixture_script_runs_end_to_end_against_a_stand_in(tmp_path, monkeypatch):
    """The script's plumbing, without TensorFlow: tests/fake_reference.py plays `models` and `tensorflow` (layer
    lists and shapes of the reference constructors, `set_weights` / `get_weights`, the inference model's call
    convention) and computes with the oracle itself -- so this pins NOTHING about parity; it shows that the layer-name
    mapping, the importer round trip, the recurrent stepping (`output_raw` -> `pre_gen`, `last_frames` shifted), the
    crop of the large case and the fixture format work, and that a fixture the script writes is one the comparison
    accepts."""
    import fake_reference
    monkeypatch.setitem(sys.modules, "tensorflow", fake_reference.fake_tensorflow())
    mod = _script_module()
    for name in ("small_autoencoder", "small_resnet", "small_brightness", "small_gen32_ae5_in2", "small_res128_in5",
                 "small_lrelu"):
        path = mod.run_case(fake_reference, name, out_dir=str(tmp_path))
        assert os.path.basename(path) == f"ref_{name}.npz"
        assert compare_fixture(path) <= 1e-6   # float32 storage of the oracle's own float64 values
        g = np.load(path)
        assert g["output"].dtype == np.uint8 and g["output_raw"].dtype == np.float32 and "fake" in str(g["generator"])
    # the crop path of the full-size case, on a geometry that takes seconds: patch the case table
    monkeypatch.setitem(mod.CASES, "full_psp_4blocks", (dict(frame_height=60, frame_width=96, gen_blocks=1), 2, "smooth"))
    path = mod.run_case(fake_reference, "full_psp_4blocks", out_dir=str(tmp_path))
    g = np.load(path)
    assert g["output_raw"].shape[1:3] == (mod.CROP, mod.CROP) and tuple(g["crop"]) != (0, 0)
    compare_fixture(path)
    # a stand-in whose layer list differs from the container's names is refused, not silently mis-loaded
    broken = types.SimpleNamespace(**{k: getattr(fake_reference, k) for k in dir(fake_reference) if k.startswith("get_")})
    real = fake_reference.get_generator_resnet

    def renamed(**kw):
        m = real(**kw)
        m.get_layer("bn_2").name = "batch_normalization_7"
        return m
    broken.get_generator_resnet = renamed
    with pytest.raises(SystemExit, match="do not match the container's tensor names"):
        mod.run_case(broken, "small_autoencoder", out_dir=str(tmp_path))


def test_generator_cases_are_loadable_and_round_trip_through_the_importer():
    """What of the off-box script can run here: every case is a configuration the loader's Python twin
    accepts, and its seeded weights survive container -> Keras layer lists -> container exactly (the
    mapping the script applies with `set_weights` / `get_weights`)."""
    mod = _script_module()
    assert "full_psp_4blocks" in mod.CASES and len(mod.CASES) >= 8
    for name, (kw, n_frames, kind) in mod.CASES.items():
        cfg = M.ModelConfig(**kw)
        wts = M.make_seeded_weights(cfg, seed=42)
        cfg_back, _ = M.deserialize(M.serialize(cfg, wts))
        assert cfg_back.gen_filters == cfg.gen_filters and cfg_back.flow_filters == tuple(cfg.flow_filters)
        gen_layers, flow_layers = keras_import.layers_from_container(wts)
        cfg2, w2 = keras_import.container_weights(gen_layers, flow_layers, cfg)
        assert cfg2 == cfg, name
        assert set(w2) == set(wts) and all(np.array_equal(w2[k], wts[k]) for k in wts), name
        assert n_frames >= 3 and kind in ("smooth", "noise")
        spec = mod.activation_arg(cfg.gen_activation, cfg.gen_negative_slope)
        assert keras_import.activation_fields(spec)[0] == cfg.gen_activation


def test_generator_script_says_what_it_needs_without_tensorflow():
    try:
        import tensorflow  # noqa: F401
        pytest.skip("TensorFlow is installed here: run the script instead")
    except ImportError:
        pass
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "scripts", "training")):
        out = subprocess.run([sys.executable, SCRIPT, "--reference", "/nonexistent"], capture_output=True, text=True)
        assert out.returncode != 0 and "models.py not found" in out.stderr
        return
    out = subprocess.run([sys.executable, SCRIPT, "--reference", ref], capture_output=True, text=True)
    assert out.returncode != 0 and "needs the reference's Python dependencies" in out.stderr, out.stderr
