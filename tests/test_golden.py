"""The restatements reproduce the committed golden vectors (CPU)."""

import hashlib
import os

import numpy as np
import pytest

from helpers import M, O, ROOT, oracle_config, small_config

GOLD = os.path.join(ROOT, "tests", "golden")
SMALL = {
    "small_autoencoder": small_config(),
    "small_resnet": small_config(flow_arch="resnet", flow_pad_factor=0, flow_res_blocks=2,
                                 frame_height=34, frame_width=50),
    "small_noise": small_config(gen_blocks=2),
    "small_lrelu": small_config(flow_activation="lrelu", gen_activation="lrelu", gen_negative_slope=0.2),
}
FULL = {"psp-quality": "full_psp_quality", "psp-fast": "full_psp_fast",
        "psp-quality-flowres": "full_psp_quality_flowres", "ps2-quality": "full_ps2_quality"}


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def test_numpy_oracle_reproduces_the_fp8_golden():
    """Pins the 8-bit restatement (e4m3 rounding, scale rules, where it quantises): any
    change to it shows here before it silently moves the GPU parity target."""
    g = load("small_fp8")
    cfg = small_config(gen_blocks=4)
    wts = M.make_seeded_weights(cfg, seed=42)
    assert hashlib.sha256(M.serialize(cfg, wts)).hexdigest() == str(g["model_sha256"])
    sess = O.Session(wts, oracle_config(cfg, fp8_tower=True))
    for t, frame in enumerate(g["frames"]):
        assert np.array_equal(sess.run(frame), g["outputs"][t]), t
    assert np.abs(sess.last.output_raw - g["output_raw_last"]).max() < 1e-6
    plain = O.Session(wts, oracle_config(cfg))
    assert not np.array_equal(plain.run(g["frames"][0]), g["outputs"][0])   # it IS a different model


def test_numpy_oracle_reproduces_the_fp8_lrelu_golden():
    """The 8-bit scheme for `activation: lrelu` generators (LeakyReLU in full precision, e4m3
    clamped at +-448): pinned like the ReLU one, and reproduced by the PyTorch restatement."""
    from torch_restatement import TorchSession
    g = load("small_fp8_lrelu")
    cfg = small_config(gen_blocks=3, gen_activation="lrelu", gen_negative_slope=0.2)
    wts = M.make_seeded_weights(cfg, seed=42)
    assert hashlib.sha256(M.serialize(cfg, wts)).hexdigest() == str(g["model_sha256"])
    sess = O.Session(wts, oracle_config(cfg, fp8_tower=True))
    tsess = TorchSession(wts, oracle_config(cfg, fp8_tower=True))
    for t, frame in enumerate(g["frames"]):
        assert np.array_equal(sess.run(frame), g["outputs"][t]), t
        assert np.array_equal(tsess.run(frame), g["outputs"][t]), t
    relu8 = O.Session(M.make_seeded_weights(small_config(gen_blocks=3)), oracle_config(small_config(gen_blocks=3), fp8_tower=True))
    assert not np.array_equal(relu8.run(g["frames"][0]), g["outputs"][0])


@pytest.mark.parametrize("name", sorted(SMALL))
def test_numpy_oracle_reproduces_small_goldens(name):
    g = load(name)
    cfg = SMALL[name]
    wts = M.make_seeded_weights(cfg, seed=42)
    assert hashlib.sha256(M.serialize(cfg, wts)).hexdigest() == str(g["model_sha256"]), \
        "seeded weights changed: regenerate with tests/golden/make_golden.py"
    sess = O.Session(wts, oracle_config(cfg))
    for t, frame in enumerate(g["frames"]):
        out = sess.run(frame)
        assert np.array_equal(out, g["outputs"][t]), (name, t)
    assert np.abs(sess.last.output_raw - g["output_raw_last"]).max() < 1e-6
    assert np.abs(sess.last.flow - g["flow_last"]).max() < 1e-5


@pytest.mark.parametrize("name", sorted(SMALL))
def test_c_restatement_matches_small_goldens(name):
    from oracle.c_binding import CSession
    g = load(name)
    cfg = SMALL[name]
    cs = CSession(M.serialize(cfg, M.make_seeded_weights(cfg, seed=42)), cfg.frame_height,
                  cfg.frame_width)
    for t, frame in enumerate(g["frames"]):
        out = cs.run(frame)
        d = np.abs(out.astype(int) - g["outputs"][t].astype(int))
        assert d.max() <= 1 and np.mean(d > 0) < 0.01, (name, t)   # fp32 vs fp64 truncation
    assert np.abs(cs.output_raw() - g["output_raw_last"]).max() < 5e-5


@pytest.mark.parametrize("preset", sorted(FULL))
def test_full_size_goldens_were_generated_twice_and_agree(preset):
    """Every full-size fixture was produced by BOTH restatements (numpy float64 oracle and
    the independent PyTorch one, tests/golden/make_golden.py): whole-frame digests of each
    are committed and must agree -- the only independent anchor available while the
    reference itself cannot run here."""
    g = load(FULL[preset])
    cfg = M.PRESETS[preset]
    assert len(g["out_sha256_numpy"]) == int(g["n_frames"]) == len(g["out_sha256_torch"])
    assert [str(x) for x in g["out_sha256_numpy"]] == [str(x) for x in g["out_sha256_torch"]]
    assert not g["bytes_differing"].any() and g["raw_max_diff"].max() < 1e-9
    assert g["crops_u8"].shape == (int(g["n_frames"]), 6, 64, 64, 3)
    H, W = 4 * cfg.frame_height, 4 * cfg.frame_width
    assert all(0 <= y <= H - 64 and 0 <= x <= W - 64 for y, x in g["crops"])


FULL_FP8 = {"psp-quality": "full_psp_quality_fp8", "ps2-quality": "full_ps2_quality_fp8"}


@pytest.mark.parametrize("preset", sorted(FULL_FP8))
def test_full_size_8bit_goldens_were_generated_twice_and_agree(preset):
    """The 8-bit tower (BASELINE.json config 5) at the sizes it is quoted on, 480x270 and
    640x448: whole-frame digests of the numpy oracle's restatement of the scheme and of the
    PyTorch one (its own float8_e4m3fn rounding) agree byte for byte; same clip and crops as
    the float fixtures, and it IS a different model (the crops differ from the float ones)."""
    g = load(FULL_FP8[preset])
    gf = load(FULL[preset])
    n = int(g["n_frames"])
    assert n == int(gf["n_frames"]) and str(g["model_sha256"]) == str(gf["model_sha256"])
    assert str(g["frames_sha256"]) == str(gf["frames_sha256"])       # the SAME clip as the float fixture
    assert [str(x) for x in g["out_sha256_numpy"]] == [str(x) for x in g["out_sha256_torch"]]
    assert not g["bytes_differing"].any() and g["raw_max_diff"].max() < 1e-9
    assert np.array_equal(g["crops"], gf["crops"]) and int(g["seed"]) == int(gf["seed"])
    d = g["crops_u8"].astype(int) - gf["crops_u8"][:n].astype(int)
    psnr = 10 * np.log10(255.0 ** 2 / np.mean(d.astype(float) ** 2))
    assert 40.0 < psnr < 60.0, psnr          # quantisation noise of 48 e4m3 convolutions, not a broken model


@pytest.mark.parametrize("preset", ["psp-quality", "psp-fast"])
def test_c_restatement_matches_full_size_golden_first_frame(preset):
    """One 480x270 frame of the BASELINE.json configurations (a few seconds each)."""
    from oracle.c_binding import CSession
    g = load(FULL[preset])
    cfg = M.PRESETS[preset]
    blob = M.serialize(cfg, M.make_seeded_weights(cfg, seed=42))
    assert hashlib.sha256(blob).hexdigest() == str(g["model_sha256"])
    frames = M.synthetic_frames(int(g["n_frames"]), 270, 480, seed=int(g["seed"]), kind="smooth")
    assert hashlib.sha256(frames.tobytes()).hexdigest() == str(g["frames_sha256"])
    cs = CSession(blob, 270, 480)
    out = cs.run(frames[0])
    raw = cs.output_raw()
    for k, (y, x) in enumerate(g["crops"]):
        d = np.abs(out[y:y + 64, x:x + 64, :3].astype(int) - g["crops_u8"][0, k].astype(int))
        assert d.max() <= 1
        assert np.abs(raw[y:y + 64, x:x + 64] - g["crops_raw"][0, k]).max() < 1e-4
    assert np.abs(out[..., :3].reshape(-1, 3).mean(0) - g["means"][0]).max() < 0.01
