"""The three restatements agree with each other (CPU)."""

import os

import numpy as np
import pytest

from helpers import M, O, err, oracle_config, small_config, u8_stats
from torch_restatement import TorchSession


LRELU = dict(flow_activation="lrelu", gen_activation="lrelu", gen_negative_slope=0.2)


@pytest.mark.parametrize("arch,pad,h,w,extra", [
    ("autoencoder", 8, 30, 48, {}), ("resnet", 0, 20, 24, {}), ("autoencoder", 8, 17, 33, {}),
    ("autoencoder", 8, 30, 48, LRELU), ("resnet", 0, 20, 24, LRELU),
    ("autoencoder", 8, 17, 33, dict(gen_activation="lrelu"))])
def test_numpy_oracle_vs_torch_restatement(arch, pad, h, w, extra):
    cfg = small_config(frame_height=h, frame_width=w, flow_arch=arch, flow_pad_factor=pad,
                       flow_res_blocks=2, **extra)
    wts = M.make_seeded_weights(cfg)
    s = O.Session(wts, oracle_config(cfg))
    ts = TorchSession(wts, oracle_config(cfg))
    frames = M.synthetic_frames(3, h, w, kind="smooth")
    for t in range(3):
        a = s.run(frames[t])
        b = ts.run(frames[t])
        raw = ts.output_raw[0].permute(1, 2, 0).numpy()
        assert np.abs(raw - s.last.output_raw).max() <= 1e-9   # float64 both sides
        assert np.abs(a.astype(int) - b.astype(int)).max() <= 1  # truncation boundary only
        assert (a[..., 3] == 0).all()


# model shapes beside the defaults (the reference constructors are parametric in all of them:
# models.py:257-263, 334-339, 364-365, 449-468, 484-491): all three restatements must agree there too
WIDTHS = [
    dict(gen_filters=32), dict(gen_filters=128), dict(gen_blocks=0),
    dict(flow_filters=(32, 64, 32)), dict(flow_filters=(32, 64, 64, 32)), dict(flow_filters=(32, 32)),
    dict(flow_filters=(64, 96, 128, 96, 64), frame_height=34, frame_width=50),
    dict(flow_arch="resnet", flow_pad_factor=0, flow_res_filters=32, flow_res_blocks=2, frame_height=34, frame_width=50),
    dict(flow_arch="resnet", flow_pad_factor=8, flow_res_filters=96, flow_res_blocks=1),
    dict(num_flow_inputs=1), dict(num_flow_inputs=2), dict(num_flow_inputs=5),
    dict(num_flow_inputs=1, flow_arch="resnet", flow_pad_factor=0, flow_res_blocks=1),
    dict(gen_filters=128, flow_filters=(32, 64, 128, 64, 32), num_flow_inputs=2, **LRELU),
]


@pytest.mark.parametrize("kw", WIDTHS, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items())[:60])
def test_restatements_agree_on_nondefault_widths(kw):
    from oracle.c_binding import CSession
    cfg = small_config(**kw)
    wts = M.make_seeded_weights(cfg)
    s = O.Session(wts, oracle_config(cfg))
    ts = TorchSession(wts, oracle_config(cfg))
    cs = CSession(M.serialize(cfg, wts), cfg.frame_height, cfg.frame_width)
    for f in M.synthetic_frames(3, cfg.frame_height, cfg.frame_width, seed=7, kind="smooth"):
        a, b, c = s.run(f), ts.run(f), cs.run(f)
        raw = ts.output_raw[0].permute(1, 2, 0).numpy()
        assert np.abs(raw - s.last.output_raw).max() <= 1e-9
        assert np.abs(a.astype(int) - b.astype(int)).max() <= 1
        assert err(cs.output_raw(), s.last.output_raw)["max_abs"] < 2e-5
        assert u8_stats(c, a)["max"] <= 1 and (c[..., 3] == 0).all()


def test_numpy_oracle_vs_torch_restatement_8bit_tower():
    """The 8-bit tower scheme (csrc/fp8.h) restated twice: the oracle's frexp-based e4m3
    rounding against PyTorch's own float8_e4m3fn conversion, scales derived independently.
    Both fold BatchNorm the way the loader does (float64, one rounding to float32), so the
    discontinuous quantiser sees identical inputs and the frames are EQUAL."""
    amax = np.linspace(0.8, 9.0, 8).astype(np.float32)          # a calibration tensor too
    for extra, leaky in (({}, {}), ({"generator/fp8_amax": amax}, {}),
                         ({}, dict(gen_activation="lrelu", gen_negative_slope=0.2))):   # + a LeakyReLU generator
        cfg = small_config(gen_blocks=4, **leaky)
        wts = M.make_seeded_weights(cfg)
        w = dict(wts, **extra)
        s = O.Session(w, oracle_config(cfg, fp8_tower=True))
        ts = TorchSession(w, oracle_config(cfg, fp8_tower=True))
        for f in M.synthetic_frames(3, 30, 48, seed=14, kind="smooth"):
            a, b = s.run(f), ts.run(f)
            raw = ts.output_raw[0].permute(1, 2, 0).numpy()
            assert np.abs(raw - s.last.output_raw).max() <= 1e-9
            assert np.array_equal(a, b)


@pytest.mark.parametrize("arch,pad,extra", [("autoencoder", 8, {}), ("resnet", 0, {}),
                                            ("autoencoder", 8, LRELU), ("resnet", 0, LRELU),
                                            ("autoencoder", 8, dict(normalize_brightness=True)),
                                            ("resnet", 0, dict(normalize_brightness=True, num_flow_inputs=2))])
def test_c_restatement_vs_numpy_oracle(arch, pad, extra):
    from oracle.c_binding import CSession
    cfg = small_config(flow_arch=arch, flow_pad_factor=pad, flow_res_blocks=2, **extra)
    wts = M.make_seeded_weights(cfg)
    s = O.Session(wts, oracle_config(cfg))
    cs = CSession(M.serialize(cfg, wts), cfg.frame_height, cfg.frame_width)
    frames = M.synthetic_frames(4, cfg.frame_height, cfg.frame_width, kind="smooth")
    for t in range(4):
        a = s.run(frames[t])
        b = cs.run(frames[t])
        assert err(cs.output_raw(), s.last.output_raw)["max_abs"] < 2e-5  # fp32 vs fp64
        st = u8_stats(b, a)
        assert st["max"] <= 1 and (b[..., 3] == 0).all()


def test_x_byte_is_ignored_by_the_oracle():
    cfg = small_config(gen_blocks=1)
    wts = M.make_seeded_weights(cfg)
    f = M.synthetic_frames(1, 30, 48)[0]
    g = f.copy()
    g[..., 3] = 0
    a = O.Session(wts, oracle_config(cfg)).run(f)
    b = O.Session(wts, oracle_config(cfg)).run(g)
    assert np.array_equal(a, b)


def test_lrelu_changes_the_result_and_relu_is_the_default():
    """The activation really is a model parameter (reference models.py:24-27, 261, 337, 489)."""
    frames = M.synthetic_frames(2, 30, 48, kind="smooth")
    wts = M.make_seeded_weights(small_config())
    outs = {}
    for key, extra in [("relu", {}), ("lrelu", LRELU), ("gen-only", dict(gen_activation="lrelu"))]:
        cfg = small_config(**extra)
        s = O.Session(wts, oracle_config(cfg))
        outs[key] = [s.run(f) for f in frames][-1]
    assert not np.array_equal(outs["relu"], outs["lrelu"])
    assert not np.array_equal(outs["lrelu"], outs["gen-only"])
    assert oracle_config(small_config()).gen_activation == "relu"


_FORMS_SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
from helpers import M, small_config
from oracle.c_binding import CSession
h = hashlib.sha256()
bits = None
for kw in (dict(gen_blocks=2), dict(gen_blocks=1, gen_filters=32, flow_filters=(32, 64, 128, 64, 32)),
           dict(gen_blocks=1, gen_filters=96, flow_arch="resnet", flow_pad_factor=0, flow_res_filters=32, flow_res_blocks=1),
           dict(gen_blocks=1, frame_height=17, frame_width=33), dict(gen_blocks=1, frame_height=9, frame_width=7)):
    cfg = small_config(**kw)
    cs = CSession(M.serialize(cfg, M.make_seeded_weights(cfg)), cfg.frame_height, cfg.frame_width)
    bits = cs.vector_bits
    for f in M.synthetic_frames(2, cfg.frame_height, cfg.frame_width, seed=5, kind="noise"):
        h.update(cs.run(f).tobytes()); h.update(cs.output_raw().tobytes())
print(bits, h.hexdigest())
"""


def test_c_restatement_gives_the_same_bytes_in_every_form_of_its_convolution():
    """oracle/ju_oracle_c.c computes interior pixels in register blocks (6 pixels x 16 channels with AVX2, 8 x 32 where the
    CPU has AVX-512) and everything else one pixel at a time; every output element accumulates the same terms in the same
    order with one multiply and one add each, so the frames and the float `output_raw` are EQUAL whichever form ran --
    the plain form (JUO_VECTOR_BITS=0) is the restatement, the blocks are only its speed."""
    import subprocess
    import sys
    from helpers import ROOT
    seen = {}
    for want in ("0", "256", "512"):
        out = subprocess.run([sys.executable, "-c", _FORMS_SCRIPT.format(root=ROOT)], capture_output=True, text=True,
                             env=dict(os.environ, JUO_VECTOR_BITS=want, OMP_NUM_THREADS="4"))
        assert out.returncode == 0, out.stderr
        bits, digest = out.stdout.split()
        seen[int(bits)] = digest
        if want != "512":
            assert int(bits) == int(want)
    assert {0, 256} <= set(seen) and len(set(seen.values())) == 1, seen
