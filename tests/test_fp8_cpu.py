"""8-bit tower (BASELINE.json config 5), CPU side: the oracle's e4m3 restatement
against an enumerated code table, the loader's C++ quantiser against the oracle's, and
the quantisation helpers' invariants.  The scheme is the BUILD's (joshupscale_amd/csrc/
fp8.h): the reference's own 8-bit deployment is TensorRT INT8
(scripts/inference/tensorrt/quantize_int8.py:140-209) and cannot run here."""

import ctypes as C

import numpy as np
import pytest

from helpers import M, O, oracle_config, small_config


def e4m3_table() -> np.ndarray:
    """All 127 non-negative finite OCP e4m3fn values, by code."""
    vals = []
    for code in range(0x7f):
        e, m = (code >> 3) & 15, code & 7
        vals.append(m * 2.0 ** -9 if e == 0 else (1 + m / 8) * 2.0 ** (e - 7))
    return np.array(vals)


def nearest_even(table: np.ndarray, x: float) -> int:
    d = np.abs(table - min(abs(x), 448.0))
    best = np.flatnonzero(d == d.min())
    if len(best) > 1:
        best = [k for k in best if k % 2 == 0]  # tie: even code = even mantissa
    return int(best[0])


def sample_values(rng) -> np.ndarray:
    t = e4m3_table()
    x = np.concatenate([
        rng.uniform(0, 500, 4000), rng.uniform(0, 0.05, 4000), rng.uniform(0, 4, 4000),
        t, (t[1:] + t[:-1]) / 2,                       # every exact value and every tie
        np.nextafter((t[1:] + t[:-1]) / 2, 0), np.nextafter((t[1:] + t[:-1]) / 2, 1e9),
        [448.0, 464.0, 1e9, 2.0 ** -10, 2.0 ** -11, 0.0]])
    return np.concatenate([x, -x])


def test_table_is_the_ocp_format():
    t = e4m3_table()
    assert t[0] == 0 and t[1] == 2.0 ** -9 and t[8] == 2.0 ** -6 and t[-1] == 448.0
    assert np.all(np.diff(t) > 0)


def test_oracle_e4m3_round_is_nearest_even_and_saturating():
    x = sample_values(np.random.default_rng(0))
    t = e4m3_table()
    want = np.array([np.copysign(t[nearest_even(t, v)], v) for v in x])
    got = O.e4m3_round(x)
    assert np.array_equal(got, want)
    assert np.array_equal(np.signbit(got), np.signbit(x))


def test_loader_quantiser_matches_the_oracle(hip_library):
    """csrc/fp8.h e4m3FromFloat (used for the weights) == oracle e4m3_round, on float32
    inputs, code for code."""
    x = sample_values(np.random.default_rng(1)).astype(np.float32)
    codes = np.zeros(len(x), np.uint8)
    assert hip_library.ju_debug_e4m3(x.ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p),
                                     len(x)) == 0
    t = e4m3_table()
    decoded = np.where(codes & 0x80, -1.0, 1.0) * t[np.minimum(codes & 0x7f, 0x7e)]
    assert not np.any((codes & 0x7f) == 0x7f)            # never NaN for finite inputs
    assert np.array_equal(decoded, O.e4m3_round(x.astype(np.float64)))
    # the sign of zero survives, as in the hardware conversion
    assert np.array_equal((codes & 0x80) != 0, np.signbit(x))


def test_activation_exponent_leaves_one_bit_of_headroom():
    for amax in [0.01, 0.3, 1.0, 1.3, 7.0, 13.9, 14.0, 100.0, 1e4]:
        e = O.fp8_activation_exponent(amax)
        assert 112.0 < amax * 2.0 ** e <= 224.0
    assert O.fp8_activation_exponent(0.0) == 0 and O.fp8_activation_exponent(float("nan")) == 0
    assert O.fp8_activation_exponent(1e-30) == 16 and O.fp8_activation_exponent(1e30) == -16
    assert O.fp8_activation_exponent(O.FP8_DEFAULT_AMAX) == 5


def test_weight_quantisation_is_per_output_channel_and_tight():
    rng = np.random.default_rng(2)
    k = rng.normal(0, 0.05, (3, 3, 64, 64)) * rng.uniform(0.01, 30, 64)   # very different channels
    k[..., 5] = 0.0
    q = O.fp8_quantize_weights(k)
    assert q.shape == k.shape and not q[..., 5].any()
    amax = np.abs(k).reshape(-1, 64).max(axis=0)
    rel = np.abs(q - k).reshape(-1, 64).max(axis=0) / np.maximum(amax, 1e-30)
    assert rel[np.arange(64) != 5].max() <= 2.0 ** -4     # half an ulp of a 3-bit mantissa
    # scaling one output channel by a power of two scales its quantised weights exactly
    k2 = k.copy()
    k2[..., 7] *= 8.0
    assert np.array_equal(O.fp8_quantize_weights(k2)[..., 7], q[..., 7] * 8.0)


def test_fp8_oracle_degrades_gracefully():
    """The 8-bit restatement stays close to the float one (the quantisation noise of a
    3-bit mantissa, not a wiring error) and is bit-identical outside the tower."""
    cfg = small_config()
    wts = M.make_seeded_weights(cfg)
    frames = M.synthetic_frames(2, cfg.frame_height, cfg.frame_width, seed=5, kind="smooth")
    f, q = O.Session(wts, oracle_config(cfg)), O.Session(wts, oracle_config(cfg, fp8_tower=True))
    for frame in frames:
        tf, tq = {}, {}
        a, b = f.run(frame, tf), q.run(frame, tq)
        d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
        assert 10 * np.log10(255.0 ** 2 / max(np.mean(d * d), 1e-12)) > 50.0
        assert np.abs(tf["trunk"] - tq["trunk"]).max() < 0.05 * np.abs(tf["trunk"]).max()
        assert np.abs(tf["trunk"] - tq["trunk"]).max() > 0
    # frame 0: nothing upstream of the tower depends on it
    f.reset(), q.reset()
    tf, tq = {}, {}
    f.run(frames[0], tf), q.run(frames[0], tq)
    assert np.array_equal(tf["gen_head"], tq["gen_head"]) and np.array_equal(tf["flow"], tq["flow"])


def test_container_accepts_the_fp8_dtype_and_the_calibration_tensor():
    from joshupscale_amd import runtime as R
    cfg = small_config(compute_dtype=2)
    wts = M.make_seeded_weights(cfg)
    wts["generator/fp8_amax"] = np.full(2 * cfg.gen_blocks, 3.0, np.float32)
    blob = M.serialize(cfg, wts)
    R.validate_model(blob)
    back_cfg, back = M.deserialize(blob)
    assert back_cfg.compute_dtype == 2 and np.array_equal(back["generator/fp8_amax"], wts["generator/fp8_amax"])
    with pytest.raises(R.JoshUpscaleError):
        R.validate_model(M.serialize(small_config(compute_dtype=3), M.make_seeded_weights(cfg)))
