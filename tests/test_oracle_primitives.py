"""Known-answer tests of the oracle's primitives (CPU).

The reference has no tests, so every primitive of the
restatement is pinned by a property the reference's own definition implies."""

import numpy as np
import pytest

from helpers import O


def test_preprocess_postprocess_roundtrip_all_256_values():
    # keras_layers.py:208, 227-230 + truncating cast: document the exact behaviour
    v = np.arange(256, dtype=np.uint8).reshape(1, 256, 1).repeat(3, axis=2)
    back64 = O.postprocess(O.preprocess(v, np.float64))
    back32 = np.trunc((O.preprocess(v, np.float32) + np.float32(0.5)) * np.float32(255)).astype(np.uint8)
    # float64: (v/255 - 0.5 + 0.5) * 255 can land just below v and truncate to v-1
    assert np.abs(back64.astype(int) - v.astype(int)).max() <= 1
    assert np.abs(back32.astype(int) - v.astype(int)).max() <= 1
    assert (back64 <= v).all()


def test_space_to_depth_channel_order_and_inverse():
    x = np.arange(8 * 12 * 3, dtype=np.float64).reshape(8, 12, 3)
    y = O.space_to_depth(x, 4)
    assert y.shape == (2, 3, 48)
    for i in range(4):
        for j in range(4):
            for c in range(3):
                assert y[1, 2, (i * 4 + j) * 3 + c] == x[4 + i, 8 + j, c]
    assert np.array_equal(O.depth_to_space(y, 4), x)


def test_depth_to_space_dcr_order():
    x = np.arange(2 * 3 * 32, dtype=np.float64).reshape(2, 3, 32)
    y = O.depth_to_space(x, 4)
    assert y.shape == (8, 12, 2)
    for i in range(4):
        for j in range(4):
            for c in range(2):
                assert y[4 + i, 8 + j, c] == x[1, 2, (i * 4 + j) * 2 + c]


def test_resize_bilinear_tf1_ramp_and_edge_clamp():
    # asymmetric mapping src = dst/scale: a linear ramp stays linear until the
    # last `scale-1` outputs, which clamp to the last input sample
    x = np.arange(6, dtype=np.float64).reshape(1, 6, 1).repeat(2, axis=0)
    y = O.resize_bilinear_tf1(x, 4)
    assert y.shape == (8, 24, 1)
    expect = np.minimum(np.arange(24) / 4.0, 5.0)
    assert np.allclose(y[0, :, 0], expect)
    assert np.allclose(y[7, :, 0], expect)  # rows identical -> vertical lerp is a no-op
    assert np.array_equal(y[::4, ::4], x)   # phase 0 reproduces the input


def test_warp_zero_flow_is_identity():
    rng = np.random.default_rng(0)
    img = rng.normal(size=(9, 11, 3))
    out = O.dense_image_warp(img, np.zeros((9, 11, 2)))
    # interior: alpha == 0, exact.  Last row/column: floor is clamped to size-2 and
    # alpha == 1, so the reference formula gives (b - a) + a, equal only to rounding.
    assert np.array_equal(out[:-1, :-1], img[:-1, :-1])
    assert np.allclose(out, img, rtol=0, atol=1e-15)


def test_warp_integer_shift_with_border_clamp():
    rng = np.random.default_rng(1)
    img = rng.normal(size=(8, 10, 3))
    flow = np.zeros((8, 10, 2))
    flow[..., 0] = 2.0   # out[y, x] = img[y - 2, x + 3], clamped to the border
    flow[..., 1] = -3.0
    out = O.dense_image_warp(img, flow)
    ys = np.clip(np.arange(8) - 2, 0, 7)
    xs = np.clip(np.arange(10) + 3, 0, 9)
    assert np.allclose(out, img[ys][:, xs])


def test_warp_fractional_matches_manual_bilinear():
    img = np.arange(5 * 6, dtype=np.float64).reshape(5, 6, 1)
    flow = np.zeros((5, 6, 2))
    flow[..., 0] = 0.25
    flow[..., 1] = -0.5
    out = O.dense_image_warp(img, flow)
    y, x = 2, 3  # query (1.75, 3.5)
    manual = (img[1, 3, 0] * 0.5 + img[1, 4, 0] * 0.5) * 0.25 + (img[2, 3, 0] * 0.5 + img[2, 4, 0] * 0.5) * 0.75
    assert np.isclose(out[y, x, 0], manual)


def test_conv2d_same_against_direct_loops():
    rng = np.random.default_rng(2)
    x = rng.normal(size=(5, 7, 3))
    k = rng.normal(size=(3, 3, 3, 4))
    y = O.conv2d_same(x, k)
    xp = np.pad(x, ((1, 1), (1, 1), (0, 0)))
    ref = np.zeros((5, 7, 4))
    for h in range(5):
        for w in range(7):
            for o in range(4):
                ref[h, w, o] = np.sum(xp[h:h + 3, w:w + 3, :] * k[:, :, :, o])
    assert np.allclose(y, ref)


def test_conv_transpose_no_overlap_definition():
    rng = np.random.default_rng(3)
    x = rng.normal(size=(3, 4, 5))
    k = rng.normal(size=(2, 2, 6, 5))
    b = rng.normal(size=6)
    y = O.conv2d_transpose_k2s2(x, k, b)
    assert y.shape == (6, 8, 6)
    for a in range(2):
        for bb in range(2):
            assert np.allclose(y[2 * 1 + a, 2 * 2 + bb], k[a, bb] @ x[1, 2] + b)


def test_max_pool_and_batch_norm():
    x = np.arange(4 * 6 * 2, dtype=np.float64).reshape(4, 6, 2)
    p = O.max_pool_2x2(x)
    assert p.shape == (2, 3, 2)
    assert p[0, 0, 0] == x[1, 1, 0] and p[1, 2, 1] == x[3, 5, 1]
    y = O.batch_norm(np.ones((1, 1, 2)), np.array([2.0, 3.0]), np.array([0.5, -0.5]),
                     np.array([0.0, 1.0]), np.array([1.0, 4.0]), 0.0)
    assert np.allclose(y[0, 0], [2.5, -0.5])


def test_state_shift_register_and_padding():
    from helpers import M, oracle_config, small_config
    cfg = small_config(gen_blocks=1)
    oc = oracle_config(cfg)
    assert (oc.padded_height, oc.padded_width) == (32, 48)
    sess = O.Session(M.make_seeded_weights(cfg), oc)
    frames = M.synthetic_frames(3, 30, 48, kind="noise")
    for t in range(3):
        sess.run(frames[t])
    # last_frames[0] is the most recent padded frame; pad rows are exactly zero
    lf = sess.state.last_frames
    assert np.array_equal(lf[0][1:31], O.preprocess(frames[2][..., :3]))
    assert np.array_equal(lf[1][1:31], O.preprocess(frames[1][..., :3]))
    assert np.array_equal(lf[2][1:31], O.preprocess(frames[0][..., :3]))
    assert not lf[0][0].any() and not lf[0][31].any()
    assert np.array_equal(sess.state.pre_gen, sess.last.output_raw)


def test_macs_match_survey():
    m = O.macs_per_frame(O.ModelConfig())
    assert m["generator"] == 234391449600 and m["flow"] == 17898209280


def test_temporal_filter_gate_and_weights():
    """frame_moving_avg.py default mode: still scene blends `strength` of the warped
    previous output in, a scene cut passes the generator output through."""
    rng = np.random.default_rng(3)
    gen = rng.uniform(-0.5, 0.5, (8, 12, 3))
    near = gen + rng.uniform(-0.02, 0.02, gen.shape)          # mean |diff| = 0.01
    far = gen + np.where(rng.random(gen.shape) < 0.5, 0.4, -0.4)  # mean |diff| = 0.4
    s = 0.25
    still = O.temporal_filter(gen, near, s, 0.1)
    assert np.allclose(still, s * near + (1 - s) * gen, atol=1e-15)
    cut = O.temporal_filter(gen, far, s, 0.1)
    assert np.array_equal(cut, gen)
    # exactly at the threshold the sign is 0: weights s/2 and 1 - s/2
    pw = gen + 0.125
    edge = O.temporal_filter(gen, pw, s, float(np.mean(np.abs(gen - pw))))
    assert np.allclose(edge, (s / 2) * pw + (1 - s / 2) * gen, atol=1e-15)


def test_leaky_relu_known_answers():
    """keras.layers.LeakyReLU(negative_slope) (reference models.py:24-27 "lrelu")."""
    x = np.array([-2.0, -0.5, 0.0, 0.25, 3.0])
    assert np.allclose(O.leaky_relu(x, 0.3), [-0.6, -0.15, 0.0, 0.25, 3.0])
    assert np.array_equal(O.leaky_relu(x, 0.0), O.relu(x))
    assert np.array_equal(O.leaky_relu(x, 1.0), x)
    cfg = O.ModelConfig(gen_activation="lrelu", gen_negative_slope=0.2)
    assert np.allclose(cfg.act("generator")(x), [-0.4, -0.1, 0.0, 0.25, 3.0])
    assert np.array_equal(cfg.act("flow")(x), O.relu(x))          # per sub-model
    with pytest.raises(ValueError, match="Unknown activation"):
        O.ModelConfig(flow_activation="gelu").act("flow")


def test_temporal_filter_modes_known_answers():
    """frame_moving_avg.py:157-296 on hand-made tensors."""
    rng = np.random.default_rng(3)
    gen = rng.uniform(-0.5, 0.5, (8, 12, 3))
    pw = gen + 0.2                                  # |d| = 0.2 everywhere
    s = 0.5
    # global sign gate: mean 0.2 > threshold 0.1 -> scene cut -> generator output
    assert np.allclose(O.temporal_filter(gen, pw, s, 0.1), gen)
    # still scene (threshold above the mean): s of pre_warp blended in
    assert np.allclose(O.temporal_filter(gen, pw, s, 0.3), pw * s + gen * (1 - s))
    # exactly at the threshold: sign(0) = 0 -> half of s
    assert np.allclose(O.temporal_filter(np.zeros((4, 4, 3)), np.full((4, 4, 3), 0.25), s, 0.25),
                       0.25 * (s / 2))
    # tanh gate: c = tanh(gain * (m - t))
    c = np.tanh(10.0 * (0.2 - 0.3))
    assert np.allclose(O.temporal_filter(gen, pw, s, 0.3, gain=10.0),
                       pw * (s / 2 - c * s / 2) + gen * (c * s / 2 + 1 - s / 2))
    # L2 norm: mean of d^2 = 0.04
    tr = {}
    O.temporal_filter(gen, pw, s, 0.3, norm="L2", trace=tr)
    assert np.isclose(tr["temporal_mean"], 0.04)
    # luma weighting: each channel's term times BGR_LUMA*3 (squared for L2)
    O.temporal_filter(gen, pw, s, 0.3, luma=True, trace=tr)
    assert np.isclose(tr["temporal_mean"], 0.2 * np.mean(O.BGR_LUMA * 3))
    O.temporal_filter(gen, pw, s, 0.3, norm="L2", luma=True, trace=tr)
    assert np.isclose(tr["temporal_mean"], 0.04 * np.mean((O.BGR_LUMA * 3) ** 2))
    # limit: pre_warp is clipped to +-0.5 before everything else
    far = np.full((4, 4, 3), 2.0)
    out = O.temporal_filter(np.zeros((4, 4, 3)), far, s, 1.0, limit=True)
    assert np.allclose(out, 0.5 * s)
    # windowed gate: 8x12 frame, window 4 -> 2x3 grid of block means (no padding here);
    # left two thirds still, right third cut -> per-pixel gate, linearly interpolated
    d = np.zeros((8, 12, 3))
    d[:, 8:] = 0.4
    tr = {}
    out = O.temporal_filter(np.zeros((8, 12, 3)), d, 1.0, 0.2, window=4, trace=tr)
    assert tr["temporal_mean"].shape == (2, 3)
    assert np.allclose(tr["temporal_mean"], [[0, 0, 0.4], [0, 0, 0.4]])
    # gate grid [-1, -1, +1] resized x4 asymmetric-linear: columns 0..4 = -1, then a ramp
    # -1 -> +1 over columns 4..8, then +1; out = pre_warp * (1/2 - c/2)
    cexp = np.concatenate([np.full(5, -1.0), [-0.5, 0.0, 0.5], np.full(4, 1.0)])
    assert np.allclose(out[..., 0], d[..., 0] * (0.5 - cexp / 2)[None, :])
    # ragged frame: 6x10 with window 4 pads to 8x12 (1 px leading on each axis); the
    # divisor stays 3*window^2
    tr = {}
    O.temporal_filter(np.zeros((6, 10, 3)), np.full((6, 10, 3), 0.3), 1.0, 0.5, window=4, trace=tr)
    assert tr["temporal_mean"].shape == (2, 3)
    assert np.isclose(tr["temporal_mean"][0, 0], 0.3 * (3 * 3) / 16)    # 3x3 real pixels of 4x4
    assert np.isclose(tr["temporal_mean"][0, 1], 0.3 * (3 * 4) / 16)
    with pytest.raises(ValueError, match="Unknown norm"):
        O.temporal_filter(gen, pw, s, 0.1, norm="L3")
