"""Model container round trip and header layout (CPU)."""

import struct

import numpy as np
import pytest

from helpers import M


def test_roundtrip_all_presets():
    for name, cfg in M.PRESETS.items():
        small = M.ModelConfig(**{**cfg.__dict__, "gen_blocks": 1, "flow_res_blocks": 1})
        w = M.make_seeded_weights(small)
        cfg2, w2 = M.deserialize(M.serialize(small, w))
        floats = {"bn_eps": 0, "temporal_strength": 0, "temporal_threshold": 0, "temporal_gain": 0,
                  "flow_negative_slope": 0, "gen_negative_slope": 0}  # f32 in the header
        assert cfg2.__dict__ | floats == small.__dict__ | floats, name
        assert cfg2.bn_eps == pytest.approx(small.bn_eps)
        assert cfg2.temporal_threshold == pytest.approx(small.temporal_threshold)
        assert list(w) == list(w2)
        assert all(np.array_equal(w[k], w2[k]) for k in w)


def test_temporal_filter_fields_live_in_the_reserved_words():
    cfg = M.ModelConfig(gen_blocks=1, temporal_strength=0.25, temporal_threshold=0.05)
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    assert struct.unpack_from("<2f3I", blob, 108) == (0.25, pytest.approx(0.05), 0, 0, 0)
    cfg2, _ = M.deserialize(blob)
    assert cfg2.temporal_strength == 0.25 and cfg2.temporal_threshold == pytest.approx(0.05)
    plain = M.serialize(M.ModelConfig(gen_blocks=1), M.make_seeded_weights(cfg))
    assert struct.unpack_from("<f", plain, 108) == (0.0,)      # filter off = a version-1 file as before


def test_activation_fields_live_in_the_reserved_words():
    """`activation` of the sub-model constructors (reference models.py:24-27, 261, 337, 489)."""
    cfg = M.ModelConfig(gen_blocks=1, flow_activation="lrelu", gen_activation="lrelu",
                        gen_negative_slope=0.2)
    w = M.make_seeded_weights(cfg)
    blob = M.serialize(cfg, w)
    acts, fs, gs = struct.unpack_from("<Iff", blob, 116)
    assert acts == 0x0101 and fs == pytest.approx(0.3) and gs == pytest.approx(0.2)
    cfg2, _ = M.deserialize(blob)
    assert (cfg2.flow_activation, cfg2.gen_activation) == ("lrelu", "lrelu")
    assert cfg2.gen_negative_slope == pytest.approx(0.2) and cfg2.flow_negative_slope == pytest.approx(0.3)
    mixed = M.ModelConfig(gen_blocks=1, gen_activation="lrelu")
    assert struct.unpack_from("<Iff", M.serialize(mixed, w), 116) == (0x0100, 0.0, pytest.approx(0.3))
    plain = M.serialize(M.ModelConfig(gen_blocks=1), w)
    assert struct.unpack_from("<3I", plain, 116) == (0, 0, 0)   # all-ReLU = a file as before the field
    with pytest.raises(KeyError):
        M.serialize(M.ModelConfig(gen_blocks=1, gen_activation="gelu"), w)


def test_extended_temporal_fields_grow_the_header():
    """--window / --gain / --norm / --limit / --luma-normalize of frame_moving_avg.py: only a
    non-default mode makes the header 160 bytes; everything else stays a 128-byte header."""
    w = M.make_seeded_weights(M.ModelConfig(gen_blocks=1))
    cfg = M.ModelConfig(gen_blocks=1, temporal_strength=0.5, temporal_window=16, temporal_gain=4.0,
                        temporal_norm="L2", temporal_limit=True, temporal_luma=True)
    blob = M.serialize(cfg, w)
    assert struct.unpack_from("<I", blob, 12) == (160,)
    assert struct.unpack_from("<IfI", blob, 128) == (16, 4.0, 7)
    cfg2, w2 = M.deserialize(blob)
    assert (cfg2.temporal_window, cfg2.temporal_gain, cfg2.temporal_norm, cfg2.temporal_limit,
            cfg2.temporal_luma) == (16, 4.0, "L2", True, True)
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    default = M.serialize(M.ModelConfig(gen_blocks=1, temporal_strength=0.5), w)
    assert struct.unpack_from("<I", default, 12) == (128,)
    off = M.serialize(M.ModelConfig(gen_blocks=1, temporal_window=16), w)     # filter off: fields unused
    assert struct.unpack_from("<I", off, 12) == (128,)


def test_header_layout_and_alignment():
    cfg = M.ModelConfig(gen_blocks=1)
    w = M.make_seeded_weights(cfg)
    blob = M.serialize(cfg, w)
    assert blob[:8] == b"JUPWGT\x00\x01"
    version, header = struct.unpack_from("<2I", blob, 8)
    assert (version, header) == (1, 128)
    assert struct.unpack_from("<4I", blob, 16) == (270, 480, 4, 4)
    n = struct.unpack_from("<I", blob, 104)[0]
    assert n == len(w)
    assert struct.calcsize("<92sI4IQQ") == M.ENTRY_BYTES
    for i in range(n):
        off = struct.unpack_from("<Q", blob, 128 + i * 128 + 112)[0]
        assert off % 64 == 0


def test_rejects_foreign_files():
    with pytest.raises(ValueError):
        M.deserialize(b"\x00" * 256)          # e.g. a TensorRT engine
    with pytest.raises(ValueError):
        M.deserialize(b"JUPWGT")


def test_python_twin_rejects_what_the_cpp_loader_rejects():
    """csrc/model.cpp refuses unknown activation codes, non-zero upper bits of the activations
    word and unknown temporal flag bits with 'Invalid model: ...'; so does the Python twin, with
    ValueError like every other malformed field (not a bare KeyError)."""
    cfg = M.ModelConfig(frame_height=30, frame_width=48, gen_blocks=1, temporal_strength=0.5,
                        temporal_window=16)
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))

    def patched(offset, fmt, *vals):
        b = bytearray(blob)
        struct.pack_into(fmt, b, offset, *vals)
        return bytes(b)
    for bad, match in [(patched(116, "<I", 2), "flow activation"), (patched(116, "<I", 0x0300), "generator activation"),
                       (patched(116, "<I", 0x10000), "activation field"), (patched(136, "<I", 8), "temporal filter flags")]:
        with pytest.raises(ValueError, match=match):
            M.deserialize(bad)
    M.deserialize(patched(136, "<I", 7))     # every known flag bit set: accepted


def test_seeded_weights_are_deterministic_and_named_like_keras():
    cfg = M.ModelConfig(gen_blocks=2)
    a = M.make_seeded_weights(cfg, seed=42)
    b = M.make_seeded_weights(cfg, seed=42)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert a["generator/conv_1/kernel"].shape == (3, 3, 51, 64)
    assert a["generator/conv_trans_1/kernel"].shape == (2, 2, 32, 64)
    assert a["generator/conv_trans_2/kernel"].shape == (2, 2, 3, 32)
    assert a["flow/block_1/conv_1/kernel"].shape == (3, 3, 12, 32)
    assert a["flow/conv_2/kernel"].shape == (3, 3, 32, 32)
    assert "generator/block_2/bn_2/moving_variance" in a


def test_synthetic_frames_contract():
    f = M.synthetic_frames(2, 30, 48, seed=1234)
    assert f.shape == (2, 30, 48, 4) and f.dtype == np.uint8 and (f[..., 3] == 255).all()
    assert np.array_equal(f, M.synthetic_frames(2, 30, 48, seed=1234))
    s = M.synthetic_frames(3, 30, 48, kind="smooth")
    assert np.array_equal(s[0, :, 1:, :3], s[1, :, :-1, :3])  # translates 1 px / frame
