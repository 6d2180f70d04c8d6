"""The C++ container loader through ``ju_validate_model`` (no GPU needed): what it
accepts, what it rejects and that hostile bytes never crash it.  The same code runs
first inside ``ju_create*``."""
import struct

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from helpers import M, small_config
from joshupscale_amd import runtime as R

CFG = small_config(gen_blocks=1)
WTS = M.make_seeded_weights(CFG)
BLOB = M.serialize(CFG, WTS)


def rejects(blob: bytes, match: str):
    with pytest.raises(R.JoshUpscaleError, match=match) as e:
        R.validate_model(blob)
    assert e.value.code == 1                       # JU_ERR_INVALID_ARGUMENT


def patched(offset: int, fmt: str, *vals) -> bytes:
    b = bytearray(BLOB)
    struct.pack_into(fmt, b, offset, *vals)
    return bytes(b)


def test_accepts_every_preset_and_variant(hip_library):
    for cfg in [CFG, small_config(flow_arch="resnet", flow_pad_factor=0, flow_res_blocks=2),
                small_config(normalize_brightness=True), small_config(temporal_strength=0.25),
                small_config(flow_activation="lrelu", gen_activation="lrelu", gen_negative_slope=0.2),
                small_config(temporal_strength=0.5, temporal_window=16, temporal_gain=4.0, temporal_norm="L2",
                             temporal_limit=True, temporal_luma=True),
                M.ModelConfig(gen_blocks=1), M.ModelConfig(frame_height=448, frame_width=640, gen_blocks=1)]:
        R.validate_model(M.serialize(cfg, M.make_seeded_weights(cfg)))


def test_header_fields_are_range_checked(hip_library):
    rejects(b"", "too small")
    rejects(b"ptrt" + b"\0" * 4096, "TensorRT")
    rejects(patched(8, "<I", 2), "version")
    rejects(patched(24, "<I", 2), "scale")
    rejects(patched(16, "<I", 0), "frame size")
    rejects(patched(20, "<I", 1 << 20), "frame size")
    rejects(patched(28, "<I", 9), "flow inputs")
    rejects(patched(32, "<I", 7), "flow architecture")
    rejects(patched(36, "<I", 7), "divisible")            # pad factor 7: 35 x 49, not / 8
    rejects(patched(44, "<I", 48), "gen_filters")
    rejects(patched(48, "<I", 100000), "gen_blocks")
    rejects(patched(64, "<I", 33), "flow filters")
    rejects(patched(96, "<f", float("nan")), "bn_eps")
    rejects(patched(100, "<I", 5), "compute dtype")
    rejects(patched(108, "<f", 1.5), "temporal")
    rejects(patched(116, "<I", 2), "flow activation")          # ACTIVATIONS = {relu, lrelu}
    rejects(patched(116, "<I", 0x0300), "generator activation")
    rejects(patched(116, "<I", 0x10000), "activation field")
    rejects(patched(116, "<If", 1, -0.5), "negative_slope")    # non-monotonic: rejected
    rejects(patched(120, "<f", 0.3), "negative_slope set on a relu model")
    rejects(patched(104, "<I", 1 << 24), "truncated tensor table")


def test_tensor_table_is_bounds_checked(hip_library):
    entry = 128                                         # first table entry
    rejects(patched(entry + 92, "<I", 5), "rank")
    rejects(patched(entry + 96, "<I", 0), "dimension")
    rejects(patched(entry + 96, "<4I", 65536, 65536, 65536, 65536), "bad tensor entry")  # no wrap-around
    rejects(patched(entry + 112, "<Q", len(BLOB) - 8), "bad tensor entry")               # runs past the end
    rejects(patched(entry + 112, "<Q", 1 << 62), "bad tensor entry")
    rejects(patched(entry + 120, "<Q", 7), "bad tensor entry")                           # count != prod(dims)
    dup = bytearray(BLOB)
    dup[entry + 128:entry + 128 + 92] = dup[entry:entry + 92]
    rejects(bytes(dup), "duplicate tensor")
    rejects(BLOB[:len(BLOB) // 2], "bad tensor entry|truncated")


def test_graph_consistency_is_checked(hip_library):
    def with_tensor(name, arr):
        w = dict(WTS)
        if arr is None:
            del w[name]
        else:
            w[name] = np.ascontiguousarray(arr, np.float32)
        return M.serialize(CFG, w)
    rejects(with_tensor("generator/block_1/bn_2/moving_mean", None), "missing tensor generator/block_1/bn_2")
    rejects(with_tensor("flow/conv_2/bias", np.zeros(31)), "flow/conv_2/bias")
    k = WTS["generator/block_1/conv_2/kernel"]
    rejects(with_tensor("generator/block_1/conv_2/kernel", k[:, :, :32]), "generator/block_1/conv_2 is 32 -> 64")
    k = WTS["flow/block_2/conv_1/kernel"]
    rejects(with_tensor("flow/block_2/conv_1/kernel", k[..., :32]), "flow/block_2/")
    rejects(with_tensor("generator/conv_1/kernel", WTS["generator/conv_1/kernel"][:, :, :48]), "generator/conv_1")
    rejects(with_tensor("generator/conv_trans_2/kernel", np.zeros((2, 2, 32, 3))), "conv_trans_2")
    rejects(with_tensor("flow/conv_2/kernel", WTS["flow/conv_2/kernel"][:1, :1]), "flow/conv_2")
    # header says 2 generator blocks, file has 1
    rejects(patched(48, "<I", 2), "missing tensor generator/block_2")


@settings(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(data=st.data())
def test_mutated_headers_and_tables_never_crash(hip_library, data):
    """Random byte / word edits in the header and the tensor table: the loader must
    return (accept or reject), never fault or hang.  The data area is left alone: any
    float is a legal weight."""
    b = bytearray(BLOB)
    table_end = 128 + 128 * len(WTS)
    for _ in range(data.draw(st.integers(1, 6))):
        off = data.draw(st.integers(8, table_end - 8))
        kind = data.draw(st.sampled_from(["byte", "u32", "u64", "extreme"]))
        if kind == "byte":
            b[off] = data.draw(st.integers(0, 255))
        elif kind == "u32":
            struct.pack_into("<I", b, off & ~3, data.draw(st.integers(0, 2 ** 32 - 1)))
        elif kind == "u64":
            struct.pack_into("<Q", b, off & ~7, data.draw(st.integers(0, 2 ** 64 - 1)))
        else:
            struct.pack_into("<I", b, off & ~3, data.draw(st.sampled_from(
                [0, 1, 0x7fffffff, 0x80000000, 0xffffffff, 65536, 65537])))
    cut = data.draw(st.one_of(st.none(), st.integers(0, len(b))))
    blob = bytes(b if cut is None else b[:cut])
    lib = hip_library
    rc = lib.ju_validate_model(blob, len(blob))
    assert rc in (0, 1), (rc, lib.ju_last_error())


# Widths outside what the GPU parity tests run (test_gpu_parity.py test_nondefault_widths_match_oracle)
# are refused by BOTH loaders with the SAME message: nothing loads that no test has compared with the oracle.
BAD_WIDTHS = [
    (dict(gen_filters=48), "gen_filters must be a multiple of 32 (at most 256)"),
    (dict(gen_filters=288), "gen_filters must be a multiple of 32 (at most 256)"),
    (dict(gen_filters=1024), "gen_filters must be a multiple of 32 (at most 256)"),
    (dict(flow_filters=(32, 40, 32)), "flow filters must be multiples of 32 (at most 512)"),
    (dict(flow_filters=(64, 128, 256, 1024, 256, 128, 64)), "flow filters must be multiples of 32 (at most 512)"),
    (dict(flow_filters=(32,)), "padded frame size must be divisible by 2^(flow depth)"),
    (dict(flow_filters=(32, 64, 128, 256, 256, 128, 64, 32), frame_width=50), "padded frame size must be divisible by 2^(flow depth)"),
    (dict(flow_arch="resnet", flow_pad_factor=0, flow_res_filters=48), "flow_res_filters must be a multiple of 32 (at most 256)"),
    (dict(flow_arch="resnet", flow_pad_factor=0, flow_res_filters=512), "flow_res_filters must be a multiple of 32 (at most 256)"),
    (dict(num_flow_inputs=6), "1..5 flow inputs supported"),
    (dict(num_flow_inputs=0), "1..5 flow inputs supported"),
]


@pytest.mark.parametrize("kw,message", BAD_WIDTHS, ids=[str(sorted(k.items()))[:50] for k, _ in BAD_WIDTHS])
def test_untested_widths_are_refused_by_both_loaders_with_one_message(hip_library, kw, message):
    cfg = small_config(gen_blocks=1, **kw)
    with pytest.raises(ValueError) as py:
        M.validate_config(cfg)
    assert str(py.value) == "Invalid model: " + message
    # the same header through the C++ loader (weights of a valid model: the header is checked first)
    blob = M.serialize(cfg, WTS, validate=False)
    with pytest.raises(R.JoshUpscaleError) as cc:
        R.validate_model(blob)
    assert cc.value.code == 1 and str(py.value) in str(cc.value), str(cc.value)


# No activation tensor may reach 4 GiB (csrc/model.cpp validateConfig: several kernels address with 32-bit byte offsets).
# The GPU suite runs the engines just under the limit (test_gpu_presets.py test_large_frames_agree_with_crops_of_themselves).
SIZE_CASES = [
    (dict(frame_height=8192, frame_width=8192), False),
    (dict(frame_height=8192, frame_width=4064), True),
    (dict(frame_height=5632, frame_width=5888), True),
    (dict(frame_height=5800, frame_width=5800), False),
    (dict(frame_height=4096, frame_width=4096, gen_filters=256), False),
    (dict(frame_height=2048, frame_width=4000, gen_filters=256), True),
    (dict(frame_height=4096, frame_width=4000, gen_filters=128), True),
    (dict(frame_height=2048, frame_width=2048, flow_filters=(512, 64, 32)), False),
    (dict(frame_height=2048, frame_width=2048, flow_filters=(64, 512, 32)), False),     # the decoder block's output is upsampled to full size
    (dict(frame_height=2048, frame_width=2048, flow_filters=(64, 128, 512, 128, 32)), True),  # 512 channels at a quarter of the pixels at most
    (dict(frame_height=4096, frame_width=4096, flow_arch="resnet", flow_pad_factor=0, flow_res_filters=256), False),
]


@pytest.mark.parametrize("kw,ok", SIZE_CASES, ids=[str(sorted(k.items()))[:60] for k, _ in SIZE_CASES])
def test_frames_whose_tensors_would_reach_4_gib_are_refused_by_both_loaders(hip_library, kw, ok):
    cfg = small_config(gen_blocks=1, **kw)
    blob = M.serialize(cfg, M.make_seeded_weights(cfg), validate=False)
    if ok:
        M.validate_config(cfg)
        R.validate_model(blob)
        return
    with pytest.raises(ValueError) as py:
        M.validate_config(cfg)
    assert "frame too large for this model: an activation tensor would reach 4 GiB" in str(py.value)
    with pytest.raises(R.JoshUpscaleError) as cc:
        R.validate_model(blob)
    assert cc.value.code == 1 and str(py.value) in str(cc.value), str(cc.value)


def test_every_width_the_gpu_suite_runs_is_accepted_by_both_loaders(hip_library):
    import ast
    import os
    src = open(os.path.join(os.path.dirname(__file__), "test_gpu_parity.py")).read()
    tree = ast.parse(src)
    ns = {"LRELU": dict(flow_activation="lrelu", gen_activation="lrelu", gen_negative_slope=0.2)}
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") in ("AE7_WIDE", "WIDTH_CASES"):
            exec(compile(ast.Module([node], []), "widths", "exec"), ns)
    assert len(ns["WIDTH_CASES"]) >= 20
    for name, h, w, kw in ns["WIDTH_CASES"]:
        kw = dict(kw)
        kw.setdefault("gen_blocks", 1)
        cfg = small_config(frame_height=h, frame_width=w, **kw)
        M.validate_config(cfg)
        R.validate_model(M.serialize(cfg, M.make_seeded_weights(cfg)))
