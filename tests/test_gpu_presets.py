"""Every BASELINE.json configuration at its FULL size, on both paths through the boundary
(needs an MI355X): host frames (staging + graph replay, what the reference's AviSynth
caller uses) and JU_LOC_DEVICE frames (no staging; the path bench.py times), in bf16 and
fp16, against

  * whole frames of the float32 C restatement (oracle/ju_oracle_c.c, run on this box's
    host cores), and
  * the committed crops of the float64 oracle, which were generated twice (numpy and the
    independent PyTorch restatement; tests/golden/make_golden.py, tests/test_golden.py).

Tolerances: tests/gpu_common.py.
"""

import hashlib
import os

import numpy as np
import pytest

from gpu_common import GOLD, TOL, check_u8, make, record
from helpers import M, u8_stats
from joshupscale_amd import runtime as R

pytestmark = pytest.mark.gpu

FULL = {"psp-quality": "full_psp_quality", "psp-fast": "full_psp_fast",
        "psp-quality-flowres": "full_psp_quality_flowres", "ps2-quality": "full_ps2_quality"}
_C_REF = {}


def c_reference(preset):
    """Whole-frame outputs of the C restatement on the golden clip (computed once per preset)."""
    if preset not in _C_REF:
        from oracle.c_binding import CSession
        g = np.load(os.path.join(GOLD, FULL[preset] + ".npz"))
        cfg = M.PRESETS[preset]
        blob = M.serialize(cfg, M.make_seeded_weights(cfg))
        assert hashlib.sha256(blob).hexdigest() == str(g["model_sha256"])
        n = int(g["n_frames"])
        frames = M.synthetic_frames(n, cfg.frame_height, cfg.frame_width, seed=int(g["seed"]), kind="smooth")
        assert hashlib.sha256(frames.tobytes()).hexdigest() == str(g["frames_sha256"])
        cs = CSession(blob, cfg.frame_height, cfg.frame_width)
        outs = [cs.run(f).copy() for f in frames]
        # the C restatement itself against the doubly-generated float64 crops: float32 vs
        # float64 can only differ at a truncation boundary
        for t in range(n):
            for k, (y, x) in enumerate(g["crops"]):
                d = np.abs(outs[t][y:y + 64, x:x + 64, :3].astype(int) - g["crops_u8"][t, k].astype(int))
                assert d.max() <= 1 and np.mean(d > 0) < 0.01, (preset, t, k)
        _C_REF[preset] = (g, frames, outs)
    return _C_REF[preset]


@pytest.mark.parametrize("dtype", [R.DTYPE_BF16, R.DTYPE_F16])
@pytest.mark.parametrize("preset", sorted(FULL))
def test_full_size_preset_on_host_and_device_paths(preset, dtype):
    import torch
    g, frames, refs = c_reference(preset)
    cfg = M.PRESETS[preset]
    h, w = cfg.frame_height, cfg.frame_width
    _, blob, rt = make(cfg, dtype)
    assert (rt.input_width, rt.input_height, rt.output_width, rt.output_height) == (w, h, 4 * w, 4 * h)
    n = len(frames)
    # ---- host frames: staging + replay of the captured graph ----
    host = []
    for t in range(n):
        out = rt.process_image(frames[t]).copy()
        st = check_u8(out, refs[t], dtype, ("full", preset, "host", t))
        got = np.stack([out[y:y + 64, x:x + 64] for y, x in g["crops"]])
        ref = np.concatenate([g["crops_u8"][t], np.zeros(g["crops_u8"][t].shape[:3] + (1,), np.uint8)], -1)
        check_u8(got, ref, dtype, ("full-crops", preset, t))
        assert np.abs(out[..., :3].reshape(-1, 3).mean(0) - g["means"][t]).max() < 0.25
        state = rt.read_tensor("state").reshape(4 * h, 4 * w, 4)
        raw_err = max(np.abs(state[y:y + 64, x:x + 64, :3] - g["crops_raw"][t, k]).max()
                      for k, (y, x) in enumerate(g["crops"]))
        record(("full-raw", preset, t), dtype, {"raw": raw_err})
        assert raw_err <= TOL[dtype]["raw"]
        host.append(out)
    assert rt.stat("graph_replays") == n and rt.stat("eager_runs") == 0
    # ---- device frames: the kernels read / write the caller's memory; the path bench.py times ----
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(frames).to(dev)
    d_out = torch.empty((4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for rep in range(3):        # 1st pass eager (tuples seen once), 2nd captures, 3rd replays
        rt.reset()
        for t in range(n):
            rt.process(rt.device_image(d_in[t].data_ptr(), w, h), rt.device_image(d_out.data_ptr(), 4 * w, 4 * h))
            assert np.array_equal(d_out.cpu().numpy(), host[t]), (preset, "device", rep, t)
    assert rt.stat("eager_runs") == n and rt.stat("direct_graphs") == n
    assert rt.stat("graph_replays") == n + 2 * n
    rt.close()


_BENCH_REF = {}


def bench_clip_reference(preset, n=4):
    """Whole-frame outputs of the C restatement on the clip bench.py times: uniform-random
    frames, seed 1234 (rank 0), computed once per preset."""
    if preset not in _BENCH_REF:
        from oracle.c_binding import CSession
        cfg = M.PRESETS[preset]
        blob = M.serialize(cfg, M.make_seeded_weights(cfg, seed=42))
        frames = M.synthetic_frames(16, cfg.frame_height, cfg.frame_width, seed=1234, kind="noise")[:n]
        cs = CSession(blob, cfg.frame_height, cfg.frame_width)
        _BENCH_REF[preset] = (blob, frames, [cs.run(f).copy() for f in frames])
    return _BENCH_REF[preset]


@pytest.mark.parametrize("dtype", [R.DTYPE_BF16, R.DTYPE_F16])
@pytest.mark.parametrize("preset", ["psp-quality", "ps2-quality", "psp-quality-lrelu"])
def test_benchmark_clip_at_full_size_on_host_and_device_paths(preset, dtype):
    """The workload bench.py headlines -- kind="noise", seed 1234, the full frame size -- whole
    frames against the C restatement, 4 recurrent frames, through the host path and the
    device-frame path (registered buffers: graph replay from the first frame).  bf16 is held
    to the noise-clip bound of gpu_common (<= 0.035 % of bytes off by more than 1; measured
    0.022-0.023 %), fp16 to the common one.  `psp-quality-lrelu`: the LEAKY instantiation of
    the resident tower at full size."""
    import torch
    blob, frames, refs = bench_clip_reference(preset)
    cfg = M.PRESETS[preset]
    h, w = cfg.frame_height, cfg.frame_width
    rt = R.Runtime(blob, 0, dtype)
    assert rt.stat("resident_tower") == (0 if preset == "ps2-quality" else 1)
    host = []
    for t, f in enumerate(frames):
        out = rt.process_image(f).copy()
        check_u8(out, refs[t], dtype, ("bench-clip", preset, "host", t), clip="noise")
        host.append(out)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(frames).to(dev)
    d_out = torch.empty((4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    rt.reset()
    ins = [rt.device_image(d_in[t].data_ptr(), w, h) for t in range(len(frames))]
    out_img = rt.device_image(d_out.data_ptr(), 4 * w, 4 * h)
    assert sum(rt.prepare_frames(i, out_img) for i in ins) == 2 * len(frames)
    for t in range(len(frames)):
        rt.process(ins[t], out_img)
        assert np.array_equal(d_out.cpu().numpy(), host[t]), (preset, "device", t)
    assert rt.stat("eager_runs") == 0 and rt.stat("graph_captures") == 0
    rt.close()


@pytest.mark.parametrize("dtype", [R.DTYPE_BF16, R.DTYPE_F16])
@pytest.mark.parametrize("h,w,extra", [(135, 241, {}), (203, 310, dict(gen_activation="lrelu", gen_negative_slope=0.2))])
def test_mid_size_ragged_geometry_against_the_c_restatement(h, w, extra, dtype):
    """Between the small cases the float64 oracle steps through and the presets: ragged sizes with
    many partial 32 x 16 regions (135 x 241: 8 x 9 regions, the last column 17 px wide, the last
    row 7 high; 203 x 310: 10 x 13), padded flow input (135 -> 136 rows), partial 30-px flow tiles,
    six residual blocks -- whole frames against the C restatement, ReLU and LeakyReLU."""
    from oracle.c_binding import CSession
    cfg = M.ModelConfig(frame_height=h, frame_width=w, gen_blocks=6, **extra)
    blob = M.serialize(cfg, M.make_seeded_weights(cfg, seed=7))
    frames = M.synthetic_frames(4, h, w, seed=77, kind="noise")
    cs = CSession(blob, h, w)
    rt = R.Runtime(blob, 0, dtype)
    assert rt.stat("resident_tower") == 1
    for t, f in enumerate(frames):
        check_u8(rt.process_image(f), cs.run(f), dtype, ("mid-size", h, w, sorted(extra), t), clip="noise")
    rt.close()


@pytest.mark.parametrize("dtype", [R.DTYPE_BF16, R.DTYPE_F16, R.DTYPE_FP8])
@pytest.mark.parametrize("h,w", [(16, 8192), (4096, 32), (2, 8192), (4095, 31), (17, 4090), (129, 889), (512, 256), (9, 8161)],
                         ids=lambda v: str(v))
def test_resident_tower_at_the_edges_of_its_geometry(h, w, dtype):
    """The one-launch tower takes any frame of at most 256 regions of 32 x 16 pixels: here the shapes at the edges of
    that rule -- one row of 256 regions (16 x 8192; 2 x 8192: regions two rows high; 9 x 8161: ragged in both directions),
    one column of 256 (4096 x 32, 4095 x 31), two rows of 128 (17 x 4090: 9- and 8-row regions), exactly 256 full
    regions (512 x 256), 9 x 28 ragged (129 x 889) -- whole frames against the C restatement (the 8-bit engine: within
    the quantisation noise of 8 e4m3 convolutions, 40 dB, and it must take its one-launch tower too)."""
    from oracle.c_binding import CSession
    cfg = M.ModelConfig(frame_height=h, frame_width=w, gen_blocks=4)
    blob = M.serialize(cfg, M.make_seeded_weights(cfg, seed=11))
    frames = M.synthetic_frames(3, h, w, seed=5, kind="smooth")
    cs = CSession(blob, h, w)
    rt = R.Runtime(blob, 0, dtype)
    assert rt.stat("resident_tower") == 1
    for t, f in enumerate(frames):
        out, ref = rt.process_image(f), cs.run(f)
        if dtype == R.DTYPE_FP8:
            st = u8_stats(out, ref)
            assert st["psnr"] >= 40.0 and not out[..., 3].any(), (h, w, t, st)
        else:
            check_u8(out, ref, dtype, ("resident-edge", h, w, t))
    rt.close()


@pytest.mark.parametrize("dtype", [R.DTYPE_BF16, R.DTYPE_F16])
def test_full_size_resident_tower_against_the_per_block_kernels(monkeypatch, dtype):
    """The resident tower and the one-launch-per-block kernels share no exchange code and
    accumulate the taps in a different order (the resident kernel runs dx = 1 first for its
    pre-run), so over 49 layers their 16-bit roundings drift apart by a unit here and there:
    at the benchmark size at most 2 LSB, on well under 0.1 % of the bytes (measured: bf16 max 2
    on 0.004 %, fp16 max 1).  A missed halo, a stale fragment or a wrong weight set shows as
    far more than that."""
    cfg = M.PRESETS["psp-quality"]
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames = M.synthetic_frames(3, cfg.frame_height, cfg.frame_width, seed=5, kind="smooth")
    rt = R.Runtime(blob, 0, dtype)
    assert rt.stat("resident_tower") == 1
    resident = [rt.process_image(f).copy() for f in frames]
    rt.close()
    monkeypatch.setenv("JU_TOWER", "layers")
    rt = R.Runtime(blob, 0, dtype)
    assert rt.stat("resident_tower") == 0
    layered = [rt.process_image(f).copy() for f in frames]
    rt.close()
    for t, (a, b) in enumerate(zip(resident, layered)):
        st = u8_stats(a, b)
        record(("resident-vs-blocks", t), dtype, st)
        assert st["max"] <= (2 if dtype == R.DTYPE_BF16 else 1) and st["frac_gt1"] <= 1e-3, (t, st)


@pytest.mark.parametrize("dtype", [R.DTYPE_BF16, R.DTYPE_F16])
def test_pipelined_residual_block_kernel_gives_the_plain_kernels_frames(dtype, monkeypatch):
    """res_block_pipe_kernel (a row pair's epilogue spread over the next pair's MFMAs, ReLU on the packed
    16-bit values, two accumulator sets, the last pair of a tile finished inside the next tile's first K loops,
    the next tile's X issued behind the first conv B loop) against res_block_kernel (JU_RES_BLOCK=plain):
    same arithmetic per element up to the sign of zeros in the intermediate tensors, so the frames and the
    recurrent state must be EQUAL -- at the PS2 size (704 tiles: interior and all four kinds of edge tiles,
    three tiles per workgroup), on frames small enough for one tile per workgroup and ragged in both
    directions, and with the resident tower switched off at 480x270 (JU_TOWER=layers)."""
    lib = R.load_library()
    from helpers import small_config
    monkeypatch.setenv("JU_TOWER", "layers")
    cases = [(M.PRESETS["ps2-quality"], 3), (M.PRESETS["psp-quality"], 2),
             (small_config(frame_height=61, frame_width=97, gen_blocks=3), 4),
             (small_config(frame_height=14, frame_width=30, gen_blocks=2), 3),
             (small_config(frame_height=33, frame_width=31, gen_blocks=2), 3)]
    for cfg, n in cases:
        blob = M.serialize(cfg, M.make_seeded_weights(cfg))
        frames = M.synthetic_frames(n, cfg.frame_height, cfg.frame_width, seed=21, kind="noise")
        outs = {}
        for plain in (0, 1):
            lib.ju_debug_set(b"res_block_plain", plain)
            try:
                rt = R.Runtime(blob, 0, dtype)
                assert rt.stat("resident_tower") == 0
                outs[plain] = [rt.process_image(f).copy() for f in frames]
                outs[(plain, "state")] = rt.read_tensor("state").copy()
                rt.close()
            finally:
                lib.ju_debug_set(b"res_block_plain", 0)
        key = (cfg.frame_height, cfg.frame_width)
        assert np.array_equal(outs[(0, "state")], outs[(1, "state")]), key
        for a, b in zip(outs[0], outs[1]):
            assert np.array_equal(a, b), key


@pytest.mark.parametrize("dtype", [R.DTYPE_BF16, R.DTYPE_F16])
def test_resident_tower_schedule_does_not_change_the_bytes(dtype, monkeypatch):
    """The product schedule of tower_resident_kernel (two units' halo-independent steps run around
    the halo loads with their accumulators kept across the sweep, the next layer's weights refilled
    behind the last unit's MFMAs, hand-issued fragment reads under 480 registers) against the PLAIN
    schedule of the same kernel (debug variant 8: no pre-run, weight burst, natural unit order).
    Both execute the same arithmetic per output element, so the frames must be EQUAL -- a lost
    halo, a fragment register copied before its data landed, or a weight register refilled too
    early shows as a difference here, where the comparison against the per-block kernels (other
    summation order) only bounds it."""
    lib = R.load_library()
    cases = [(M.PRESETS["psp-quality"], 4), (M.PRESETS["psp-fast"], 3)]
    from helpers import small_config
    cases.append((small_config(frame_height=34, frame_width=50, gen_blocks=3), 4))
    cases.append((small_config(frame_height=17, frame_width=33, gen_blocks=2), 3))
    # a small frame whose regions all have the fast schedule's shape (rows 16 / 16 / 14, columns 32 / 32: since round 5 --
    # the 16x16x32 form -- the fast schedule takes full-width regions only and writes its groups without a lane mask) ...
    cases.append((small_config(frame_height=46, frame_width=64, gen_blocks=3), 4))
    # ... and one with a ragged last column (32 / 32 / 6), which therefore runs the general schedule
    cases.append((small_config(frame_height=46, frame_width=70, gen_blocks=3), 4))
    # the LEAKY instantiation (`activation: lrelu`: 16-bit epoch beside the values, f32 LeakyReLU)
    cases.append((M.PRESETS["psp-quality-lrelu"], 3))
    cases.append((small_config(frame_height=34, frame_width=50, gen_blocks=3, gen_activation="lrelu",
                               gen_negative_slope=0.2), 4))
    for cfg, n in cases:
        blob = M.serialize(cfg, M.make_seeded_weights(cfg))
        frames = M.synthetic_frames(n, cfg.frame_height, cfg.frame_width, seed=11, kind="noise")
        # default: the product kernel carries the fused tail (no trunk tensor; the plain-schedule
        # variant has no such form: it writes the trunk and the same tail code runs as a launch of
        # its own) -- frames and state must be equal; JU_TAIL=fused: both write the trunk -- equal too
        for tail_mode in ("tower", "fused"):
            monkeypatch.setenv("JU_TAIL", tail_mode)
            outs = {}
            # 0: the product (the FAST instantiation where every region has its shape: epilogues behind the
            # next unit's MFMAs), "general": the general schedule forced, 8: the plain schedule
            fast_expected = (cfg.frame_height, cfg.frame_width) in ((270, 480), (46, 64))   # (ReLU and LeakyReLU models)
            for variant in (0, "general", 8):
                lib.ju_debug_set(b"tower_variant", 8 if variant == 8 else 0)
                lib.ju_debug_set(b"tower_fast", 0 if variant == "general" else 1)
                try:
                    rt = R.Runtime(blob, 0, dtype)
                    assert rt.stat("resident_tower") == 1 and rt.stat("tower_variant") == (8 if variant == 8 else 0)
                    assert rt.stat("tower_fast") == (1 if fast_expected and variant != "general" else 0), (variant, cfg.frame_height)
                    outs[variant] = [rt.process_image(f).copy() for f in frames]
                    outs[(variant, "state")] = rt.read_tensor("state").copy()
                    outs[(variant, "trunk")] = rt.read_tensor("trunk").copy()
                    rt.close()
                finally:
                    lib.ju_debug_set(b"tower_variant", 0)
                    lib.ju_debug_set(b"tower_fast", 1)
            key = (cfg.frame_height, cfg.frame_width, cfg.gen_activation, tail_mode)
            assert np.array_equal(outs[("general", "state")], outs[(0, "state")]), key
            for a, b in zip(outs[0], outs["general"]):
                assert np.array_equal(a, b), key
            if tail_mode == "fused" or cfg.gen_activation != "relu":   # (lrelu: no fused-tail form, the trunk is written)
                assert np.array_equal(outs[(0, "trunk")], outs[(8, "trunk")]), key
            assert np.array_equal(outs[(0, "state")], outs[(8, "state")]), key
            for a, b in zip(outs[0], outs[8]):
                assert np.array_equal(a, b), key
        monkeypatch.delenv("JU_TAIL")


def test_device_frame_graphs_equal_eager_launches(monkeypatch):
    """Config 3 (psp-fast, fp16, latency-optimised graph capture): on device frames the
    cached graph of a frame-buffer tuple replays exactly what the eager launches do, for a
    clip whose buffers are reused, including bottom-up (negative stride) device frames."""
    import torch
    cfg = M.PRESETS["psp-fast"]
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames = M.synthetic_frames(4, h, w, seed=33, kind="smooth")
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(np.ascontiguousarray(frames[:, ::-1])).to(dev)     # stored bottom-up
    d_out = torch.empty((2, 4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)   # two output buffers
    torch.cuda.synchronize()

    def run(rt, loops):
        outs = []
        for i in range(loops * 4):
            t = i % 4
            src = rt.device_image(d_in[t].data_ptr() + (h - 1) * w * 4, w, h, stride=-w * 4)
            dst = rt.device_image(d_out[i % 2].data_ptr(), 4 * w, 4 * h)
            rt.process(src, dst)
            outs.append(d_out[i % 2].cpu().numpy())
        return outs

    monkeypatch.setenv("JU_DIRECT_GRAPH", "0")
    eager_rt = R.Runtime(blob, 0, R.DTYPE_F16)
    eager = run(eager_rt, 3)
    assert eager_rt.stat("graph_replays") == 0 and eager_rt.stat("eager_runs") == 12
    eager_rt.close()
    monkeypatch.delenv("JU_DIRECT_GRAPH")
    rt = R.Runtime(blob, 0, R.DTYPE_F16)
    graph = run(rt, 3)
    # 4 inputs x 2 outputs... tuple (t, i % 2, idx): t = i % 4 fixes i % 2 and the binding set
    # alternates with i, so there are 4 tuples: seen in loop 1, captured in loop 2, replayed in 3
    assert rt.stat("eager_runs") == 4 and rt.stat("graph_replays") == 8 and rt.stat("direct_graphs") == 4
    for a, b in zip(eager, graph):
        assert np.array_equal(a, b)
    rt.close()


def test_prepare_frames_captures_in_setup_and_the_loop_only_replays():
    """ju_prepare_frames: the graphs of a registered device frame-buffer pair are captured at
    registration (both binding sets), as the reference captures its graphs in the constructor
    (tensorrt_backend.cc:257-263); the frame loop then replays from its first frame on -- no
    eager first sighting, no capture inside ju_process -- and produces the bytes of an
    unregistered runtime.  A dropped cache (fallback to the per-block path) re-captures a
    registered pair at its first use."""
    import torch
    cfg = M.PRESETS["psp-fast"]
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames = M.synthetic_frames(5, h, w, seed=35, kind="noise")
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(frames).to(dev)
    d_out = torch.empty((4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    plain = R.Runtime(blob, 0, R.DTYPE_BF16)
    want = []
    for i in range(20):   # (in_t, out, binding set) repeats with period 10: eager, then captured inline
        plain.process(plain.device_image(d_in[i % 5].data_ptr(), w, h), plain.device_image(d_out.data_ptr(), 4 * w, 4 * h))
        want.append(d_out.cpu().numpy())
    assert plain.stat("graph_captures") == 10 and plain.stat("eager_runs") == 10     # second-sighting policy
    plain.close()
    rt = R.Runtime(blob, 0, R.DTYPE_BF16)
    ins = [rt.device_image(d_in[t].data_ptr(), w, h) for t in range(5)]
    out = rt.device_image(d_out.data_ptr(), 4 * w, 4 * h)
    assert [rt.prepare_frames(i, out) for i in ins] == [2] * 5                   # one graph per binding set
    assert rt.prepare_frames(ins[0], out) == 0                                    # registered already
    assert rt.prepare_frames(R.host_image(frames[0]), R.host_image(np.empty((4 * h, 4 * w, 4), np.uint8))) == 0
    with pytest.raises(R.JoshUpscaleError):
        rt.prepare_frames(rt.device_image(d_in[0].data_ptr(), w - 1, h), out)
    assert rt.stat("prepared_captures") == 10 and rt.stat("registered_pairs") == 5
    for i in range(10):
        rt.process(ins[i % 5], out)
        assert np.array_equal(d_out.cpu().numpy(), want[i]), i
    assert rt.stat("graph_replays") == 10 and rt.stat("eager_runs") == 0 and rt.stat("graph_captures") == 0
    # the cache is dropped when the engine leaves the resident kernel; a registered pair is
    # captured again at its FIRST use afterwards (one inline capture, no eager run)
    lib = R.load_library()
    rt.reset()
    d_other = torch.empty_like(d_out)
    lib.ju_debug_set(b"resident_fault", 1)            # (acts on new launches: an unregistered pair runs eagerly)
    try:
        rt.process(ins[0], rt.device_image(d_other.data_ptr(), 4 * w, 4 * h))   # times out, falls back, re-runs
    finally:
        lib.ju_debug_set(b"resident_fault", 0)
    assert rt.stat("resident_tower") == 0 and rt.stat("direct_graphs") == 0
    rt.process(ins[1], out)
    assert rt.stat("graph_captures") == 1 and rt.stat("eager_runs") == 2      # (the faulted frame and its re-run)
    rt.close()


def test_registered_pairs_are_bounded_and_the_oldest_is_forgotten():
    """ju_prepare_frames keeps at most 256 pairs: a caller that registers new buffers for ever does
    not grow the graph cache without bound; a forgotten pair still works (it is captured again at
    its second use, like any unregistered pair) and produces the same bytes."""
    import torch
    from helpers import small_config
    cfg = small_config()
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    h, w = cfg.frame_height, cfg.frame_width
    frames = M.synthetic_frames(3, h, w, seed=61, kind="smooth")
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(frames).to(dev)
    outs = torch.empty((260, 4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    rt = R.Runtime(blob, 0, R.DTYPE_F16)
    want = [rt.process_image(f).copy() for f in frames]
    rt.reset()
    src = rt.device_image(d_in[0].data_ptr(), w, h)
    for k in range(260):
        assert rt.prepare_frames(src, rt.device_image(outs[k].data_ptr(), 4 * w, 4 * h)) == 2
    assert rt.stat("registered_pairs") == 256 and rt.stat("direct_graphs") == 512
    # pair 0 was forgotten (registered first, never used): its first use is an eager run again
    for t in range(3):
        rt.process(rt.device_image(d_in[t].data_ptr(), w, h), rt.device_image(outs[0].data_ptr(), 4 * w, 4 * h))
        assert np.array_equal(outs[0].cpu().numpy(), want[t]), t
    assert rt.stat("eager_runs") == 3        # (three different inputs: three tuples seen once)
    # a pair that is still registered replays from its first frame
    rt.reset()
    before = rt.stat("graph_replays")
    rt.process(src, rt.device_image(outs[259].data_ptr(), 4 * w, 4 * h))
    assert rt.stat("graph_replays") == before + 1 and np.array_equal(outs[259].cpu().numpy(), want[0])
    rt.close()


def test_second_runtime_is_created_while_the_first_has_frames_in_flight():
    """Advisor finding: a runtime created while another resident runtime has enqueued frames
    must not run its constructor's tower launches beside them, and two threads calling the
    synchronous ju_process on two resident runtimes must serialise wait + launches + record
    per device.  Both streams produce their solo frames; neither falls back."""
    import threading
    import torch
    cfg = M.PRESETS["psp-fast"]
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    n = 20
    clips = [M.synthetic_frames(n, h, w, seed=91 + k, kind="smooth") for k in range(2)]
    dev = torch.device("cuda", 0)
    d_in = [torch.from_numpy(c).to(dev) for c in clips]
    solo = []
    for k in range(2):
        rt = R.Runtime(blob, 0, R.DTYPE_F16)
        d_out = torch.empty((n, 4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for t in range(n):
            rt.process(rt.device_image(d_in[k][t].data_ptr(), w, h), rt.device_image(d_out[t].data_ptr(), 4 * w, 4 * h))
        solo.append(d_out.cpu())
        rt.close()
    outs = [torch.empty((n, 4 * h, 4 * w, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    first = R.Runtime(blob, 0, R.DTYPE_F16)
    for t in range(n // 2):                                  # in flight, not waited for
        first.enqueue(first.device_image(d_in[0][t].data_ptr(), w, h), first.device_image(outs[0][t].data_ptr(), 4 * w, 4 * h))
    second = R.Runtime(blob, 0, R.DTYPE_F16)                 # constructor runs both programs eagerly
    rts = [first, second]
    errors = []

    def worker(k, start):
        try:
            for t in range(start, n):
                rts[k].process(rts[k].device_image(d_in[k][t].data_ptr(), w, h),
                               rts[k].device_image(outs[k][t].data_ptr(), 4 * w, 4 * h))
        except Exception as e:   # noqa: BLE001
            errors.append((k, e))

    threads = [threading.Thread(target=worker, args=(0, n // 2)), threading.Thread(target=worker, args=(1, 0))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not errors, errors
    for k in range(2):
        assert rts[k].stat("resident_tower") == 1 and rts[k].stat("fallbacks") == 0, k
        assert torch.equal(outs[k].cpu(), solo[k]), k
        rts[k].close()


def test_resident_failure_recovers_with_graphs_enabled():
    """The default configuration (graphs on): a bounded wait of the resident tower expires
    on an eagerly launched device frame; the engine runs the per-layer programs once
    eagerly, captures them, re-runs the frame, and later host frames replay the new graphs."""
    import torch
    from helpers import small_config
    cfg = small_config()
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames = M.synthetic_frames(5, 30, 48, seed=41, kind="smooth")
    ref_rt = R.Runtime(blob, 0, R.DTYPE_BF16)
    want = [ref_rt.process_image(f).copy() for f in frames]
    ref_rt.close()
    rt = R.Runtime(blob, 0, R.DTYPE_BF16)
    assert rt.stat("resident_tower") == 1
    lib = R.load_library()
    seen = []
    cb = R.LOG_CALLBACK(lambda tag, lvl, msg, user: seen.append((lvl, msg)))
    lib.ju_set_log_callback(cb, None)
    got = [rt.process_image(frames[0]).copy(), rt.process_image(frames[1]).copy()]   # graph replays
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(frames[2]).to(dev)
    d_out = torch.empty((120, 192, 4), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    assert lib.ju_debug_set(b"resident_fault", 1) == 0
    try:
        rt.process(rt.device_image(d_in.data_ptr(), 48, 30), rt.device_image(d_out.data_ptr(), 192, 120))
    finally:
        lib.ju_debug_set(b"resident_fault", 0)
        lib.ju_set_log_callback(R.LOG_CALLBACK(0), None)
    got.append(d_out.cpu().numpy())
    assert rt.stat("resident_tower") == 0
    replays = rt.stat("graph_replays")
    got += [rt.process_image(f).copy() for f in frames[3:]]      # the re-captured per-layer graphs
    assert rt.stat("graph_replays") == replays + 2
    assert any(lvl == 1 and b"per-layer" in msg for lvl, msg in seen), seen
    assert all(u8_stats(a, b)["max"] <= 1 for a, b in zip(want, got))
    rt.close()


def test_resident_failure_is_recoverable(monkeypatch):
    """After a bounded-wait failure the engine uses the per-block kernels, but not for good:
    after JU_RESIDENT_RETRY clean frames it goes back to the one-launch tower (with the
    mailbox and the slot epochs reset), and backs off x4 when that fails again."""
    from helpers import small_config
    monkeypatch.setenv("JU_RESIDENT_RETRY", "3")
    monkeypatch.setenv("JU_NO_GRAPH", "1")       # the fault hook acts on new launches
    cfg = small_config()
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames = M.synthetic_frames(12, 30, 48, seed=43, kind="smooth")
    ref_rt = R.Runtime(blob, 0, R.DTYPE_BF16)
    want = [ref_rt.process_image(f).copy() for f in frames]
    ref_rt.close()
    rt = R.Runtime(blob, 0, R.DTYPE_BF16)
    lib = R.load_library()
    got = [rt.process_image(frames[0]).copy()]
    assert rt.stat("resident_tower") == 1 and rt.stat("fallbacks") == 0
    lib.ju_debug_set(b"resident_fault", 1)
    try:
        got.append(rt.process_image(frames[1]).copy())          # times out, falls back, re-runs
    finally:
        lib.ju_debug_set(b"resident_fault", 0)
    assert rt.stat("resident_tower") == 0 and rt.stat("fallbacks") == 1
    got.append(rt.process_image(frames[2]).copy())
    got.append(rt.process_image(frames[3]).copy())
    assert rt.stat("resident_tower") == 0
    got.append(rt.process_image(frames[4]).copy())               # third clean frame: resident again
    assert rt.stat("resident_tower") == 1
    got += [rt.process_image(f).copy() for f in frames[5:8]]     # ... and it works (epochs start over)
    assert rt.stat("resident_tower") == 1 and rt.stat("fallbacks") == 1
    lib.ju_debug_set(b"resident_fault", 1)
    try:
        got.append(rt.process_image(frames[8]).copy())
    finally:
        lib.ju_debug_set(b"resident_fault", 0)
    assert rt.stat("resident_tower") == 0 and rt.stat("fallbacks") == 2
    got += [rt.process_image(f).copy() for f in frames[9:]]      # 3 clean frames < the backed-off 12
    assert rt.stat("resident_tower") == 0
    assert all(u8_stats(a, b)["max"] <= 1 for a, b in zip(want, got))
    rt.close()


def test_two_resident_runtimes_enqueue_concurrently():
    """Two runtimes on one GPU, each on its own thread and stream (two OBS filters): two
    resident towers cannot be co-resident, so the runtimes chain their frames through
    events.  Both streams must produce exactly their solo frames, without a fallback."""
    import threading
    import torch
    cfg = M.PRESETS["psp-fast"]
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    n = 24
    clips = [M.synthetic_frames(n, h, w, seed=81 + k, kind="smooth") for k in range(2)]
    dev = torch.device("cuda", 0)
    d_in = [torch.from_numpy(c).to(dev) for c in clips]
    solo = []
    for k in range(2):
        rt = R.Runtime(blob, 0, R.DTYPE_F16)
        d_out = torch.empty((n, 4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for t in range(n):
            rt.enqueue(rt.device_image(d_in[k][t].data_ptr(), w, h), rt.device_image(d_out[t].data_ptr(), 4 * w, 4 * h))
        rt.synchronize()
        solo.append(d_out.cpu())
        rt.close()
    rts = [R.Runtime(blob, 0, R.DTYPE_F16) for _ in range(2)]
    outs = [torch.empty((n, 4 * h, 4 * w, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    errors = []

    def worker(k):
        try:
            for t in range(n):
                rts[k].enqueue(rts[k].device_image(d_in[k][t].data_ptr(), w, h),
                               rts[k].device_image(outs[k][t].data_ptr(), 4 * w, 4 * h))
            rts[k].synchronize()
        except Exception as e:   # noqa: BLE001
            errors.append((k, e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not errors, errors
    for k in range(2):
        assert rts[k].stat("resident_tower") == 1 and rts[k].stat("fallbacks") == 0
        assert torch.equal(outs[k].cpu(), solo[k]), k
        rts[k].close()


def test_fp8_resident_failure_falls_back_to_the_block_kernels(monkeypatch):
    """The one-launch 8-bit tower has the same bounded waits and the same way out as the
    16-bit one: on a timeout the engine re-runs the frame on the per-block kernels -- which
    compute the same bytes -- and comes back after JU_RESIDENT_RETRY clean frames."""
    from helpers import small_config
    monkeypatch.setenv("JU_RESIDENT_RETRY", "2")
    monkeypatch.setenv("JU_NO_GRAPH", "1")
    cfg = small_config()
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames = M.synthetic_frames(6, 30, 48, seed=47, kind="smooth")
    ref_rt = R.Runtime(blob, 0, R.DTYPE_FP8)
    want = [ref_rt.process_image(f).copy() for f in frames]
    ref_rt.close()
    rt = R.Runtime(blob, 0, R.DTYPE_FP8)
    assert rt.stat("resident_tower") == 1
    lib = R.load_library()
    got = [rt.process_image(frames[0]).copy()]
    lib.ju_debug_set(b"resident_fault", 1)
    try:
        got.append(rt.process_image(frames[1]).copy())
    finally:
        lib.ju_debug_set(b"resident_fault", 0)
    assert rt.stat("resident_tower") == 0 and rt.stat("fallbacks") == 1
    got += [rt.process_image(f).copy() for f in frames[2:4]]
    assert rt.stat("resident_tower") == 1                       # back after two clean frames
    got += [rt.process_image(f).copy() for f in frames[4:]]
    assert all(np.array_equal(a, b) for a, b in zip(want, got))  # every form computes the same bytes
    rt.close()


# ---------------------------------------------------------------------------
# frames far larger than any CPU oracle finishes: a size-independent property
# ---------------------------------------------------------------------------
_CROP, _MARGIN = 384, 128


def _crop_window(y0, x0, H, W):
    """rows / columns of a crop (LR) farther than _MARGIN from every border the crop does NOT share with the frame"""
    return ((0 if y0 == 0 else _MARGIN), (_CROP if y0 + _CROP == H else _CROP - _MARGIN),
            (0 if x0 == 0 else _MARGIN), (_CROP if x0 + _CROP == W else _CROP - _MARGIN))


@pytest.mark.parametrize("h,w,dtype,kw", [
    (5632, 5888, R.DTYPE_BF16, {}),                       # 4.25 GB tensors, just under the loader's limit: res_block_pipe_kernel
    (4104, 4104, R.DTYPE_F16, {}),                        # just over 2 GiB (where that kernel's descriptors used to end)
    (5632, 5888, R.DTYPE_FP8, {}),                        # the 8-bit block kernels' 32-bit offsets near their end
    (5632, 5888, R.DTYPE_BF16, dict(gen_activation="lrelu", gen_negative_slope=0.2)),   # the plain res_block_kernel
    (4096, 4000, R.DTYPE_BF16, dict(gen_filters=128)),    # the generic convolution kernels at 4.2 GB
    (2176, 3840, R.DTYPE_BF16, dict(flow_arch="resnet", flow_pad_factor=0, flow_res_blocks=1)),
    # normalize_brightness: the frame's mean enters every pixel.  A bright pattern tiled 15 x 15 times has the mean of
    # one tile -- and channel sums of 7.5e9, which one 32-bit word does not hold (frame_sums_kernel's did not, until
    # this test)
    (5760, 5760, R.DTYPE_BF16, dict(normalize_brightness=True)),
    # the temporal filter with a gate that never calls a scene cut (threshold 1): the blend is local, its statistic
    # (1.6e9 elements into 32.32 fixed point) must not wrap or flip it
    (5632, 5888, R.DTYPE_F16, dict(temporal_strength=0.6, temporal_threshold=1.0, temporal_norm="L2", temporal_luma=True,
                                   temporal_limit=True)),
], ids=["bf16-33Mpx", "fp16-over-2GiB", "fp8-33Mpx", "lrelu-33Mpx", "gen128-16Mpx", "flowres-8Mpx", "brightness-33Mpx",
        "temporal-33Mpx"])
def test_large_frames_agree_with_crops_of_themselves(h, w, dtype, kw):
    """No CPU restatement finishes 33 M-pixel frames, but the network is local: farther than its receptive field (flow
    auto-encoder + warp + a two-block generator: about 70 LR pixels) from a crop's inner borders, the big frame's output
    IS the crop's output.  Crops at the top-left corner, the centre and the bottom-right corner -- the highest addresses of
    every tensor, where a 32-bit offset, a descriptor's range or an int index gives out first.  (It did: until round 4
    res_block_pipe_kernel's buffer descriptors ended at 2 GiB and frames beyond 4096 x 4096 came out wrong below that
    line, 20-40 LSB, with no error.)  Two recurrent frames; the second may differ by 1 LSB in a few per cent of the bytes
    (kernel choice and with it the fp32 summation order of a layer depends on the geometry)."""
    cfg = M.ModelConfig(frame_height=h, frame_width=w, gen_blocks=2, **kw)
    wts = M.make_seeded_weights(cfg, seed=42)
    rng = np.random.default_rng(3)
    bright = bool(kw.get("normalize_brightness"))
    if bright:
        tile = rng.integers(200, 256, size=(2, _CROP, _CROP, 4), dtype=np.uint8)
        frames = np.tile(tile, (1, h // _CROP, w // _CROP, 1))
        assert frames.shape[1:3] == (h, w) and int(frames[0, ..., 0].sum(dtype=np.uint64)) > 2 ** 32
    else:
        frames = rng.integers(0, 256, size=(2, h, w, 4), dtype=np.uint8)
    # (brightness: crops that ARE tiles -- every crop then has the frame's mean -- and only interior ones: a tile at the
    # frame's corner has the frame border on two sides, its twin run alone has it on four)
    spots = {"top-left": (0, 0), "centre": ((h - _CROP) // 2 // 8 * 8, (w - _CROP) // 3 // 8 * 8),
             "bottom-right": (h - _CROP, w - _CROP)}
    if bright:
        spots = {"tile-1-1": (_CROP, _CROP), "tile-7-5": (7 * _CROP, 5 * _CROP), "tile-13-13": (13 * _CROP, 13 * _CROP)}
    rt = R.Runtime(M.serialize(cfg, wts), 0, dtype)
    keep = {k: [] for k in spots}
    for f in frames:
        out = rt.process_image(f)
        assert not out[::97, ::89, 3].any()
        for k, (y0, x0) in spots.items():
            ya, yb, xa, xb = _crop_window(y0, x0, h, w)
            keep[k].append(out[4 * (y0 + ya):4 * (y0 + yb), 4 * (x0 + xa):4 * (x0 + xb), :3].copy())
        del out
    rt.close()
    small = M.ModelConfig(frame_height=_CROP, frame_width=_CROP, gen_blocks=2, **kw)
    for k, (y0, x0) in spots.items():
        rs = R.Runtime(M.serialize(small, wts), 0, dtype)
        ya, yb, xa, xb = _crop_window(y0, x0, h, w)
        for t, f in enumerate(frames):
            o = rs.process_image(np.ascontiguousarray(f[y0:y0 + _CROP, x0:x0 + _CROP]))[4 * ya:4 * yb, 4 * xa:4 * xb, :3]
            d = np.abs(o.astype(np.int16) - keep[k][t].astype(np.int16))
            assert o.std() > 10.0                                   # (a frame, not a constant)
            # (brightness: float(sum) / N of the frame and of the tile round differently in the last place)
            assert d.max() <= (0 if t == 0 and not bright else 1), (k, t, int(d.max()), float(np.mean(d > 0)))
            assert np.mean(d > 0) <= 0.12, (k, t, float(np.mean(d > 0)))
        rs.close()
    record(("large-frame crops", h, w, sorted(kw)), dtype, {"crops": 3, "frames": 2, "max_lsb": 1})


@pytest.mark.parametrize("dtype", [R.DTYPE_BF16, R.DTYPE_FP8])
def test_creating_and_destroying_runtimes_leaks_nothing(dtype):
    """The OBS plugin destroys and recreates its Runtime on every model or size change (obs_plugin/src/filter.cc:
    createRuntime in the update callback): 24 cycles of create -> host frames -> registered device frames (hipGraphs
    captured and replayed) -> destroy must leave the device's free memory, the process's resident set and its open
    file descriptors where they were."""
    import torch

    def rss_mib():
        with open("/proc/self/statm") as f:
            return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2 ** 20

    cfg = M.PRESETS["psp-quality"]
    blob = M.serialize(cfg, M.make_seeded_weights(cfg, seed=42))
    h, w = cfg.frame_height, cfg.frame_width
    frames = M.synthetic_frames(4, h, w, seed=1, kind="noise")
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(frames).to(dev)
    d_out = torch.empty((4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)

    def cycle():
        rt = R.Runtime(blob, 0, dtype)
        for f in frames[:2]:
            rt.process_image(f)
        ins = [rt.device_image(d_in[t].data_ptr(), w, h) for t in range(4)]
        out = rt.device_image(d_out.data_ptr(), 4 * w, 4 * h)
        assert sum(rt.prepare_frames(i, out) for i in ins) > 0
        for t in range(8):
            rt.process(ins[t % 4], out)
        rt.close()

    cycle()
    cycle()            # (first uses: kernel modules loaded, allocator pools grown)
    torch.cuda.synchronize()
    free0, rss0, fds0 = torch.cuda.mem_get_info()[0], rss_mib(), len(os.listdir("/proc/self/fd"))
    for _ in range(24):
        cycle()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] <= 8 * 2 ** 20
    assert rss_mib() - rss0 <= 32.0
    assert len(os.listdir("/proc/self/fd")) - fds0 <= 2


def _mid_fuzz(n, seed):
    """Seeded random mid-size models (the sizes where the one-launch tower has several rows of regions, the flow blocks
    take their tall tiles and the coarse levels their split-K kernels): against the C restatement."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        arch = "autoencoder" if rng.random() < 0.7 else "resnet"
        kw = dict(frame_height=int(rng.integers(64, 300)), frame_width=int(rng.integers(64, 520)),
                  gen_filters=int(rng.choice([64, 64, 64, 32, 128])), gen_blocks=int(rng.integers(1, 7)),
                  num_flow_inputs=int(rng.integers(1, 5)), normalize_brightness=bool(rng.random() < 0.25), flow_arch=arch)
        if arch == "autoencoder":
            if rng.random() < 0.5:
                kw["flow_filters"] = tuple(int(x) for x in rng.choice([32, 64, 128], size=int(rng.choice([3, 5, 7]))))
            kw["flow_pad_factor"] = int(rng.choice([8, 8, 16]))
        else:
            kw.update(flow_pad_factor=int(rng.choice([0, 8])), flow_res_filters=int(rng.choice([64, 64, 32])),
                      flow_res_blocks=int(rng.integers(0, 3)))
        if rng.random() < 0.3:
            kw.update(gen_activation="lrelu", gen_negative_slope=0.2)
        if rng.random() < 0.3:
            kw.update(flow_activation="lrelu", flow_negative_slope=0.1)
        try:
            M.validate_config(M.ModelConfig(**kw))
        except ValueError:
            continue
        out.append(kw)
    return out


MID_FUZZ = _mid_fuzz(int(os.environ.get("JU_FUZZ_N", "6")), seed=int(os.environ.get("JU_FUZZ_SEED", "20260412")))


@pytest.mark.parametrize("k", range(len(MID_FUZZ)), ids=[f"{i}-{c['frame_height']}x{c['frame_width']}-{c['flow_arch'][:3]}-g{c['gen_filters']}"
                                                        for i, c in enumerate(MID_FUZZ)])
def test_random_mid_size_models_against_the_c_restatement(k):
    from oracle.c_binding import CSession
    kw = MID_FUZZ[k]
    dtype = R.DTYPE_BF16 if k % 2 else R.DTYPE_F16
    cfg = M.ModelConfig(**kw)
    blob = M.serialize(cfg, M.make_seeded_weights(cfg, seed=300 + k))
    h, w = cfg.frame_height, cfg.frame_width
    cs = CSession(blob, h, w)
    rt = R.Runtime(blob, 0, dtype)
    kind = "smooth" if k % 3 else "noise"
    for t, f in enumerate(M.synthetic_frames(3, h, w, seed=400 + k, kind=kind)):
        check_u8(rt.process_image(f), cs.run(f), dtype, ("mid-fuzz", k, t), clip=kind)
    rt.close()
