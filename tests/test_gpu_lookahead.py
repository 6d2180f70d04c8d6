"""Frame look-ahead (ju_process_batch; engine.cpp "Frame look-ahead") -- needs an MI355X.

The flow net reads LR frames only (reference models.py:790, 823: its input is the current frame and the history of the last
num_flow_inputs frames), so the flow fields of several consecutive frames are computed in one pass of the flow net's
launches; warp, tower and tail stay frame by frame.  The contract is byte equality with ju_process called frame by
frame -- frames, recurrent state and frame history -- for every pass length, both parities, across passes, mixed with
plain calls, after a reset, and through the resident tower's fallback.
"""

import dataclasses

import numpy as np
import pytest

from helpers import M, small_config
from joshupscale_amd import runtime as R

pytestmark = pytest.mark.gpu


def _device_clip(cfg, n, seed, outs=1):
    import torch
    h, w = cfg.frame_height, cfg.frame_width
    frames = M.synthetic_frames(n, h, w, seed=seed, kind="noise")
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(frames).to(dev)
    d_out = torch.zeros((outs, 4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    return frames, d_in, d_out


def _frame_by_frame(blob, dtype, cfg, d_in, d_out, n):
    h, w = cfg.frame_height, cfg.frame_width
    want = []
    with R.Runtime(blob, 0, dtype) as rt:
        out = rt.device_image(d_out[0].data_ptr(), 4 * w, 4 * h)
        for t in range(n):
            rt.process(rt.device_image(d_in[t].data_ptr(), w, h), out)
            want.append(d_out[0].cpu().numpy())
        state = rt.read_tensor("state").copy()
    return want, state


@pytest.mark.parametrize("preset,dtype", [("psp-quality", R.DTYPE_BF16), ("psp-fast", R.DTYPE_F16),
                                          ("ps2-quality", R.DTYPE_FP8), ("psp-quality-lrelu", R.DTYPE_BF16)])
def test_look_ahead_passes_give_the_frame_by_frame_bytes(preset, dtype):
    """Passes of every length 2..8 back to back (21 frames + a plain call between them: both parities of the binding
    set at a pass's start, the state and history handed from pass to pass and to ju_process and back), one output
    buffer per frame.  Every frame, and the recurrent state at the end, equal the frame-by-frame runtime's."""
    cfg = M.PRESETS[preset]
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    n = 2 + 3 + 1 + 4 + 5 + 8 + 1 + 6 + 7
    frames, d_in, d_out = _device_clip(cfg, n, seed=71, outs=8)
    want, want_state = _frame_by_frame(blob, dtype, cfg, d_in, d_out, n)
    with R.Runtime(blob, 0, dtype) as rt:
        assert rt.stat("lookahead_max") == 8
        t = 0
        for k in (2, 3, 1, 4, 5, 8, 1, 6, 7):
            ins = [rt.device_image(d_in[t + i].data_ptr(), w, h) for i in range(k)]
            outs = [rt.device_image(d_out[i].data_ptr(), 4 * w, 4 * h) for i in range(k)]
            if k == 1:
                rt.process(ins[0], outs[0])
            else:
                rt.process_batch(ins, outs)
            got = d_out[:k].cpu().numpy()
            for i in range(k):
                assert np.array_equal(got[i], want[t + i]), (preset, "pass of", k, "frame", t + i)
            t += k
        assert np.array_equal(rt.read_tensor("state"), want_state)
        # (split-K, the generic kernel and the upsampling launch -- the 128-filter blocks of frames larger than 640x448 --
        # all have an item dimension too: test_models_without_... covers what has none)
        assert rt.stat("lookahead_frames") == n - 2, "the passes did not take the look-ahead path"
        assert rt.stat("fallbacks") == 0


def test_look_ahead_replays_graphs_and_takes_long_and_ragged_calls():
    """A call of any length is cut into passes of at most JU_LOOKAHEAD frames; a tuple of frame buffers seen the
    second time is a captured graph from then on (first sighting eager, like ju_process), and all frames of a pass
    may share one output buffer (the caller's business: the last frame's pixels remain)."""
    cfg = M.PRESETS["psp-fast"]
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames, d_in, d_out = _device_clip(cfg, 8, seed=5)
    want = []
    with R.Runtime(blob, 0, R.DTYPE_BF16) as rt:
        out = rt.device_image(d_out[0].data_ptr(), 4 * w, 4 * h)
        for t in range(40):
            rt.process(rt.device_image(d_in[t % 8].data_ptr(), w, h), out)
            if t % 4 == 3 or t == 39:
                want.append(d_out[0].cpu().numpy())
    with R.Runtime(blob, 0, R.DTYPE_BF16) as rt:
        out = rt.device_image(d_out[0].data_ptr(), 4 * w, 4 * h)
        ring = [rt.device_image(d_in[t].data_ptr(), w, h) for t in range(8)]
        for p in range(10):  # passes of four over a ring of eight: two tuples x (one binding set each)
            rt.process_batch([ring[(4 * p + i) % 8] for i in range(4)], [out] * 4)
            assert np.array_equal(d_out[0].cpu().numpy(), want[p]), p
        assert rt.stat("lookahead_frames") == 40
        # tuples: (frames 0-3 | 4-7) x the binding set at the pass's start, which flips once per pass and so repeats
        # with the ring: 2 tuples -> 2 eager passes, 2 captures, 8 replays
        assert rt.stat("eager_runs") == 2 and rt.stat("graph_captures") == 2 and rt.stat("graph_replays") == 8
    with R.Runtime(blob, 0, R.DTYPE_BF16) as rt:   # one call with 19 frames: passes of 8, 8, 3
        out = rt.device_image(d_out[0].data_ptr(), 4 * w, 4 * h)
        rt.process_batch([rt.device_image(d_in[t % 8].data_ptr(), w, h) for t in range(19)], [out] * 19)
        assert rt.stat("lookahead_frames") == 19
        rt.process(rt.device_image(d_in[19 % 8].data_ptr(), w, h), out)
        assert np.array_equal(d_out[0].cpu().numpy(), want[4])
        rt.process_batch([], [])


def test_look_ahead_mixes_with_host_frames_and_respects_the_cap(monkeypatch):
    """A host frame between device frames rides in the same pass (round 6: staged through the pass's own device
    buffers, its output copied out while the next frame runs); the cap -- ju_set_lookahead, or JU_LOOKAHEAD as the
    default of new runtimes -- turns the passes off altogether.  Same bytes every way."""
    import torch
    cfg = M.PRESETS["psp-fast"]
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames, d_in, d_out = _device_clip(cfg, 7, seed=23, outs=7)
    want, _ = _frame_by_frame(blob, R.DTYPE_BF16, cfg, d_in, d_out, 7)
    host_out = np.zeros((4 * h, 4 * w, 4), np.uint8)
    with R.Runtime(blob, 0, R.DTYPE_BF16) as rt:
        ins = [rt.device_image(d_in[t].data_ptr(), w, h) for t in range(7)]
        outs = [rt.device_image(d_out[t].data_ptr(), 4 * w, 4 * h) for t in range(7)]
        ins[3], outs[3] = R.host_image(frames[3]), R.host_image(host_out)   # device x3, host, device x3
        rt.process_batch(ins, outs)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        for t in range(7):
            assert np.array_equal(host_out if t == 3 else got[t], want[t]), t
        assert rt.stat("lookahead_frames") == 7 and rt.stat("lookahead_host_frames") == 1
        # the setter: passes of at most 3 from now on (7 frames = 3 + 3 + 1), then off
        rt.reset()
        rt.set_lookahead(3)
        assert rt.stat("lookahead_max") == 3
        rt.process_batch(ins, outs)
        got = d_out.cpu().numpy()
        for t in range(7):
            assert np.array_equal(host_out if t == 3 else got[t], want[t]), t
        assert rt.stat("lookahead_frames") == 7 + 6
        rt.reset()
        rt.set_lookahead(0)                                                 # clamped to 1 = frame by frame
        assert rt.stat("lookahead_max") == 1
        rt.process_batch(ins, outs)
        assert rt.stat("lookahead_frames") == 7 + 6
        assert all(np.array_equal(host_out if t == 3 else d_out[t].cpu().numpy(), want[t]) for t in range(7))
        rt.set_lookahead(99)
        assert rt.stat("lookahead_max") == 8
    monkeypatch.setenv("JU_LOOKAHEAD", "1")
    with R.Runtime(blob, 0, R.DTYPE_BF16) as rt:
        assert rt.stat("lookahead_max") == 1
        rt.process_batch([rt.device_image(d_in[t].data_ptr(), w, h) for t in range(7)],
                         [rt.device_image(d_out[t].data_ptr(), 4 * w, 4 * h) for t in range(7)])
        got = d_out.cpu().numpy()
        assert all(np.array_equal(got[t], want[t]) for t in range(7)) and rt.stat("lookahead_frames") == 0


@pytest.mark.parametrize("preset,dtype", [("psp-quality", R.DTYPE_BF16), ("psp-fast", R.DTYPE_F16), ("ps2-quality", R.DTYPE_FP8)])
def test_all_host_passes_give_the_frame_by_frame_bytes(preset, dtype):
    """The AviSynth caller's frames (host memory, bottom-up: avisynth_plugin/src/main.cc:113-144) through look-ahead
    passes: every input uploaded up front, every output copied out behind its frame while the next one runs.  Plain,
    padded-stride and bottom-up (negative stride) frames, in and out independently, passes of 8, 5, 2 and 3 -- every
    frame equals ju_process on the same host frames, and so does the recurrent state."""
    cfg = M.PRESETS[preset]
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    n = 18
    frames = M.synthetic_frames(n, h, w, seed=97, kind="noise")
    with R.Runtime(blob, 0, dtype) as rt:
        want = [rt.process_image(frames[t]).copy() for t in range(n)]
        want_state = rt.read_tensor("state").copy()

    def host_views(kind, count, rows, cols):
        """`count` frame buffers of one stride kind: (array to index as [row, col, 4] logically, JuImage)."""
        views = []
        for _ in range(count):
            if kind == "plain":
                a = np.zeros((rows, cols, 4), np.uint8)
            elif kind == "padded":
                a = np.zeros((rows, cols + 24, 4), np.uint8)[:, :cols]
            else:  # bottom-up: the first logical row is the LAST row in memory
                a = np.zeros((rows, cols + 8, 4), np.uint8)[::-1, :cols]
            views.append(a)
        return views

    for in_kind, out_kind in (("plain", "plain"), ("bottom-up", "bottom-up"), ("padded", "bottom-up"), ("bottom-up", "padded")):
        ins = host_views(in_kind, n, h, w)
        outs = host_views(out_kind, 8, 4 * h, 4 * w)
        for t in range(n):
            ins[t][...] = frames[t]
        with R.Runtime(blob, 0, dtype) as rt:
            t = 0
            for k in (8, 5, 2, 3):
                rt.process_batch([R.host_image(ins[t + i]) for i in range(k)], [R.host_image(outs[i]) for i in range(k)])
                for i in range(k):
                    assert np.array_equal(outs[i], want[t + i]), (preset, in_kind, out_kind, "pass of", k, "frame", t + i)
                t += k
            assert np.array_equal(rt.read_tensor("state"), want_state)
            assert rt.stat("lookahead_frames") == n and rt.stat("lookahead_host_frames") == n and rt.stat("fallbacks") == 0
            # one graph per (pass length, binding set, orientation) whatever the caller's addresses: the second pass of a
            # length replays.  8, 5, 2, 3 again on fresh host arrays: four captures at most, then replays only
            captures = rt.stat("graph_captures")
            ins2 = host_views(in_kind, n, h, w)
            for t2 in range(n):
                ins2[t2][...] = frames[t2]
            rt.reset()
            t = 0
            for k in (8, 5, 2, 3):
                rt.process_batch([R.host_image(ins2[t + i]) for i in range(k)], [R.host_image(outs[i]) for i in range(k)])
                t += k
            assert np.array_equal(outs[2], want[n - 1])
            rt.reset()
            before = rt.stat("graph_replays")
            rt.process_batch([R.host_image(ins2[i]) for i in range(8)], [R.host_image(outs[i]) for i in range(8)])
            assert rt.stat("graph_replays") == before + 1 and rt.stat("graph_captures") <= captures + 4
            assert all(np.array_equal(outs[i], want[i]) for i in range(8))


def test_host_frames_in_a_pass_whose_tower_times_out_are_run_again_and_an_overwritten_input_splits_the_pass():
    """(1) The fallback inside an all-host pass: the frames are run again one by one through the staging path and the
    caller's buffers end up with the per-block tower's bytes.  (2) advisor, round 5: an OUTPUT that overlaps an EARLIER
    frame's input would be harmless on the normal path (written after that input was read) but not for the re-run,
    which reads the inputs again -- such a frame starts a new pass."""
    import torch
    cfg = M.PRESETS["psp-fast"]
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames = M.synthetic_frames(6, h, w, seed=59, kind="noise")
    lib = R.load_library()
    outs = [np.zeros((4 * h, 4 * w, 4), np.uint8) for _ in range(6)]
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(frames).to(dev)
    d_out = torch.zeros((4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with R.Runtime(blob, 0, R.DTYPE_BF16) as rt:   # reference: the same fault frame by frame
        # (device frames, a fresh input pointer per call: eager launches -- the test hook acts at LAUNCH time, a replayed
        # graph of host staging would never see it; host and device frames give the same bytes)
        want = []
        for t in range(6):
            if t == 2:
                lib.ju_debug_set(b"resident_fault", 1)
            try:
                rt.process(rt.device_image(d_in[t].data_ptr(), w, h), rt.device_image(d_out.data_ptr(), 4 * w, 4 * h))
                want.append(d_out.cpu().numpy())
            finally:
                lib.ju_debug_set(b"resident_fault", 0)
        assert rt.stat("fallbacks") == 1
        want_state = rt.read_tensor("state").copy()
    with R.Runtime(blob, 0, R.DTYPE_BF16) as rt:
        rt.process_batch([R.host_image(frames[t]) for t in range(2)], [R.host_image(outs[t]) for t in range(2)])
        lib.ju_debug_set(b"resident_fault", 1)
        try:
            rt.process_batch([R.host_image(frames[t]) for t in range(2, 6)], [R.host_image(outs[t]) for t in range(2, 6)])
        finally:
            lib.ju_debug_set(b"resident_fault", 0)
        assert rt.stat("fallbacks") == 1 and rt.stat("lookahead_frames") == 2
        assert all(np.array_equal(outs[t], want[t]) for t in range(6))
        assert np.array_equal(rt.read_tensor("state"), want_state)
    # (2) device frames: frame 1's output lies over frame 0's input
    arena = torch.zeros((4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
    arena.view(-1)[: h * w * 4] = d_in[0].view(-1)
    torch.cuda.synchronize()
    with R.Runtime(blob, 0, R.DTYPE_BF16) as rt:
        ins = [rt.device_image(arena.data_ptr(), w, h), rt.device_image(d_in[1].data_ptr(), w, h),
               rt.device_image(d_in[2].data_ptr(), w, h)]
        outs_d = [rt.device_image(d_out.data_ptr(), 4 * w, 4 * h), rt.device_image(arena.data_ptr(), 4 * w, 4 * h),
                  rt.device_image(d_out.data_ptr(), 4 * w, 4 * h)]
        rt.process_batch(ins, outs_d)
        # frame 0 alone (a "pass" of one frame is a plain call), then frames 1-2 as a pass
        assert rt.stat("lookahead_frames") == 2


@pytest.mark.parametrize("variant", ["generic-flow", "brightness", "flow-resnet", "temporal"])
def test_models_without_a_batched_flow_plan_run_frame_by_frame(variant, monkeypatch):
    """The pass needs the flow auto-encoder's one-launch plan with the input packing in its first block: other
    models and developer switches (per-convolution flow kernels, normalize_brightness, the flow res-net) take the
    same call frame by frame; the temporal output filter rides on the passes (its accumulators see the frames in
    order).  Same bytes in every case."""
    cfg = small_config()
    if variant == "generic-flow":
        monkeypatch.setenv("JU_FLOW_CONV", "generic")
    elif variant == "brightness":
        cfg = small_config(normalize_brightness=True)
    elif variant == "flow-resnet":
        cfg = dataclasses.replace(M.PRESETS["psp-quality-flowres"], frame_height=64, frame_width=96)
    else:
        cfg = dataclasses.replace(M.PRESETS["psp-fast"], temporal_strength=0.6, temporal_window=3)
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames, d_in, d_out = _device_clip(cfg, 6, seed=3, outs=6)
    want, want_state = _frame_by_frame(blob, R.DTYPE_F16, cfg, d_in, d_out, 6)
    with R.Runtime(blob, 0, R.DTYPE_F16) as rt:
        rt.process_batch([rt.device_image(d_in[t].data_ptr(), w, h) for t in range(6)],
                         [rt.device_image(d_out[t].data_ptr(), 4 * w, 4 * h) for t in range(6)])
        got = d_out.cpu().numpy()
        for t in range(6):
            assert np.array_equal(got[t], want[t]), (variant, t)
        assert np.array_equal(rt.read_tensor("state"), want_state)
        assert rt.stat("lookahead_frames") == (6 if variant == "temporal" else 0), variant


def test_a_pass_whose_resident_tower_times_out_is_run_again_frame_by_frame():
    """The resident tower's bounded neighbour wait can expire (CUs taken by another process).  A look-ahead pass
    never writes what it reads -- it starts from m_State[set] / m_Packed[set] and leaves its results in the other
    halves and in buffers of its own -- so the engine falls back to the per-block kernels and runs the SAME frames
    again one by one: the caller sees the frame-by-frame bytes and no error."""
    cfg = M.PRESETS["psp-fast"]
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames, d_in, d_out = _device_clip(cfg, 9, seed=41, outs=5)
    lib = R.load_library()
    with R.Runtime(blob, 0, R.DTYPE_BF16) as rt:
        ins = [rt.device_image(d_in[t].data_ptr(), w, h) for t in range(9)]
        outs = [rt.device_image(d_out[t].data_ptr(), 4 * w, 4 * h) for t in range(5)]
        rt.process_batch(ins[:4], outs[:4])
        lib.ju_debug_set(b"resident_fault", 1)
        try:
            rt.process_batch(ins[4:9], outs)          # times out inside the pass, falls back, re-runs
        finally:
            lib.ju_debug_set(b"resident_fault", 0)
        assert rt.stat("fallbacks") == 1 and rt.stat("resident_tower") == 0 and rt.stat("lookahead_frames") == 4
        got = d_out.cpu().numpy()
        state = rt.read_tensor("state").copy()
        rt.process_batch(ins[:3], outs[:3])           # look-ahead goes on over the per-block tower
        assert rt.stat("lookahead_frames") == 7
    # reference: the same fault in a frame-by-frame runtime (the per-block tower's bytes from the faulted frame on)
    with R.Runtime(blob, 0, R.DTYPE_BF16) as rt:
        out = rt.device_image(d_out[0].data_ptr(), 4 * w, 4 * h)
        for t in range(9):
            if t == 4:
                lib.ju_debug_set(b"resident_fault", 1)
            try:
                rt.process(rt.device_image(d_in[t].data_ptr(), w, h), out)
            finally:
                lib.ju_debug_set(b"resident_fault", 0)
            if t >= 4:
                assert np.array_equal(d_out[0].cpu().numpy(), got[t - 4]), t
        assert np.array_equal(rt.read_tensor("state"), state)


def test_bad_arguments_are_errors_not_crashes():
    cfg = small_config()
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    h, w = cfg.frame_height, cfg.frame_width
    frames, d_in, d_out = _device_clip(cfg, 2, seed=1)
    lib = R.load_library()
    with R.Runtime(blob, 0, R.DTYPE_F16) as rt:
        good_in = rt.device_image(d_in[0].data_ptr(), w, h)
        good_out = rt.device_image(d_out[0].data_ptr(), 4 * w, 4 * h)
        with pytest.raises(R.JoshUpscaleError):
            rt.process_batch([good_in, rt.device_image(d_in[1].data_ptr(), w - 1, h)], [good_out, good_out])
        assert lib.ju_process_batch(rt._h, None, None, 2) != 0
        assert lib.ju_process_batch(rt._h, None, None, -1) != 0
        assert lib.ju_process_batch(None, None, None, 0) != 0
        rt.reset()
        rt.process_batch([good_in, good_in], [good_out, good_out])


def test_prepare_batch_captures_in_setup_and_the_passes_only_replay():
    """ju_prepare_batch is to ju_process_batch what ju_prepare_frames is to ju_process: the graphs of a registered
    tuple (one per binding set) exist before its first pass, which therefore replays -- no eager first sighting, no
    capture inside the call -- and writes the bytes of an unregistered runtime.  Tuples that cannot run as one pass
    (a single frame, a wrong size, more frames than the cap) register nothing."""
    cfg = M.PRESETS["psp-fast"]
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames, d_in, d_out = _device_clip(cfg, 8, seed=77, outs=4)
    want, _ = _frame_by_frame(blob, R.DTYPE_BF16, cfg, d_in, d_out, 8)
    with R.Runtime(blob, 0, R.DTYPE_BF16) as rt:
        ins = [rt.device_image(d_in[t].data_ptr(), w, h) for t in range(8)]
        outs = [rt.device_image(d_out[t].data_ptr(), 4 * w, 4 * h) for t in range(4)]
        assert rt.prepare_batch(ins[:4], outs) == 2 and rt.prepare_batch(ins[4:], outs) == 2
        assert rt.prepare_batch(ins[:4], outs) == 0                         # registered already
        assert rt.prepare_batch(ins[:1], outs[:1]) == 0
        assert rt.prepare_batch(ins + ins[:1], outs + outs + outs[:1]) == 0  # nine frames: not one pass
        assert rt.prepare_batch([rt.device_image(d_in[0].data_ptr(), w - 1, h), ins[1]], outs[:2]) == 0   # a wrong size
        assert rt.stat("prepared_captures") == 4 and rt.stat("eager_runs") == 0
        for p in range(2):
            rt.process_batch(ins[4 * p:4 * p + 4], outs)
            got = d_out.cpu().numpy()
            for i in range(4):
                assert np.array_equal(got[i], want[4 * p + i]), (p, i)
        assert rt.stat("graph_replays") == 2 and rt.stat("eager_runs") == 0 and rt.stat("graph_captures") == 0


def test_an_input_that_an_earlier_frame_of_the_call_overwrites_starts_a_new_pass():
    """Frame by frame, an input buffer that overlaps an EARLIER frame's output is read after that write; a pass's flow
    sweep would read it before.  The call keeps the frame-by-frame meaning: such a frame starts a new pass.  (Input
    rows of 480 x 4 bytes inside the first rows of a 1920-pixel-wide output: a caller recycling one big arena.)"""
    import torch
    cfg = M.PRESETS["psp-fast"]
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames = M.synthetic_frames(4, h, w, seed=13, kind="noise")
    dev = torch.device("cuda", 0)

    def run(batched):
        d_in = torch.from_numpy(frames).to(dev)
        arena = torch.zeros((2, 4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
        arena[1].view(-1)[: h * w * 4] = d_in[2].view(-1)        # frame 2's pixels live inside output buffer 1 ...
        torch.cuda.synchronize()
        got = []
        with R.Runtime(blob, 0, R.DTYPE_BF16) as rt:
            ins = [rt.device_image(d_in[0].data_ptr(), w, h), rt.device_image(d_in[1].data_ptr(), w, h),
                   rt.device_image(arena[1].data_ptr(), w, h), rt.device_image(d_in[3].data_ptr(), w, h)]
            # ... which frame 1 writes: frame 2 then reads frame 1's first output rows as its pixels
            outs = [rt.device_image(arena[0].data_ptr(), 4 * w, 4 * h), rt.device_image(arena[1].data_ptr(), 4 * w, 4 * h),
                    rt.device_image(arena[0].data_ptr(), 4 * w, 4 * h), rt.device_image(arena[0].data_ptr(), 4 * w, 4 * h)]
            if batched:
                rt.process_batch(ins, outs)
                assert rt.stat("lookahead_frames") == 4          # two passes: frames 0-1 and 2-3
            else:
                for i, o in zip(ins, outs):
                    rt.process(i, o)
            got = [arena[0].cpu().numpy(), rt.read_tensor("state").copy()]
        return got
    a, b = run(False), run(True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("switch", ["JU_NO_GRAPH", "JU_DIRECT_GRAPH"])
def test_passes_without_graphs_launch_eagerly_and_give_the_same_bytes(switch, monkeypatch):
    """JU_NO_GRAPH=1 / JU_DIRECT_GRAPH=0 (developer switches): a pass is then the same launches issued one by one."""
    cfg = M.PRESETS["psp-fast"]
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames, d_in, d_out = _device_clip(cfg, 6, seed=29, outs=6)
    want, want_state = _frame_by_frame(blob, R.DTYPE_F16, cfg, d_in, d_out, 6)
    monkeypatch.setenv(switch, "1" if switch == "JU_NO_GRAPH" else "0")
    with R.Runtime(blob, 0, R.DTYPE_F16) as rt:
        ins = [rt.device_image(d_in[t].data_ptr(), w, h) for t in range(6)]
        outs = [rt.device_image(d_out[t].data_ptr(), 4 * w, 4 * h) for t in range(6)]
        assert rt.prepare_batch(ins[:3], outs[:3]) == 0           # nothing to capture
        for rnd in range(2):
            rt.reset()
            rt.process_batch(ins[:3], outs[:3])
            rt.process_batch(ins[3:], outs[3:])
            got = d_out.cpu().numpy()
            assert all(np.array_equal(got[t], want[t]) for t in range(6)), rnd
        assert np.array_equal(rt.read_tensor("state"), want_state)
        assert rt.stat("lookahead_frames") == 12 and rt.stat("graph_captures") == 0


def test_registered_tuples_keep_their_graphs_under_pressure_and_host_tuples_register_too():
    """(advisor, round 5) A tuple handed to ju_prepare_batch is exempt from the LRU of unregistered tuples: after 70 other
    tuples have come and gone (64 cached at most) its pass still replays -- no eager run, no capture inside the call.
    Host tuples register as well (their frames ride in the pass's own device buffers: one graph per pass length and
    binding set whatever the caller's addresses)."""
    import torch
    cfg = small_config()
    h, w = cfg.frame_height, cfg.frame_width
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames, d_in, _ = _device_clip(cfg, 2, seed=7)
    dev = torch.device("cuda", 0)
    outs = torch.zeros((72, 2, 4 * h, 4 * w, 4), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with R.Runtime(blob, 0, R.DTYPE_F16) as rt:
        ins = [rt.device_image(d_in[t].data_ptr(), w, h) for t in range(2)]
        tup = lambda k: [rt.device_image(outs[k][t].data_ptr(), 4 * w, 4 * h) for t in range(2)]
        assert rt.prepare_batch(ins, tup(0)) == 2
        # (round 6: registering a LONGER pass afterwards must not drop this one -- the passes' tensors are allocated for
        # the whole cap at once; bench.py lost a registered short pass to a later registration of passes of 8)
        long_ins = [ins[t % 2] for t in range(8)]
        long_outs = [rt.device_image(outs[60 + t][0].data_ptr(), 4 * w, 4 * h) for t in range(8)]
        assert rt.prepare_batch(long_ins, long_outs) == 2
        c0, e0 = rt.stat("graph_captures"), rt.stat("eager_runs")
        rt.process_batch(ins, tup(0))
        assert (rt.stat("graph_captures"), rt.stat("eager_runs")) == (c0, e0)
        want = outs[0].cpu().numpy().copy()
        for k in range(1, 71):                       # 70 unregistered tuples, each seen twice: eager, then captured
            for _ in range(2):
                rt.reset()
                rt.process_batch(ins, tup(k))
        assert np.array_equal(outs[70].cpu().numpy(), want)
        rt.reset()
        c0, e0, r0 = rt.stat("graph_captures"), rt.stat("eager_runs"), rt.stat("graph_replays")
        outs[0].zero_()
        rt.process_batch(ins, tup(0))                # the registered tuple: still a replay
        assert (rt.stat("graph_captures"), rt.stat("eager_runs"), rt.stat("graph_replays")) == (c0, e0, r0 + 1)
        assert np.array_equal(outs[0].cpu().numpy(), want)
        # host tuples
        h_out = [np.zeros((4 * h, 4 * w, 4), np.uint8) for _ in range(2)]
        assert rt.prepare_batch([R.host_image(frames[t]) for t in range(2)], [R.host_image(o) for o in h_out]) == 2
        rt.reset()
        c0, e0 = rt.stat("graph_captures"), rt.stat("eager_runs")
        copies = [frames[t].copy() for t in range(2)]           # (other addresses than the registered ones)
        rt.process_batch([R.host_image(c) for c in copies], [R.host_image(o) for o in h_out])
        assert (rt.stat("graph_captures"), rt.stat("eager_runs")) == (c0, e0)
        assert np.array_equal(np.stack(h_out), want) and rt.stat("lookahead_host_frames") == 2
