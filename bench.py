#!/usr/bin/env python3
"""Headline benchmark: frames/s of 480x270 -> 1920x1080 recurrent SR on MI355X.

A "step" is one frame through the drop-in boundary (`ju_process`, the C ABI form
of `Runtime::processImage`, reference core/src/tensorrt_backend.cc:270-278):
stage-in, one recurrent step (flow net, warp, generator), stage-out,
synchronise.  Frames are synthetic random BGRX and already resident in HBM when
the timed region starts (JU_LOC_DEVICE images); weights are seeded random-init of
the reference's default architecture ("PSP quality", see BASELINE.json).

`value` is ALWAYS taken through that one-frame-per-call boundary unless `--lookahead N`
or `--location host` is given explicitly (A/B work; the line then says so).  The same
line carries, each timed with the same barrier / synchronise / max-over-ranks method in a
region of its own after the contract's timed region:
  frame_by_frame_value   ju_process over >= 200 frames whatever --steps was
  lookahead_value        ju_process_batch, passes of 8 device frames (an extension of the
                         boundary for callers that can read ahead; one output buffer per frame)
  host_frames_value      ju_process on host frames (PCIe-inclusive; the reference's own timing
                         loop, scripts/inference/tensorrt/inference.py:245-251)
  host_frames_lookahead_value   ju_process_batch, passes of 8 host frames

N GPUs = N independent streams, one process per GPU, one RCCL broadcast of the
model container at start-up, no per-frame communication ("replicas only").

Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)



def pin_rank_to_cpus(local_rank: int, local_world: int) -> dict:
    """Per-rank CPU affinity, set BEFORE anything touches the GPU (torch is not imported
    yet): each rank's host thread -- it polls its stream while a frame runs -- gets its own
    slice of the cores of the NUMA node its GPU hangs off.  sysfs only (amdgpu PCI functions
    in bus order = the HIP runtime's default device order); when that does not describe
    `local_world` or more GPUs the allowed cores are simply split evenly by local rank."""
    info = {"how": "none"}
    try:
        allowed = sorted(os.sched_getaffinity(0))
        if local_world <= 1 and "RANK" not in os.environ:
            return {"how": "unpinned (single process)", "cpus": len(allowed)}
        cpus = None
        base = "/sys/bus/pci/drivers/amdgpu"
        gpus = []
        if os.path.isdir(base):
            for d in sorted(os.listdir(base)):
                node = os.path.join(base, d, "numa_node")
                cl = os.path.join(base, d, "local_cpulist")
                if d.count(":") == 2 and os.path.exists(node) and os.path.exists(cl):
                    gpus.append((d, int(open(node).read().strip() or -1), open(cl).read().strip()))
        visible = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if gpus and len(gpus) >= local_world and not visible and local_rank < len(gpus):
            bdf, node, cpulist = gpus[local_rank]
            local = set()
            for part in cpulist.split(","):
                if part:
                    lo, _, hi = part.partition("-")
                    local.update(range(int(lo), int(hi or lo) + 1))
            local = sorted(local & set(allowed))
            peers = [i for i, g in enumerate(gpus[:local_world]) if g[1] == node]
            if local and local_rank in peers:
                k, n = peers.index(local_rank), len(peers)
                cpus = local[k * len(local) // n:(k + 1) * len(local) // n] or local
                info = {"how": "sysfs numa_node of the rank's GPU", "gpu": bdf, "numa_node": node}
        if cpus is None:
            n = max(local_world, 1)
            cpus = allowed[local_rank * len(allowed) // n:(local_rank + 1) * len(allowed) // n] or allowed
            info = {"how": "allowed cores split evenly by local rank"}
        os.sched_setaffinity(0, cpus)
        info.update(cpus=len(cpus), first_cpu=cpus[0], last_cpu=cpus[-1])
    except Exception as e:  # never fail a run over placement
        info = {"how": f"unpinned ({type(e).__name__}: {e})"}
    return info


AFFINITY = pin_rank_to_cpus(int(os.environ.get("LOCAL_RANK", "0")),
                            int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from joshupscale_amd import dist as jdist  # noqa: E402
from joshupscale_amd import model_file as M  # noqa: E402
from joshupscale_amd import runtime as R  # noqa: E402

PEAK_MFMA_TFLOPS = 2500.0  # dense bf16/fp16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0  # HBM3E, same guide
PEAK_FP8_TFLOPS = 5000.0  # dense block-scaled e4m3 MFMA (same guide: twice the bf16 rate)
# ~0.13 s of frames in front of --warmup: the GPU's clocks ramp while the first frames run (measured
# on one box, `--steps 20 --warmup 5`: 1966 / 1972 frames/s with the pre-roll, 1778 / 1763 without --
# 25 frames are 12 ms).  It is NOT part of the contract's W and is therefore reported at the TOP level
# of the JSON line (`preroll`, `untimed_frames`), not hidden in `config`; `--preroll 0` gives the bare contract.
PREROLL_FRAMES = 256
# PMC summaries (tools/pmc_all.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bytes = (2 x FETCH_SIZE +
# WRITE_SIZE) x 1024, the gfx950 correction of MI355X_MICROARCH.md) per (preset, dtype), committed under profiles/;
# bench.py cannot run rocprofv3 around itself, so `roofline.traffic` quotes the committed figure of the SAME kernel
# -- only while the summary was collected on the kernel source that is being measured (joshupscale_amd/provenance.py:
# the collector stores a digest of the kernel's .hip + headers; a figure with another digest, or none, is dropped)
PMC_PROFILE = "r06_pmc_per_kernel_{preset}_{dtype}.json"


class SclkSampler:
    """The shader clock of this rank's GPU while the pre-roll runs (amdgpu sysfs `pp_dpm_sclk`, a side thread, every
    10 ms; stopped BEFORE the warm-up and the timed region).  The MFMA peak is quoted at 2400 MHz; under this load the
    chip holds about 2250 (profiles/r04_clock_under_load.txt) -- `roofline.peak` stays the guide's figure, this field
    says what the clock was.  None wherever sysfs is not readable."""

    def __init__(self, device_index: int):
        import glob
        import threading
        self.path, self.samples, self._stop = None, [], threading.Event()
        try:
            import torch
            p = torch.cuda.get_device_properties(device_index)
            bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
                if os.path.basename(os.path.realpath(os.path.dirname(f))) == bdf:
                    self.path = f
        except Exception:  # noqa: BLE001 -- an optional observation must never break the benchmark
            self.path = None
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            try:
                for line in open(self.path):
                    if line.rstrip().endswith("*"):
                        self.samples.append(int(line.split(":")[1].strip().split("M")[0]))
            except Exception:  # noqa: BLE001
                return
            self._stop.wait(0.01)

    def __enter__(self):
        if self.path:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.path:
            self._thread.join(timeout=1.0)

    def result(self):
        if not self.samples:
            return None
        v = sorted(self.samples)[len(self.samples) // 4:]       # (the first quarter: the ramp from idle)
        return {"median": v[len(v) // 2], "min": v[0], "max": v[-1], "samples": len(self.samples),
                "how": "amdgpu sysfs pp_dpm_sclk sampled every 10 ms during the untimed pre-roll only"}


def cpu_baseline(blob: bytes, cfg, frames: np.ndarray, budget_s: float) -> dict:
    """The C restatement of the reference path (oracle/ju_oracle_c.c), timed on
    this box's host cores on a bounded sample of the same workload.  Threads = the CPUs the process may
    really use (oracle/c_binding.usable_cpus: the affinity mask cut to the cgroup's CPU quota -- OpenMP's own
    default, one thread per logical CPU of the host, ran 4x slower under the boxes' 16-CPU quota,
    profiles/r04_cpu_scaling.txt)."""
    from oracle.c_binding import CSession
    sess = CSession(blob, cfg.frame_height, cfg.frame_width)
    t0 = time.perf_counter()
    sess.run(frames[0])
    first = time.perf_counter() - t0
    n = int(max(1, min(64, budget_s // max(first, 1e-3))))
    t0 = time.perf_counter()
    for i in range(n):
        sess.run(frames[(i + 1) % len(frames)])
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": sess.threads, "kind": "port",
            "sample": f"{n} frame(s) of the same {cfg.frame_width}x{cfg.frame_height} clip "
                      f"after 1 warm-up frame, C fp32 restatement with OpenMP, {sess.vector_bits}-bit blocks "
                      f"({dt / n:.2f} s/frame; {sess.threads} threads = the process's CPU quota, "
                      f"host has {os.cpu_count()} logical CPUs)"}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--preset", default="psp-quality", choices=sorted(M.PRESETS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp8"],
                    help="fp8: e4m3 block convolutions over fp16 (BASELINE.json config 5)")
    ap.add_argument("--location", default="device", choices=["device", "host"],
                    help="host = PCIe-inclusive (not the headline value)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-iters", type=int, default=20)
    ap.add_argument("--preroll", type=int, default=PREROLL_FRAMES,
                    help="untimed clock-warm frames in front of --warmup (fixed; not part of the contract's W)")
    ap.add_argument("--lookahead", type=int, default=None,
                    help="A/B only: time `value` through ju_process_batch passes of N frames (2..8) instead of the "
                         "reference's one-frame boundary; the line's config.boundary then says so.  Default: `value` is "
                         "ju_process frame by frame, and the look-ahead figure is reported beside it as lookahead_value")
    ap.add_argument("--extra-frames", type=int, default=256,
                    help="frames per secondary region (frame_by_frame_value, lookahead_value, host_frames_*); 0 = none")
    ap.add_argument("--no-prepare", action="store_true",
                    help="do not register the frame buffers (ju_prepare_frames): graphs are then captured "
                         "inside ju_process at a pair's second use, and a timed region that contains such a "
                         "capture or an eager run is refused")
    args = ap.parse_args()

    rank, local_rank, world = jdist.env_world()
    if world != args.gpus:
        raise SystemExit(
            f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 as `python -m torch.distributed.run "
            f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py "
            f"--gpus {args.gpus} ...` (one rank per GPU); a bare `python bench.py --gpus {args.gpus}` "
            "would silently measure one GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:  # under torchrun, also with a single rank
        jdist.init("nccl")  # RCCL

    cfg = M.PRESETS[args.preset]
    blob = None
    if rank == 0:
        blob = M.serialize(cfg, M.make_seeded_weights(cfg, seed=42))
    t_b = time.perf_counter()
    blob = jdist.broadcast_model(blob, device)
    broadcast_s = time.perf_counter() - t_b
    dt = {"bf16": R.DTYPE_BF16, "fp16": R.DTYPE_F16, "fp8": R.DTYPE_FP8}[args.dtype]
    # The frames are timed through the PRODUCT library (libJoshUpscale.so: no test hooks); the per-kernel times of
    # the roofline come afterwards from a second runtime of its test flavour (libJoshUpscale_test.so: the same
    # objects + ju_time_steps).  JU_LIBRARY (A/B of developer builds) replaces both.
    look = 1 if args.lookahead is None else max(1, min(args.lookahead, 8))
    rt = R.Runtime(blob, device=local_rank, dtype=dt, hooks=False)
    rt.set_lookahead(8)  # (the runtime's cap; what a pass really holds is what each call hands over)

    h, w = cfg.frame_height, cfg.frame_width
    RING = 16
    clip = M.synthetic_frames(RING, h, w, seed=1234 + rank, kind="noise")
    # ONE OUTPUT BUFFER PER FRAME of the ring, on both sides: a caller that reads ahead keeps every frame of a pass
    d_in = torch.from_numpy(clip).to(device)
    d_out = torch.empty((RING, 4 * h, 4 * w, 4), dtype=torch.uint8, device=device)
    dev_ins = [rt.device_image(d_in[i].data_ptr(), w, h) for i in range(RING)]
    dev_outs = [rt.device_image(d_out[i].data_ptr(), 4 * w, 4 * h) for i in range(RING)]
    out_host = np.empty((8, 4 * h, 4 * w, 4), np.uint8)
    host_ins = [R.host_image(clip[i]) for i in range(RING)]
    host_outs = [R.host_image(out_host[i % 8]) for i in range(RING)]
    torch.cuda.synchronize()

    def passes(count: int, per_call: int):
        """The boundary calls of a phase of `count` frames: (first ring position, frames) per call -- passes of
        `per_call` frames from the phase's start, a shorter one at its end."""
        i = 0
        while i < count:
            n = min(per_call, count - i)
            yield i % RING, n
            i += n

    def register(runtime, counts, per_call: int, host_too: bool = False) -> int:
        """Set-up: every device frame-buffer pair and every pass tuple the phases will use, so that their hipGraphs are
        captured HERE (the reference captures its graphs in the constructor, tensorrt_backend.cc:257-263) and the
        timed regions only replay."""
        got = 0
        for i in range(RING):
            got += runtime.prepare_frames(dev_ins[i], dev_outs[i])
        seen = set()
        for count in counts:
            for pos, n in passes(count, per_call):
                if n > 1 and (pos, n) not in seen:
                    seen.add((pos, n))
                    got += runtime.prepare_batch([dev_ins[(pos + k) % RING] for k in range(n)],
                                                 [dev_outs[(pos + k) % RING] for k in range(n)])
                    if host_too:  # (host frames ride in the pass's own device buffers: one graph per pass length and set)
                        got += runtime.prepare_batch(host_ins[:n], host_outs[:n])
        return got

    import ctypes as C
    refs = {id(lst): [C.byref(x) for x in lst] for lst in (dev_ins, dev_outs, host_ins, host_outs)}

    def run_phase(runtime, count: int, per_call: int, host: bool = False, lat=None) -> None:
        ins, outs = (host_ins, host_outs) if host else (dev_ins, dev_outs)
        if per_call == 1 and lat is None:
            # the frame-by-frame loop as a C caller would write it: the descriptors' addresses taken once, the entry point
            # looked up once (the Python binding's per-call work -- two byref objects, a method dispatch, the generator above
            # -- is ~1.5 us of a 264 us psp-fast frame, and none of it is the product's)
            proc, hnd, rin, rout = runtime._lib.ju_process, runtime._h, refs[id(ins)], refs[id(outs)]
            for i in range(count):
                pos = i % RING
                if proc(hnd, rin[pos], rout[pos]):
                    runtime.process(ins[pos], outs[pos])   # (raises with the runtime's message)
            return
        for pos, n in passes(count, per_call):
            t1 = time.perf_counter() if lat is not None else 0.0
            if n == 1:
                runtime.process(ins[pos], outs[pos])
            else:
                runtime.process_batch([ins[(pos + k) % RING] for k in range(n)], [outs[(pos + k) % RING] for k in range(n)])
            if lat is not None:
                lat.append((time.perf_counter() - t1) * 1e3)

    def counters() -> dict:
        return {"replays": rt.stat("graph_replays"), "eager": rt.stat("eager_runs"),
                "captures": rt.stat("graph_captures")}

    def timed(count: int, per_call: int, host: bool = False, lat=None):
        """One timed region, the contract's method: barrier + synchronise on both sides, the slowest rank's time.
        Returns (max-over-ranks seconds, this rank's seconds, what the region consisted of)."""
        jdist.barrier()
        torch.cuda.synchronize()
        c0, la0 = counters(), rt.stat("lookahead_frames")
        t0 = time.perf_counter()
        run_phase(rt, count, per_call, host, lat)
        torch.cuda.synchronize()
        own = time.perf_counter() - t0
        jdist.barrier()
        elapsed = time.perf_counter() - t0
        c1 = counters()
        region = {k: int(c1[k] - c0[k]) for k in c0}
        region["lookahead_frames"] = int(rt.stat("lookahead_frames") - la0)
        return jdist.max_over_ranks(elapsed, device), own, region

    host_main = args.location == "host"
    extra = max(0, args.extra_frames)
    phases = (args.preroll, args.warmup, args.steps)
    prepared = 0
    if not args.no_prepare:
        prepared = register(rt, phases, look, host_too=host_main)
        if extra:
            prepared += register(rt, (extra,), 8, host_too=True)

    with SclkSampler(local_rank) as sclk:
        run_phase(rt, args.preroll, look, host_main)  # clock-warm, fixed, outside the contract's warm-up
    run_phase(rt, args.warmup, look, host_main)
    # ---- the contract's timed region: EXACTLY --steps frames through the boundary -----------------------------------
    elapsed, own_elapsed, timed_region = timed(args.steps, look, host_main)
    rank_fps = jdist.gather_floats(args.steps / own_elapsed, device)

    # ---- secondary regions (same runtime, same box, same method; each after a short warm-up of its own kind) --------
    def secondary(per_call: int, host: bool):
        if not extra:
            return None
        run_phase(rt, 2 * 8 * 2, per_call, host)   # (32 frames: host passes capture their graph at a tuple's second use)
        lat = []
        t, _, region = timed(extra, per_call, host, lat)
        lat.sort()
        return {"value": world * extra / t, "frames": extra, "frames_per_call": per_call, "timed_region": region,
                "call_latency_ms": {"p50": lat[len(lat) // 2], "p99": lat[min(len(lat) - 1, int(len(lat) * 0.99))],
                                    "max": lat[-1], "calls": len(lat)}}

    fbf = secondary(1, False)
    la_dev = secondary(8, False)
    host_fbf = secondary(1, True)
    host_la = secondary(8, True)
    # the synchronous call's latency: the frame-by-frame region's own per-call timers (the boundary `value` is taken through)
    # (per-call timers run in the secondary regions only; with --location host the latency is that of host-frame calls)
    src = host_fbf if host_main else fbf
    latency = dict(src["call_latency_ms"], boundary="ju_process", frames=src["frames"]) if src is not None else None

    result = None
    # a timed region that is not pure steady state is not a measurement of it: refuse
    graphs_on = os.environ.get("JU_NO_GRAPH", "0") != "1"
    # (a resident-tower fallback -- CUs held by another process -- re-captures its graphs: that run is
    # reported as it went, with config.tower saying "per-layer launches", not refused)
    fell_back = rt.stat("fallbacks") > 0
    dirty = graphs_on and not fell_back and (timed_region["captures"] > 0 or timed_region["eager"] > 0)
    if dirty:
        print(f"bench.py: rank {rank}: the timed region contained {timed_region['captures']} graph capture(s) and "
              f"{timed_region['eager']} eager run(s) ({timed_region}); refusing to report a value. Register the "
              "frame buffers (default) or raise --warmup past two uses of every buffer pair.", file=sys.stderr)
    if max(jdist.gather_floats(float(dirty), device)) > 0:  # every rank leaves together
        rt.close()
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return 3
    if rank == 0:
        fps = world * args.steps / elapsed
        # dominant kernel: the 3x3 64->64 convolution of the residual tower,
        # timed with HIP events on the engine's own stream
        fp8 = args.dtype == "fp8"
        # product-side facts first: the measuring runtime below shares the device with `rt` (two resident towers of
        # different libraries must not run at once: `rt` is idle from here on)
        product = {"fallbacks": int(rt.stat("fallbacks")), "graph_replays": rt.stat("graph_replays"),
                   "eager_runs": rt.stat("eager_runs"), "direct_graphs": rt.stat("direct_graphs"),
                   "graph_captures": rt.stat("graph_captures"), "resident_tower": rt.stat("resident_tower"),
                   "library": os.path.basename(R.library_path(False))}
        rt.synchronize()
        rt_timed, rt = rt, R.Runtime(blob, device=local_rank, dtype=dt, hooks=True)
        # The GPU idled while that runtime was set up, and ju_time_steps runs its frames on the runtime's own staging
        # buffer: the measuring runtime first goes through exactly what the timed one had gone through by the end of its
        # timed region -- the SAME frames through ju_process -- so that the kernel is timed in the same power state and
        # on live recurrent data.  (Measured, round 5: a pre-roll of ju_time_steps frames instead -- a constant, zero
        # input frame, the state decays -- read 325 us in-frame for a kernel that takes ~350 in the timed frames: MFMAs
        # on degenerate data draw less power and the chip clocks higher, tools/probes/two_runtimes2.py.)
        rt.set_lookahead(8)
        if not args.no_prepare:
            register(rt, phases, look)
        for count in phases:
            run_phase(rt, count, look, host_main)
        # (timed inside whole frames: the kernel in the clock / cache context of the workload --
        # what `rocprofv3 --kernel-trace --stats` of this command averages; the back-to-back
        # figure of the launches alone is reported beside it)
        # (inside look-ahead passes when the timed region ran them: the towers of consecutive frames then follow one
        # another with only the warp in between -- another power state than frame-by-frame calls)
        in_passes = timed_region["lookahead_frames"] > 0
        ctx = "@pass" if in_passes else "@frame"
        ms, launches, flops = rt.time_steps("tower" + ctx, args.roofline_iters)
        ms_alone = rt.time_steps("tower", args.roofline_iters)[0]
        tower_in_pass_ms = None
        if not in_passes and la_dev is not None and la_dev["timed_region"]["lookahead_frames"] > 0:
            try:  # (beside the headline's figure: the same kernel inside look-ahead passes -- another power state)
                tower_in_pass_ms = rt.time_steps("tower@pass", max(2, args.roofline_iters // 4))[0]
            except R.JoshUpscaleError:
                tower_in_pass_ms = None
        flops_per_launch = flops / max(launches, 1)
        achieved = flops_per_launch / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = PEAK_MFMA_TFLOPS
        fp8_roofline = None
        if fp8 and launches == 1:
            # the one-launch 8-bit tower: MFMA-bound by arithmetic, priced against the block-scaled e4m3 peak
            peak = PEAK_FP8_TFLOPS
        elif fp8 and launches == 1 + cfg.gen_blocks:
            # one launch per residual block (more regions than CUs): step 0 is the quantise kernel.  The block kernel
            # is HBM-bound: 2 x 9.55 GFLOP x (H W / 129600) against the fp16 stream in and out + the e4m3 copy in and
            # out = 384 B per pixel (+ 72 KB of weights): 335 FLOP/B at 640x448, below the 625 FLOP/B ridge of
            # 5 PFLOP/s over 8 TB/s (round 4 mislabelled it MFMA-bound).  The MFMA fraction is reported beside it.
            ms, _, fl1 = rt.time_steps("tower#1" + ctx, args.roofline_iters)
            launches, flops_per_launch = cfg.gen_blocks, fl1
            achieved = fl1 / (ms * 1e-3) / 1e12
            peak = PEAK_FP8_TFLOPS
            bytes_blk = h * w * (128 + 128 + 64 + 64) + 2 * 9 * 64 * 64
            fp8_roofline = {
                "kernel": "res_block_fp8_kernel: one residual block (two 3x3 64->64 e4m3 convs) per launch; fp16 stream "
                          "and e4m3 copy in and out",
                "bound": "hbm", "achieved": bytes_blk / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": bytes_blk / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None,
                "launch_ms": ms, "launches_per_frame": cfg.gen_blocks, "bytes_per_launch": bytes_blk,
                "flop_per_byte": fl1 / bytes_blk, "ridge_flop_per_byte": PEAK_FP8_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9),
                "launch_ms_how": "HIP events around the kernel's launches inside whole frames on the engine's stream "
                                 "(ju_time_steps tag%s), mean over %d %s" % (ctx, args.roofline_iters, "passes" if in_passes else "frames"),
                "mfma": {"achieved": achieved, "peak": PEAK_FP8_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP8_TFLOPS,
                         "flops_per_launch": fl1},
            }
        elif fp8:
            # 8-bit tower: step 0 of the stage is the quantise kernel, then per block the first
            # convolution (e4m3 in, e4m3 out: MFMA-bound by arithmetic) and the second (+ the
            # fp16 residual stream in and out: memory-bound, and the one most time goes to)
            ms1, _, fl1 = rt.time_steps("tower#1" + ctx, args.roofline_iters)
            ms2, _, _ = rt.time_steps("tower#2" + ctx, args.roofline_iters)
            px = h * w
            bytes2 = px * (64 + 128 + 128 + 64) + 9 * 64 * 64   # t8 in, stream in/out, x8 out, weights
            fp8_roofline = {
                "kernel": "conv_tower_fp8_kernel<stream>: second 3x3 64->64 conv of a residual block, e4m3 operands, "
                          "fp16 skip connection in and out",
                "bound": "hbm", "achieved": bytes2 / (ms2 * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": bytes2 / (ms2 * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None,
                "launch_ms": ms2, "launches_per_frame": cfg.gen_blocks, "bytes_per_launch": bytes2,
                "first_conv": {"kernel": "conv_tower_fp8_kernel: first conv of a block (e4m3 in, e4m3 out)",
                               "bound": "mfma", "launch_ms": ms1, "achieved": fl1 / (ms1 * 1e-3) / 1e12,
                               "peak": PEAK_FP8_TFLOPS, "unit": "TFLOP/s",
                               "frac": fl1 / (ms1 * 1e-3) / 1e12 / PEAK_FP8_TFLOPS},
            }
        total_flops = rt.time_steps("", 0)[2]
        tail_inside = rt.time_steps("tail", 0)[1] == 0
        # HBM traffic of the dominant kernel from the committed PMC summary (bench.py cannot
        # run rocprofv3 around itself); only used when it describes the kernel measured here
        traffic = None
        traffic_source = None
        pmc_name = PMC_PROFILE.format(preset=args.preset, dtype=args.dtype)
        tpath = os.path.join(ROOT, "profiles", pmc_name)
        if fp8 and launches == 1:
            pmc_kernel = "tower8_resident_kernel"
        elif fp8:
            pmc_kernel = "res_block_fp8_kernel" if launches == cfg.gen_blocks else "conv_tower_fp8_kernel<stream>"
        else:
            pmc_kernel = "tower_resident_kernel" if launches == 1 else "res_block_pipe_kernel"
        if os.path.exists(tpath):
            from joshupscale_amd.provenance import kernel_source_digest
            with open(tpath) as f:
                entry = json.load(f).get(pmc_kernel, {})
            traffic = entry.get("hbm_bytes_per_launch")
            now = kernel_source_digest(pmc_kernel)
            if traffic is not None and (now is None or entry.get("source_sha256") != now):
                traffic_source = (f"profiles/{pmc_name} was collected on another version of {pmc_kernel}'s source "
                                  f"(digest {str(entry.get('source_sha256'))[:12]} there, {str(now)[:12]} now): its byte count is "
                                  "not quoted; re-run tools/pmc_all.sh")
                traffic = None
            elif traffic is not None:
                traffic_source = (f"committed profile profiles/{pmc_name}, kernel {pmc_kernel} (rocprofv3 --pmc passes of "
                                  f"this command line on this kernel source, digest {now[:12]}, collected separately; NOT "
                                  "measured by this run)")
        if fp8_roofline:
            fp8_roofline["traffic"], fp8_roofline["traffic_source"] = traffic, traffic_source
            fp8_roofline["traffic_unit"] = "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)"
            if traffic and fp8_roofline.get("launch_ms"):
                fp8_roofline["frac_of_counted_traffic"] = traffic / (fp8_roofline["launch_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS
        result = {
            "metric": "frames/sec 480x270->1920x1080 recurrent SR" if args.preset.startswith("psp")
                      else f"frames/sec {w}x{h}->{4 * w}x{4 * h} recurrent SR",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "preroll": args.preroll, "untimed_frames": args.preroll + args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            # Secondary figures: the same runtime, the same box, each in a timed region of its own right after the
            # contract's (same barrier / synchronise / max-over-ranks method, `extra_frames` frames each, one output
            # buffer per frame).  In reading order: the drop-in boundary over a longer sample than --steps; the look-ahead
            # extension of the boundary (callers that can hand over 8 consecutive frames); host frames (PCIe-inclusive:
            # the reference's own timing loop, scripts/inference/tensorrt/inference.py:245-251) both ways.
            "frame_by_frame_value": fbf and fbf["value"],
            "lookahead_value": la_dev and la_dev["value"],
            "host_frames_value": host_fbf and host_fbf["value"],
            "host_frames_lookahead_value": host_la and host_la["value"],
            "extra_frames": extra,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {
                "workload": f"{args.preset}: {w}x{h}->{4 * w}x{4 * h} recurrent loop, 1 stream per GPU, "
                            f"{args.dtype} MFMA operands, fp32 accumulate, frames resident in "
                            f"{'HBM' if args.location == 'device' else 'host memory (PCIe-inclusive)'}",
                "weights": "seeded random-init (seed 42), reference default architecture",
                "streams": world, "parallelism": f"replicas x{world}",
                "model_broadcast": dict(jdist.LAST_BROADCAST, seconds=broadcast_s),
                "per_rank_fps": {"values": rank_fps, "min": min(rank_fps), "max": max(rank_fps)},
                "affinity": AFFINITY,
                "boundary": ("ju_process_batch (synchronous; passes of %d consecutive frames: the flow fields of a pass in one "
                             "sweep of the flow net's launches, warp / tower / tail frame by frame; bytes equal "
                             "ju_process) -- NOT the reference's boundary: an explicit --lookahead run" % look) if in_passes
                            else "ju_process (synchronous processImage, one frame per call: core.h:68-69)",
                "secondary": {
                    "how": "each figure: its own timed region after the contract's (barrier + synchronise on both sides, slowest "
                           "rank), %d frames after 32 warm-up frames of its kind, a ring of %d inputs and one output buffer per "
                           "frame" % (extra, RING),
                    "frame_by_frame": fbf, "lookahead": la_dev, "host_frames": host_fbf, "host_frames_lookahead": host_la,
                    "tower_launch_ms_inside_passes": tower_in_pass_ms,
                },
                "library": {"timed": product["library"], "kernel_times": os.path.basename(R.library_path(True))},
                "timed_region": timed_region,
                "resident_fallbacks": product["fallbacks"],
                "preroll_frames": args.preroll,
                "sclk_mhz_during_preroll": sclk.result(),
                "submission": {"graph_replays": product["graph_replays"], "eager_runs": product["eager_runs"],
                               "cached_graphs": product["direct_graphs"], "prepared_captures": prepared,
                               "inline_captures": product["graph_captures"],
                               "how": "one hipGraph per (input, output, binding set) tuple of device frames -- per tuple "
                                      "of a look-ahead pass's frame buffers when the boundary is ju_process_batch --, "
                                      "captured in set-up by ju_prepare_frames / ju_prepare_batch (unregistered tuples: at "
                                      "their second use), replayed afterwards"},
                "tower": "resident (one launch)" if product["resident_tower"] else "per-layer launches",
                "latency_ms": latency,
                "gflop_per_frame": total_flops / 1e9,
                "whole_frame_tflops": total_flops * fps / world / 1e12,
            },
            "roofline": {
                "kernel": ("tower8_resident_kernel: all 3x3 64->64 e4m3 convs of the residual blocks, one launch" if fp8
                           else "tower_resident_kernel: generator conv_1 + all 3x3 64->64 convs of the residual blocks"
                                + (" + the generator tail on the LDS-resident last layer" if tail_inside else "") + ", one launch")
                          if launches == 1 else
                          ("res_block_fp8_kernel: one residual block (two 3x3 64->64 e4m3 convs) per launch" if fp8
                           else "res_block_pipe_kernel (ReLU blocks; res_block_kernel otherwise): one residual block "
                                "(two 3x3 64->64 convs) per launch"),
                "bound": "mfma", "achieved": achieved, "peak": peak,
                "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                "traffic_unit": "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                "traffic_source": traffic_source,
                "launch_ms": ms, "launches_per_frame": launches,
                "launch_ms_how": "HIP events around the kernel's launches inside whole frames on the engine's stream "
                                 "(ju_time_steps tag%s), mean over %d %s" % (ctx, args.roofline_iters, "passes" if in_passes else "frames"),
                "launch_ms_back_to_back": ms_alone,
                # rounds 1-2 priced `frac` on the back-to-back figure; both are kept side by side
                "frac_back_to_back": (flops_per_launch / (ms_alone * 1e-3) / 1e12 / peak) if ms_alone > 0 and launches == 1 else None,
                "flops_per_launch": flops_per_launch,
            },
        }
        if fp8_roofline:
            result["roofline"] = fp8_roofline
            result["config"]["fp8_scheme"] = (
                "self-defined (csrc/fp8.h): e4m3 operands with power-of-two per-channel weight / per-tensor "
                "activation scales, fp16 residual stream; the gfx950 counterpart of the reference's TensorRT INT8 "
                "engines, not a restatement of them -- parity is against the oracle's restatement of THIS scheme")
        clk = result["config"]["sclk_mhz_during_preroll"]
        if clk and clk["median"] > 0 and result["roofline"].get("bound") == "mfma":
            # `peak` is the guide's figure, quoted at 2400 MHz; the chip held `median` MHz under this load (power): the
            # same achieved rate against the peak AT THAT CLOCK, beside `frac`, never instead of it (an HBM-bound
            # kernel's fraction does not scale with the shader clock: not emitted there)
            result["roofline"]["frac_at_measured_clock"] = result["roofline"]["frac"] * 2400.0 / clk["median"]
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(blob, cfg, clip, args.cpu_seconds)
        print(json.dumps(result), flush=True)
        rt.close()
        rt = rt_timed
    jdist.barrier()
    rt.close()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
