#!/usr/bin/env python3
"""What would the flow net gain from frame look-ahead?  (developer probe, needs a GPU)

The flow net reads LR frames only -- never the HR state -- so the flow fields of frames t+1 .. t+n can be
computed in ONE pass of the eight launches.  This probe prices that before anything is built: it times the flow
stage on a frame n times as tall (n x the tiles per launch, the same kernels), against n times the 480x270 stage.
usage: flow_batch_estimate.py [n ...]"""
import dataclasses
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("JU_TEST_HOOKS", "1")
from joshupscale_amd import model_file as M  # noqa: E402
from joshupscale_amd import runtime as R  # noqa: E402

ns = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 8]
base = None
for n in ns:
    cfg = dataclasses.replace(M.PRESETS["psp-quality"], frame_height=272 * n if n > 1 else 270)
    rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_BF16)
    _, k, _ = rt.time_steps("flow", 1)
    per = [min(rt.time_steps(f"flow#{i}", 20)[0] for _ in range(3)) * 1e3 for i in range(k)]
    ms, k, _ = rt.time_steps("flow", 30)
    tot = ms * k * 1e3
    if base is None:
        base = tot
    print(f"n={n}: {k} launches, flow stage {tot:7.1f} us = {tot / n:6.1f} us per frame  (wide={os.environ.get('JU_FLOW_WIDE', '-')})  "
          + " ".join(f"{p / n:5.1f}" for p in per), flush=True)
    del rt
