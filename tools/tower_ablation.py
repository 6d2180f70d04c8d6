#!/usr/bin/env python3
"""Interleaved A/B timing of the tower kernel's ablation variants in ONE process
(developer tool, needs a GPU).  Variant 0 is the product kernel; the others drop
one phase each and compute garbage by design."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M  # noqa: E402
from joshupscale_amd import runtime as R  # noqa: E402

NAMES = {0: "full", 1: "no halo exchange", 2: "no MFMA loop", 3: "neither"}
if os.environ.get("JU_TOWER") == "layers":
    NAMES = {0: "full", 1: "no MFMA loop", 2: "no epilogue", 3: "no tile staging", 4: "no weight staging"}
cfg = M.PRESETS["psp-quality"]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_BF16)
lib = R.load_library()
res = {v: [] for v in NAMES}
for rnd in range(5):
    for v in NAMES:
        lib.ju_debug_set(b"tower_variant", v)
        ms, n, fl = rt.time_steps("tower", 10)
        res[v].append(ms * 1e3)
lib.ju_debug_set(b"tower_variant", 0)
for v, name in NAMES.items():
    xs = sorted(res[v])
    print(f"variant {v} {name:18s}: median {xs[len(xs) // 2]:7.2f} us  min {xs[0]:7.2f} us per launch")
for tag in ["pack", "flow", "warp", "gen_head", "tower", "tail", ""]:
    ms, n, fl = rt.time_steps(tag, 10)
    print(f"stage {tag or 'ALL':9s}: {n:3d} launches, {ms * n * 1e3:8.1f} us per frame, {fl / 1e9:7.1f} GFLOP")
