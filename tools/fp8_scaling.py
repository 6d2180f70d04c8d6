#!/usr/bin/env python3
"""Launch time of the two 8-bit block convolutions against the number of tiles
(developer tool, needs a GPU): the fixed per-launch cost is what is left at one tile row."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M  # noqa: E402
from joshupscale_amd import runtime as R  # noqa: E402

for h in (8, 64, 104, 200, 208, 408, 416, 448, 616, 624):
    cfg = M.ModelConfig(frame_height=h, frame_width=640, gen_blocks=1)
    rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_FP8)
    t1 = min(rt.time_steps("tower#1", 30)[0] for _ in range(3)) * 1e3
    t2 = min(rt.time_steps("tower#2", 30)[0] for _ in range(3)) * 1e3
    tiles = 20 * ((h + 7) // 8)
    print(f"H {h:4d}: {tiles:5d} tiles ({tiles / 512:.2f} per workgroup slot)  first conv {t1:6.2f} us  second {t2:6.2f} us")
    rt.close()
