#!/usr/bin/env python3
"""Developer tool, GPU box: in-kernel phase sums of res_block_pipe_kernel (workgroup 0), printed by a
library built with -DJU_RB_PROF:
    JU_LIBRARY=build/ab/lib_rbprof.so JU_RES_BLOCK=pipe python3 tools/rb_pipe_profile.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("JU_TEST_HOOKS", "1")  # developer tool: works through libJoshUpscale_test.so (the product library exports no hooks)
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["ps2-quality"]
rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_BF16)
print("us per block (eager, back to back): %.1f" % (rt.time_steps("tower", 12)[0] * 1e3))
