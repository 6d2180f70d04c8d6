#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# developer tool, GPU box: byte check (product schedule against the plain one, bf16 psp-quality frames) and
# interleaved timing of developer builds of the resident tower (tools/dev_tower_lib.sh)
# usage: tools/ab_tower_dev.sh <lib.so>...
for L in "$@"; do
JU_LIBRARY=$L python3 - <<PY
import os, sys, hashlib
import numpy as np
sys.path.insert(0, ".")
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["psp-quality"]
blob = M.serialize(cfg, M.make_seeded_weights(cfg))
frames = list(M.synthetic_frames(3, cfg.frame_height, cfg.frame_width, seed=1234, kind="noise"))
lib = R.load_library()
def run(variant):
    lib.ju_debug_set(b"tower_variant", variant)
    s = R.Session(blob, 0, R.DTYPE_BF16)
    h = hashlib.sha256()
    for f in frames: h.update(s.run(f).tobytes())
    lib.ju_debug_set(b"tower_variant", 0)
    return h.hexdigest()[:16]
a, b = run(0), run(8)
print("%-24s product %s plain %s %s" % (os.path.basename(os.environ["JU_LIBRARY"]), a, b, "EQUAL" if a == b else "DIFFERENT"))
PY
done
bash tools/ab_tower_libs.sh "$@"
