#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# developer tool, GPU box: the pipelined residual-block kernel against the plain one (JU_RES_BLOCK=plain) --
# frame digests (must be equal) and us per tower (24 launches) at 640x448, bf16 and fp16
for dt in BF16 F16; do
for mode in plain pipe; do
JU_RES_BLOCK=$mode python3 - <<PY
import os, sys, hashlib
sys.path.insert(0, ".")
from joshupscale_amd import model_file as M, runtime as R
cfg = M.PRESETS["ps2-quality"]
blob = M.serialize(cfg, M.make_seeded_weights(cfg))
frames = list(M.synthetic_frames(3, cfg.frame_height, cfg.frame_width, seed=1234, kind="noise"))
rt = R.Runtime(blob, 0, R.DTYPE_$dt)
h = hashlib.sha256()
for f in frames: h.update(rt.process_image(f).tobytes())
h.update(rt.read_tensor("state").tobytes())
ms = min(rt.time_steps("tower", 10)[0] for _ in range(3))
print("%-5s %-5s digest %s  tower %.1f us" % ("$dt", os.environ["JU_RES_BLOCK"], h.hexdigest()[:16], ms * 1e3))
PY
done
done
