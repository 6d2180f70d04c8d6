#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# GPU box: do two builds of the library produce the same bytes?  Frame and flow-head digests of three frames per
# preset / dtype, then interleaved stage timings.   usage: bash tools/ab_digest.sh <libA.so> <libB.so>
for L in "$1" "$2"; do
JU_LIBRARY=$PWD/$L python3 - <<PY
import os, sys, hashlib
sys.path.insert(0, ".")
from joshupscale_amd import model_file as M, runtime as R
from tests.helpers import small_config
out = []
for name, cfg in [("psp-quality", M.PRESETS["psp-quality"]), ("ps2-quality", M.PRESETS["ps2-quality"]),
                  ("small-ragged", small_config(frame_height=135, frame_width=241, gen_blocks=2)),
                  ("small-lrelu", small_config(frame_height=64, frame_width=96, gen_blocks=1, flow_activation="lrelu"))]:
    for dt in (R.DTYPE_BF16, R.DTYPE_F16):
        rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, dt)
        h = hashlib.sha256()
        for f in M.synthetic_frames(3, cfg.frame_height, cfg.frame_width, seed=3, kind="noise"):
            h.update(rt.process_image(f).tobytes()); h.update(rt.read_tensor("flow").tobytes())
        out.append(f"{name}/{dt}:{h.hexdigest()[:12]}")
        rt.close()
print(os.path.basename(os.path.dirname(os.environ["JU_LIBRARY"])), " ".join(out))
PY
done
bash tools/ab_libs.sh $1 $2 3 "flow|ALL"
