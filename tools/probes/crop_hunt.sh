#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# hunt: crop consistency over random mid sizes and engines
python3 - <<'PY' > /tmp/sizes.txt
import numpy as np
rng = np.random.default_rng(5)
for i in range(36):
    h = int(rng.integers(49, 300)) * 8; w = int(rng.integers(49, 400)) * 8
    dt = ["bf16", "fp16", "fp8"][i % 3]
    extra = ["", "gen_activation=lrelu", "flow_arch=resnet flow_pad_factor=0 flow_res_blocks=1", "", "gen_filters=96", ""][i % 6]
    if dt == "fp8" and "gen_filters" in extra: extra = ""
    print(h, w, dt, extra)
PY
while read line; do
  echo "== $line"; timeout 200 python3 tools/probes/crop_consistency.py $line 2>&1 | grep -E "WORST|Error|error" | tail -2
done < /tmp/sizes.txt
