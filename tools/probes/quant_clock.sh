#!/bin/bash
export JU_TEST_HOOKS=1  # the inline python below uses the hooks of libJoshUpscale_test.so
# Is the uniform 3 % slowdown of the block launches behind a conv_1 that also quantises (docs/quant_in_conv1.patch, built
# as build/ab/lib_quant.so) a CLOCK effect?  Shader clock (amdgpu sysfs, this GPU's pp_dpm_sclk) and socket power sampled
# beside a long run in each mode of the same library, interleaved.
CARD=$(python3 - <<'PY'
import glob, os, torch
p = torch.cuda.get_device_properties(0)
bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
    if os.path.basename(os.path.realpath(os.path.dirname(f))) == bdf:
        print(f)
PY
)
echo "clock file: $CARD"
for i in 1 2; do for m in split fused; do
  if [ $m = split ]; then export JU_QUANT=split; else unset JU_QUANT; fi
  JU_LIBRARY=build/ab/lib_quant.so python3 bench.py --preset ps2-quality --dtype fp8 --steps 12000 --warmup 30 --no-cpu-baseline > /tmp/qc.json 2>/dev/null &
  BP=$!
  sleep 6
  S=""
  for k in $(seq 40); do S="$S $(grep '\*' $CARD | sed 's/.*: \([0-9]*\)Mhz.*/\1/')"; sleep 0.1; done
  P=$(amd-smi metric --power 2>/dev/null | grep SOCKET_POWER | head -1 | awk '{print $2}')
  wait $BP
  python3 - "$m" "$P" $S <<'PY'
import json, sys
v = sorted(int(x) for x in sys.argv[3:])
d = json.loads(open("/tmp/qc.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:6s} {d['value']:7.1f} frames/s  block launch {d['roofline']['launch_ms'] * 1e3:6.2f} us  sclk median {v[len(v) // 2]} MHz (min {v[0]}, max {v[-1]}, {len(v)} samples)  socket {sys.argv[2]} W")
PY
done; done
