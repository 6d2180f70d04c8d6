#!/usr/bin/env python3
"""Compile one HIP source for gfx950 and print a table of per-kernel resources
(VGPR/AGPR/SGPR, scratch, LDS, occupancy) from hipcc's kernel-resource-usage
remarks.  Usage: tools/kernel_resources.py [source] [name filter]"""
import os
import re
import subprocess
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "joshupscale_amd/csrc/tower_kernels.hip")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
       f"-I{root}/include", f"-I{root}/joshupscale_amd/csrc", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage", "-mllvm", "-amdgpu-mfma-vgpr-form"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True,
                          text=True).stdout.strip()
    name = re.sub(r"ju::\(anonymous namespace\)::", "", name).replace("(ju::ConvParams)", "")
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} vgpr {r.get('VGPRs', '?'):>4s} agpr {r.get('AGPRs', '?'):>4s} "
          f"scratch {r.get('ScratchSize', '?'):>5s} lds {r.get('LDS Size', '?'):>7s} "
          f"occ {r.get('Occupancy', '?')}")
