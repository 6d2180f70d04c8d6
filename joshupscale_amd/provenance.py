"""Which source a kernel's measurements belong to.

``bench.py`` quotes the HBM traffic of its dominant kernel from a committed PMC summary (it cannot run
rocprofv3 around itself).  A summary is only valid for the kernel source it was collected on: the collector
(tools/pmc_all.sh) stores ``kernel_source_digest`` of every kernel it lists, and ``bench.py`` drops a figure
whose digest is not the current source's (``roofline.traffic`` = null, the reason in ``traffic_source``)."""

import hashlib
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")

# kernel (the short names of tools/pmc_all.sh, without template arguments) -> the .hip file that defines it
KERNEL_SOURCES = {
    "tower_resident_kernel": "tower_kernels.hip",
    "conv_tower_kernel": "tower_kernels.hip",
    "tower8_resident_kernel": "tower8_kernels.hip",
    "res_block_fp8_kernel": "fp8_kernels.hip",
    "conv_tower_fp8_kernel": "fp8_kernels.hip",
    "quantize_tower_kernel": "fp8_kernels.hip",
    "res_block_pipe_kernel": "res_block_kernels.hip",
    "res_block_kernel": "res_block_kernels.hip",
    "flow_block_kernel": "flow_kernels.hip",
    "conv_splitk_kernel": "splitk_kernels.hip",
    "conv_mfma_kernel": "conv_kernels.hip",
    "tail_fused_kernel": "frame_kernels.hip",
    "warp_pack_kernel": "frame_kernels.hip",
    "pack_frames_kernel": "frame_kernels.hip",
    "upsample2_kernel": "frame_kernels.hip",
    "maxpool2_kernel": "frame_kernels.hip",
}


def _closure(name: str, seen: set) -> None:
    """``name`` and every header of csrc/ it includes, transitively."""
    if name in seen:
        return
    path = os.path.join(CSRC, name)
    if not os.path.exists(path):
        return
    seen.add(name)
    for inc in re.findall(r'^\s*#include\s+"([^"]+)"', open(path).read(), re.M):
        _closure(os.path.basename(inc), seen)


def kernel_source_digest(kernel: str):
    """SHA-256 over the kernel's .hip file and the csrc/ headers it includes (sorted by name), or None for a
    kernel this table does not know.  ``kernel`` may carry template arguments (``conv_tower_fp8_kernel<stream>``)."""
    base = kernel.split("<")[0]
    src = KERNEL_SOURCES.get(base)
    if src is None or not os.path.exists(os.path.join(CSRC, src)):
        return None
    files: set = set()
    _closure(src, files)
    h = hashlib.sha256()
    for name in sorted(files):
        h.update(name.encode() + b"\0")
        h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()
