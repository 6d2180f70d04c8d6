#!/bin/bash
# Reproducer of the "stale tile" report of the per-convolution 8-bit kernel.
# Three probe builds of libJoshUpscale.so differ only in how conv_tower_fp8_kernel moves its tiles:
#   mubuf_nowait   tile loads as `buffer_load ... lds`, stores as buffer_store, NO explicit DMA wait
#   mubuf_wait     the same instructions + `s_waitcnt vmcnt(0)` in front of the two barriers
#   global_nowait  global_load_lds / global_store (the product's instructions), no explicit wait
# Build (here, no GPU needed):   tools/probes/fp8_stale_tile.sh build
# Run (GPU box):                 tools/probes/fp8_stale_tile.sh
# Each build runs the grid-independence stress test (PS2 size, grids 256 / default / 1024 / 96 must give
# the same bytes).  Expected: mubuf_nowait FAILS (hipcc emits `s_waitcnt vmcnt(23)` in front of the first
# barrier: it does not count a buffer load to LDS as something the barrier's fence must wait for, so the
# first tile is read before it has landed); the other two pass.
set -u
cd "$(dirname "$0")/../.."
VARIANTS="mubuf_nowait:-DJU_FP8_MUBUF -DJU_FP8_NOWAIT|mubuf_wait:-DJU_FP8_MUBUF|mubuf_ld:-DJU_FP8_MUBUF_LD|mubuf_st:-DJU_FP8_MUBUF_ST|global_nowait:-DJU_FP8_NOWAIT"
# round 3: bisecting the store anomaly (PROBE_SET=bisect): which store group, soffset or not, hazard / drain
if [ "${PROBE_SET:-}" = bisect ]; then
VARIANTS="st_stream:-DJU_FP8_MUBUF_ST -DJU_FP8_ST_GROUPS=1|st_e4m3:-DJU_FP8_MUBUF_ST -DJU_FP8_ST_GROUPS=2|st_nosoff:-DJU_FP8_MUBUF_ST -DJU_FP8_ST_NOSOFF|st_nop:-DJU_FP8_MUBUF_ST -DJU_FP8_ST_NOP|st_drain:-DJU_FP8_MUBUF_ST -DJU_FP8_ST_DRAIN|mubuf_st:-DJU_FP8_MUBUF_ST"
fi
if [ "${1:-}" = build ]; then
  make -s || exit 1
  mkdir -p build/probes
  IFS='|'; for v in $VARIANTS; do unset IFS
    n=${v%%:*}; f=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude -Ijoshupscale_amd/csrc \
      -Wno-unused-parameter -mllvm -amdgpu-mfma-vgpr-form $f -c joshupscale_amd/csrc/fp8_kernels.hip -o build/probes/fp8_$n.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/probes/libju_fp8_$n.so \
      $(ls build/obj/*.o | grep -v fp8_kernels.o) build/probes/fp8_$n.o -Wl,--exclude-libs,ALL -ldl || exit 1
    echo "built build/probes/libju_fp8_$n.so"
  done
  exit 0
fi
DEFAULT_RUN="mubuf_nowait mubuf_wait mubuf_ld mubuf_st global_nowait"
[ "${PROBE_SET:-}" = bisect ] && DEFAULT_RUN="mubuf_st st_stream st_e4m3 st_nosoff st_nop st_drain"
for n in ${PROBE_VARIANTS:-$DEFAULT_RUN}; do
  for rep in 1 2 3; do
    JU_LIBRARY=build/probes/libju_fp8_$n.so timeout 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu \
      -k test_fp8_tower_bytes_do_not_depend_on_the_grid 2>&1 | grep -E "passed|failed" | sed "s/^/$n run $rep: /"
  done
done
