#!/bin/bash
# developer tool (no GPU needed): build/ab/lib_<name>.so = the current objects with tower_kernels.hip
# recompiled as a developer build (-DJU_TOWER_DEV: bf16 ReLU instantiations only, ~1 min) plus extra flags
# (linked with the -DJU_TEST_HOOKS objects: the A/B scripts use ju_time_steps / ju_debug_set)
# usage: [SRC=other_tower_source.hip] tools/dev_tower_lib.sh <name> [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
n=$1; shift
SRC=${SRC:-joshupscale_amd/csrc/tower_kernels.hip}  # (e.g. a `git show <rev>:...` copy of the kernel as the baseline)
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude -Ijoshupscale_amd/csrc \
  -Wno-unused-parameter -mllvm -amdgpu-mfma-vgpr-form -DJU_TOWER_DEV "$@" -c $SRC -o build/ab/tower_$n.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/lib_$n.so \
  $(ls build/obj/*.o | grep -v -e tower_kernels.o -e /c_api.o -e /graphics.o -e /dev_switch.o) build/ab/tower_$n.o -Wl,--exclude-libs,ALL -ldl
echo built build/ab/lib_$n.so
