#!/bin/bash
# developer tool (no GPU needed): build/ab/lib_<name>.so = the current objects (test-hook flavour) with ONE kernel file
# recompiled with extra flags -- interleaved A/B runs of a kernel variant (tools/ab_libs.sh, tools/ab_bench_libs.sh)
# usage: [SRC=other_source.hip] tools/dev_kernel_lib.sh <file.hip> <name> [extra hipcc flags]        (run `make` first;
# SRC: e.g. a `git show <rev>:...` copy of the kernel file as the baseline)
set -e
cd "$(dirname "$0")/.."
f=$1; n=$2; shift 2
base=$(basename $f .hip)
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude -Ijoshupscale_amd/csrc \
  -Wno-unused-parameter -mllvm -amdgpu-mfma-vgpr-form "$@" -c ${SRC:-joshupscale_amd/csrc/$base.hip} -o build/ab/${base}_$n.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/lib_$n.so \
  $(ls build/obj/*.o | grep -v -e /$base.o -e /c_api.o -e /graphics.o -e /dev_switch.o) build/ab/${base}_$n.o -Wl,--exclude-libs,ALL -ldl
echo built build/ab/lib_$n.so
