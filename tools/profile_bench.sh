#!/bin/bash
# Runs on the GPU box: bench line + rocprofv3 kernel-trace stats of the same command.
# usage: bash tools/profile_bench.sh <tag> [extra bench.py arguments, e.g. --preset ps2-quality --dtype fp8]
TAG=${1:-run}
shift
EXTRA="$*"
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 python3 $R/bench.py $EXTRA > $R/gpurun_out/bench_$TAG.json 2> $R/gpurun_out/bench_$TAG.err
tail -c 2500 $R/gpurun_out/bench_$TAG.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py $EXTRA --steps 100 --warmup 10 --extra-frames 0 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
tail -1 $R/gpurun_out/prof_$TAG.log | cut -c1-300
