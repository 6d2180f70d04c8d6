#!/bin/bash
# What the GPU box's host gives the CPU baseline: affinity, cgroup quota, and the C restatement's
# frame time against the OpenMP thread count (bench.py's cpu_baseline uses what this shows is best).
OUT=gpurun_out/cpu_scaling.txt
{
echo "nproc $(nproc); affinity $(python3 -c 'import os; print(len(os.sched_getaffinity(0)))'); logical $(python3 -c 'import os; print(os.cpu_count())')"
echo "OMP_NUM_THREADS=${OMP_NUM_THREADS:-unset}"
for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us /sys/fs/cgroup/cpuset.cpus.effective /sys/fs/cgroup/cpuset/cpuset.cpus; do
  [ -r $f ] && echo "$f: $(cat $f)"
done
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)|MHz" 
echo "default (oracle/c_binding.usable_cpus)"
python3 tools/probes/cpu_oracle_time.py
for t in 8 16 32 64 128; do
  for bits in 256 512; do
    echo "threads $t, ${bits}-bit blocks"
    OMP_NUM_THREADS=$t JUO_VECTOR_BITS=$bits timeout 300 python3 tools/probes/cpu_oracle_time.py
  done
done
} > $OUT 2>&1
tail -40 $OUT
