#!/bin/bash
TAG=r03v
mkdir -p gpurun_out
export TMPDIR=/tmp
for dec in greedy_batch alsd; do
  timeout 300 python scripts/b1_profile.py --decoding=$dec 2>&1 | grep -v amdgpu | tee -a gpurun_out/${TAG}_b1_latency.txt
done
OUT=gpurun_out/prof_${TAG}
timeout 300 rocprofv3 --kernel-trace -d $OUT -o trace -- python scripts/b1_profile.py > gpurun_out/${TAG}_prof.log 2>&1
DB=$(find $OUT -name "*.db" | head -1); python scripts/rocprof_summary.py $DB > gpurun_out/${TAG}_kernel_stats_b1.txt 2>&1
head -30 gpurun_out/${TAG}_kernel_stats_b1.txt | cut -c1-140; tail -2 gpurun_out/${TAG}_kernel_stats_b1.txt
grep latency gpurun_out/${TAG}_prof.log
rm -rf $OUT
