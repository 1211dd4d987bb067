#!/bin/bash
TAG=r03x
mkdir -p gpurun_out
export TMPDIR=/tmp
for sec in 21 25; do
  OUT=gpurun_out/prof_${TAG}_$sec
  timeout 300 rocprofv3 --kernel-trace -d $OUT -o trace -- python scripts/b1_profile.py --seconds=$sec > gpurun_out/${TAG}_b1_${sec}s.log 2>&1
  grep -E "latency|encoder" gpurun_out/${TAG}_b1_${sec}s.log
  DB=$(find $OUT -name "*.db" | head -1); python scripts/rocprof_summary.py $DB > gpurun_out/${TAG}_kernel_stats_b1_${sec}s.txt 2>&1
  head -16 gpurun_out/${TAG}_kernel_stats_b1_${sec}s.txt | cut -c1-130
  rm -rf $OUT
done
