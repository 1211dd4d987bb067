#!/bin/bash
# round 3, GPU call: where the B = 32 step (BASELINE configs[1]) goes — kernel trace of the pipelined and sequential schedule
TAG=r03u
mkdir -p gpurun_out
export TMPDIR=/tmp
for mode in pipe seq; do
  OUT=gpurun_out/prof_${TAG}_$mode
  extra=""; [ $mode == seq ] && extra="--no-pipeline"
  timeout 600 rocprofv3 --kernel-trace -d $OUT -o trace -- python bench.py --batch 32 --steps 30 --warmup 3 $extra --no-cpu-baseline --no-extra-configs --api-batches 0 --no-profile > gpurun_out/${TAG}_$mode.log 2>&1
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_$mode.log | head -1
  DB=$(find $OUT -name "*.db" | head -1); python scripts/rocprof_summary.py $DB 30 > gpurun_out/${TAG}_kernel_stats_b32_$mode.txt 2>&1
  head -28 gpurun_out/${TAG}_kernel_stats_b32_$mode.txt | cut -c1-150; tail -2 gpurun_out/${TAG}_kernel_stats_b32_$mode.txt
  rm -rf $OUT
done
