#!/bin/bash
# rocprofv3 kernel trace of the bench command (run via gpurun).  Output -> gpurun_out/prof_<tag>/
TAG=${1:-r01}
STEPS=${2:-3}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && cd - >/dev/null
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-extra-configs --api-batches 0 --no-profile > gpurun_out/prof_$TAG.log 2>&1
tail -3 gpurun_out/prof_$TAG.log
find $OUT -name "*stats*" | head
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then head -30 "$F" | cut -c1-220; cp "$F" gpurun_out/kernel_stats_$TAG.csv; fi
# keep the merged-back payload small
find $OUT -name "*kernel_trace.csv" -size +30M -delete
