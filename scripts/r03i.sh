#!/bin/bash
TAG=r03i
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do for v in 0 2000 4000 6000 9000 14000; do RS_ATTN_SKEW_CYCLES=$v timeout 120 python scripts/attn_bench.py 2>/dev/null; done; done | tee gpurun_out/${TAG}_attn_skew_ab.txt
REPS=2 bash scripts/bench_ab.sh RS_ATTN_SKEW_CYCLES 0 6000 > gpurun_out/${TAG}_bench_attn_skew_ab.txt 2>&1
cat gpurun_out/${TAG}_bench_attn_skew_ab.txt
