#!/bin/bash
# round-2 A/B on one box, whole path: decode kernels (screened joint, narrow tiles) x GEMM scheduling
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "0 0 2 0" "0 1 2 0" "1 1 2 0" "1 1 1 0" "1 1 1 8" "1 1 1 16"; do
  set -- $cfg
  echo "== RS_DECODE_SCREEN=$1 RS_DECODE_NARROW=$2 RS_GEMM_PERSISTENT=$3 RS_GEMM_RESERVE_CUS=$4 (rep $rep)"
  RS_DECODE_SCREEN=$1 RS_DECODE_NARROW=$2 RS_GEMM_PERSISTENT=$3 RS_GEMM_RESERVE_CUS=$4 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'), 'gemm share', r.get('share_of_step'))
"
done; done > gpurun_out/r02j_bench_decode_ab.txt 2>&1
for S in "0 0" "1 1"; do set -- $S; echo "== no-pipeline screen=$1 narrow=$2"; RS_DECODE_SCREEN=$1 RS_DECODE_NARROW=$2 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  ', d['ms_per_step'], d.get('ms_per_step_median'))
"; done >> gpurun_out/r02j_bench_decode_ab.txt 2>&1
