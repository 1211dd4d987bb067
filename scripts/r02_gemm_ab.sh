#!/bin/bash
# round-2 A/B on one box: persistent GEMM kernel vs one-tile-per-workgroup, isolated shapes and the whole path
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k gemm 2>&1 | tail -5 > gpurun_out/r02b_gemm_tests.log
timeout 900 python scripts/gemm_bench.py 30 40 32 42 > gpurun_out/r02b_gemm_bench.txt 2>&1
for rep in 1 2; do
for cfg in "0 0" "1 0" "1 8" "1 16" "1 32"; do
  set -- $cfg
  echo "== RS_GEMM_PERSISTENT=$1 RS_GEMM_RESERVE_CUS=$2 (rep $rep)"
  RS_GEMM_PERSISTENT=$1 RS_GEMM_RESERVE_CUS=$2 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'), 'gemm share', r.get('share_of_step'))
"
done; done > gpurun_out/r02b_bench_ab.txt 2>&1
