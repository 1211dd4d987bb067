#!/bin/bash
# round-2 A/B on one box, whole path: RS_GEMM_PERSISTENT = 0 (round-1 kernels) / 2 (whole-line kernel, one tile per
# workgroup) / 1 (whole-line kernel, persistent grid with N reserved CUs)
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "0 0" "2 0" "1 0" "1 16" "1 32"; do
  set -- $cfg
  echo "== RS_GEMM_PERSISTENT=$1 RS_GEMM_RESERVE_CUS=$2 (rep $rep)"
  RS_GEMM_PERSISTENT=$1 RS_GEMM_RESERVE_CUS=$2 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'), 'gemm share', r.get('share_of_step'))
"
done; done > gpurun_out/r02g_bench_ab.txt 2>&1
