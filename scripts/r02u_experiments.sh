#!/bin/bash
# round-2 (second session) experiment pack on one box: GPU tests, per-tile timeline of the whole-line GEMM kernel,
# A/B of the deep residual prefetch in the f32 epilogue (microbench + whole path), three resident batches
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r02u
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > ${O}_pytest_gpu.log
timeout 300 python scripts/gemm_trace_lmf16.py > ${O}_gemm_tile_timeline.txt 2>&1
timeout 300 python scripts/gemm_bench.py 1062 1092 1082 1060 > ${O}_gemm_res_prefetch_ab.txt 2>&1
sum() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'), 'seq', r.get('achieved_sequential_schedule'), 'share', r.get('share_of_step'))
        print('   intervals', d.get('step_intervals_ms'))
"; }
for rep in 1 2; do
for cfg in "RS_GEMM_RES_PREFETCH=1 RS_BUFFER_SETS=2" "RS_GEMM_RES_PREFETCH=6 RS_BUFFER_SETS=2" "RS_GEMM_RES_PREFETCH=3 RS_BUFFER_SETS=2" "RS_GEMM_RES_PREFETCH=6 RS_BUFFER_SETS=3"; do
  echo "== $cfg (rep $rep)"
  env $cfg timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | sum
done; done > ${O}_bench_ab.txt 2>&1
cat ${O}_pytest_gpu.log; cat ${O}_gemm_tile_timeline.txt; cat ${O}_gemm_res_prefetch_ab.txt; cat ${O}_bench_ab.txt
