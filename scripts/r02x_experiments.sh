#!/bin/bash
# round-2 (second session) experiment pack 4: wave-group epilogues side by side, split-ring GEMM kernel
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r02x
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" 2>&1 | tail -12 > ${O}_pytest_gpu_gemm.log
timeout 300 python scripts/gemm_trace_lmf16.py 1060 1200 1092 1202 > ${O}_gemm_tile_timeline.txt 2>&1
timeout 300 python scripts/gemm_bench.py 1060 1200 1210 1092 1202 1212 > ${O}_gemm_split_ring_ab.txt 2>&1
sum() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'), 'seq', r.get('achieved_sequential_schedule'), 'share', r.get('share_of_step'))
"; }
for rep in 1 2; do
for cfg in "RS_GEMM_RING=0" "RS_GEMM_RING=1" "RS_GEMM_RING=2" "RS_GEMM_RING=1 RS_DEC_STREAMS=2 RS_DECODE_PRIORITY=0"; do
  echo "== $cfg (rep $rep)"
  env $cfg timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | sum
done; done > ${O}_bench_ab.txt 2>&1
cat ${O}_pytest_gpu_gemm.log; grep -v amdgpu ${O}_gemm_tile_timeline.txt; grep -v amdgpu ${O}_gemm_split_ring_ab.txt; cat ${O}_bench_ab.txt
