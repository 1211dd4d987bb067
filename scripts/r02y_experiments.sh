#!/bin/bash
# NOTE: variants 1300 / 1302 and RS_GEMM_RING=3/4 (gemm_tmf16_kernel, one 64-deep phase per K tile) were rejected by this
# run and removed from the tree afterwards; they exist in the history at e1dfbef (records: profiles/r02y_*).
# round-2 (second session) experiment pack 5: one 64-deep phase per K tile on the split ring (gemm_tmf16_kernel)
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r02y
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" 2>&1 | tail -12 > ${O}_pytest_gpu_gemm.log
timeout 200 python scripts/gemm_trace_lmf16.py --shape=qkv --shape=ffn_down 1210 1300 1212 1302 > ${O}_gemm_tile_timeline.txt 2>&1
timeout 200 python scripts/gemm_bench.py 1210 1300 1212 1302 > ${O}_gemm_one_phase_ab.txt 2>&1
sum() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'), 'seq', r.get('achieved_sequential_schedule'), 'share', r.get('share_of_step'))
"; }
for rep in 1 2; do
for cfg in "RS_GEMM_RING=2" "RS_GEMM_RING=3" "RS_GEMM_RING=4"; do
  echo "== $cfg (rep $rep)"
  env $cfg timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | sum
done; done > ${O}_bench_ab.txt 2>&1
cat ${O}_pytest_gpu_gemm.log; grep -v amdgpu ${O}_gemm_tile_timeline.txt | grep -E "^==|main loop|epilogue issue|stall|tile wall"; grep -v amdgpu ${O}_gemm_one_phase_ab.txt | grep -v sub_; cat ${O}_bench_ab.txt
