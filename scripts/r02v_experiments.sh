#!/bin/bash
# NOTE: variants 1102 / 1112 / 1100 / 1110 and RS_GEMM_L2PF (L2 operand prefetch) were rejected by this run and removed
# from the tree afterwards; they exist in the history at e1dfbef (records: profiles/r02v_*).
# round-2 (second session) experiment pack 2: L2 prefetch of the long-K GEMM operands, GLU fused into the pw1 GEMM
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r02v
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -q -m gpu -k "gemm or glu or encoder_matches or encoder_619m" 2>&1 | tail -15 > ${O}_pytest_gpu_subset.log
timeout 300 python scripts/gemm_trace_lmf16.py --shape=ffn_down 1092 1102 1112 > ${O}_gemm_tile_timeline_l2pf.txt 2>&1
timeout 300 python scripts/gemm_bench.py 1092 1102 1112 1060 1100 1110 > ${O}_gemm_l2pf_ab.txt 2>&1
sum() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'), 'seq', r.get('achieved_sequential_schedule'), 'share', r.get('share_of_step'))
"; }
for rep in 1 2; do
for cfg in "RS_FUSE_GLU=0 RS_GEMM_L2PF=0" "RS_FUSE_GLU=1 RS_GEMM_L2PF=0" "RS_FUSE_GLU=1 RS_GEMM_L2PF=2" "RS_FUSE_GLU=1 RS_GEMM_L2PF=3" "RS_FUSE_GLU=1 RS_GEMM_L2PF=2 RS_GEMM_L2PF_MIN_K=1024"; do
  echo "== $cfg (rep $rep)"
  env $cfg timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | sum
done; done > ${O}_bench_ab.txt 2>&1
cat ${O}_pytest_gpu_subset.log; cat ${O}_gemm_tile_timeline_l2pf.txt; cat ${O}_gemm_l2pf_ab.txt; cat ${O}_bench_ab.txt
