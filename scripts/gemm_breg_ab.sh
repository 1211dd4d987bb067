#!/bin/bash
# same-box A/B of the register-resident weight form of gemm_smf16 ($RS_GEMM_BREG): per shape in isolation
# (variants interleaved inside one process), then the whole-path bench, alternating, 2 repetitions
#   bash scripts/gemm_breg_ab.sh <tag>
TAG=${1:-breg}
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== per shape, B = 256 (M = 35328): LDS form (0) vs register-resident weights (0b), forced 256- and 192-row tiles too"
  timeout 600 python scripts/gemm_bench.py 0 0b 256 256b 192 192b 2>&1 | grep -v amdgpu
  for rep in 1 2; do
    for v in 0 1; do
      echo "== whole path, RS_GEMM_BREG=$v (rep $rep)"
      RS_GEMM_BREG=$v timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs --api-batches 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'), 'frac', r.get('frac'))
        for k,v in (r.get('per_shape') or {}).items(): print('      ', k, v)
"
    done
  done
} 2>&1 | tee gpurun_out/${TAG}_gemm_breg_ab.txt
