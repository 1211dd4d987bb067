#!/bin/bash
# same-box A/B of two builds of librs_asr.so (scripts/_ab/librs_asr_old.so vs _new.so, built beforehand):
#   bash scripts/lib_ab.sh <tag>      -> attention micro-bench and the whole-path bench, alternating, 2 repetitions
TAG=${1:-ab}
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
  for v in old new; do
    cp scripts/_ab/librs_asr_$v.so reazonspeech_amd/lib/librs_asr.so
    echo "== $v (rep $rep)"
    timeout 200 python scripts/attn_bench.py 2>&1 | grep -v amdgpu
    timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs --api-batches 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'))
"
  done
done 2>&1 | tee gpurun_out/${TAG}_lib_ab.txt
cp scripts/_ab/librs_asr_new.so reazonspeech_amd/lib/librs_asr.so
