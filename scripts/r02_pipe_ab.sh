#!/bin/bash
# round-2 A/B on one box, whole path: persistent decode kernel (G workgroups) vs one launch per phase
mkdir -p gpurun_out
run() { env "$@" timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'gemm TF/s', r.get('achieved'), 'gemm share', r.get('share_of_step'))
"; }
for rep in 1 2; do
for cfg in "RS_DECODE_PERSIST_WGS=0 RS_DECODE_SCREEN=0 RS_DECODE_NARROW=0" "RS_DECODE_PERSIST_WGS=32" "RS_DECODE_PERSIST_WGS=64" "RS_DECODE_PERSIST_WGS=96" "RS_DECODE_PERSIST_WGS=128" \
           "RS_DECODE_PERSIST_WGS=64 RS_GEMM_PERSISTENT=1 RS_GEMM_RESERVE_CUS=32" "RS_DECODE_PERSIST_WGS=64 RS_GEMM_PERSISTENT=1 RS_GEMM_RESERVE_CUS=64"; do
  echo "== $cfg (rep $rep)"; run $cfg
done; done > gpurun_out/r02p_pipe_ab.txt 2>&1
for W in 0 64; do echo "== no-pipeline persist_wgs=$W"; RS_DECODE_PERSIST_WGS=$W timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  ', d['ms_per_step'], d.get('ms_per_step_median'))
"; done >> gpurun_out/r02p_pipe_ab.txt 2>&1
