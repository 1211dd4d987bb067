#!/bin/bash
# round-2 (second session) experiment pack 3: do power-of-two operand pitches camp on a few L2 / HBM channels?
# (row pitch of A / W / out padded by 64 elements), and two decode streams next to the encoder
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r02w
{
for pads in "" "--pad-a=64" "--pad-w=64" "--pad-a=64 --pad-w=64" "--pad-a=64 --pad-w=64 --pad-c=64" "--pad-a=32 --pad-w=32"; do
  timeout 200 python scripts/gemm_bench.py 1092 --shape=ffn_down --shape=out/pw2 $pads 2>/dev/null
  timeout 200 python scripts/gemm_bench.py 1060 --shape=ffn_up --shape=qkv --shape=pw1 $pads 2>/dev/null
done
} > ${O}_gemm_pitch_ab.txt 2>&1
sum() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'), 'seq', r.get('achieved_sequential_schedule'), 'share', r.get('share_of_step'))
        print('   intervals', d.get('step_intervals_ms'))
"; }
for rep in 1 2; do
for cfg in "RS_DEC_STREAMS=1" "RS_DEC_STREAMS=2" "RS_DEC_STREAMS=2 RS_DECODE_PRIORITY=0" "RS_DEC_STREAMS=2 RS_DECODE_NARROW=1 RS_DECODE_SCREEN=0"; do
  echo "== $cfg (rep $rep)"
  env $cfg timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | sum
done; done > ${O}_bench_ab.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -k "decode or pipelined or two_decode" 2>&1 | tail -5 > ${O}_pytest_gpu_subset.log
cat ${O}_gemm_pitch_ab.txt; cat ${O}_bench_ab.txt; cat ${O}_pytest_gpu_subset.log
