#!/bin/bash
# last same-box A/B of the round: with two decode lanes the decode chain has slack — does the cheaper (screened / narrow)
# decode family now win next to the encoder?
mkdir -p gpurun_out
sum() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'), 'share', r.get('share_of_step'))
"; }
for cfg in "RS_DECODE_SCREEN=0 RS_DECODE_NARROW=0" "RS_DECODE_SCREEN=1 RS_DECODE_NARROW=0" "RS_DECODE_SCREEN=1 RS_DECODE_NARROW=1" "RS_DECODE_SCREEN=0 RS_DECODE_NARROW=1" "RS_DECODE_SCREEN=0 RS_DECODE_NARROW=0 RS_BUFFER_SETS=4"; do
  echo "== $cfg"
  env $cfg timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | sum
done > gpurun_out/r02zz_decode_family_ab.txt 2>&1
cat gpurun_out/r02zz_decode_family_ab.txt
