#!/bin/bash
mkdir -p gpurun_out
for cfg in "RS_BUFFER_SETS=3" "RS_BUFFER_SETS=4"; do
  echo "== $cfg"
  env $cfg timeout 60 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'), 'frac', r.get('frac'), 'seq', r.get('achieved_sequential_schedule'))
"
done > gpurun_out/r02zz_final_policy_ab.txt 2>&1
cat gpurun_out/r02zz_final_policy_ab.txt
