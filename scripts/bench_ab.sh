#!/bin/bash
# same-box A/B of an environment knob on the full bench:  bash scripts/bench_ab.sh VAR val1 val2 ... [-- extra bench args]
mkdir -p gpurun_out
var=$1; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
for rep in $(seq 1 ${REPS:-2}); do
for v in "${vals[@]}"; do
  echo "== $var=$v (rep $rep) $*"
  env $var=$v timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs --api-batches 0 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'), 'gemm share', r.get('share_of_step'))
"
done; done 2>&1 | tee -a gpurun_out/bench_ab.log
