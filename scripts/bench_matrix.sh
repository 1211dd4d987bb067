#!/bin/bash
# several bench configurations on ONE box (box-to-box variance is ~5-8 %)
mkdir -p gpurun_out
run() { echo "== $1"; shift; env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d.get('roofline', {})
print('   ms/step %.2f  RTFx %.0f  gemm %.0f TF/s (share %.2f)' % (d['ms_per_step'], d['value'], r.get('achieved', 0), r.get('share_of_step', 0)))"; }
run "default (split encoder)" A=1
run "no split" RS_SPLIT_ENCODER=0
run "default again" A=1
