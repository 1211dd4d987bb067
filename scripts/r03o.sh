#!/bin/bash
# round 3, GPU call: whole-path bench with the early-B split ring, the three-stage ring for low tiles, alsd4_sharp, median transcribe_batch
TAG=r03o
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03o_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_median','value_host_to_ids','value_transcribe_batch')})
print(d['roofline']['achieved'],d['roofline']['frac'],d['roofline'].get('achieved_sequential_schedule'))
print(json.dumps(d['host_boundary'])[:400])
for k,v in d['configs'].items(): print(k, {a:b for a,b in v.items() if a not in ('workload','parity')})
print(d['parity'].get('flip_audit'))
PY
tail -3 gpurun_out/${TAG}_bench.err
