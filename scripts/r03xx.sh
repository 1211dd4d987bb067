#!/bin/bash
# round 3, GPU call: (1) the launch line the driver uses for N > 1 (torch.distributed.run, RCCL process group) at world 1;
# (2) 100 timed steps: pipeline fill and drain amortised
TAG=r03x
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-extra-configs --api-batches 0 --no-cpu-baseline > gpurun_out/${TAG}_bench_torchrun_world1.json 2> gpurun_out/${TAG}_bench_torchrun_world1.err
tail -c 600 gpurun_out/${TAG}_bench_torchrun_world1.json | cut -c1-600; grep "^rank" gpurun_out/${TAG}_bench_torchrun_world1.err
timeout 600 python bench.py --steps 100 --warmup 5 --no-extra-configs --api-batches 0 --no-cpu-baseline > gpurun_out/${TAG}_bench_steps100.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03x_bench_steps100.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','steps','ms_per_step','ms_per_step_median')}, d['roofline']['achieved'], d['roofline']['frac'])
PY
