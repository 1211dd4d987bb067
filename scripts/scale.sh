#!/bin/bash
# Weak-scaling sweep of bench.py on one node: N = 1, 2, 4, 8 ranks (one process per GPU, RCCL over xGMI), 256 utterances
# per rank per step (BASELINE.json configs[2] at N = 8: 2048 = 8 x 256).  Prints, per N, the bench line's whole-job
# RTFx, ms/step (max over ranks) and the scaling efficiency against the N = 1 line.  Per-rank step times come from each
# rank's own stderr line ("rank R: ... ms/step").
#
#   bash scripts/scale.sh [STEPS] [WARMUP] [N ...]        (default 20 5 "1 2 4 8")
#
# NUMA / host placement: every rank allocates its pinned staging buffers itself after torch.cuda.set_device(LOCAL_RANK),
# so first-touch puts them on the NUMA node of the CPU that runs that rank; bind each rank near its GPU (the GPU's PCIe
# root) to keep H2D off the inter-socket link, e.g. with numactl when it is installed:
#   RS_NUMA_BIND=1 bash scripts/scale.sh      ->  numactl --cpunodebind=$((LOCAL_RANK * NODES / N)) --preferred=<same>
STEPS=${1:-20}; WARMUP=${2:-5}; shift 2 2>/dev/null
NS=${*:-1 2 4 8}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
have=$(python -c "import torch; print(torch.cuda.device_count())")
base=""
for N in $NS; do
  if [ "$N" -gt "$have" ]; then echo "N=$N: only $have GPU(s) visible — skipped"; continue; fi
  port=$((29500 + N))
  if [ "$N" -eq 1 ]; then
    line=$(python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --no-extra-configs --api-batches 0 2> gpurun_out/scale_n1.err | tail -1)
  else
    line=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $port \
             bench.py --gpus $N --steps $STEPS --warmup $WARMUP 2> gpurun_out/scale_n$N.err | grep '^{' | tail -1)
  fi
  echo "$line" > gpurun_out/scale_n$N.json
  python - "$N" "$base" <<PY
import json, sys
n, base = int(sys.argv[1]), sys.argv[2]
d = json.loads(open(f"gpurun_out/scale_n{n}.json").read())
eff = f"  efficiency {d['value'] / (n * float(base)):.3f}" if base else ""
print(f"N={n}: RTFx {d['value']:.0f}  ms/step {d['ms_per_step']:.2f} (median {d.get('ms_per_step_median')}){eff}")
PY
  grep -h "^rank " gpurun_out/scale_n$N.err 2>/dev/null
  if [ "$N" -eq 1 ]; then base=$(python -c "import json; print(json.loads(open('gpurun_out/scale_n1.json').read())['value'])"); fi
done
