#!/bin/bash
# last call of the round: the GPU suite, smoke and a short bench on the final tree
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee gpurun_out/r03_final_pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee gpurun_out/r03_final_smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra-configs --api-batches 0 --no-cpu-baseline 2>/dev/null | cut -c1-330 | tee gpurun_out/r03_final_bench_head.txt
