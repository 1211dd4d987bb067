#!/bin/bash
# round 3, first GPU call: the unified GEMM family + host pipeline on hardware
TAG=r03a
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -6 > gpurun_out/${TAG}_gpu.txt; nproc >> gpurun_out/${TAG}_gpu.txt
timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tail -120 > gpurun_out/${TAG}_pytest_gpu.log
tail -15 gpurun_out/${TAG}_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
timeout 600 python scripts/gemm_bench.py 0:0 256:0 192:0 256:2 192:2 256:1 192:1 > gpurun_out/${TAG}_gemm_sched_ab.txt 2>&1
timeout 300 python scripts/gemm_bench.py --batch=32 0:0 256:0 192:0 128:0 64:0 > gpurun_out/${TAG}_gemm_b32_tiles.txt 2>&1
timeout 300 python scripts/gemm_bench.py --batch=1 --quick 0:0 256:0 64:0 > gpurun_out/${TAG}_gemm_b1_tiles.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cat gpurun_out/${TAG}_bench.json | cut -c1-3000
REPS=1 bash scripts/bench_ab.sh RS_GEMM_SCHED 0 2 1 > gpurun_out/${TAG}_bench_sched_ab.txt 2>&1
cat gpurun_out/${TAG}_bench_sched_ab.txt
