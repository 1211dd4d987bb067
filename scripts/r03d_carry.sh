#!/bin/bash
# round 3, GPU call 4: the LDS ring carried across the two tiles of a pair — correctness, isolated and whole-path A/B
TAG=r03d
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "gemm or frontend" 2>&1 | tail -30 > gpurun_out/${TAG}_pytest_gemm.log
tail -5 gpurun_out/${TAG}_pytest_gemm.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider -k "alone or flip or taps or frontend or invariance" 2>&1 | tail -30 > gpurun_out/${TAG}_pytest_fullsize.log
tail -5 gpurun_out/${TAG}_pytest_fullsize.log
timeout 600 python scripts/gemm_bench.py 0p1 0p2 0p1 0p2 0p0 > gpurun_out/${TAG}_gemm_carry_ab.txt 2>&1
REPS=2 bash scripts/bench_ab.sh RS_GEMM_PAIRS 2 1 > gpurun_out/${TAG}_bench_carry_ab.txt 2>&1
cat gpurun_out/${TAG}_bench_carry_ab.txt
