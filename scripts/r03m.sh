#!/bin/bash
# round 3, GPU call: wave group 1 issues its B pieces one phase early (split ring, 192- / 256-row tiles)
TAG=r03m
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "gemm" 2>&1 | tail -20 > gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
timeout 600 python scripts/gemm_bench.py 0 0e 0 0e 0 0e 256 256e 192 192e > gpurun_out/${TAG}_gemm_early_b_ab.txt 2>&1
grep -v amdgpu gpurun_out/${TAG}_gemm_early_b_ab.txt
timeout 300 python scripts/gemm_trace.py > gpurun_out/${TAG}_gemm_trace.txt 2>&1
grep -E "^==|stall|main loop" gpurun_out/${TAG}_gemm_trace.txt
