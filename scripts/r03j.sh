#!/bin/bash
# round 3, GPU call: 128-row tiles on a three-stage ring (both operands two K tiles ahead)
TAG=r03j
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "three_stage or invariant" 2>&1 | tail -20 > gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
timeout 600 python scripts/gemm_bench.py 0 128 128r 192 128 128r > gpurun_out/${TAG}_gemm_ring3_ab.txt 2>&1
grep -v amdgpu gpurun_out/${TAG}_gemm_ring3_ab.txt
timeout 300 python scripts/gemm_bench.py --batch=32 0 0r 0 0r > gpurun_out/${TAG}_gemm_ring3_ab_b32.txt 2>&1
grep -v amdgpu gpurun_out/${TAG}_gemm_ring3_ab_b32.txt
