#!/bin/bash
# round 3, GPU call: three-stage ring for 128- and 64-row tiles adopted; tile-height data at B = 32 / 64 / 128 / 8 to refit the rule
TAG=r03k
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "gemm" 2>&1 | tail -20 > gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
for b in 32 64 128 8; do
  timeout 300 python scripts/gemm_bench.py --batch=$b 0 256 192 128 64 256 192 128 64 > gpurun_out/${TAG}_gemm_tiles_b$b.txt 2>&1
  grep -v amdgpu gpurun_out/${TAG}_gemm_tiles_b$b.txt | grep -v "sub_pw"
done
