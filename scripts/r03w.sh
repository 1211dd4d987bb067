#!/bin/bash
# round 3, GPU call: attention prologue — loads in consumption order, K by LDS-DMA, Q fragments + first position block
# built while K / V are in flight
TAG=r03w
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider -k "attention or long_form or encoder_matches or full_size or batch_invariance" 2>&1 | tail -15 > gpurun_out/${TAG}_pytest.log
tail -6 gpurun_out/${TAG}_pytest.log
timeout 200 python scripts/attn_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/${TAG}_attn_bench.txt
timeout 300 python scripts/attn_trace.py 2>&1 | grep -v amdgpu | tee gpurun_out/${TAG}_attn_trace.txt | tail -8
