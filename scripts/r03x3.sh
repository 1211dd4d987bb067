#!/bin/bash
TAG=r03x
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_alsd.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/${TAG}_pytest_pipeline.log
tail -4 gpurun_out/${TAG}_pytest_pipeline.log
timeout 600 python scripts/host_timeline.py --batches=4 --ragged --reps=2 2>&1 | grep -v amdgpu | grep -v WARNING | tee gpurun_out/${TAG}_host_timeline_ragged_contiguous.txt
timeout 600 python scripts/ragged_order_ab.py 2>&1 | grep -v amdgpu | grep -v WARNING | tee gpurun_out/${TAG}_ragged_order_ab2.txt
