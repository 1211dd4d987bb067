#!/bin/bash
TAG=r03x
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scripts/varied_lengths_latency.py 2>&1 | grep -v amdgpu | grep -v WARNING | tee gpurun_out/${TAG}_varied_lengths_latency_reuse.txt
timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|error" | tail -5 | tee gpurun_out/${TAG}_pytest_gpu_after_reuse.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
