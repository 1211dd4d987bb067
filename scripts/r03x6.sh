#!/bin/bash
# round 3, GPU call: greedy decode of small batches as a replayed hipGraph of 16 steps
TAG=r03x
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -q -m gpu --tb=short -p no:cacheprovider -k "decode or boundary or invariance or end_to_end or full_size or long_form" 2>&1 | grep -E "passed|failed|Error" | tail -3
for g in 0 1 0 1; do
  echo "RS_DECODE_GRAPH=$g"
  RS_DECODE_GRAPH=$g timeout 300 python scripts/b1_profile.py 2>&1 | grep -E "latency|encoder"
done 2>&1 | tee gpurun_out/${TAG}_decode_graph_b1_ab.txt
RS_DECODE_GRAPH=1 timeout 300 python scripts/varied_lengths_latency.py 2>&1 | grep round | tee gpurun_out/${TAG}_varied_lengths_latency_graph.txt
