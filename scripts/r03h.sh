#!/bin/bash
# round 3, GPU call: DMA-staged depthwise kernel, native host staging — tests, A/B, bench
TAG=r03h
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider -k "glu or dwconv or encoder or long_list or python_boundary or invariance or evaluation" 2>&1 | tail -30 > gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --tb=short -p no:cacheprovider -k "taps or alone" 2>&1 | tail -8 > gpurun_out/${TAG}_pytest_fullsize.log
tail -3 gpurun_out/${TAG}_pytest_fullsize.log
timeout 300 python scripts/elementwise_bench.py > gpurun_out/${TAG}_elementwise_ab.txt 2>&1
cat gpurun_out/${TAG}_elementwise_ab.txt | grep -v amdgpu
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d.get("host_boundary"))
PY
