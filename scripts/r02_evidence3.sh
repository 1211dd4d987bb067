#!/bin/bash
# short evidence pass on the final tree (what fits in the GPU minutes left): all GPU tests, smoke, the headline bench line
TAG=${1:-r02zz}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/${TAG}_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cat gpurun_out/${TAG}_pytest_gpu.log; tail -1 gpurun_out/${TAG}_smoke.log; cat gpurun_out/${TAG}_bench.json
