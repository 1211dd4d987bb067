#!/bin/bash
# short evidence pass (what fits in the GPU minutes left): all GPU tests, the headline bench line, kernel trace
TAG=${1:-r02zz}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/${TAG}_pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rm -rf gpurun_out/prof_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/prof_$TAG.log 2>&1
F=$(find gpurun_out/prof_$TAG -name "*results.db" | head -1)
python scripts/rocprof_summary.py $F 3 > gpurun_out/${TAG}_kernel_stats.txt
find gpurun_out/prof_$TAG -size +20M -delete
cat gpurun_out/${TAG}_pytest_gpu.log; cat gpurun_out/${TAG}_bench.json; head -12 gpurun_out/${TAG}_kernel_stats.txt
