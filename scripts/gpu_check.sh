#!/bin/bash
# Run on the GPU box via gpurun: parity tests (all, no -x), smoke, a short bench.  Logs -> gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt
echo "== pytest gpu" 
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -80 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -15 | tee gpurun_out/smoke.log
if [ "$1" == "bench" ]; then
echo "== bench"
timeout 900 python bench.py --steps ${2:-5} --warmup 2 2>&1 | tail -5 | tee gpurun_out/bench.log
fi
