#!/bin/bash
# round 3, GPU call: deferred output norm (the f32 rows of a layer's output LayerNorm are not stored; the next layer's
# first residual GEMM normalises from per-row statistics)
TAG=r03q
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/bench_ab.log
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "deferred or layernorm or encoder_matches or full_size or invariant" 2>&1 | tail -15 > gpurun_out/${TAG}_pytest.log
tail -6 gpurun_out/${TAG}_pytest.log
REPS=3 bash scripts/bench_ab.sh RS_DEFER_OUT_NORM 0 1
mv gpurun_out/bench_ab.log gpurun_out/${TAG}_bench_defer_norm_ab.txt
