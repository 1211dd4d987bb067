#!/bin/bash
# round 3, second GPU call: two tiles per workgroup (pairs) — correctness, isolated A/B, whole-path A/B, fresh kernel trace
TAG=r03b
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k gemm 2>&1 | tail -30 > gpurun_out/${TAG}_pytest_gemm.log
tail -5 gpurun_out/${TAG}_pytest_gemm.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --tb=short -p no:cacheprovider -k "alone or flip or taps" 2>&1 | tail -30 > gpurun_out/${TAG}_pytest_fullsize.log
tail -5 gpurun_out/${TAG}_pytest_fullsize.log
timeout 600 python scripts/gemm_bench.py 0p0 0p1 0p0 0p1 > gpurun_out/${TAG}_gemm_pairs_ab.txt 2>&1
timeout 300 python scripts/gemm_bench.py --batch=32 0p0 0p1 0p0 0p1 > gpurun_out/${TAG}_gemm_pairs_ab_b32.txt 2>&1
REPS=2 bash scripts/bench_ab.sh RS_GEMM_PAIRS 1 0 > gpurun_out/${TAG}_bench_pairs_ab.txt 2>&1
cat gpurun_out/${TAG}_bench_pairs_ab.txt
bash scripts/gpu_profile.sh ${TAG} 3 > gpurun_out/${TAG}_profile.log 2>&1
DB=$(find gpurun_out/prof_${TAG} -name "*.db" | head -1); python scripts/rocprof_summary.py $DB 3 > gpurun_out/${TAG}_kernel_stats.txt 2>&1
head -40 gpurun_out/${TAG}_kernel_stats.txt
