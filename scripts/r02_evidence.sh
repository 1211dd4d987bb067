#!/bin/bash
# round-2 evidence pass on one box: GPU tests, bench lines (B = 256 and B = 32), kernel trace, PMC passes
TAG=${1:-r02n}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/${TAG}_pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --steps 20 --warmup 3 --batch 32 --no-cpu-baseline > gpurun_out/${TAG}_bench_b32.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --steps 6 --warmup 2 --decoding alsd --beam 4 --no-profile > gpurun_out/${TAG}_bench_alsd4.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1
rm -rf gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_alsd
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_alsd -o trace -- python bench.py --steps 2 --warmup 1 --decoding alsd --beam 4 --no-pipeline --no-profile > gpurun_out/prof_${TAG}_alsd.log 2>&1
F=$(find gpurun_out/prof_${TAG}_alsd -name "*results.db" | head -1)
python scripts/rocprof_summary.py $F 3 > gpurun_out/${TAG}_kernel_stats_alsd.txt
find gpurun_out/prof_${TAG}_alsd -size +20M -delete
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/prof_$TAG.log 2>&1
F=$(find gpurun_out/prof_$TAG -name "*results.db" | head -1)
python scripts/rocprof_summary.py $F 3 > gpurun_out/${TAG}_kernel_stats.txt
find gpurun_out/prof_$TAG -size +20M -delete
bash scripts/gpu_pmc.sh $TAG python bench.py --steps 1 --warmup 1 --no-pipeline --no-cpu-baseline --no-profile > gpurun_out/${TAG}_pmc.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_${TAG}_1.csv gpurun_out/pmc_${TAG}_2.csv gpurun_out/pmc_${TAG}_3.csv gpurun_out/pmc_${TAG}_4.csv > gpurun_out/${TAG}_pmc_per_kernel.txt 2>&1
rm -f gpurun_out/pmc_${TAG}_*.csv
cat gpurun_out/${TAG}_pytest_gpu.log; cat gpurun_out/${TAG}_smoke.log | tail -2; cat gpurun_out/${TAG}_bench_alsd4.json; head -14 gpurun_out/${TAG}_kernel_stats_alsd.txt; cat gpurun_out/${TAG}_bench.json; cat gpurun_out/${TAG}_bench_b32.json; head -12 gpurun_out/${TAG}_kernel_stats.txt; head -12 gpurun_out/${TAG}_pmc_per_kernel.txt
