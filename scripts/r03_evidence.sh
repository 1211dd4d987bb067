#!/bin/bash
# evidence pass on the current tree: all GPU tests, smoke, the bench line (with configs), kernel trace, PMC passes
#   bash scripts/r03_evidence.sh <tag> [pmc]
TAG=${1:-r03z}
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -6 > gpurun_out/${TAG}_gpu.txt; nproc >> gpurun_out/${TAG}_gpu.txt
timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/${TAG}_pytest_gpu.log
tail -4 gpurun_out/${TAG}_pytest_gpu.log
cp gpurun_out/parity_fullsize.json gpurun_out/${TAG}_parity_fullsize.json 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cut -c1-1200 gpurun_out/${TAG}_bench.json
bash scripts/gpu_profile.sh ${TAG} 3 > gpurun_out/${TAG}_profile.log 2>&1
DB=$(find gpurun_out/prof_${TAG} -name "*.db" | head -1); python scripts/rocprof_summary.py $DB 3 > gpurun_out/${TAG}_kernel_stats.txt 2>&1
head -24 gpurun_out/${TAG}_kernel_stats.txt
rm -rf gpurun_out/prof_${TAG}
if [ "$2" == "pmc" ]; then
  bash scripts/gpu_pmc.sh ${TAG} python bench.py --steps 1 --warmup 1 --no-pipeline --no-cpu-baseline --no-extra-configs --api-batches 0 --no-profile > gpurun_out/${TAG}_pmc.log 2>&1
  python scripts/pmc_summary.py gpurun_out/pmc_${TAG}_1.csv gpurun_out/pmc_${TAG}_2.csv gpurun_out/pmc_${TAG}_3.csv gpurun_out/pmc_${TAG}_4.csv > gpurun_out/${TAG}_pmc_per_kernel.txt 2>&1
  head -20 gpurun_out/${TAG}_pmc_per_kernel.txt | cut -c1-260
  rm -f gpurun_out/pmc_${TAG}_*.csv
fi
