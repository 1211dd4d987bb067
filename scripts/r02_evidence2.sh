#!/bin/bash
# round-2 final evidence pass on one box: GPU tests, PMC passes (-> traffic), bench lines (B = 256, B = 32, ALSD),
# smoke, kernel trace, and three same-box A/B lines of this session's schedule / kernel choices
TAG=${1:-r02z}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/${TAG}_pytest_gpu.log
cp gpurun_out/parity_fullsize.json gpurun_out/${TAG}_parity_fullsize.json 2>/dev/null
bash scripts/gpu_pmc.sh $TAG python bench.py --steps 1 --warmup 1 --no-pipeline --no-cpu-baseline --no-profile > gpurun_out/${TAG}_pmc.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_${TAG}_1.csv gpurun_out/pmc_${TAG}_2.csv gpurun_out/pmc_${TAG}_3.csv gpurun_out/pmc_${TAG}_4.csv > gpurun_out/${TAG}_pmc_per_kernel.txt 2>&1
rm -f gpurun_out/pmc_${TAG}_*.csv
python scripts/pmc_to_traffic.py gpurun_out/${TAG}_pmc_per_kernel.txt > /dev/null 2>&1 && cp profiles/gemm_traffic.json gpurun_out/${TAG}_gemm_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --steps 20 --warmup 3 --batch 32 --no-cpu-baseline > gpurun_out/${TAG}_bench_b32.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --steps 6 --warmup 2 --decoding alsd --beam 4 --no-profile > gpurun_out/${TAG}_bench_alsd4.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1
rm -rf gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/prof_$TAG.log 2>&1
F=$(find gpurun_out/prof_$TAG -name "*results.db" | head -1)
python scripts/rocprof_summary.py $F 3 > gpurun_out/${TAG}_kernel_stats.txt
find gpurun_out/prof_$TAG -size +20M -delete
sum() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('   ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'RTFx', d['value'], 'gemm TF/s', r.get('achieved'), 'seq', r.get('achieved_sequential_schedule'), 'share', r.get('share_of_step'))
"; }
for cfg in "RS_DEC_STREAMS=2" "RS_DEC_STREAMS=1" "RS_GEMM_RING=1" "RS_GEMM_RING=0" "RS_FUSE_GLU=0"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | sum
done > gpurun_out/${TAG}_bench_ab.txt 2>&1
cat gpurun_out/${TAG}_pytest_gpu.log; tail -2 gpurun_out/${TAG}_smoke.log; cat gpurun_out/${TAG}_bench.json; cat gpurun_out/${TAG}_bench_b32.json; cat gpurun_out/${TAG}_bench_alsd4.json; head -14 gpurun_out/${TAG}_kernel_stats.txt; head -14 gpurun_out/${TAG}_pmc_per_kernel.txt; cat gpurun_out/${TAG}_bench_ab.txt; tail -5 gpurun_out/${TAG}_bench.err
