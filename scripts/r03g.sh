#!/bin/bash
# round 3, GPU call: front-end with the signal span staged in LDS, post-processing thread — tests, kernel time, bench
TAG=r03g
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_alsd.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/${TAG}_pytest_pipeline.log
tail -5 gpurun_out/${TAG}_pytest_pipeline.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
bash scripts/gpu_profile.sh ${TAG} 3 > gpurun_out/${TAG}_profile.log 2>&1
DB=$(find gpurun_out/prof_${TAG} -name "*.db" | head -1); python scripts/rocprof_summary.py $DB 3 > gpurun_out/${TAG}_kernel_stats.txt 2>&1
grep -E "logmel|feat_norm|sub_conv0|total kernel" gpurun_out/${TAG}_kernel_stats.txt
rm -rf gpurun_out/prof_${TAG}
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d.get("host_boundary"))
PY
