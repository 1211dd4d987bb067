#!/bin/bash
# round 3, GPU call: kernel trace with inter-kernel gaps on the encoder stream (pipelined and sequential schedule)
TAG=r03r
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_profile.sh ${TAG} 3 > gpurun_out/${TAG}_profile.log 2>&1
DB=$(find gpurun_out/prof_${TAG} -name "*.db" | head -1); python scripts/rocprof_summary.py $DB 3 > gpurun_out/${TAG}_kernel_stats.txt 2>&1
python - <<PY
import sqlite3
con=sqlite3.connect("$DB"); cur=con.cursor()
print([r[1] for r in cur.execute("pragma table_info(kernels)")])
PY
tail -4 gpurun_out/${TAG}_kernel_stats.txt
rm -rf gpurun_out/prof_${TAG}
OUT=gpurun_out/prof_${TAG}s
timeout 600 rocprofv3 --kernel-trace -d $OUT -o trace -- python bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-extra-configs --api-batches 0 --no-profile > gpurun_out/${TAG}s_profile.log 2>&1
DB=$(find $OUT -name "*.db" | head -1); python scripts/rocprof_summary.py $DB 2 > gpurun_out/${TAG}_kernel_stats_sequential.txt 2>&1
tail -3 gpurun_out/${TAG}_kernel_stats_sequential.txt
rm -rf $OUT
