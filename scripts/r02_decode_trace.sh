#!/bin/bash
# per-kernel durations of the decode loop (sequential schedule, rocprofv3 kernel trace)
TAG=${1:-r02}
export TMPDIR=/tmp
rm -rf gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-pipeline > gpurun_out/prof_$TAG.log 2>&1
F=$(find gpurun_out/prof_$TAG -name "*results.db" | head -1)
python scripts/rocprof_summary.py $F 2 > gpurun_out/${TAG}_kernel_stats.txt
head -14 gpurun_out/${TAG}_kernel_stats.txt
python - <<PY
import sqlite3,glob,statistics
db=glob.glob("gpurun_out/prof_$TAG/*results.db")[0]
con=sqlite3.connect(db); cur=con.cursor()
cols=[r[1] for r in cur.execute("pragma table_info(kernels)")]
nc="name" if "name" in cols else "kernel_name"
for k in ("rnnt_verify_kernel","rnnt_screen_kernel","rnnt_lstm4_kernel","rnnt_pred16_kernel","rnnt_tile_kernel","rnnt_lstm_kernel","rnnt_finalize"):
    rows=cur.execute(f"select start,end from kernels where {nc} like ? order by start", (f"%{k}%",)).fetchall()
    if not rows: continue
    d=[(e-s)/1e3 for s,e in rows]; n=len(d)//3; seg=d[:n]
    print(k, "calls/batch", n, "us @0-4", [round(x,1) for x in seg[:5]], "@100", [round(x,1) for x in seg[100:105]], "@300", [round(x,1) for x in seg[300:305]])
PY
find gpurun_out/prof_$TAG -size +20M -delete
