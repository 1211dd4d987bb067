#!/bin/bash
# The experiment VERDICT r3 #7 asks for: L2-miss (fabric) read bytes per GEMM launch as a function of the tile walk inside
# an XCD.  RS_GEMM_GROUP_M = g_m row panels per XCD tile group: the ~32 tiles an XCD runs at a time are g_m A panels x
# 32/g_m weight tiles; g_m = 2 (1) is the "N-major" walk — one or two A row panels against ALL column tiles — g_m = 32 is
# "M-major" (one weight tile against 32 row panels).  One rocprofv3 --pmc FETCH_SIZE pass per (shape, g_m); time from the
# same dispatches.   bash scripts/gemm_traffic_sweep.sh <tag>     -> gpurun_out/<tag>_gemm_traffic_sweep.txt
TAG=${1:?tag}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && cd - >/dev/null
OUTF=gpurun_out/${TAG}_gemm_traffic_sweep.txt
: > $OUTF
for SHAPE in ffn_up qkv ffn_down out/pw2 glu; do
  for GM in 1 2 4 8 16 32; do
    D=gpurun_out/pmc_sweep; rm -rf $D
    timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D -o pmc --output-format csv -- \
        python scripts/gemm_bench.py 0 --quick --shape=$SHAPE --group-m=$GM > gpurun_out/pmc_sweep.log 2>&1
    F=$(find $D -name "*counter_collection.csv" | head -1)
    python - "$F" "$SHAPE" "$GM" >> $OUTF <<'PY'
import csv, sys
path, shape, gm = sys.argv[1:4]
rows = [r for r in csv.DictReader(open(path)) if "gemm_smf16" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"] if path else []
if rows:
    rd = [2.0 * float(r["Counter_Value"]) * 1024 / 1e6 for r in rows]
    us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    print(f"{shape:12s} group_m {int(gm):2d}: {len(rows)} launches, fabric reads {sum(rd) / len(rd):8.1f} MB per launch (FETCH_SIZE x 2), {sum(us) / len(us):7.1f} us under the profiler")
else:
    print(f"{shape:12s} group_m {gm}: no counter rows")
PY
    rm -rf $D
  done
done
cat $OUTF
