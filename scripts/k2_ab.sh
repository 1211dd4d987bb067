#!/bin/bash
# sequential-schedule A/B of the Zipformer encoder's kernel forms (same box): bash scripts/k2_ab.sh <tag>
TAG=${1:?tag}; mkdir -p gpurun_out; OUT=gpurun_out/${TAG}_k2_forms_ab.txt; : > $OUT
for REP in 1 2; do
  for V in "" RS_K2_PV_OLD=1 RS_K2_CONV1_OLD=1 RS_K2_CNX_OLD=1 RS_K2_CNX_FH5=1; do
    L=$(env $V timeout 300 python scripts/k2_bench.py 6 --seq 2>/dev/null | tail -1)
    echo "rep $REP ${V:-default}: $L" | tee -a $OUT
  done
done
