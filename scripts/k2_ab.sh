#!/bin/bash
# sequential-schedule A/B of the Zipformer encoder's kernel forms (same box): bash scripts/k2_ab.sh <tag>
# (profiles/r05x_k2_forms_ab.txt was taken with this script on a working tree that held both forms of the product and depthwise
#  kernels behind RS_K2_PV_OLD / RS_K2_CNX_OLD / RS_K2_CNX_FH5; the first forms are k2_pv_kernel / k2_cnx_dw_kernel of the tree at
#  2a83200^ and were dropped in 2a83200.  conv1's first form stays for other channel counts; round 6 added RS_K2_ATTW_SWEEPS=3 = the
#  three-sweep attention-weights kernel against the one-sweep default.)
TAG=${1:?tag}; mkdir -p gpurun_out; OUT=gpurun_out/${TAG}_k2_forms_ab.txt; : > $OUT
for REP in 1 2; do
  for V in "" RS_K2_ATTW_SWEEPS=3 RS_K2_CONV1_OLD=1; do
    L=$(env $V timeout 300 python scripts/k2_bench.py 6 --seq 2>/dev/null | tail -1)
    echo "rep $REP ${V:-default}: $L" | tee -a $OUT
  done
done
