#!/bin/bash
# PMC passes over a command (each counter set in its own run, kernel-trace only — never combined
# with sys/hip/hsa traces).  usage: gpu_pmc.sh <tag> <cmd...>
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && cd - >/dev/null
# PASSES="2 3" restricts the run to those counter sets (default: all four)
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
  OUT=gpurun_out/pmc_${TAG}_$i
  rm -rf $OUT
  timeout 600 rocprofv3 --kernel-trace --pmc $SET -d $OUT -o pmc --output-format csv -- "$@" > gpurun_out/pmc_${TAG}_$i.log 2>&1
  echo "== pass $i: $SET (rc $?)"; tail -2 gpurun_out/pmc_${TAG}_$i.log
  F=$(find $OUT -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then cp "$F" gpurun_out/pmc_${TAG}_$i.csv; fi
  rm -rf $OUT
done
ls -la gpurun_out/ | grep pmc_${TAG}
