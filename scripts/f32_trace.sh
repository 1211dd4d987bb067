#!/bin/bash
# rocprofv3 kernel traces of the float32 modes (run via gpurun): bash scripts/f32_trace.sh <tag> [families...]
TAG=${1:-r06}
shift
FAMS=${@:-nemo espnet k2 avsr}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && cd - >/dev/null
for f in $FAMS; do
  OUT=gpurun_out/prof_${TAG}_$f
  rm -rf $OUT
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python scripts/family_f32_run.py $f > gpurun_out/${TAG}_f32_$f.log 2>&1
  tail -2 gpurun_out/${TAG}_f32_$f.log
  DB=$(find $OUT -name "*.db" | head -1); python scripts/rocprof_summary.py $DB 2 > gpurun_out/${TAG}_f32_${f}_kernel_stats.txt 2>&1
  head -22 gpurun_out/${TAG}_f32_${f}_kernel_stats.txt | cut -c1-200
  rm -rf $OUT
done
