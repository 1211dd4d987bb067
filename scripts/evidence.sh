#!/bin/bash
# One parameterised evidence pass on the current tree (run on the GPU box through gpurun).  Replaces the per-experiment
# scripts of earlier rounds (scripts/r03*.sh: in the history at 9e028b2); every output goes to gpurun_out/<tag>_*.
#
#   bash scripts/evidence.sh <tag> [tests] [smoke] [bench[=STEPS]] [trace] [pmc] [gemm[=ARGS]] [espnet[=BATCH]] [beam[=ARGS]] [ab:VAR=a,b[:STEPS]]
#
#   tests        python -m pytest tests -m gpu            -> <tag>_pytest_gpu.log, <tag>_parity_fullsize.json
#   smoke        __graft_entry__.smoke()                  -> <tag>_smoke.log
#   bench=N      python bench.py --steps N --warmup 3     -> <tag>_bench.json
#   trace        rocprofv3 --kernel-trace --stats of a short bench -> <tag>_kernel_stats.txt
#   pmc          four --pmc passes (sequential schedule)  -> <tag>_pmc_per_kernel.txt (scripts/pmc_to_traffic.py turns it into profiles/gemm_traffic.json)
#   gemm=ARGS    scripts/gemm_bench.py ARGS (commas = spaces; e.g. gemm=0,--yardstick) -> <tag>_gemm.txt
#   espnet=B     scripts/espnet_bench.py --batch=B        -> <tag>_espnet_bench.txt
#   k2           scripts/k2_bench.py (pipelined bench line of configs.k2_zipformer_159m) + a kernel trace of the sequential schedule -> <tag>_k2_bench.json, <tag>_k2_kernel_stats.txt
#   espnettrace  kernel trace of scripts/espnet_bench.py -> <tag>_espnet_kernel_stats.txt
#   beam[=ARGS]  scripts/espnet_beam_bench.py ARGS ('+' = space; default --bias=16+--beam=5,20) -> <tag>_beam_bench.txt;
#                with RS_BEAM_TRACE=1 in the environment the step kernel's phase times are appended
#   ab:VAR=a,b   the bench with environment VAR set to a, then b (same box, interleaved twice) -> <tag>_ab_VAR.txt
TAG=${1:?tag}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && cd - >/dev/null
rocm-smi --showproductname 2>/dev/null | head -6 > gpurun_out/${TAG}_gpu.txt; nproc >> gpurun_out/${TAG}_gpu.txt
QUIET="--no-cpu-baseline --no-extra-configs --api-batches 0 --no-fp32-parity"
for WHAT in "$@"; do
  case "$WHAT" in
    tests)
      timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/${TAG}_pytest_gpu.log
      tail -4 gpurun_out/${TAG}_pytest_gpu.log
      cp gpurun_out/parity_fullsize.json gpurun_out/${TAG}_parity_fullsize.json 2>/dev/null ;;
    smoke)
      timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log ;;
    bench|bench=*)
      N=${WHAT#bench=}; [ "$N" == "bench" ] && N=20
      timeout 900 python bench.py --steps $N --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
      cut -c1-600 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err ;;
    trace)
      OUT=gpurun_out/prof_$TAG; rm -rf $OUT
      timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 3 --warmup 1 $QUIET --no-profile > gpurun_out/${TAG}_trace.log 2>&1
      DB=$(find $OUT -name "*.db" | head -1); python scripts/rocprof_summary.py $DB 3 > gpurun_out/${TAG}_kernel_stats.txt 2>&1
      head -24 gpurun_out/${TAG}_kernel_stats.txt; rm -rf $OUT ;;
    pmc)
      bash scripts/gpu_pmc.sh ${TAG} python bench.py --steps 1 --warmup 1 --no-pipeline $QUIET --no-profile > gpurun_out/${TAG}_pmc.log 2>&1
      python scripts/pmc_summary.py gpurun_out/pmc_${TAG}_1.csv gpurun_out/pmc_${TAG}_2.csv gpurun_out/pmc_${TAG}_3.csv gpurun_out/pmc_${TAG}_4.csv > gpurun_out/${TAG}_pmc_per_kernel.txt 2>&1
      head -20 gpurun_out/${TAG}_pmc_per_kernel.txt | cut -c1-260; rm -f gpurun_out/pmc_${TAG}_*.csv ;;
    gemm|gemm=*)
      A=${WHAT#gemm=}; [ "$A" == "gemm" ] && A=0
      timeout 600 python scripts/gemm_bench.py ${A//,/ } > gpurun_out/${TAG}_gemm.txt 2>&1; cat gpurun_out/${TAG}_gemm.txt ;;
    espnet|espnet=*)
      B=${WHAT#espnet=}; [ "$B" == "espnet" ] && B=256
      timeout 600 python scripts/espnet_bench.py --batch=$B > gpurun_out/${TAG}_espnet_bench.txt 2>&1; cat gpurun_out/${TAG}_espnet_bench.txt ;;
    k2)
      timeout 600 python scripts/k2_bench.py 8 > gpurun_out/${TAG}_k2_bench.json 2> gpurun_out/${TAG}_k2_bench.err; cut -c1-400 gpurun_out/${TAG}_k2_bench.json
      OUT=gpurun_out/prof_${TAG}_k2; rm -rf $OUT
      timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python scripts/k2_bench.py 3 --seq > gpurun_out/${TAG}_k2_trace.log 2>&1
      DB=$(find $OUT -name "*.db" | head -1); python scripts/rocprof_summary.py $DB 4 > gpurun_out/${TAG}_k2_kernel_stats.txt 2>&1
      head -16 gpurun_out/${TAG}_k2_kernel_stats.txt; rm -rf $OUT ;;
    espnettrace)
      OUT=gpurun_out/prof_${TAG}_espnet; rm -rf $OUT
      timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python scripts/espnet_bench.py --batch=256 > gpurun_out/${TAG}_espnet_trace.log 2>&1
      DB=$(find $OUT -name "*.db" | head -1); python scripts/rocprof_summary.py $DB > gpurun_out/${TAG}_espnet_kernel_stats.txt 2>&1
      head -16 gpurun_out/${TAG}_espnet_kernel_stats.txt; rm -rf $OUT ;;
    beam|beam=*)
      A=${WHAT#beam=}; [ "$A" == "beam" ] && A="--bias=16+--beam=5,20"
      timeout 600 python scripts/espnet_beam_bench.py ${A//+/ } > gpurun_out/${TAG}_beam_bench.txt 2>&1; grep -v amdgpu.ids gpurun_out/${TAG}_beam_bench.txt ;;
    ab:*)
      SPEC=${WHAT#ab:}; VAR=${SPEC%%=*}; REST=${SPEC#*=}; VALS=${REST%%:*}; N=20; [[ "$REST" == *:* ]] && N=${REST##*:}
      : > gpurun_out/${TAG}_ab_${VAR}.txt
      for REP in 1 2; do for V in ${VALS//,/ }; do
        L=$(env $VAR=$V timeout 600 python bench.py --steps $N --warmup 3 $QUIET 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('achieved'))")
        echo "$VAR=$V rep $REP: ms_per_step value gemm_TF/s = $L" | tee -a gpurun_out/${TAG}_ab_${VAR}.txt
      done; done ;;
    *) echo "unknown item $WHAT" ;;
  esac
done
