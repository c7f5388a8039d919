#!/bin/bash
# one gpurun call of round 3: GPU test suite, A/B of the launch-plan switches on ONE box, bench line, kernel trace.
#   STAGES="tests ab bench prof" AB="GRL_NO_EXACT_TAP=1 GRL_NO_LPT_ORDER=1" bash scripts/gpu_round3.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
LOG=$R/gpurun_out/round3.log
: > $LOG
STAGES=${STAGES-tests ab bench prof}
for st in $STAGES; do
  case $st in
    tests)
      echo "== pytest ${PYTEST_ARGS-}" >> $LOG
      timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q ${PYTEST_ARGS-} ${PYTEST_K:+-k "$PYTEST_K"} --timeout 900 -rA 2>&1 | grep -v "^PASSED\|^$" | tail -${PYTEST_TAIL:-60} >> $LOG
      ;;
    ab)
      echo "== A/B (updates/s | per-launch us, eager pass)" >> $LOG
      bash scripts/ab_env.sh ${AB-GRL_NO_EXACT_TAP=1 GRL_NO_LPT_ORDER=1} >> $LOG 2>&1
      ;;
    bench)
      for W in ${WORKLOADS-sac_depth}; do
        echo "== bench $W" >> $LOG
        timeout 900 python bench.py --workload $W --steps ${BENCH_STEPS:-200} --warmup 20 ${BENCH_ARGS-} > $R/gpurun_out/bench_$W.json 2> $R/gpurun_out/bench_$W.err
        echo "rc=$?" >> $LOG; tail -3 $R/gpurun_out/bench_$W.err >> $LOG; cut -c1-3000 $R/gpurun_out/bench_$W.json >> $LOG
      done
      ;;
    prof)
      echo "== rocprofv3 kernel trace (graph replay)" >> $LOG
      NAME=${PROF_NAME:-sac_depth} PMC=${PMC:-0} BENCH_ARGS="${PROF_BENCH_ARGS-}" bash scripts/profile_round.sh >> $LOG 2>&1
      ;;
    *) echo "== custom: $st" >> $LOG; bash -c "$st" >> $LOG 2>&1 ;;
  esac
done
tail -${TAIL:-220} $LOG
