#!/bin/bash
# one gpurun call of round 4: stages run in order, everything lands in gpurun_out/round4.log
#   STAGES="tests bench dp rehearsal" bash scripts/gpu_round4.sh      (any other word is run as a shell command)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
LOG=$R/gpurun_out/${LOGNAME_R4:-round4}.log
: > $LOG
STAGES=${STAGES-tests bench dp}
for st in $STAGES; do
  S0=$(date +%s)
  case $st in
    tests)
      echo "== pytest ${PYTEST_ARGS-} ${PYTEST_K:+-k $PYTEST_K}" >> $LOG
      timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q ${PYTEST_ARGS-} ${PYTEST_K:+-k "$PYTEST_K"} --timeout 600 --durations=15 -rA 2>&1 | grep -v "^PASSED\|^$" | tail -${PYTEST_TAIL:-80} >> $LOG
      ;;
    bench)
      for W in ${WORKLOADS-sac_depth}; do
        echo "== bench $W" >> $LOG
        timeout 900 python bench.py --workload $W --steps ${BENCH_STEPS:-200} --warmup 20 ${BENCH_ARGS-} > $R/gpurun_out/bench_$W.json 2> $R/gpurun_out/bench_$W.err
        echo "rc=$?" >> $LOG; tail -3 $R/gpurun_out/bench_$W.err >> $LOG; cut -c1-3500 $R/gpurun_out/bench_$W.json >> $LOG
      done
      ;;
    dp)
      echo "== data-parallel overhead at world 1 (scripts/dp_overhead.py)" >> $LOG
      timeout 600 python scripts/dp_overhead.py >> $LOG 2>&1
      ;;
    rehearsal)
      for N in ${REHEARSAL_N-2 4 8}; do
        echo "== bench.py --gpus $N --dist-backend gloo --same-device (dress rehearsal of the driver's SCALE run)" >> $LOG
        timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py \
          --gpus $N --steps 64 --warmup 16 --repeats 3 --dist-backend gloo --same-device --no-profile ${REHEARSAL_ARGS-} > $R/gpurun_out/rehearsal_$N.json 2> $R/gpurun_out/rehearsal_$N.err
        echo "rc=$?" >> $LOG; grep -v "^W0\|^\[Gloo\]\|^$" $R/gpurun_out/rehearsal_$N.err | tail -8 >> $LOG; cut -c1-1500 $R/gpurun_out/rehearsal_$N.json >> $LOG
      done
      ;;
    act)
      echo "== scripts/act_bench.py (grl_act / grl_replay_add / grl_observe latencies)" >> $LOG
      timeout 300 python scripts/act_bench.py 2>&1 | grep -v amdgpu.ids >> $LOG
      ;;
    phases)
      echo "== scripts/learn_loop_phases.py" >> $LOG
      timeout 300 python scripts/learn_loop_phases.py 2>&1 | grep -v amdgpu.ids >> $LOG
      ;;
    loop)
      echo "== scripts/profile_learn_loop.py --strict --device-norm" >> $LOG
      timeout 300 python scripts/profile_learn_loop.py --strict --device-norm 2>&1 | grep -v amdgpu.ids | head -45 >> $LOG
      echo "== scripts/profile_learn_loop.py --strict" >> $LOG
      timeout 300 python scripts/profile_learn_loop.py --strict 2>&1 | grep -v amdgpu.ids | head -12 >> $LOG
      ;;
    prof)
      echo "== rocprofv3 kernel trace (graph replay)" >> $LOG
      for W in ${PROF_WORKLOADS-sac_depth}; do
        A="${PROF_BENCH_ARGS-}"; [ "$W" != "sac_depth" ] && A="$A --workload $W"
        P=${PMC:-0}; [ "$W" != "sac_depth" ] && P=0          # counters for the headline workload only
        NAME=$W PMC=$P BENCH_ARGS="$A" bash scripts/profile_round.sh >> $LOG 2>&1
        # the same build on the same box WITHOUT the profiler (bench stage of this call): what bench.py's staleness check compares with
        [ -f $R/gpurun_out/bench_$W.json ] && python3 -c "
import json,sys
d=json.loads(open('$R/gpurun_out/bench_$W.json').read().strip().splitlines()[-1]); print('unprofiled_value: %.2f  (ms_per_step %.4f, same box, same build, no profiler)' % (d['value'], d['ms_per_step']))" >> $R/gpurun_out/prof_$W/kernel_summary.txt
      done
      ;;
    *) echo "== custom: $st" >> $LOG; bash -c "$st" >> $LOG 2>&1 ;;
  esac
  echo "-- stage $st took $(( $(date +%s) - S0 )) s" >> $LOG
done
tail -${TAIL:-260} $LOG
