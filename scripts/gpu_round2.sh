#!/bin/bash
# one gpurun call: full GPU test suite + every bench workload (JSON lines under gpurun_out/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
LOG=$R/gpurun_out/round2.log
echo "== pytest" > $LOG
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS-} --timeout 900 >> $LOG 2>&1
  echo "pytest rc=$?" >> $LOG
fi
for W in ${WORKLOADS-sac_depth sac_rgbd sac_nature bdq_per ae_train}; do
  echo "== bench $W" >> $LOG
  timeout 600 python bench.py --workload $W --steps ${BENCH_STEPS:-200} --warmup 20 > $R/gpurun_out/bench_$W.json 2> $R/gpurun_out/bench_$W.err
  echo "rc=$?" >> $LOG; tail -3 $R/gpurun_out/bench_$W.err >> $LOG; cut -c1-1500 $R/gpurun_out/bench_$W.json >> $LOG
done
if [ "${GB128:-1}" = "1" ]; then
  echo "== bench sac_depth --global-batch 128 (per-rank workload of configs[4])" >> $LOG
  timeout 600 python bench.py --global-batch 128 --steps ${BENCH_STEPS:-200} --warmup 20 --no-cpu-baseline > $R/gpurun_out/bench_gb128.json 2> $R/gpurun_out/bench_gb128.err
  echo "rc=$?" >> $LOG; tail -3 $R/gpurun_out/bench_gb128.err >> $LOG; cut -c1-1500 $R/gpurun_out/bench_gb128.json >> $LOG
fi
tail -150 $LOG
