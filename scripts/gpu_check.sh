#!/bin/bash
# one gpurun call: smoke + GPU parity tests + short bench (+ optional rocprofv3 kernel trace); logs under gpurun_out/
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
LOG=$R/gpurun_out/check.log
echo "== smoke" > $LOG
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $LOG 2>&1
echo "smoke rc=$?" >> $LOG
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest" >> $LOG
timeout 1200 python -m pytest tests -m gpu -q ${PYTEST_ARGS--x} --timeout 600 >> $LOG 2>&1
echo "pytest rc=$?" >> $LOG
fi
echo "== bench" >> $LOG
timeout 600 python bench.py --steps ${BENCH_STEPS:-200} --warmup 20 > $R/gpurun_out/bench.json 2> $R/gpurun_out/bench.err
echo "bench rc=$?" >> $LOG
tail -5 $R/gpurun_out/bench.err >> $LOG
cat $R/gpurun_out/bench.json >> $LOG
if [ "${ROCPROF:-0}" = "1" ]; then
  echo "== rocprofv3" >> $LOG
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
  echo "rocprof rc=$?" >> $LOG
  cd $R
  find gpurun_out/prof -name "*stats*" | head >> $LOG
  for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do head -30 $f >> $LOG; done
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
tail -120 $LOG
