#!/bin/bash
# one gpurun call: smoke + GPU parity tests + short bench; logs under gpurun_out/
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke" > gpurun_out/check.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/check.log 2>&1
echo "smoke rc=$?" >> gpurun_out/check.log
echo "== pytest" >> gpurun_out/check.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 >> gpurun_out/check.log 2>&1
echo "pytest rc=$?" >> gpurun_out/check.log
echo "== bench" >> gpurun_out/check.log
timeout 600 python bench.py --steps ${BENCH_STEPS:-100} --warmup 10 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench rc=$?" >> gpurun_out/check.log
tail -5 gpurun_out/bench.err >> gpurun_out/check.log
cat gpurun_out/bench.json >> gpurun_out/check.log
tail -60 gpurun_out/check.log
