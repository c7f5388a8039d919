#!/bin/bash
# same-box sweep of one GRL_TUNE key over bench workloads:  KEY=gather_rows VALUES="1 2 4 8" WORKLOADS="sac_depth sac_rgbd" bash scripts/tune_sweep.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for W in ${WORKLOADS:-sac_depth}; do
  for V in ${VALUES:-1}; do
    for rep in $(seq 1 ${REPS:-1}); do
      GRL_TUNE="${KEY}=${V}${EXTRA_TUNE:+,$EXTRA_TUNE}" python bench.py --workload $W --steps ${STEPS:-200} --warmup 20 --repeats 3 --no-cpu-baseline --no-learn-loop --no-success ${BENCH_ARGS-} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['step_kernel_ms'] if d.get('roofline') else {}
print('%-14s %s=%-4s value %9.1f  ms %.4f  | eager: %s' % ('$W', '$KEY', '$V', d['value'], d['ms_per_step'], ' '.join('%s %.1f' % (a, 1e3*b) for a,b in sorted(k.items(), key=lambda kv:-kv[1])[:12])))"
    done
  done
done
