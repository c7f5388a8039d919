#!/bin/bash
# development: A/B environment switches of the launch plan on one box, e.g.  scripts/ab_env.sh GRL_TUNE=fused_adam=0 "GRL_TUNE=wg_split=72/12/6,graph_updates=1"
q() { python bench.py --no-learn-loop --no-success --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['step_kernel_ms']
print('%9.1f updates/s | ' % d['value'] + ' '.join('%s %.1f' % (n.replace('_fwd','F').replace('_bwd','B'), 1e3*v) for n, v in sorted(k.items(), key=lambda kv: -kv[1])))"; }
for setting in "_DUMMY=0" "$@"; do
  printf "%-34s: " "$setting"; env $setting bash -c "$(declare -f q); q"
done
