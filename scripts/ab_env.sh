#!/bin/bash
# development: A/B an environment switch of the launch plan, e.g.  scripts/ab_env.sh GRL_NO_XCD=1
for setting in "_DUMMY=0" "$@"; do
  echo "== $setting"
  env $setting timeout 200 python bench.py --steps 600 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print(d['value'], d['ms_per_step'])
for k, v in sorted(d['roofline']['step_kernel_ms'].items(), key=lambda kv: -kv[1]): print('   %-14s %.1f' % (k, 1e3 * v))
"
done
