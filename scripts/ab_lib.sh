#!/bin/bash
# development: A/B two builds of the library on one box:  scripts/ab_lib.sh path/to/other.so [bench args]
other=$1; shift
q() { python bench.py --no-learn-loop --no-cpu-baseline --repeats 3 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['step_kernel_ms']
print('%9.1f | ' % d['value'] + ' '.join('%s %.1f' % (n.replace('_fwd','F').replace('_bwd','B'), 1e3*v) for n, v in sorted(k.items(), key=lambda kv: -kv[1])))"; }
for i in 1 2; do
  echo -n "this  : "; q "$@"
  echo -n "other : "; GRL_LIBRARY=$other q "$@"
done
