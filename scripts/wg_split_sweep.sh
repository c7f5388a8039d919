#!/bin/bash
# Development aid: updates/s and the weight-gradient launch time for a list of GRL_TUNE=wg_split=a/b/c settings (conv1 / conv2 / conv3
# reduction splits), same box, same process order.  Usage: scripts/wg_split_sweep.sh 72/12/6 36/12/6 ...
q() { python bench.py --no-learn-loop --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['step_kernel_ms']
print('%9.1f updates/s  wgrad %.1f us  reduce %.1f us  conv2_bwd %.1f us' % (d['value'], 1e3*k.get('wgrad_conv',0), 1e3*k.get('reduce_adam',0), 1e3*k.get('conv2_bwd',0)))"; }
echo -n "default                     : "; q
for s in "$@"; do echo -n "wg_split=$s : "; GRL_TUNE=wg_split=$s q; done
