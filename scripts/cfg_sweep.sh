#!/bin/bash
# development aid: per-launch workgroup-shape sweep (GRL_TUNE=i2cfg_<tag>=c) using the graph-replay throughput
run() { env "$@" python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-34s %7.1f' % (' '.join(sys.argv[1:]) or 'baseline', d['value']))
" "$@"; }
run
for spec in conv2_fwd=3 conv2_fwd=1 conv3_fwd=0 fc_fwd=0 fc_bwd=0 conv3_bwd=0 conv2_bwd=0 conv2_bwd=3 heads_dfeat=0 wgrad_conv=1; do run GRL_TUNE=i2cfg_$spec; done
run
