#!/bin/bash
# development aid: per-launch workgroup-shape sweep (GRL_TUNE=i2cfg_<tag>=c) at another per-GPU batch:  B=128 bash scripts/cfg_sweep_b.sh
B=${B:-128}
run() { env "$@" python bench.py --global-batch $B --steps 300 --warmup 30 --repeats 3 --no-cpu-baseline --no-profile --no-learn-loop --no-success 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-34s %7.1f' % (' '.join(sys.argv[1:]) or 'baseline', d['value']))
" "$@"; }
run
for spec in conv2_fwd=3 conv2_fwd=1 conv3_fwd=0 conv3_fwd=1 fc_fwd=0 fc_bwd=0 conv3_bwd=0 conv3_bwd=1 conv2_bwd=0 conv2_bwd=3 wgrad_conv=1 wgrad_conv=3; do run GRL_TUNE=i2cfg_$spec; done
run
