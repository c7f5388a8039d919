#!/bin/bash
# development aid: per-launch workgroup-shape sweep (GRL_I2CFG_<tag>) using bench.py's eager per-op timings
run() { env "$@" python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['step_kernel_ms']
print('%-34s value %7.1f  ' % (' '.join(sys.argv[1:]) or 'baseline', d['value']) + ' '.join('%s=%.1f' % (t, 1e3*k[t]) for t in sorted(k) if t in ('conv1_fwd','conv2_fwd','conv3_fwd','fc_fwd','heads_l0','heads_dfeat','fc_bwd','conv3_bwd','conv2_bwd','wgrad_dense','wgrad_conv')))
" "$@"; }
run
for tag in conv2_fwd conv3_fwd fc_fwd heads_l0 heads_dfeat fc_bwd conv3_bwd; do
  for c in 0 2 3; do run GRL_I2CFG_$tag=$c; done
done
