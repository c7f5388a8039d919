#!/bin/bash
# one gpurun call of round 6 (stage runner of scripts/gpu_round4.sh plus the dress rehearsals of the driver's N-GPU command)
#   STAGES="tests bench rehearsal6" bash scripts/gpu_round6.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
REST=""
for st in ${STAGES-tests bench}; do
  if [ "$st" = "rehearsal6" ]; then
    LOG=$R/gpurun_out/r06_dp_rehearsal.log; : > $LOG
    reh() {   # name, N, extra args
      local name=$1 N=$2; shift 2
      echo "== python -m torch.distributed.run --nproc-per-node $N bench.py --gpus $N --steps 20 --warmup 5 --dist-backend gloo --same-device $*" >> $LOG
      timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29700 + N + RANDOM % 50)) bench.py \
        --gpus $N --steps 20 --warmup 5 --dist-backend gloo --same-device --no-profile "$@" > $R/gpurun_out/r06_dp_rehearsal_$name.json 2> $R/gpurun_out/r06_dp_rehearsal_$name.err
      echo "rc=$?" >> $LOG; grep -v "^W0\|^\[Gloo\]\|^$\|amdgpu.ids" $R/gpurun_out/r06_dp_rehearsal_$name.err | tail -12 >> $LOG
      python3 -c "
import json,sys
try:
    d=json.loads(open('$R/gpurun_out/r06_dp_rehearsal_$name.json').read().strip().splitlines()[-1])
except Exception as e:
    print('no JSON line:', e); sys.exit()
c=d['config']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'block_ms', d['repeats']['block_ms'], 'calls_per_block', d['repeats']['calls_per_block'])
print('parallelism:', c['parallelism'])
print('exchange:', c.get('exchange'), '| distinct_devices:', c['devices']['distinct_devices'])
for r in c['devices']['ranks']: print('  rank', r['rank'], r['device'], r['name'], r.get('pci'), r.get('uuid'), 'peer_access', r['peer_access'])
c4=c.get('configs4')
if c4: print('configs4:', c4['value'], c4['unit'], 'ms_per_step', c4['ms_per_step'], 'per_gpu_batch', c4['per_gpu_batch'], '|', c4['parallelism'][:160])
print('validation_only:', c.get('validation_only'))
" >> $LOG 2>&1
    }
    reh w2 2
    reh w8 8 --replay 20000
    reh w2_ipc_failure_rank1 2 --inject-ipc-failure 1 --replay 20000
    reh w8_ipc_failure_rank5 8 --inject-ipc-failure 5 --replay 20000 --no-configs4
    cat $LOG
  else
    REST="$REST $st"
  fi
done
[ -n "$REST" ] && LOGNAME_R4=${LOGNAME_R6:-round6} STAGES="$REST" bash $R/scripts/gpu_round4.sh
