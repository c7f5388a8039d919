#!/bin/bash
# one gpurun call of round 5 (same stage runner as scripts/gpu_round4.sh, plus the conv-stack gate)
#   STAGES="gate tests bench" bash scripts/gpu_round5.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
if [[ " ${STAGES-gate tests bench} " == *" gate "* ]]; then
  echo "== conv-stack gate (scripts/conv_stack_bench.hip)" > $R/gpurun_out/gate.log
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value $R/scripts/conv_stack_bench.hip -o /tmp/conv_stack_bench >> $R/gpurun_out/gate.log 2>&1 \
    && timeout 120 /tmp/conv_stack_bench ${GATE_REPS:-300} >> $R/gpurun_out/gate.log 2>&1
  echo "rc=$?" >> $R/gpurun_out/gate.log
  cat $R/gpurun_out/gate.log
fi
REST=$(echo " ${STAGES-gate tests bench} " | sed 's/ gate / /')
LOGNAME_R4=${LOGNAME_R5:-round5} STAGES="$REST" bash $R/scripts/gpu_round4.sh
