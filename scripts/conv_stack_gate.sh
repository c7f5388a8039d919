#!/bin/bash
# conv-stack gate on the GPU box: variants of scripts/conv_stack_bench.hip (macros), phase stamps, SQ counters.
#   VARIANTS="base -DCS_PREFETCH=3 ..." bash scripts/conv_stack_gate.sh     -> gpurun_out/gate.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
LOG=$R/gpurun_out/gate.log
: > $LOG
IFS='|' read -ra VS <<< "${VARIANTS-base|-DCS_STAMPS|-DCS_PREFETCH=3|-DCS_PREFETCH=4|-DCS_NOSTORE}"
k=0
for v in "${VS[@]}"; do
  k=$((k + 1))
  [ "$v" = base ] && f="" || f="$v"
  echo "== variant: $v" >> $LOG
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value $f $R/scripts/conv_stack_bench.hip -o /tmp/csb_$k >> $LOG 2>&1 \
    && timeout 120 /tmp/csb_$k ${GATE_REPS:-300} ${GATE_ONLY:-} >> $LOG 2>&1
  echo "rc=$?" >> $LOG
done
if [ "${GATE_RESIDENCY:-1}" = 1 ]; then      # how much do co-resident workgroups help?  pad the LDS so that 3 / 2 / 1 share a CU
  for pad in 0 30000 90000; do
    echo "== base variant, headline shape, dynamic LDS pad $pad" >> $LOG
    timeout 60 /tmp/csb_1 ${GATE_REPS:-300} 256 $pad >> $LOG 2>&1
  done
fi
if [ "${GATE_PMC:-1}" = 1 ]; then
  cd /tmp && export TMPDIR=/tmp
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE"; do
    n=$(echo $set | cut -d' ' -f1)
    rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$n -o p -- /tmp/csb_1 20 256 > /dev/null 2>/tmp/pmc_$n.err
  done
  cd $R
  python3 - >> $LOG <<'PY'
import csv, glob, collections
print("== SQ counters (base variant, C=1 B=256 only: per launch averages)")
for f in sorted(glob.glob('/tmp/pmc_*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:50], r.get('Grid_Size'))
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
    for k, v in agg.items():
        if 'conv_stack' in k[0]:
            print(k, {a: int(b / cnt[(k, a)]) for a, b in v.items()})
PY
fi
cat $LOG
