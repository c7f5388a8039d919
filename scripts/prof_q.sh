#!/bin/bash
# rocprofv3 kernel-trace summary of the BDQ / DQN update (graph replay)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/profq
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profq -o trace -- python $R/scripts/q_bench.py --replay 200000 --steps 400 > $R/gpurun_out/profq.json 2> $R/gpurun_out/profq.err
cd $R
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/profq/trace_kernel_stats.csv')))
for r in rows:
    if 'grl::' in r['Name']:
        n = r['Name'].replace('void ', '').split('(')[0][:64]
        print("%-66s calls %6s avg %8.2f us  total %9.1f us" % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
find gpurun_out/profq -name "*kernel_trace.csv" -size +20M -delete
