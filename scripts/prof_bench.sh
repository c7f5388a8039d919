#!/bin/bash
# rocprofv3 kernel-trace summary of the bench (graph replay): average kernel durations as launched in production
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
cd $R
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof/trace_kernel_stats.csv')))
tot = 0
for r in rows:
    if 'grl::' in r['Name']:
        n = r['Name'].replace('void ', '').split('(')[0][:70]
        print("%-72s calls %5s avg %8.2f us  min %7.2f  total %8.1f us" % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
cat gpurun_out/prof_bench.json | head -c 400
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
