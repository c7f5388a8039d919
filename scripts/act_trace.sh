#!/bin/bash
# kernel trace of scripts/act_trace.py: per-kernel durations of one grl_act graph and the gaps between its kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/act_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $R/scripts/act_trace.py 2>/dev/null | grep "act on"
python3 - $OUT <<'PY'
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/t/**/*kernel_trace.csv', recursive=True)[0])))
rows = [r for r in rows if 'grl::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-400 * 6:]                       # the timed calls: 6 kernels each
dur, gap = collections.defaultdict(list), []
for i, r in enumerate(rows):
    name = r['Kernel_Name'].replace('void ', '').split('(')[0].replace('grl::', '')[:40]
    dur[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    if i % 6: gap.append((int(r['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])) / 1e3)
span = [(int(rows[i + 5]['End_Timestamp']) - int(rows[i]['Start_Timestamp'])) / 1e3 for i in range(0, len(rows) - 5, 6)]
for k, v in dur.items(): print("  %-42s avg %6.2f us  (%d)" % (k, sum(v) / len(v), len(v)))
print("  gap between consecutive kernels of a call: avg %.2f us;  first kernel start -> last kernel end: avg %.2f us" % (sum(gap) / len(gap), sum(span) / len(span)))
PY
