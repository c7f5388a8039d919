#!/bin/bash
# rocprofv3 evidence for one bench workload: (1) kernel-trace --stats of the graph-replay run, (2) HBM traffic per launch
# (FETCH_SIZE / WRITE_SIZE, separate --pmc passes), (3) SQ counters per kernel (two --pmc passes).  Summaries land in
# gpurun_out/prof_<name>/ as text / csv; copy the ones to be kept into profiles/.
#   NAME=sac_depth BENCH_ARGS="" bash scripts/profile_round.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=${NAME:-sac_depth}
ARGS="${BENCH_ARGS:-} --no-cpu-baseline --no-profile --no-learn-loop --no-success --repeats 1"
OUT=$R/gpurun_out/prof_$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --steps 100 --warmup 10 $ARGS > $OUT/bench_under_trace.json 2> $OUT/trace.err
if [ "${PMC:-1}" = "1" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmct_$c -o p -- python $R/bench.py --steps 20 --warmup 5 $ARGS > /dev/null 2> $OUT/pmct_$c.err
done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmcb_$i -o p -- python $R/bench.py --steps 20 --warmup 5 $ARGS > /dev/null 2> $OUT/pmcb_$i.err
done
fi
cd $R
GRL_PLAN_DUMP=1 timeout 200 python bench.py --steps 2 --warmup 1 $ARGS 2> $OUT/plan.txt > /dev/null
python3 - "$OUT" <<'PY'
import csv, glob, collections, re, sys, os
out = sys.argv[1]
# ---- launch tags by grid size
tag_of = {}
for line in open(out + '/plan.txt'):
    m = re.match(r'grl plan: (\S+) .* tiles (\d+)', line)
    if m: tag_of.setdefault(str(int(m.group(2)) * 256), m.group(1))
    m2 = re.match(r"grl plan: (\S+) +carries (\d+) filler tiles of '(\S+)' behind its own (\d+)", line)   # pair launch: one grid
    if m2: tag_of[str((int(m2.group(2)) + int(m2.group(4))) * 256)] = m2.group(1) + '+' + m2.group(3)
# ---- (1) kernel stats of the grl:: kernels
rows = list(csv.DictReader(open(glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True)[0])))
with open(out + '/kernel_summary.txt', 'w') as f:
    tot = 0.0
    for r in rows:
        if 'grl::' in r['Name']:
            n = r['Name'].replace('void ', '').split('(')[0][:96]
            f.write("%-98s calls %5s avg %8.2f us  min %7.2f  total %9.1f us\n" % (n, r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
    f.write(open(out + '/bench_under_trace.json').read()[:600] + "\n")
# ---- (1b) per-LAUNCH averages under graph replay (kernel trace grouped by kernel + grid -> launch tag): the lines bench.py's
# roofline.frac_graph reads ("launch <tag> calls <n> avg <us> us ...")
tr = glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True)
if tr:
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        if 'grl::' not in r['Kernel_Name']: continue
        nm = r['Kernel_Name'].replace('void ', '').split('(')[0][:60]
        g = str(int(r.get('Grid_Size', r.get('Grid_Size_X', '0'))))
        t = tag_of.get(g, '') if 'igemm' in nm else nm.replace('grl::', '').replace('_kernel', '').split('<')[0]
        per[(t or nm + ':' + g).split('+')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    with open(out + '/kernel_summary.txt', 'a') as f:
        f.write("per launch tag (kernel trace grouped by kernel + grid size; tags from GRL_PLAN_DUMP):\n")
        for t, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            f.write("launch %-16s calls %5d avg %8.2f us  min %7.2f  total %9.1f us\n" % (t, len(v), sum(v) / len(v), min(v), sum(v)))
def collect(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for fpath in sorted(glob.glob(out + pattern, recursive=True)):
        for r in csv.DictReader(open(fpath)):
            if 'grl::' not in r['Kernel_Name']: continue
            k = (r['Kernel_Name'].replace('void ', '').split('(')[0][:60], r.get('Grid_Size'))
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
    return agg, cnt
def tag(k):
    return tag_of.get(str(k[1]), '') if 'igemm' in k[0] else k[0].replace('grl::', '').replace('_kernel', '').split('<')[0]
agg, cnt = collect('/pmct_*/**/*counter_collection.csv')
if agg:
    with open(out + '/pmc_hbm_traffic.csv', 'w') as f:
        f.write("kernel,grid,calls,FETCH_SIZE_KB_per_launch_raw,WRITE_SIZE_KB_per_launch_raw,HBM_MB_per_launch(fetch x2 gfx950 correction + write),launch_tag\n")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('FETCH_SIZE', 0)):
            n = max(1, cnt[k]['FETCH_SIZE']); fe = v['FETCH_SIZE'] / n; w = v['WRITE_SIZE'] / max(1, cnt[k]['WRITE_SIZE'])
            f.write("\"%s\",%s,%d,%.1f,%.1f,%.2f,%s\n" % (k[0], k[1], n, fe, w, (2 * fe + w) / 1024.0, tag(k)))
agg, cnt = collect('/pmcb_*/**/*counter_collection.csv')
if agg:
    with open(out + '/pmc_sq_counters.txt', 'w') as f:
        f.write("per kernel (average per launch; the SQ counters of this rocprofv3 cover ONE of the 8 XCDs -- 32 CUs x 4 SIMDs -- so mfma%% = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)): MFMA-pipe busy %% of GPU-active cycles, wave cycles spent waiting (any / on an instruction), issuing\n")
        f.write("%-62s %-12s %8s %6s | %9s %7s %6s %6s %6s | %8s %8s %8s %8s\n" % ("kernel", "tag", "grid", "calls", "gui_cyc", "mfma%", "wait%", "stall%", "act%", "valu/mfma", "lds/mfma", "salu/mfma", "mfma_mops_f32"))
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0)):
            n = cnt[k]['SQ_WAVE_CYCLES'] or 1
            gui = v['GRBM_GUI_ACTIVE'] / max(1, cnt[k]['GRBM_GUI_ACTIVE'])
            mf = v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1, cnt[k]['SQ_VALU_MFMA_BUSY_CYCLES'])
            wc = v['SQ_WAVE_CYCLES']; im = max(1, v['SQ_INSTS_MFMA'])
            f.write("%-62s %-12s %8s %6d | %9.0f %6.1f%% %5.1f%% %5.1f%% %5.1f%% | %8.1f %8.1f %8.1f %12.0f\n" % (
                k[0], tag(k), k[1], n, gui, 100 * mf / (gui * 128) if gui else 0, 100 * v['SQ_WAIT_ANY'] / wc if wc else 0,
                100 * v['SQ_WAIT_INST_ANY'] / wc if wc else 0, 100 * v['SQ_ACTIVE_INST_ANY'] / wc if wc else 0,
                v['SQ_INSTS_VALU'] / im, v['SQ_INSTS_LDS'] / im, v['SQ_INSTS_SALU'] / im,
                v['SQ_INSTS_VALU_MFMA_MOPS_F32'] / max(1, cnt[k]['SQ_INSTS_VALU_MFMA_MOPS_F32'])))
PY
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
cat $OUT/kernel_summary.txt | cut -c1-170
[ -f $OUT/pmc_hbm_traffic.csv ] && cat $OUT/pmc_hbm_traffic.csv | cut -c1-200
[ -f $OUT/pmc_sq_counters.txt ] && cat $OUT/pmc_sq_counters.txt | cut -c1-220
