#!/bin/bash
# per-kernel PMC counters of the bench (separate passes; --pmc only, no trace domains)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcb_$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcb_$i -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > /dev/null 2>$R/gpurun_out/pmcb_$i.err
done
cd $R
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.Counter())
for f in sorted(glob.glob('gpurun_out/pmcb_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'grl::' not in r['Kernel_Name']: continue
        k=(r['Kernel_Name'].replace('void ','').split('(')[0][:44], r.get('Grid_Size'))
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
print("%-46s %8s %6s | %9s %7s %6s %6s %6s | %7s %7s %7s %7s" % ("kernel","grid","calls","gui_cyc","mfma%","wait%","stall%","act%","valu/m","lds/m","salu/m","bankcf"))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get('GRBM_GUI_ACTIVE',0)):
    n=cnt[k]['SQ_WAVE_CYCLES'] or 1
    gui=v['GRBM_GUI_ACTIVE']/max(1,cnt[k]['GRBM_GUI_ACTIVE'])
    mf=v['SQ_VALU_MFMA_BUSY_CYCLES']/max(1,cnt[k]['SQ_VALU_MFMA_BUSY_CYCLES'])
    wc=v['SQ_WAVE_CYCLES']; im=max(1,v['SQ_INSTS_MFMA'])
    print("%-46s %8s %6d | %9.0f %6.1f%% %5.1f%% %5.1f%% %5.1f%% | %7.1f %7.1f %7.1f %7.0f" % (k[0],k[1],n,gui, 100*mf/(gui*1024) if gui else 0,
          100*v['SQ_WAIT_ANY']/wc if wc else 0, 100*v['SQ_WAIT_INST_ANY']/wc if wc else 0, 100*v['SQ_ACTIVE_INST_ANY']/wc if wc else 0,
          v['SQ_INSTS_VALU']/im, v['SQ_INSTS_LDS']/im, v['SQ_INSTS_SALU']/im, v['SQ_LDS_BANK_CONFLICT']/max(1,cnt[k]['SQ_LDS_BANK_CONFLICT'])))
PY
