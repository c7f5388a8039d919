#!/bin/bash
# HBM traffic per launch (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, MI355X_MICROARCH.md "HBM")
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmct_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmct_$c -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > /dev/null 2>$R/gpurun_out/pmct_$c.err
done
cd $R
GRL_PLAN_DUMP=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2> gpurun_out/pmct_plan.txt > /dev/null
python3 - <<'PY'
import csv,glob,collections,re
tag_of={}                      # grid size (threads) of an implicit-GEMM launch -> launch tag
for line in open('gpurun_out/pmct_plan.txt'):
    m=re.match(r'grl plan: (\S+) .* tiles (\d+)', line)
    if m: tag_of.setdefault(str(int(m.group(2))*256), m.group(1))
    m2 = re.match(r"grl plan: (\S+) +carries (\d+) filler tiles of '(\S+)' behind its own (\d+)", line)   # pair launch: one grid
    if m2: tag_of[str((int(m2.group(2)) + int(m2.group(4))) * 256)] = m2.group(1) + '+' + m2.group(3)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.Counter())
for f in sorted(glob.glob('gpurun_out/pmct_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'grl::' not in r['Kernel_Name']: continue
        k=(r['Kernel_Name'].replace('void ','').split('(')[0][:44], r.get('Grid_Size'))
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
print("kernel,grid,calls,FETCH_SIZE_KB_per_launch_raw,WRITE_SIZE_KB_per_launch_raw,HBM_MB_per_launch(fetch x2 gfx950 correction + write),launch_tag")
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get('FETCH_SIZE',0)):
    n=max(1,cnt[k]['FETCH_SIZE']); f=v['FETCH_SIZE']/n; w=v['WRITE_SIZE']/max(1,cnt[k]['WRITE_SIZE'])
    tag=tag_of.get(str(k[1]),'') if 'igemm' in k[0] else k[0].replace('grl::','').replace('_kernel','')
    print("\"%s\",%s,%d,%.1f,%.1f,%.2f,%s" % (k[0],k[1],n,f,w,(2*f+w)/1024.0,tag))
PY
