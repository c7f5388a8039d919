cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > $R/gpurun_out/sq_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_WAVE32_LDS SQ_WAVES"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_$n -o p -- $R/tests/_build/igemm_bench "ksweep-4wg" 0 > /dev/null 2>$R/gpurun_out/pmc_$n.err
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmc_*/**/*counter_collection.csv', recursive=True)):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in rows:
        k=(r['Kernel_Name'][:60], r.get('Grid_Size'))
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    for k,v in agg.items():
        if 'igemm2' in k[0]:
            print(k, {a:int(b) for a,b in v.items()})
PY
