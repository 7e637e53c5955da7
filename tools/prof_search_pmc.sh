cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2e
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex search_phase -d $O/sq --output-format csv -- python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-include-regex search_phase -d $O/sq2 --output-format csv -- python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex search_phase -d $O/fetch --output-format csv -- python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex search_phase -d $O/write --output-format csv -- python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $O/write.log 2>&1
find $O -name "*counter_collection.csv" | while read f; do echo $f; python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if 'search_phase' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, sum(v)/len(v), len(v))
if rows: print('VGPR', rows[0].get('VGPR_Count'), 'LDS', rows[0].get('LDS_Block_Size'), 'grid', rows[0].get('Grid_Size'))
PY
done
