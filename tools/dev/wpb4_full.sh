# GPU box, development: the 4-wave (register-tape) search build forced onto the full 512-observation launch:
# time and WRITE_SIZE / FETCH_SIZE against the 8-wave default.  Restores the normal build.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
export RIP_EXTRA_HIPCC_FLAGS="-DRIP_FORCE_WPB=4"  # (bench.py rebuilds when the flags differ from the last build)
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
BENCH="python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3"
$BENCH 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('WPB=4 forced: search ms', round(d['roofline']['ms_per_launch'],3), 'calls/s', round(d['value']))"
for c in WRITE_SIZE FETCH_SIZE; do
O=$R/gpurun_out/wpb4_$c; rm -rf $O
rocprofv3 --kernel-trace --pmc $c --kernel-include-regex search_phase -d $O --output-format csv -- $BENCH > /dev/null 2>&1
python - "$O" $c <<'PY'
import csv,glob,sys,os
f=glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True)[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"]==sys.argv[2] and "false" in r["Kernel_Name"]]
print(sys.argv[2], "per launch (KiB): %.4g over %d launches" % (sum(v)/len(v), len(v)))
PY
done
unset RIP_EXTRA_HIPCC_FLAGS
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
