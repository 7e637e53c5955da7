# GPU box: the round's rocprofv3 evidence (summaries land in gpurun_out/prof/; copy what is judged into profiles/rN/).
#   bash tools/prof_round.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd $R
BENCH="python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3"
# 1. per-kernel time of the bench command
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- $BENCH > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null
# 2. SQ counters of the plan-search kernel
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex 'search_split|search_phase' -d $O/sq --output-format csv -- $BENCH > $O/sq.log 2>&1
# 3. / 4. memory-side traffic of every kernel of the step (separate passes: FETCH_SIZE and WRITE_SIZE do not fit together)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- $BENCH > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- $BENCH > $O/write.log 2>&1
python - "$O" <<'PY'
import csv, sys, glob, collections, os
O = sys.argv[1]
def load(sub):
  f = glob.glob(os.path.join(O, sub, "**", "*counter_collection.csv"), recursive=True)
  return list(csv.DictReader(open(f[0]))) if f else []
def short(n):
  n = n.replace("void rip::(anonymous namespace)::", "").replace("rip::(anonymous namespace)::", "")
  return n.split("(")[0][:70]
with open(os.path.join(O, "pmc_summary.csv"), "w") as out:
  w = csv.writer(out)
  w.writerow(["kernel", "counter", "mean_per_dispatch", "dispatches", "sum_over_run"])
  for sub in ("sq", "fetch", "write"):
    agg = collections.defaultdict(list)
    for r in load(sub):
      agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
      w.writerow([k, c, "%.6g" % (sum(v) / len(v)), len(v), "%.6g" % sum(v)])
print(open(os.path.join(O, "pmc_summary.csv")).read()[:6000])
PY
# HBM-side bytes per launch in the form bench.py reads (copy to profiles/measured.json with the summary it condenses)
python tools/pmc_measured.py $O/pmc_summary.csv "profiles/${ROUND:-r6}/pmc_summary_${1:-vN}.csv" $(python -c "import bench; print(bench.DEFAULT_OBS_BATCH)") > $O/measured.json
cat $O/measured.json
head -40 $O/bench_kernel_stats.csv
