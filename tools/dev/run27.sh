cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for f in -1 0 3 6 10 17; do echo "fused=$f"; python tools/stage_times.py --obs-batch 512 --iters 10 --enc fp32 --fused $f 2>&1 | grep "B="; done
mkdir -p gpurun_out/r3n
rocprofv3 --kernel-trace --stats -d gpurun_out/r3n/fp32 --output-format csv -- python tools/stage_times.py --obs-batch 512 --iters 10 --enc fp32 > gpurun_out/r3n/log.txt 2>&1
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r3n/fp32/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]: print("%-90s calls %5s total_ms %8.2f avg_us %8.1f  %s%%"%(r["Name"].replace("void rip::(anonymous namespace)::","")[:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
