cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trainprof; rm -rf $O; mkdir -p $O; cd $R
cat > /tmp/tp.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
class A: channels=2
dev=torch.device("cuda",0)
import time
def timed(step, steps, warm, events=None):
    for i in range(warm): step(i,None)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(steps): step(i,None)
    torch.cuda.synchronize(); return time.perf_counter()-t0
print(bench._bench_train(A, dev, timed))
PY
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python /tmp/tp.py > $O/log.txt 2>&1
tail -2 $O/log.txt
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/trainprof/stats/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:30]: print("%-70s calls %5s total_ms %8.2f avg_us %8.1f  %s%%"%(r["Name"].replace("void rip::(anonymous namespace)::","")[:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
