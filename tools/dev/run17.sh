# round-3 GPU call: kernel trace of the online path (one observation per call): durations and gaps per launch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3j/trace -- python tools/online_probe.py 120 > gpurun_out/r3j/probe.log 2>&1
grep "graph=" gpurun_out/r3j/probe.log
f=$(find gpurun_out/r3j/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "dispatches")
# find the last 20 calls of the graph phase: split by transform kernel occurrences
idx = [i for i, r in enumerate(rows) if "transform" in r["Kernel_Name"]]
print(len(idx), "transform launches")
def show(i0, i1):
  prev = None
  for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) if prev else 0
    print("%-60s dur %6.2f us gap %6.2f us grid %s wg %s" % (r["Kernel_Name"][:60], (e - s) / 1e3, gap / 1e3, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
    prev = e
for which in (100, 250):
  if which + 1 < len(idx):
    print("---- call", which)
    show(idx[which], idx[which + 1])
PY
