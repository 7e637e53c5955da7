#!/bin/bash
# GPU box: kernel trace of a few act() iterations: per-kernel durations AND the gaps between consecutive kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ktrace32; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace -d $O/t --output-format csv -- python tools/stage_times.py --obs-batch 512 --iters 3 --enc fp32 > $O/log.txt 2>&1
python - $O <<'PY'
import csv, glob, os, sys
O = sys.argv[1]
f = glob.glob(os.path.join(O, "t", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "rip" in r["Kernel_Name"]]
last = rows[-70:]
prev_end = None
for r in last:
  s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
  name = r["Kernel_Name"].replace("void rip::(anonymous namespace)::", "").replace("rip::(anonymous namespace)::", "").split("(")[0][:44]
  print("%-46s dur %8.1f us   gap %7.1f us" % (name, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
  prev_end = e
PY
