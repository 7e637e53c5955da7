#!/bin/bash
# GPU box: per-kernel time of the fp32 (parity-mode) encoder at 512 observations x 4 models
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fp32prof; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O/t --output-format csv -- python tools/stage_times.py --obs-batch 512 --iters 5 --enc fp32 > $O/log.txt 2>&1
python - $O <<'PY'
import csv, glob, os, sys, re
f = glob.glob(os.path.join(sys.argv[1], "t", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:24]:
  m = re.search(r"rip::\(anonymous namespace\)::(\w+(?:<[^>]*>)?)", r["Name"])
  if m: print("%-50s calls %5s total %9.1f us avg %8.1f us %6s%%" % (m.group(1).replace(" ", "")[:48], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
PY
