"""Per-launch timeline of the last training step in a rocprofv3 kernel trace (steps are delimited by stem_fwd_kernel)."""
import csv, glob, os, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
fs = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stem_fwd_kernel" in r["Kernel_Name"]]
seg = rows[idx[-2]:idx[-1]]
t_first, t_last = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
tot = 0.0
for r in seg:
  dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
  tot += dur
  n = r["Kernel_Name"].replace("void rip::(anonymous namespace)::", "").replace("rip::(anonymous namespace)::", "")
  if flt in n:
    print(f"{n[:60]:60s} grid={r['Grid_Size_X']:>8s},{r['Grid_Size_Y']:>4s} {dur:8.1f}")
print("launches %d, sum of kernel durations %.1f us, first start -> last end %.1f us" % (len(seg), tot, (t_last - t_first) / 1e3))
