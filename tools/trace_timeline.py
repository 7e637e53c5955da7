"""Print the per-kernel timeline of the last full step in a rocprofv3 kernel_trace.csv (newest under the given dir)."""
import csv, glob, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
fs = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "transform_kernel" in r["Kernel_Name"]]
tot = 0.0
for r in rows[idx[-2]:idx[-1]]:
  dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
  tot += dur
  n = r["Kernel_Name"].replace("void rip::(anonymous namespace)::", "").replace("rip::(anonymous namespace)::", "")
  print(f"{n[:64]:64s} grid={r['Grid_Size_X']:>8s},{r['Grid_Size_Y']:>3s} {dur:8.1f}")
print("sum of kernel durations: %.1f us  (%s)" % (tot, fs[-1]))
