#!/bin/bash
# GPU box: memory-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the fp32 encoder's kernels at 512 observations x 4 models,
# with the split-f16 kernels (variant 0) and with the layer-wise true-fp32 kernels (variant 16)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fp32pmc; rm -rf $O; mkdir -p $O; cd $R
for v in 0 16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $O/$v/$c --output-format csv -- python tools/stage_times.py --obs-batch 512 --iters 3 --enc fp32 --variant $v > $O/$v.$c.log 2>&1
  done
done
python - $O <<'PY'
import csv, glob, os, sys, collections
O = sys.argv[1]
def short(n):
  n = n.replace("void rip::(anonymous namespace)::", "").replace("rip::(anonymous namespace)::", "")
  return n.split("(")[0].replace(" ", "")[:56]
for v in ("0", "16"):
  agg = collections.defaultdict(lambda: collections.defaultdict(list))
  for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(O, v, c, "**", "*counter_collection.csv"), recursive=True)
    for r in csv.DictReader(open(f[0])):
      agg[short(r["Kernel_Name"])][c].append(float(r["Counter_Value"]))
  enc = max(1, len(agg.get("transform_kernel<2,true,float>", {}).get("FETCH_SIZE", [0])))
  print("== RIP_OPT_ENCODER_VARIANT %s (%d encodes): MB per dispatch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE counts 64-byte units as 32), dispatches per encode" % (v, enc))
  tot = 0.0
  rows = []
  for k, d in agg.items():
    if any(t in k for t in ("search", "select", "prefix")): continue
    fe, wr = d.get("FETCH_SIZE", [0.0]), d.get("WRITE_SIZE", [0.0])
    mb_f, mb_w = 2.0 * sum(fe) / len(fe) * 1024 / 1e6, sum(wr) / len(wr) * 1024 / 1e6
    per = len(fe) / enc
    tot += (mb_f + mb_w) * per
    rows.append((-(mb_f + mb_w) * per, "%-58s read %9.1f MB write %9.1f MB  x %4.1f per encode" % (k, mb_f, mb_w, per)))
  for _, l in sorted(rows): print(l)
  print("   memory-side traffic of one fp32 encode (512 observations x 4 models): %.2f GB" % (tot / 1e3))
PY
