"""Development: instruction mix between `; RIPMARK name` comment lines of a kernel's ISA listing (built with -DRIP_ISA_MARKS).
   python tools/dev/isa_regions.py file.s kernel-substring"""
import collections, sys


def klass(op):
  if op.startswith("v_mfma"): return "mfma_f16" if "f16" in op or "bf16" in op else "mfma_f32"
  if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")): return "trans"
  if op.startswith(("v_readlane", "v_writelane")): return "lane_spill"
  if op.startswith("v_"): return "valu"
  if op.startswith("ds_"): return "lds"
  if op.startswith("scratch_"): return "scratch"
  if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
  if op.startswith("s_waitcnt"): return "waitcnt"
  if op.startswith("s_nop"): return "nop"
  if op.startswith("s_"): return "salu"
  return "other"


def main():
  lines = open(sys.argv[1]).read().split("\n")
  start = [i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l and ":" in l][0]
  end = [i for i, l in enumerate(lines) if i > start and l.strip().startswith("s_endpgm")][0]
  cur, regions = "prologue", []
  cnt = collections.Counter()
  for l in lines[start:end]:
    t = l.strip()
    if t.startswith("; RIPMARK"):
      regions.append((cur, cnt))
      cur, cnt = t.split()[2], collections.Counter()
      continue
    if not t or t.startswith((".", ";")) or t.endswith(":"):
      continue
    cnt[klass(t.split()[0])] += 1
  regions.append((cur, cnt))
  keys = ["mfma_f16", "mfma_f32", "trans", "valu", "lane_spill", "lds", "scratch", "vmem", "salu", "waitcnt", "nop"]
  print("%-16s" % "after mark" + "".join("%11s" % k for k in keys) + "   est.cycles")
  for name, c in regions:
    est = 16 * c["mfma_f16"] + 32 * c["mfma_f32"] + 16 * c["trans"] + 4 * (c["valu"] + c["lane_spill"])
    print("%-16s" % name + "".join("%11d" % c[k] for k in keys) + "   %8d" % est)


if __name__ == "__main__":
  main()
