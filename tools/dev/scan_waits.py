"""Development: compile every csrc/*.hip to gfx950 assembly (the flags of __graft_entry__) and list, per kernel,
(a) vector-memory loads that are followed by `s_waitcnt vmcnt(0)` within eight instructions — a load the wave waits
    for on its own, usually because it sits behind a per-lane branch (the compiler's wait insertion falls back to
    vmcnt(0) at every control-flow merge), and
(b) scratch (private segment) use — e.g. a ternary on HIP's float4 struct that went through a stack slot.
   python tools/dev/scan_waits.py [min_count]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def scan(sources=None):
  """{(source, kernel): (loads waited for on their own, loads, scratch bytes)} for every kernel of `sources`
  (file names under csrc/; None = all)."""
  out = tempfile.mkdtemp(prefix="rip_asm_")
  procs = []
  for src in sorted(glob.glob(os.path.join(G.CSRC, "*.hip"))):
    name = os.path.basename(src)
    if sources is not None and name not in sources:
      continue
    flags = [f for f in G.FLAGS if f != "-fPIC"] + G.SOURCE_FLAGS.get(name, [])
    procs.append((name, subprocess.Popen(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-o",
                                         os.path.join(out, name + ".s"), src], stderr=subprocess.DEVNULL)))
  result = {}
  for name, pr in procs:
    pr.wait()
    path = os.path.join(out, name + ".s")
    if not os.path.exists(path):
      continue
    lines = open(path).read().splitlines()
    kern, stats, last = None, {}, None
    for l in lines:
      m = re.match(r"^(_Z\w+):", l)
      if m:
        kern, last = m.group(1), None
        stats[kern] = [0, 0]
        continue
      t = l.strip()
      if kern is None or not t or t[0] in ";.":
        continue
      op = t.split()[0]
      if op.startswith(("global_load", "buffer_load", "flat_load")) and "lds" not in op:
        last = 0
        stats[kern][1] += 1
      elif last is not None:
        last += 1
        if op == "s_waitcnt" and "vmcnt(0)" in t and last <= 8:
          stats[kern][0] += 1
          last = None
        elif last > 8:
          last = None
    scratch = {}
    cur = None
    for l in lines:
      m = re.match(r"\s*\.name:\s+(\S+)", l)
      if m:
        cur = m.group(1)
      m = re.match(r"\s*\.private_segment_fixed_size:\s+(\d+)", l)
      if m and cur:
        scratch[cur] = int(m.group(1))
    for k, (a, b) in stats.items():
      if k in scratch:  # kernels only (device functions have no metadata entry)
        result[(name, k)] = (a, b, scratch[k])
  return result


def main():
  min_count = int(sys.argv[1]) if len(sys.argv) > 1 else 6
  for (name, k), (a, b, sc) in sorted(scan().items()):
    if a >= min_count or sc > 0:
      print("%-26s %3d of %3d loads waited for on their own, scratch %4d B  %s" % (name, a, b, sc, k[:100]))


if __name__ == "__main__":
  main()
