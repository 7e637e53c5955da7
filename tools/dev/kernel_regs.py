"""Development: registers / static LDS / scratch of every kernel in gfx950 assembly files (hipcc -S --cuda-device-only).
   python tools/dev/kernel_regs.py /tmp/asm/*.s"""
import re, subprocess, sys
for f in sys.argv[1:]:
  cur = {}
  for l in open(f):
    m = re.match(r"\s+\.(name|vgpr_count|agpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|vgpr_spill_count):\s+(\S+)", l)
    if m:
      cur[m.group(1)] = m.group(2)
      if m.group(1) == "vgpr_spill_count":
        n = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        n = n.replace("void rip::(anonymous namespace)::", "").replace("rip::(anonymous namespace)::", "").split("(")[0][:76]
        print("%-78s vgpr %3s agpr %3s lds %6s scratch %s" % (n, cur.get("vgpr_count"), cur.get("agpr_count"), cur.get("group_segment_fixed_size"), cur.get("private_segment_fixed_size")))
