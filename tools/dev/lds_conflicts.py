"""LDS bank-conflict check for the row-streaming encoder block kernel (round 1's encoder_bf16_irb.hip, retired in round 6: kept as the record of that sweep): which pixel-slot pitch
(ELD, bf16 elements) keeps the depthwise B-operand ds_read_b128 and the expansion's ds_write_b64 conflict-free.
Lane groups / bank rule: /opt/skills/guides/MI355X_MICROARCH.md, LDS section.  python tools/dev/lds_conflicts.py"""
G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G128 = G128 + [[l + 32 for l in g] for g in G128]
G64W = [list(range(16 * i, 16 * i + 16)) for i in range(4)]  # ds_write_b64: 4 x 16 contiguous, bank (a/4) mod 32


def cycles(addr_of_lane, groups, width, banks):
  tot = 0
  for g in groups:
    use = {}
    for l in g:
      a = addr_of_lane(l)
      for k in range(width // 4):
        use.setdefault(((a // 4) + k) % banks, set()).add((a // 4) + k)
    tot += max(len(v) for v in use.values())
  return tot


def report(S, NG):
  best = []
  for pad in range(0, 72, 8):
    ELD = 16 * NG + pad
    if ELD * 2 % 16:
      continue
    # depthwise B read: lane (n, q): pixel slot S*n (+kx for the odd half), 8-channel half q&1, tap half q>>1 (kx + 1)
    rd = cycles(lambda l: ((S * (l & 15) + (l >> 5)) * ELD + 8 * ((l >> 4) & 1)) * 2, G128, 16, 64)
    # the other tap of a pair can also sit one ring ROW below (ky + 1): a row is EW * ELD elements, take an odd EW
    rd2 = cycles(lambda l: ((S * (l & 15)) * ELD + (l >> 5) * 67 * ELD + 8 * ((l >> 4) & 1)) * 2, G128, 16, 64)
    # expansion write: lane (n, q): pixel slot n, channels 4q
    wr = cycles(lambda l: ((l & 15) * ELD + 4 * (l >> 4)) * 2, G64W, 8, 32)
    best.append((rd + rd2 + wr, ELD, rd, rd2, wr))
  best.sort()
  print("S=%d NG=%d:" % (S, NG), ["ELD=%d rd %d/%d (4 = free) wr %d (4 = free)" % (e, r, r2, w) for _, e, r, r2, w in best[:4]])


for S in (1, 2):
  for NG in (1, 2, 3):
    report(S, NG)
# projection operand row: lane (n, q) reads 16 B at pixel n, K offset 8q (+32 ks); dw epilogue writes 8 B at pixel n, 4q
for KS in (3, 5, 6):
  out = []
  for pad in range(0, 40, 8):
    DLD = 32 * KS + pad
    rd = cycles(lambda l: ((l & 15) * DLD + 8 * (l >> 4)) * 2, G128, 16, 64)
    wr = cycles(lambda l: ((l & 15) * DLD + 4 * (l >> 4)) * 2, G64W, 8, 32)
    out.append((rd + wr, DLD, rd, wr))
  out.sort()
  print("KS=%d:" % KS, ["DLD=%d rd %d wr %d" % (d, r, w) for _, d, r, w in out[:3]])
