"""pmc_summary.csv (tools/prof_round.sh) -> measured.json: the HBM-side bytes bench.py quotes as `roofline.traffic` and
`roofline.encoder.measured_hbm_bytes`.  FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE is doubled (the gfx950
correction of MI355X_MICROARCH.md's HBM / rocprofv3 section: the counter's unit is 64 B while requests are 128 B).

  python tools/pmc_measured.py <pmc_summary.csv> <source label> [obs_batch models candidates steps channels dtype algo]
"""
import csv, json, re, sys

ENC = re.compile(r"^(cls_|dw_|front|gemm_|irb|merger|transform|pw_|stem)")
SEARCH = re.compile(r"^search_(phase|split)\w*kernel<false")


def main():
  path, source = sys.argv[1], sys.argv[2]
  cfg = sys.argv[3:] + [None] * 7
  rows = list(csv.DictReader(open(path)))
  by = {}
  for r in rows:
    by.setdefault(r["kernel"], {})[r["counter"]] = (float(r["mean_per_dispatch"]), int(r["dispatches"]), float(r["sum_over_run"]))
  search = [k for k in by if SEARCH.match(k) and "FETCH_SIZE" in by[k]]
  out = {"source": source,
         "config": {"obs_batch": int(cfg[0] or 512), "models": int(cfg[1] or 4), "candidates": int(cfg[2] or 128),
                    "search_steps": int(cfg[3] or 10), "channels": int(cfg[4] or 2), "encoder_dtype": cfg[5] or "bf16",
                    "algorithm": cfg[6] or "WCM"}}
  steps = None
  if search:
    k = max(search, key=lambda n: by[n]["FETCH_SIZE"][0])  # (the operand-range fallback launch of flow_phase.hip moves ~0 bytes)
    f, w = by[k]["FETCH_SIZE"], by[k].get("WRITE_SIZE", (0.0, 0, 0.0))
    steps = f[1]
    out["search"] = {"kernel": k, "fetch_KiB": f[0], "write_KiB": w[0], "traffic_bytes": (2.0 * f[0] + w[0]) * 1024.0}
    for c in ("SQ_INSTS_MFMA", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES"):
      if c in by[k]:
        out["search"][c] = by[k][c][0]
  once = [v["FETCH_SIZE"][1] for k, v in by.items() if k.startswith(("merger_kernel", "transform_kernel", "cls_")) and "FETCH_SIZE" in v]
  if once:
    steps = min(once)  # encoder steps of the run (the search kernel also runs outside the step: bench.py's trace / stats launches)
  if steps:
    fe = sum(v["FETCH_SIZE"][2] for k, v in by.items() if ENC.match(k) and "FETCH_SIZE" in v) / steps
    wr = sum(v["WRITE_SIZE"][2] for k, v in by.items() if ENC.match(k) and "WRITE_SIZE" in v) / steps
    out["encoder"] = {"fetch_KiB": fe, "write_KiB": wr, "traffic_bytes": (2.0 * fe + wr) * 1024.0,
                      "kernels": sorted(k for k in by if ENC.match(k))}
  json.dump(out, sys.stdout, indent=1)
  print()


if __name__ == "__main__":
  main()
