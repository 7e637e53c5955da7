cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
bash tools/dev/fp32_prof.sh > gpurun_out/r6/fp32_kernels_v4.txt 2>&1; tail -22 gpurun_out/r6/fp32_kernels_v4.txt | cut -c1-150
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r6/gpu_tests_v8.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r6/gpu_tests_v8.log | cut -c1-300
timeout 1200 python bench.py > gpurun_out/r6/bench_v9.json 2> gpurun_out/r6/bench_v9.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r6/bench_v9.json") if l.startswith("{")][-1])
ro = r["roofline"]
print("value", r["value"], "ms", r["ms_per_step"], "enc", ro["encoder"]["ms_per_step"], "search", ro["ms_per_launch"], r["repeats"]["calls_per_s"])
print("frac", ro["frac"], "whole", ro["whole_act_hbm_frac"])
for k in ("hbm_resident", "strict_fp32_search", "fp32_parity", "scoring_only"):
  print(k, {a: b for a, b in (r.get(k) or {}).items() if a not in ("note", "encoder_kernels")})
print("online", {k: r["online"][k] for k in ("calls_per_s", "p50_us")} if r.get("online") and "p50_us" in r["online"] else r.get("online"))
PY
