cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
for b in 512 2048; do timeout 300 python tools/stage_times.py --obs-batch $b --iters 10 --enc bf16 2>&1 | tail -1; done
timeout 2400 python -m pytest tests/ -q -m gpu -x > gpurun_out/r6/gpu_tests_v5.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r6/gpu_tests_v5.log | cut -c1-300
timeout 1200 python bench.py > gpurun_out/r6/bench_v6.json 2> gpurun_out/r6/bench_v6.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r6/bench_v6.json") if l.startswith("{")][-1])
ro = r["roofline"]
print("value", r["value"], "ms", r["ms_per_step"], "enc", ro["encoder"]["ms_per_step"], "search", ro["ms_per_launch"], r["repeats"]["calls_per_s"])
print("frac", ro["frac"], "useful", ro.get("useful_frac"), "measured_hbm_frac", ro.get("measured_hbm_frac"), "whole", ro["whole_act_hbm_frac"], "pipe", ro["matrix_pipe_busy"], "mfma", ro.get("mfma_instructions"))
for k in ("hbm_resident", "hbm_resident_512", "two_handles_two_streams", "strict_fp32_search", "fp32_parity", "scoring_only"):
  print(k, {a: b for a, b in (r.get(k) or {}).items() if a != "note"})
print("replay", {a: b for a, b in r["replay"]["packed_cache"].items() if a != "note"} if r.get("replay") and "packed_cache" in r["replay"] else r.get("replay"))
print("online", {k: r["online"][k] for k in ("calls_per_s", "p50_us")} if r.get("online") and "p50_us" in r["online"] else r.get("online"))
PY
