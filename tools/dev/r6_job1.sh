# GPU box, round 6: outlier counts of the statistical gates (pytest -s), the new tests, a bench line with the new keys.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "search_candidates_vs_oracle or bench_configuration_parity or full_size_properties or model_parallel_gradient_mode or config3 or configs3 or eight_ranks or packed or headline_launch_shape or split_kernel_operand_ranges" > gpurun_out/r6/outliers.log 2>&1; echo "rc=$?" >> gpurun_out/r6/outliers.log
grep -n "candidates outside\|passed\|failed\|rc=\|Error\|error" gpurun_out/r6/outliers.log | cut -c1-220 | tail -80
timeout 900 python bench.py > gpurun_out/r6/bench_v1.json 2> gpurun_out/r6/bench_v1.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r6/bench_v1.json") if l.startswith("{")][-1])
ro = r["roofline"]
print("value", r["value"], "ms", r["ms_per_step"], "enc", ro["encoder"]["ms_per_step"], "search", ro["ms_per_launch"])
print("frac", ro["frac"], "useful", ro.get("useful_frac"), "measured_hbm_frac", ro.get("measured_hbm_frac"), "whole", ro["whole_act_hbm_frac"])
print("strict", r.get("strict_fp32_search"))
print("replay", r["replay"]["packed_cache"] if r.get("replay") and "packed_cache" in r["replay"] else r.get("replay"))
print("online", {k: r["online"][k] for k in ("calls_per_s", "p50_us")} if r.get("online") and "p50_us" in r["online"] else r.get("online"))
PY
