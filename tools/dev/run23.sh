# round-3 GPU call: validation after the small-launch encoder work — full GPU suite, smoke, bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r3l/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3l/gpu_tests.log
tail -3 gpurun_out/r3l/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r3l/bench.json 2> gpurun_out/r3l/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3l/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
for k in ("online", "fp32_parity", "hbm_resident"):
  print(k, json.dumps(d.get(k))[:300])
PY
