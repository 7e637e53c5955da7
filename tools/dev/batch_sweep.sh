# GPU box, round 6: observations per step vs throughput (resident observations: stage_times; whole unit: bench.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sweep
for b in 512 768 1024 1536 2048; do
  timeout 300 python tools/stage_times.py --obs-batch $b --iters 10 --enc bf16 2>&1 | grep -v amdgpu.ids | tail -1
done
for b in 512 1024 2048; do
  timeout 600 python bench.py --obs-batch $b --no-extras --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/sweep/bench_$b.json 2>gpurun_out/sweep/bench_$b.err
  python - $b <<'PY'
import json, sys
b = sys.argv[1]
try:
  r = json.loads([l for l in open("gpurun_out/sweep/bench_%s.json" % b) if l.startswith("{")][-1])
  print("bench B=%s: value %.0f calls/s, %.3f ms/step, enc %.3f, search %.3f, repeats %s" % (b, r["value"], r["ms_per_step"], r["roofline"]["encoder"]["ms_per_step"], r["roofline"]["ms_per_launch"], r["repeats"]["calls_per_s"]))
except Exception as e:
  print("bench B=%s failed: %r" % (b, e)); print(open("gpurun_out/sweep/bench_%s.err" % b).read()[-800:])
PY
done
