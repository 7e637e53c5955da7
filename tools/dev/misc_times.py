"""Development: wall times of the secondary paths in one line — training step (B=128), online call p50 (fp32 graph), fp32 encoder at B=512."""
import os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from oatomobile_amd import ImitativeModel
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
args = types.SimpleNamespace(channels=2, algorithm="WCM", candidates=128, search_steps=10, encoder_dtype="fp32", online_calls=600)
def timed(step, steps, warmup, events=None):
  for i in range(warmup): step(i, None)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(steps): step(i, None)
  torch.cuda.synchronize(); return time.perf_counter() - t0
out = []
if "train" in sys.argv or len(sys.argv) == 1:
  out.append("train %.2f ms/step" % bench._bench_train(args, dev, timed)["ms_per_step"])
if "online" in sys.argv or len(sys.argv) == 1:
  models = [ImitativeModel.synthetic(100 + k, max_batch=1).to(dev) for k in range(4)]
  hb = bench.synth_batch(np.random.default_rng(5), 8, 2)
  r = bench._bench_online(args, models, dev, hb)
  out.append("online p50 %.1f us mean %.1f us (eager p50 %.1f)" % (r["p50_us"], r["latency_us"], r["eager"]["p50_us"]))
print("; ".join(out))
