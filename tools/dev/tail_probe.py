import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from bench import synth_batch
from oatomobile_amd import ImitativeModel, RIPAgent, _lib, arch, transform_visual
dev = torch.device("cuda", 0)
K, N, B = 4, 128, 512
models = [ImitativeModel.synthetic(100 + k, max_batch=1) for k in range(K)]
agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, max_batch=B, device=dev)
lib, h = _lib.load(), agent._handle.raw
lidar, vec, goal = (torch.from_numpy(a).to(dev) for a in synth_batch(np.random.default_rng(0), B, 2))
vis = transform_visual(lidar, channels_last=True)
L = len(arch.conv_layers(2))
st = torch.cuda.current_stream()
z = torch.empty(K, B, 64, device=dev)
def timed(fn, reps=15):
  fn()
  evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
  for a, b in evs:
    a.record(st); fn(); b.record(st)
  torch.cuda.synchronize()
  return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3
E2 = _lib.ENC_DTYPES["bf16"]
full = lambda: _lib.check(lib.rip_encode(h, _lib.ptr(vis), _lib.ptr(vec), B, 0, K, E2, _lib.ptr(z), None, _lib.current_stream()))
tap = lambda i: (lambda: _lib.check(lib.rip_encode_tap_k(h, _lib.ptr(vis), B, 0, K, E2, i, None, 0, _lib.current_stream())))
for _ in range(2):
  print("full %.1f  tap51 %.1f  tap50 %.1f  full %.1f" % (timed(full), timed(tap(L - 1)), timed(tap(L - 2)), timed(full)))
