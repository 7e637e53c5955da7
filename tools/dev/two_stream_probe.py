"""GPU box (dev): does running the per-step (binned) search of two half batches on two streams hide each other's launch
tails?  python tools/dev/two_stream_probe.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth_batch
from oatomobile_amd import ImitativeModel, RIPAgent, _lib
dev = torch.device("cuda", 0)
K, N = 4, 128
lib = _lib.load()
models = [ImitativeModel.synthetic(100 + k, max_batch=1) for k in range(K)]

def setup(B, seed):
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, max_batch=B, device=dev, encoder_dtype="bf16")
  h = agent._handle.raw
  lidar, vec, goal = (torch.from_numpy(a).to(dev) for a in synth_batch(np.random.default_rng(seed), B, 2))
  z = torch.empty(K, B, 64, device=dev)
  _lib.check(lib.rip_encode_raw(h, _lib.ptr(lidar), 1, 200, 200, _lib.ptr(vec), B, 0, K, 1, _lib.ptr(z), _lib.current_stream(dev)))
  torch.cuda.synchronize()
  return dict(agent=agent, h=h, z=z, goal=goal, x0=agent._x0(B), plan=torch.empty(B, 4, 2, device=dev), loss=torch.empty(B, N, device=dev), B=B)

def search(c, stream):
  _lib.check(lib.rip_search(c["h"], _lib.ptr(c["z"]), _lib.ptr(c["goal"]), _lib.ptr(c["x0"]), c["B"], N, 10, 0, 10, 0.1, 1.0,
                            _lib.ptr(c["plan"]), None, _lib.ptr(c["loss"]), None, None, None, None, ctypes_stream(stream)))

def ctypes_stream(s):
  import ctypes
  return ctypes.c_void_p(s.cuda_stream)

def timeit(fn, iters=20):
  for _ in range(3): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(iters): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / iters * 1e3

full = setup(512, 0)
ha, hb = setup(256, 1), setup(256, 2)
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
for binned in (0, 1):
  for c in (full, ha, hb):
    _lib.check(lib.rip_set_option(c["h"], _lib.OPT_SEARCH_BINNED, binned))
  t_full = timeit(lambda: search(full, s0))
  t_two = timeit(lambda: (search(ha, s1), search(hb, s2)))
  t_half = timeit(lambda: search(ha, s1))
  print("binned=%d: one launch of 512 observations %.3f ms | two halves on two streams %.3f ms | one half alone %.3f ms" % (binned, t_full, t_two, t_half))
