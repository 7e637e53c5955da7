"""GPU box (dev): two handles on two streams, each running encode -> search on its own 512-observation batch, against the
same work on one stream: do the encoder's launch tails and the search fill each other's gaps?"""
import os, sys, time, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth_batch
from oatomobile_amd import ImitativeModel, RIPAgent, _lib
dev = torch.device("cuda", 0)
K, N, B = 4, 128, 512
lib = _lib.load()
models = [ImitativeModel.synthetic(100 + k, max_batch=1) for k in range(K)]

def setup(seed):
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, max_batch=B, device=dev, encoder_dtype="bf16")
  lidar, vec, goal = (torch.from_numpy(a).to(dev) for a in synth_batch(np.random.default_rng(seed), B, 2))
  return dict(agent=agent, h=agent._handle.raw, lidar=lidar, vec=vec, goal=goal, z=torch.empty(K, B, 64, device=dev),
              x0=agent._x0(B), plan=torch.empty(B, 4, 2, device=dev), loss=torch.empty(B, N, device=dev))

def act(c, stream):
  s = ctypes.c_void_p(stream.cuda_stream)
  _lib.check(lib.rip_encode_raw(c["h"], _lib.ptr(c["lidar"]), 1, 200, 200, _lib.ptr(c["vec"]), B, 0, K, 1, _lib.ptr(c["z"]), s))
  _lib.check(lib.rip_search(c["h"], _lib.ptr(c["z"]), _lib.ptr(c["goal"]), _lib.ptr(c["x0"]), B, N, 10, 0, 10, 0.1, 1.0,
                            _lib.ptr(c["plan"]), None, _lib.ptr(c["loss"]), None, None, None, None, s))

def timeit(fn, iters=20):
  for _ in range(3): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(iters): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / iters * 1e3

a, b = setup(1), setup(2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
t_seq = timeit(lambda: (act(a, s1), act(b, s1)))
t_par = timeit(lambda: (act(a, s1), act(b, s2)))
print("two 512-observation act() batches: one stream %.3f ms (%.0f calls/s) | two streams %.3f ms (%.0f calls/s)" %
      (t_seq, 2 * B / t_seq * 1e3, t_par, 2 * B / t_par * 1e3))
