"""GPU box experiment: encoder of batch i+1 on one stream beside the search of batch i on another (two handles), against
the sequential step.  python tools/dev/overlap_probe.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth_batch  # noqa: E402


def main():
  from oatomobile_amd import ImitativeModel, RIPAgent, _lib
  dev = torch.device("cuda", 0)
  K, N, B, S = 4, 128, 512, 10
  models = [ImitativeModel.synthetic(100 + k, max_batch=1) for k in range(K)]
  enc = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, max_batch=B, device=dev, encoder_dtype="bf16")
  sea = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, max_batch=B, device=dev, encoder_dtype="bf16")
  lib = _lib.load()
  lidar, vec, goal = (torch.from_numpy(a).to(dev) for a in synth_batch(np.random.default_rng(0), B, 2))
  x0 = sea._x0(B)
  z = [torch.empty(K, B, 64, device=dev) for _ in range(2)]
  plan = torch.empty(B, 4, 2, device=dev)
  s_e, s_s = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
  z_ready = [torch.cuda.Event() for _ in range(2)]
  z_free = [torch.cuda.Event() for _ in range(2)]

  def seq(n):
    st = torch.cuda.current_stream(dev)
    for i in range(n):
      _lib.check(lib.rip_encode_raw(enc._handle.raw, _lib.ptr(lidar), 1, 200, 200, _lib.ptr(vec), B, 0, K, 1, _lib.ptr(z[0]), st.cuda_stream))
      _lib.check(lib.rip_search(enc._handle.raw, _lib.ptr(z[0]), _lib.ptr(goal), _lib.ptr(x0), B, N, 10, 0, S, 0.1, 1.0, _lib.ptr(plan),
                                None, None, None, None, None, None, st.cuda_stream))

  def piped(n):
    for j in range(2):
      z_free[j].record(s_s)
    for i in range(n + 1):
      j = i & 1
      if i < n:
        s_e.wait_event(z_free[j])
        _lib.check(lib.rip_encode_raw(enc._handle.raw, _lib.ptr(lidar), 1, 200, 200, _lib.ptr(vec), B, 0, K, 1, _lib.ptr(z[j]), s_e.cuda_stream))
        z_ready[j].record(s_e)
      if i >= 1:
        p = (i - 1) & 1
        s_s.wait_event(z_ready[p])
        _lib.check(lib.rip_search(sea._handle.raw, _lib.ptr(z[p]), _lib.ptr(goal), _lib.ptr(x0), B, N, 10, 0, S, 0.1, 1.0, _lib.ptr(plan),
                                  None, None, None, None, None, None, s_s.cuda_stream))
        z_free[p].record(s_s)

  for name, fn in (("sequential", seq), ("encoder || search", piped), ("sequential", seq), ("encoder || search", piped)):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(20)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%-20s %.3f ms per 512-observation step -> %.0f calls/s" % (name, dt * 1e3, B / dt))


if __name__ == "__main__":
  main()
