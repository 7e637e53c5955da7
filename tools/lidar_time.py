"""Time rip_lidar_bev on a batch of synthetic CARLA-like clouds (HIP events) and report bytes / time."""
import argparse, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oatomobile_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--obs-batch", type=int, default=256)
ap.add_argument("--points", type=int, default=50000)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
rng = np.random.default_rng(0)
B, P = a.obs_batch, a.points
pts = np.c_[rng.normal(0, 18, size=(B * P, 2)), rng.normal(-2.4, 0.6, size=(B * P, 1))].astype(np.float32)
dpts = torch.from_numpy(pts).cuda()
off = torch.arange(0, (B + 1) * P, P, dtype=torch.int32, device="cuda")
bev = torch.empty(B, 200, 200, 2, device="cuda")
lib = _lib.load()
for _ in range(3):
  _lib.check(lib.rip_lidar_bev(_lib.ptr(dpts), _lib.ptr(off), B, _lib.ptr(bev), _lib.current_stream()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
  _lib.check(lib.rip_lidar_bev(_lib.ptr(dpts), _lib.ptr(off), B, _lib.ptr(bev), _lib.current_stream()))
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.iters
byts = B * (P * 12 + 200 * 200 * 2 * 4)
print("rip_lidar_bev B=%d x %d points: %.1f us -> %.2f TB/s algorithmic (12 B/point + 320 KB/observation), %.2f G points/s"
      % (B, P, us, byts / us / 1e6, B * P / us / 1e3))
