"""GPU box: event-timed stages of the act() path for one observation batch size (dev tool).
   python tools/stage_times.py --obs-batch 1 --iters 200"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batch  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--obs-batch", type=int, default=1)
  ap.add_argument("--iters", type=int, default=100)
  ap.add_argument("--models", type=int, default=4)
  ap.add_argument("--candidates", type=int, default=128)
  ap.add_argument("--algorithm", default="WCM")
  ap.add_argument("--enc", default="fp32")
  ap.add_argument("--fused", type=int, default=-1, help="RIP_OPT_ENCODER_FUSED (-1 = auto)")
  ap.add_argument("--mega", type=int, default=-1, help="RIP_OPT_ENCODER_MEGA (-1 auto, 0 never, 1 up to 4 observations)")
  ap.add_argument("--variant", type=int, default=0, help="RIP_OPT_ENCODER_VARIANT bit mask")
  ap.add_argument("--blocks", action="store_true", help="also the encoder kernel by kernel (bench._encoder_kernel_times)")
  ap.add_argument("--search-kernel", type=int, default=0, help="RIP_OPT_SEARCH_KERNEL (0 auto, 1 chain, 3 phase, 4 split)")
  args = ap.parse_args()
  from oatomobile_amd import ImitativeModel, RIPAgent, _lib
  dev = torch.device("cuda", 0)
  K, N, B = args.models, args.candidates, args.obs_batch
  models = [ImitativeModel.synthetic(100 + k, max_batch=1) for k in range(K)]
  agent = RIPAgent(None, algorithm=args.algorithm, models=models, num_candidates=N, max_batch=B, device=dev)
  lib, h = _lib.load(), agent._handle.raw
  _lib.check(lib.rip_set_option(h, 1, args.fused))
  _lib.check(lib.rip_set_option(h, 0, args.search_kernel))
  _lib.check(lib.rip_set_option(h, 3, args.mega))
  _lib.check(lib.rip_set_option(h, _lib.OPT_ENCODER_VARIANT, args.variant))
  lidar, vec, goal = (torch.from_numpy(a).to(dev) for a in synth_batch(np.random.default_rng(0), B, 2))
  x0 = agent._x0(B)
  z = torch.empty(K, B, 64, device=dev); plan = torch.empty(B, 4, 2, device=dev); loss = torch.empty(B, N, device=dev)
  st = torch.cuda.current_stream()
  ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.iters)]
  def run(e=None):
    s = _lib.current_stream()
    if e: e[0].record(st)
    _lib.check(lib.rip_encode_raw(h, _lib.ptr(lidar), 1, 200, 200, _lib.ptr(vec), B, 0, K, _lib.ENC_DTYPES[args.enc], _lib.ptr(z), s))
    if e: e[1].record(st)
    _lib.check(lib.rip_search(h, _lib.ptr(z), _lib.ptr(goal), _lib.ptr(x0), B, N, goal.shape[1], _lib.ALGORITHMS[args.algorithm],
                              10, 0.1, 1.0, _lib.ptr(plan), None, _lib.ptr(loss), None, None, None, None, s))
    if e: e[2].record(st)
  for _ in range(10): run()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(args.iters): run(ev[i])
  torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / args.iters
  enc = np.median([e[0].elapsed_time(e[1]) for e in ev]) * 1e3
  sea = np.median([e[1].elapsed_time(e[2]) for e in ev]) * 1e3
  assert lib.rip_encoder_status(h) == 0
  if args.blocks:
    from bench import _encoder_kernel_times
    for r in _encoder_kernel_times(lib, agent._handle, lidar, B, K, 2, args.enc, reps=15):
      print("  blk %-5s %7.1f us  %s" % (r["through_layer"], r["us"], " ".join(r["kernels"]) + "  " + "; ".join(r.get("launch", []))))
  print("B=%d K=%d N=%d: encode %.1f us, search %.1f us, wall/iter %.1f us -> %.0f calls/s" % (B, K, N, enc, sea, wall * 1e6, B / wall))


if __name__ == "__main__":
  main()
