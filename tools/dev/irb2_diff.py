"""GPU box: where do the fused bf16 block kernels differ from the layer-wise kernels / the bf16 oracle?  (dev tool)
   python tools/dev/irb2_diff.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oatomobile_amd import ImitativeModel, _lib, arch, weights as W, transform_visual
from oracle import bf16_encoder as BE, reference_cpu as O
from tests.helpers import synth_observation

dev = torch.device("cuda", 0)
B = 3
m = ImitativeModel.synthetic(22, max_batch=B).to(dev)
mo = O.OracleImitativeModel.from_numpy_state_dict(W.synthetic_state_dict(22))
m.encoder_dtype = "bf16"
obs = [synth_observation(np.random.default_rng(2000 + i)) for i in range(B)]
lid = torch.stack([torch.from_numpy(o["lidar"]) for o in obs]).to(dev)
vis = transform_visual(lid, channels_last=True)
L = len(arch.conv_layers())
m.fused_encoder = 17
fused, ranges, first = {}, [], 0
for i in range(L):
  try:
    fused[i] = m.encoder_layer_output(vis, i).cpu()
  except _lib.RipError:
    continue
  ranges.append((first, i)); first = i + 1
m.fused_encoder = 0
lw = {i: m.encoder_layer_output(vis, i).cpu() for i in range(L)}
want = BE.teacher_forced(mo, {i: t for i, t in fused.items() if i < L - 1}, vis.cpu(), [r for r in ranges if r[1] < L - 1])
for a, b in ranges:
  if b == L - 1 or b - a != 2:
    continue
  f, l, o = fused[b].double(), lw[b].double(), want[b].double()
  # layer-wise output of the same block fed the FUSED path's input differs from lw[b] when an earlier block differs:
  for name, ref in (("oracle(teacher-forced)", o),):
    d = (f - ref).abs()
    scale = ref.abs().max().item()
    bad = d > (2.0 ** -7 * ref.abs() + 3e-5 * scale)
    nz = d > 0
    print("layers %d..%d vs %s: shape %s scale %.3g  differ %.3f %%  beyond 1-layer tol %.4f %%  max|d| %.3g" %
          (a, b, name, tuple(f.shape), scale, 100 * nz.double().mean().item(), 100 * bad.double().mean().item(), d.max().item()))
    if bad.any():
      idx = bad.nonzero()
      ys, xs, cs = idx[:, 2].numpy(), idx[:, 3].numpy(), idx[:, 1].numpy()
      H = f.shape[2]
      print("   rows of bad elements:", np.bincount(ys, minlength=H).tolist())
      print("   cols of bad elements:", np.bincount(xs, minlength=H).tolist())
      print("   channels:", np.bincount(cs, minlength=f.shape[1]).tolist())
    big = d > 4 * 2.0 ** -8 * scale
    print("   elements with |d| > 4 ulp of the scale: %d" % int(big.sum()))
