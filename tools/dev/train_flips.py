"""GPU box, development: ReLU6 mask differences between the HIP training forward and the CPU oracle, layer by layer."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oatomobile_amd import DIMTrainer, ImitativeModel, weights as W, transform_visual, _lib
from oracle import train_cpu as TC
from tests.helpers import synth_observation
dev = torch.device("cuda", 0)
B = 9
sd = W.synthetic_state_dict(33)
rng = np.random.default_rng(330)
obs = [synth_observation(rng) for _ in range(B)]
lid = torch.stack([torch.from_numpy(o["lidar"]) for o in obs]).to(dev)
ctx = dict(visual_features=transform_visual(lid, channels_last=True),
           velocity=torch.stack([torch.from_numpy(o["velocity"]) for o in obs]).to(dev),
           is_at_traffic_light=torch.tensor([[float(o["is_at_traffic_light"])] for o in obs], device=dev),
           traffic_light_state=torch.tensor([[float(o["traffic_light_state"])] for o in obs], device=dev))
future = torch.from_numpy(np.cumsum(np.abs(rng.normal(size=(B, 4, 3))) * 2.0, axis=1).astype(np.float32))
y = future[..., :2] + 1e-2 * torch.from_numpy(rng.normal(size=(B, 4, 2)).astype(np.float32))
mask = torch.from_numpy(((rng.random((B, 1280)) >= 0.2) / 0.8).astype(np.float32))
m = ImitativeModel.synthetic(33).to(dev)
tr = DIMTrainer(m, max_batch=16, device=dev)
tr.backward(dict(ctx, player_future=future.to(dev)), y=y, dropout_mask=mask, train=True)
torch.cuda.synchronize()
mo = TC.trainable_model(sd)
bn_out = []
for name, mod in mo.named_modules():
  if isinstance(mod, torch.nn.BatchNorm2d):
    mod.register_forward_hook(lambda md, inp, out, name=name: bn_out.append((name, out.detach().clone())))
cpu = {k: v.cpu() for k, v in ctx.items()}
TC.loss_and_grads(mo, cpu["visual_features"], cpu["velocity"], cpu["is_at_traffic_light"], cpu["traffic_light_state"], y, mask)
lib = _lib.load()
lib.rip_train_debug_layer.restype = ctypes.c_int
lib.rip_train_debug_layer.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
tot = 0
for i, (name, v) in enumerate(bn_out):
  Bc, C, H, Wd = v.shape
  buf = torch.empty(Bc * H * Wd * C, device=dev)
  rc = lib.rip_train_debug_layer(tr._h, i, 1, B, ctypes.c_void_p(buf.data_ptr()), buf.numel())
  assert rc == 0, lib.rip_last_error()
  post = buf.view(Bc, H, Wd, C).permute(0, 3, 1, 2).cpu()
  pre = torch.empty_like(buf)
  lib.rip_train_debug_layer(tr._h, i, 0, B, ctypes.c_void_p(pre.data_ptr()), pre.numel())
  relu = not name.endswith((".conv.2", ".conv.3")) or "features.1.conv.2" == name[-17:]
  vref = v
  is_proj = ("conv.3" in name and not name.endswith("conv.3.1")) or name.endswith("features.1.conv.2")
  # layers followed by ReLU6: every BN except the projection BNs (features.k.conv.3 and features.1.conv.2)
  if is_proj:
    continue
  mg = (post > 0) & (post < 6)
  mc = (vref > 0) & (vref < 6)
  nf = int((mg != mc).sum())
  tot += nf
  dv = float((post - vref.clamp(0, 6)).abs().max())
  if nf:
    idx = (mg != mc).nonzero()[:3]
    print("%-40s flips %d of %d, max|post - oracle| %.2e, oracle BN values at flips: %s" % (
        name, nf, mg.numel(), dv, [float(vref[tuple(j)]) for j in idx]))
print("total flips:", tot)
