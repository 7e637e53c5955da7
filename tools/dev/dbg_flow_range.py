"""GPU box (dev): which flow quantity leaves fp32 at huge z — forward / inverse entry points against the oracle."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from oatomobile_amd import ImitativeModel, weights as W
from oracle import reference_cpu as O
dev = torch.device("cuda", 0)
for zsc, wsc in ((1e3, 10.0), (1e5, 1.0), (1e4, 1.0)):
  sd = W.synthetic_state_dict(300)
  for key in sd:
    if key.startswith("_decoder.") and key.endswith(("weight_ih", "weight_hh", "0.weight", "2.weight")):
      sd[key] = (sd[key] * np.float32(wsc)).astype(np.float32)
  m = ImitativeModel().load_numpy_state_dict(sd).to(dev)
  mo = O.OracleImitativeModel.from_numpy_state_dict(sd)
  rng = np.random.default_rng(11)
  N = 256
  z = (np.abs(rng.normal(size=(1, 64))) * zsc).astype(np.float32); z[:, ::7] = 0
  x = rng.normal(size=(N, 4, 2)).astype(np.float32)
  zt, xt = torch.from_numpy(z).to(dev), torch.from_numpy(x).to(dev)
  y, lad = m._forward(xt, zt)
  yo, lado = O.flow_forward(mo, torch.from_numpy(x), torch.from_numpy(z).repeat(N, 1))
  print("z x %g, w x %g: forward y finite %s (oracle %s) max|y| %.3g, lad finite %s (oracle %s)" %
        (zsc, wsc, bool(torch.isfinite(y).all()), bool(torch.isfinite(yo).all()), float(yo.abs().max()), bool(torch.isfinite(lad).all()), bool(torch.isfinite(lado).all())))
  bad = ~torch.isfinite(y.cpu())
  if bad.any():
    i = bad.nonzero()[0].tolist(); print("   first bad y at", i, "hip", y.cpu()[i[0]].tolist(), "oracle", yo[i[0]].tolist())
  ygood = yo.to(dev)
  xi, lp, lad2 = m._inverse(ygood, zt)
  xo, lpo, lad2o = O.flow_inverse(mo, yo, torch.from_numpy(z).repeat(N, 1))
  print("   inverse x finite %s (oracle %s), logp finite %s (oracle %s), lad finite %s (oracle %s)" %
        (bool(torch.isfinite(xi).all()), bool(torch.isfinite(xo).all()), bool(torch.isfinite(lp).all()), bool(torch.isfinite(lpo).all()), bool(torch.isfinite(lad2).all()), bool(torch.isfinite(lad2o).all())))
  for name, a, b in (("x", xi.cpu(), xo), ("logp", lp.cpu(), lpo), ("lad", lad2.cpu(), lad2o)):
    badm = ~torch.isfinite(a)
    if badm.any():
      i = badm.nonzero()[0].tolist(); print("   first bad", name, i, "hip", a[i[0]].tolist() if a.dim() > 1 else float(a[i[0]]), "oracle", b[i[0]].tolist() if b.dim() > 1 else float(b[i[0]]))
