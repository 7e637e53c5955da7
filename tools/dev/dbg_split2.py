import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from oatomobile_amd import _lib, RIPAgent, ImitativeModel, weights as W
dev = torch.device("cuda", 0)
K, N, S, algo = 3, 128, 24, "WCM"
hips = [ImitativeModel().load_numpy_state_dict(W.synthetic_state_dict(300 + k)).to(dev) for k in range(K)]
lib = _lib.load()
agent = RIPAgent(None, algorithm=algo, models=hips, num_candidates=N, seed=9, search_kernel="chain", max_batch=S)
rng = np.random.default_rng(11)
z_np = (np.abs(rng.normal(size=(K, S, 64))) * 1e5).astype(np.float32); z_np[:, :, ::7] = 0
goal_np = (np.cumsum(np.abs(rng.normal(size=(S, 10, 2))) * 2.0, axis=1) + 100.0).astype(np.float32)
x_np = rng.normal(size=(S, N, 4, 2)).astype(np.float32)
z, goal, x = (torch.from_numpy(a).to(dev) for a in (z_np, goal_np, x_np))
lb = torch.empty(S, N, device=dev); tp = torch.empty(1, K, S, N, device=dev); tg = torch.empty(1, S, N, 4, 2, device=dev)
h = agent._handle
_lib.check(lib.rip_search(h.raw, _lib.ptr(z), _lib.ptr(goal), _lib.ptr(x), S, N, 10, _lib.ALGORITHMS[algo], 1, 0.1, 1.0, None, None, _lib.ptr(lb), None, _lib.ptr(tp), None, _lib.ptr(tg), h.stream()))
p = tp.cpu().numpy()[0]
bad = np.argwhere(~np.isfinite(p[0, 0]))[:, 0]
print("bad candidates of observation 0:", bad[:10].tolist(), "values", p[:, 0, bad[0]].tolist())
n = int(bad[0])
y, lad = hips[0]._forward(x[0], z[0, 0:1])
print("y[n] =", y[n].cpu().numpy().tolist(), "lad", float(lad[n]))
rows = hips[0]._goal_likelihood_rows(y, goal[0:1])
print("goal rows finite:", bool(torch.isfinite(rows).all()), "goal row n:", float(rows[n]), "nonfinite count", int((~torch.isfinite(rows)).sum()))
for k in range(K):
  xi, lp, l2 = hips[k]._inverse(y, z[k, 0:1])
  print("model", k, "logp", float(lp[n]), "lad", float(l2[n]), "x", xi[n].abs().max().item())
