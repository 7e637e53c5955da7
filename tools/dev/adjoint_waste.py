"""GPU box experiment: how many inverse-pass adjoints does the phase-sequential search run per 16-candidate block, and
how many would it run if the candidates of a workgroup were regrouped by the model they selected in the previous Adam
step?  (Reads the kernel's own posterior trace.)  python tools/dev/adjoint_waste.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth_batch  # noqa: E402


def count(tp, order_fn, group, use_mask=False):
  """tp [S,K,B,N] posteriors.  Per step: candidates of each `group`-sized chunk are permuted by order_fn(prev ksel),
  split into blocks of 16; an inverse model's adjoint runs for a block iff it is a strict running maximum (models in
  index order, model 0 first) for some candidate of the block."""
  S, K, B, N = tp.shape
  prev = np.zeros((B, N), np.int64)
  prev_key = [np.zeros((B, N), np.int64)]
  total = 0
  hist = np.zeros(K, np.int64)
  for s in range(S):
    q = tp[s]                                   # [K,B,N]
    run = np.maximum.accumulate(q, axis=0)
    take = q[1:] > run[:-1]                     # [K-1,B,N]
    ksel = np.argmax(q, axis=0)                 # final selection (first max)
    for k in range(K):
      hist[k] += int((ksel == k).sum())
    perm = order_fn(prev_key[0] if use_mask else prev, group)                # [B,N] indices
    tk = np.take_along_axis(take, perm[None].repeat(K - 1, 0), axis=2)
    total += int(tk.reshape(K - 1, B, N // 16, 16).any(-1).sum())
    prev = ksel
    prev_key[0] = (take * (1 << np.arange(K - 1))[:, None, None]).sum(0)  # which models were running maxima
  return total, hist


def ident(prev, group):
  B, N = prev.shape
  return np.tile(np.arange(N), (B, 1))


def sort_prev(prev, group):
  B, N = prev.shape
  perm = np.empty((B, N), np.int64)
  for g0 in range(0, N, group):
    perm[:, g0:g0 + group] = g0 + np.argsort(prev[:, g0:g0 + group], axis=1, kind="stable")
  return perm


def main():
  from oatomobile_amd import ImitativeModel, RIPAgent, _lib
  dev = torch.device("cuda", 0)
  K, N, B, S = 4, 128, 512, 10
  models = [ImitativeModel.synthetic(100 + k, max_batch=1) for k in range(K)]
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, max_batch=B, device=dev, encoder_dtype="bf16")
  lib, h = _lib.load(), agent._handle.raw
  lidar, vec, goal = (torch.from_numpy(a).to(dev) for a in synth_batch(np.random.default_rng(1000), B, 2))
  z = torch.empty(K, B, 64, device=dev)
  _lib.check(lib.rip_encode_raw(h, _lib.ptr(lidar), 1, 200, 200, _lib.ptr(vec), B, 0, K, 1, _lib.ptr(z), _lib.current_stream(dev)))
  tp = torch.empty(S, K, B, N, device=dev)
  loss = torch.empty(B, N, device=dev)
  _lib.check(lib.rip_search(h, _lib.ptr(z), _lib.ptr(goal), _lib.ptr(agent._x0(B)), B, N, 10, 0, S, 0.1, 1.0, None, None,
                            _lib.ptr(loss), None, _lib.ptr(tp), None, None, _lib.current_stream(dev)))
  tp = tp.cpu().numpy().astype(np.float64)
  blocks = S * B * N // 16
  base, hist = count(tp, ident, 64)
  print("selection histogram over (step, candidate):", (hist / hist.sum()).round(3))
  ks = np.argmax(tp, axis=1)
  print("selection unchanged from one Adam step to the next: %.3f" % float((ks[1:] == ks[:-1]).mean()))
  print("as laid out:            %d inverse adjoints = %.2f per block-step" % (base, base / blocks))
  for g in (64, 128):
    t, _ = count(tp, sort_prev, g)
    print("sorted within %3d cand.: %d = %.2f per block-step" % (g, t, t / blocks))
  for g in (64, 128):
    t, _ = count(tp, sort_prev, g, use_mask=True)
    print("sorted by the previous step's record mask within %3d: %d = %.2f per block-step" % (g, t, t / blocks))
  # bound: blocks sorted by the CURRENT step's selection, adjoint only for models some candidate finally selects
  tot = 0
  for s in range(S):
    k = np.sort(np.argmax(tp[s], axis=0), axis=1).reshape(B, N // 16, 16)
    for m in range(1, K):
      tot += int((k == m).any(-1).sum())
  print("oracle bound (sorted by the final selection, no intermediate records): %.2f per block-step" % (tot / blocks))


def wg_phases(tp, group=64, global_sort=False, reorder=False):
  """Adjoint PHASES a 64-candidate workgroup walks through per Adam step (a phase is paid by the whole workgroup as
  soon as one of its candidates needs that model's adjoint).  global_sort: the launch's candidates are binned by the
  model they selected in the previous step (any observation), workgroups take 64 consecutive ones; reorder: a workgroup
  evaluates its majority's predicted model first."""
  S, K, B, N = tp.shape
  q_all = tp.reshape(S, K, B * N)
  prev = np.zeros(B * N, np.int64)
  phases = 0
  for s in range(S):
    q = q_all[s]
    perm = np.argsort(prev, kind="stable") if global_sort else np.arange(B * N)
    qs = q[:, perm].reshape(K, -1, group)            # [K, WGs, 64]
    pv = prev[perm].reshape(-1, group)
    for w in range(qs.shape[1]):
      order = list(range(1, K))
      if reorder:
        p = np.bincount(pv[w], minlength=K).argmax()
        if p != 0:
          order = [p] + [k for k in range(1, K) if k != p]
      best = qs[0, w].copy()
      for k in order:
        take = qs[k, w] > best
        if take.any():
          phases += 1
        best = np.maximum(best, qs[k, w])
    prev = np.argmax(q, axis=0)
  return phases / (S * (B * N // group))


def main2():
  from oatomobile_amd import ImitativeModel, RIPAgent, _lib
  dev = torch.device("cuda", 0)
  K, N, B, S = 4, 128, 512, 10
  models = [ImitativeModel.synthetic(100 + k, max_batch=1) for k in range(K)]
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, max_batch=B, device=dev, encoder_dtype="bf16")
  lib, h = _lib.load(), agent._handle.raw
  lidar, vec, goal = (torch.from_numpy(a).to(dev) for a in synth_batch(np.random.default_rng(1000), B, 2))
  z = torch.empty(K, B, 64, device=dev)
  _lib.check(lib.rip_encode_raw(h, _lib.ptr(lidar), 1, 200, 200, _lib.ptr(vec), B, 0, K, 1, _lib.ptr(z), _lib.current_stream(dev)))
  tp = torch.empty(S, K, B, N, device=dev)
  loss = torch.empty(B, N, device=dev)
  _lib.check(lib.rip_search(h, _lib.ptr(z), _lib.ptr(goal), _lib.ptr(agent._x0(B)), B, N, 10, 0, S, 0.1, 1.0, None, None,
                            _lib.ptr(loss), None, _lib.ptr(tp), None, None, _lib.current_stream(dev)))
  tp = tp.cpu().numpy().astype(np.float64)
  print("adjoint phases per 64-candidate workgroup and Adam step (of %d):" % (K - 1))
  print("  as laid out                                  %.2f" % wg_phases(tp))
  print("  globally binned by the previous selection    %.2f" % wg_phases(tp, global_sort=True))
  print("  + predicted model evaluated first            %.2f" % wg_phases(tp, global_sort=True, reorder=True))


if __name__ == "__main__" and os.environ.get("RIP_WASTE_WG") == "1":
  main2()

if __name__ == "__main__" and os.environ.get("RIP_WASTE_WG") != "1":
  main()
