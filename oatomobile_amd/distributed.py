"""Multi-GPU layouts of the RIP path: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code.  The path shards three ways:

* observation-parallel (replay, BASELINE config 5) and candidate-parallel (config 3 scaling): ranks own disjoint
  observations / candidate rows, every rank holds all K models (38 MB of weights) — **no data-path collective**;
  only the final plans are gathered (`gather_rows`).
* model-parallel (config 4: K=8 over 8 GPUs): rank r owns models `shard_range(K, r, world)`, encodes the same
  observations and scores the same N plans with its models; ONE all-gather of the `[K_local, B, N]` fp32 score
  block (16 KiB at K=8, N=512) gives every rank the full `[K, B, N]` matrix, which it aggregates redundantly
  (`rip_aggregate_scores`) — no second collective.  Messages are KiB-sized, i.e. latency-bound: one fused
  all-gather per call, never one per model.  `ModelParallelScorer` is the scoring mode, `ModelParallelRIP` the
  gradient mode (the reference's 10 Adam steps): per step ONE all-gather of `(q_k, dq_k/dy)` `[K_local, B, N, 9]`.
* candidate-parallel (`CandidateParallelRIP`): every rank holds all K models, searches `N / world` of the latent
  starts locally with no per-step collective, and ONE all-gather of `(best loss, plan)` per observation picks the
  winner — the near-linear mode for candidate throughput.
"""

import os
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
  """Contiguous, balanced [begin, end) of `n` items for `rank` (the first n % world ranks get one extra)."""
  if world < 1 or not 0 <= rank < world:
    raise ValueError("bad rank/world %d/%d" % (rank, world))
  base, extra = divmod(n, world)
  begin = rank * base + min(rank, extra)
  return begin, begin + base + (1 if rank < extra else 0)


def _world(group=None) -> Tuple[int, int]:
  if not dist.is_available() or not dist.is_initialized():
    return 0, 1
  return dist.get_rank(group), dist.get_world_size(group)


def _skip_collective(world: int) -> bool:
  """A one-rank world needs no exchange and normally takes none.  `RIP_DIST_ALWAYS_COLLECTIVE=1` sends a one-rank
  process group through the collectives all the same, so that the RCCL code path (`backend="nccl"`, device tensors)
  can be executed and checked on a one-GPU box (tests/test_gpu_parity.py::test_rccl_world1_*, `bench.py --gpus 1
  --mode candidates`)."""
  if world > 1:
    return False
  return not (os.environ.get("RIP_DIST_ALWAYS_COLLECTIVE") == "1" and dist.is_available() and dist.is_initialized())



def _all_gather_into(flat: torch.Tensor, block: torch.Tensor, group=None) -> None:
  """`dist.all_gather_into_tensor`; on the gloo backend device tensors are staged through the host (gloo has no
  device all-gather) — that combination only occurs in the one-GPU tests of the multi-rank paths, RCCL takes the
  device tensors as they are."""
  from oatomobile_amd import _lib
  with _lib.trace_range("rip all_gather (%d B)" % (flat.numel() * flat.element_size())):
    if block.is_cuda and dist.get_backend(group) == "gloo":
      host = torch.empty(flat.shape, dtype=flat.dtype)
      dist.all_gather_into_tensor(host, block.detach().cpu().contiguous(), group=group)
      flat.copy_(host)
      return
    dist.all_gather_into_tensor(flat, block.contiguous(), group=group)

def all_gather_scores(local_scores: torch.Tensor, num_models: int, group=None) -> torch.Tensor:
  """Model-parallel exchange: `local_scores [K_local, B, N]` (this rank's models, in `shard_range` order) ->
  `[K, B, N]` on every rank.  Ranks may own different numbers of models (K % world != 0): blocks are padded to
  the largest share for ONE fixed-size all-gather, then trimmed."""
  rank, world = _world(group)
  if _skip_collective(world):
    return local_scores
  shares = [shard_range(num_models, r, world) for r in range(world)]
  kmax = max(e - b for b, e in shares)
  k_local, B, N = local_scores.shape
  if k_local != shares[rank][1] - shares[rank][0]:
    raise ValueError("rank %d should own %d models, got %d" % (rank, shares[rank][1] - shares[rank][0], k_local))
  block = local_scores.new_zeros((kmax, B, N))
  block[:k_local] = local_scores
  flat = local_scores.new_empty((world * kmax, B, N))  # concatenated along dim 0 (the layout gloo and RCCL share)
  _all_gather_into(flat, block, group)
  out = flat.view(world, kmax, B, N)
  return torch.cat([out[r, :e - b] for r, (b, e) in enumerate(shares)], dim=0).contiguous()


def gather_rows(local_rows: torch.Tensor, total_rows: int, group=None) -> torch.Tensor:
  """Observation-/candidate-parallel epilogue: concatenates per-rank row blocks (`shard_range(total_rows, r, world)`
  order) of a `[rows_local, ...]` tensor on every rank."""
  rank, world = _world(group)
  if _skip_collective(world):
    return local_rows
  shares = [shard_range(total_rows, r, world) for r in range(world)]
  rmax = max(e - b for b, e in shares)
  block = local_rows.new_zeros((rmax,) + tuple(local_rows.shape[1:]))
  block[:local_rows.shape[0]] = local_rows
  flat = local_rows.new_empty((world * rmax,) + tuple(local_rows.shape[1:]))
  _all_gather_into(flat, block, group)
  out = flat.view((world, rmax) + tuple(local_rows.shape[1:]))
  return torch.cat([out[r, :e - b] for r, (b, e) in enumerate(shares)], dim=0).contiguous()


class ModelParallelScorer:
  """BASELINE config 4: every rank scores the same plans with ITS slice of the K models on its GPU
  (rip_encode_raw + rip_score), one all-gather, redundant aggregation + arg-best (rip_aggregate_scores)."""

  def __init__(self, models: Sequence, num_models_total: int, algorithm: str = "WCM", max_batch: int = 1,
               epsilon: float = 1.0, device: Optional[torch.device] = None, group=None) -> None:
    from oatomobile_amd import _lib
    assert algorithm in ("WCM", "MA", "BCM")
    self._lib = _lib
    self._group = group
    self._algorithm, self._epsilon = algorithm, float(epsilon)
    self._k_total = int(num_models_total)
    rank, world = _world(group)
    b, e = shard_range(self._k_total, rank, world)
    if len(models) != e - b:
      raise ValueError("rank %d/%d owns models [%d,%d) but got %d" % (rank, world, b, e, len(models)))
    self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    self._models = [m.to(self._device) for m in models]
    self._handle = _lib.Handle(len(models), self._models[0]._in_channels, max_batch,
                               self._device.index if self._device.index is not None else torch.cuda.current_device())
    for k, m in enumerate(self._models):
      self._handle.load_model(k, m.packed_weights())

  def local_scores(self, lidar: torch.Tensor, vec: torch.Tensor, goal: torch.Tensor, plans: torch.Tensor) -> torch.Tensor:
    """lidar [B,200,200,C], vec [B,5], goal [B,G,2], plans [B,N,4,2] -> S_local [K_local,B,N]."""
    lib, _lib = self._lib.load(), self._lib
    kl, B, N = len(self._models), lidar.shape[0], plans.shape[1]
    z = torch.empty(kl, B, 64, device=self._device)
    S = torch.empty(kl, B, N, device=self._device)
    st = self._handle.stream()
    _lib.check(lib.rip_encode_raw(self._handle.raw, _lib.ptr(lidar), 1, lidar.shape[1], lidar.shape[2], _lib.ptr(vec), B,
                                  0, kl, 0, _lib.ptr(z), st))
    _lib.check(lib.rip_score(self._handle.raw, 0, kl, _lib.ptr(z), _lib.ptr(plans.contiguous()), _lib.ptr(goal), B, N,
                             goal.shape[1], self._epsilon, _lib.ptr(S), st))
    return S

  def aggregate(self, scores: torch.Tensor):
    """[K,B,N] -> (loss [B,N], best [B] int32) with the HIP aggregation kernel."""
    lib, _lib = self._lib.load(), self._lib
    K, B, N = scores.shape
    loss = torch.empty(B, N, device=scores.device)
    best = torch.empty(B, device=scores.device, dtype=torch.int32)
    with torch.cuda.device(scores.device):  # stateless entry point: launches on the current device
      _lib.check(lib.rip_aggregate_scores(_lib.ptr(scores.contiguous()), K, B, N, _lib.ALGORITHMS[self._algorithm],
                                          _lib.ptr(loss), _lib.ptr(best, torch.int32), _lib.current_stream(scores.device)))
    return loss, best

  def __call__(self, lidar, vec, goal, plans):
    """Returns (best plan [B,4,2], best index [B], loss [B,N]) — identical on every rank."""
    S = all_gather_scores(self.local_scores(lidar, vec, goal, plans), self._k_total, self._group)
    loss, best = self.aggregate(S)
    idx = best.long().view(-1, 1, 1, 1).expand(-1, 1, 4, 2)
    return torch.gather(plans, 1, idx)[:, 0], best, loss


def all_gather_blocks(local: torch.Tensor, num_models: int, group=None) -> torch.Tensor:
  """`[K_local, ...]` per-model blocks in `shard_range` order -> `[K, ...]` on every rank, ONE fixed-size all-gather
  (uneven shares are padded to the largest one and trimmed)."""
  rank, world = _world(group)
  if _skip_collective(world):
    return local
  shares = [shard_range(num_models, r, world) for r in range(world)]
  kmax = max(e - b for b, e in shares)
  if local.shape[0] != shares[rank][1] - shares[rank][0]:
    raise ValueError("rank %d should own %d models, got %d" % (rank, shares[rank][1] - shares[rank][0], local.shape[0]))
  block = local.new_zeros((kmax,) + tuple(local.shape[1:]))
  block[:local.shape[0]] = local
  flat = local.new_empty((world * kmax,) + tuple(local.shape[1:]))
  _all_gather_into(flat, block, group)
  out = flat.view((world, kmax) + tuple(local.shape[1:]))
  return torch.cat([out[r, :e - b] for r, (b, e) in enumerate(shares)], dim=0).contiguous()


def select_best_plan(loss_best: torch.Tensor, plans: torch.Tensor):
  """`loss_best [B, N]`, `plans [B, N, 4, 2]` -> (plan [B,4,2], index [B]) of the lowest best-loss (first on ties),
  the arg-min `rip_search` applies on one GPU."""
  idx = torch.argmin(loss_best, dim=1)
  return plans[torch.arange(plans.shape[0], device=plans.device), idx], idx


class CandidateParallelRIP:
  """Candidate-parallel RIP search: rank r searches the latent starts
  `shard_range(N, r, world)` of every observation with ALL K models (`rip_encode_raw` + `rip_search`, no per-step
  collective: candidates are independent under per-candidate aggregation, rip/agent.py:121-135), then ONE
  all-gather of the per-rank winner `(best loss, plan)` `[B, 9]` and an arg-min over ranks (rip/agent.py:133-137).
  Identical to one GPU searching all N candidates (ties resolve to the lowest global index on both)."""

  def __init__(self, models: Sequence, num_candidates: int, algorithm: str = "WCM", num_steps: int = 10,
               lr: float = 1e-1, epsilon: float = 1.0, seed: int = 0, max_batch: int = 1,
               device: Optional[torch.device] = None, encoder_dtype: str = "fp32", search_kernel: str = "auto",
               group=None, rank: Optional[int] = None, world: Optional[int] = None) -> None:
    from oatomobile_amd import _lib, arch
    import numpy as np
    assert algorithm in ("WCM", "MA", "BCM")
    self._lib, self._group = _lib, group
    r, w = _world(group)
    self._rank = r if rank is None else int(rank)     # explicit rank/world: a rank's share without a process group
    self._world = w if world is None else int(world)  # (one-GPU "halves == whole" tests)
    self._explicit = world is not None
    self._algorithm, self._num_steps, self._lr, self._epsilon = algorithm, int(num_steps), float(lr), float(epsilon)
    self._n_total = int(num_candidates)
    self._begin, self._end = shard_range(self._n_total, self._rank, self._world)
    if self._end <= self._begin:
      raise ValueError("rank %d/%d owns no candidates (N=%d)" % (self._rank, self._world, self._n_total))
    self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if self._device.index is None:
      self._device = torch.device("cuda", torch.cuda.current_device())
    self._models = [m.to(self._device) for m in models]
    self._max_batch = int(max_batch)
    self._enc_dtype = _lib.ENC_DTYPES[encoder_dtype]
    n_local = self._end - self._begin
    self._handle = _lib.Handle(len(self._models), self._models[0]._in_channels, self._max_batch, self._device.index,
                               max_candidates=n_local)
    for k, m in enumerate(self._models):
      self._handle.load_model(k, m.packed_weights())
    self._handle.set_option(_lib.OPT_SEARCH_KERNEL, _lib.search_kernel_id(search_kernel))
    # the SAME N latent starts as RIPAgent(seed=...) draws on one GPU; this rank keeps its rows
    x0 = np.random.default_rng(seed).standard_normal((self._n_total, arch.T, 2)).astype(np.float32)
    x0[0] = 0.0
    self._x0_rows = torch.from_numpy(x0[self._begin:self._end].copy()).to(self._device)

  def local_search(self, lidar: torch.Tensor, vec: torch.Tensor, goal: torch.Tensor):
    """-> (loss_best [B, N_local], plans [B, N_local, 4, 2]) of this rank's candidates."""
    lib, _lib = self._lib.load(), self._lib
    B, nl = lidar.shape[0], self._end - self._begin
    z = torch.empty(len(self._models), B, 64, device=self._device)
    plans = torch.empty(B, nl, 4, 2, device=self._device)
    loss = torch.empty(B, nl, device=self._device)
    x0 = self._x0_rows.unsqueeze(0).expand(B, -1, -1, -1).contiguous()
    st = self._handle.stream()
    _lib.check(lib.rip_encode_raw(self._handle.raw, _lib.ptr(lidar.contiguous()), 1, lidar.shape[1], lidar.shape[2],
                                  _lib.ptr(vec.contiguous()), B, 0, len(self._models), self._enc_dtype, _lib.ptr(z), st))
    _lib.check(lib.rip_search(self._handle.raw, _lib.ptr(z), _lib.ptr(goal.contiguous()), _lib.ptr(x0), B, nl,
                              goal.shape[1], _lib.ALGORITHMS[self._algorithm], self._num_steps, self._lr, self._epsilon,
                              None, _lib.ptr(plans), _lib.ptr(loss), None, None, None, None, st))
    return loss, plans

  def __call__(self, lidar: torch.Tensor, vec: torch.Tensor, goal: torch.Tensor):
    """Returns (plan [B,4,2], global index of the winning candidate [B], its best loss [B]) — same on every rank."""
    loss, plans = self.local_search(lidar, vec, goal)
    plan_l, idx_l = select_best_plan(loss, plans)
    B = loss.shape[0]
    rec = torch.cat([loss.gather(1, idx_l[:, None]), plan_l.reshape(B, 8), (idx_l + self._begin).float()[:, None]], dim=1)
    local_only = self._explicit or _skip_collective(self._world)
    return reduce_rank_winners(rec[None] if local_only else gather_rank_winners(rec, self._group))


def gather_rank_winners(rec: torch.Tensor, group=None) -> torch.Tensor:
  """The candidate-parallel exchange: every rank's `[B, 10]` winner records -> `[world, B, 10]` on every rank (ONE
  all-gather, 40 bytes per observation and rank)."""
  rank, world = _world(group)
  if _skip_collective(world):
    return rec[None]
  allrec = rec.new_empty((world,) + tuple(rec.shape))
  _all_gather_into(allrec.view(world * rec.shape[0], rec.shape[1]), rec, group)
  return allrec


def reduce_rank_winners(allrec: torch.Tensor):
  """`[world, B, 10]` records (best loss, plan[8], global candidate index) -> (plan [B,4,2], index [B], loss [B]):
  lowest loss wins, ties go to the lower rank = the lower global index (ranks own ascending index ranges)."""
  win = torch.argmin(allrec[:, :, 0], dim=0)  # first minimum = lowest rank
  B = allrec.shape[1]
  pick = allrec[win, torch.arange(B, device=allrec.device)]
  return pick[:, 1:9].reshape(B, 4, 2).contiguous(), pick[:, 9].long(), pick[:, 0].contiguous()


class ModelParallelRIP:
  """Gradient-mode model-parallel RIP search (BASELINE configs[3]: K = 8 models on 8 GPUs).

  Rank r owns the models `shard_range(K, r, world)` (encoder + flow); every rank additionally holds the small flow
  decoder of the ensemble's model 0, through which the latent is pushed (rip/agent.py:106,137).  Per act:
    1. local encoders -> z_k; ONE all-gather of `[K_local, B, 64]` gives every rank z_0;
    2. per Adam step: `rip_mp_local` (y = F_0(x), inverse_k + adjoint for the local models) -> ONE all-gather of the
       `(q_k, dq_k/dy)` block `[K_local, B, N, 9]` (147 KiB in total at K=8, N=512: latency-bound) ->
       `rip_mp_update` on every rank redundantly (aggregation over K, F_0 adjoint, Adam, loss/x_best bookkeeping on
       rank-replicated state — bitwise the same on every rank, so no second collective);
    3. plan = F_0(x_best) of the lowest best-loss candidate.
  One rank with all K models reproduces `rip_search` (wave-per-chain kernel)."""

  def __init__(self, models: Sequence, num_models_total: int, flow0=None, num_candidates: int = 1, algorithm: str = "WCM",
               num_steps: int = 10, lr: float = 1e-1, epsilon: float = 1.0, seed: int = 0, max_batch: int = 1,
               device: Optional[torch.device] = None, group=None, rank: Optional[int] = None,
               world: Optional[int] = None) -> None:
    from oatomobile_amd import _lib, arch
    import numpy as np
    assert algorithm in ("WCM", "MA", "BCM")
    self._lib, self._group = _lib, group
    r, w = _world(group)
    self._rank = r if rank is None else int(rank)
    self._world = w if world is None else int(world)
    self._explicit = world is not None
    self._k_total = int(num_models_total)
    self._kb, self._ke = shard_range(self._k_total, self._rank, self._world)
    if len(models) != self._ke - self._kb:
      raise ValueError("rank %d/%d owns models [%d,%d) but got %d" % (self._rank, self._world, self._kb, self._ke, len(models)))
    self._owns0 = self._kb == 0
    if not self._owns0 and flow0 is None:
      raise ValueError("ranks that do not own model 0 need `flow0` (an ImitativeModel carrying model 0's flow weights)")
    self._algorithm, self._num_steps, self._lr, self._epsilon = algorithm, int(num_steps), float(lr), float(epsilon)
    self._n = int(num_candidates)
    self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if self._device.index is None:
      self._device = torch.device("cuda", torch.cuda.current_device())
    self._models = [m.to(self._device) for m in models]
    # handle layout: local models first; model 0's flow is slot 0 on its owner, an extra slot elsewhere
    held = list(self._models) + ([] if self._owns0 else [flow0.to(self._device)])
    self._k_fwd = 0 if self._owns0 else len(self._models)
    self._handle = _lib.Handle(len(held), self._models[0]._in_channels, int(max_batch), self._device.index)
    for k, m in enumerate(held):
      self._handle.load_model(k, m.packed_weights())
    x0 = np.random.default_rng(seed).standard_normal((self._n, arch.T, 2)).astype(np.float32)
    x0[0] = 0.0
    self._x0_rows = torch.from_numpy(x0).to(self._device)

  def encode_local(self, lidar: torch.Tensor, vec: torch.Tensor) -> torch.Tensor:
    lib, _lib = self._lib.load(), self._lib
    kl, B = len(self._models), lidar.shape[0]
    z = torch.empty(kl, B, 64, device=self._device)
    _lib.check(lib.rip_encode_raw(self._handle.raw, _lib.ptr(lidar.contiguous()), 1, lidar.shape[1], lidar.shape[2],
                                  _lib.ptr(vec.contiguous()), B, 0, kl, 0, _lib.ptr(z), self._handle.stream()))
    return z

  def local_block(self, z_local: torch.Tensor, z0: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """One step's `(q_k, dq_k/dy)` of this rank's models: [K_local, B, N, 9]."""
    lib, _lib = self._lib.load(), self._lib
    kl, B, N = len(self._models), x.shape[0], x.shape[1]
    out = torch.empty(kl, B, N, 9, device=self._device)
    _lib.check(lib.rip_mp_local(self._handle.raw, self._k_fwd, 0, kl, int(self._owns0), _lib.ptr(z0), _lib.ptr(z_local),
                                _lib.ptr(x), B, N, _lib.ptr(out), self._handle.stream()))
    return out

  def update(self, gathered: torch.Tensor, z0, goal, step: int, state) -> None:
    lib, _lib = self._lib.load(), self._lib
    x, m, v, xb, lb = state
    B, N = x.shape[0], x.shape[1]
    _lib.check(lib.rip_mp_update(self._handle.raw, self._k_fwd, _lib.ptr(z0), _lib.ptr(gathered), self._k_total,
                                 _lib.ptr(goal), B, N, goal.shape[1], _lib.ALGORITHMS[self._algorithm], int(step),
                                 self._lr, self._epsilon, _lib.ptr(x), _lib.ptr(m), _lib.ptr(v), _lib.ptr(xb),
                                 _lib.ptr(lb), None, self._handle.stream()))

  def __call__(self, lidar: torch.Tensor, vec: torch.Tensor, goal: torch.Tensor, exchange=None):
    """Returns (plan [B,4,2], best candidate index [B], loss_best [B,N]) — identical on every rank.
    `exchange(block)` replaces the all-gather (tests: concatenate the blocks of emulated ranks)."""
    if exchange is not None:
      gather = exchange
    elif self._explicit or _skip_collective(self._world):  # one rank holds every model (also when constructed
      gather = lambda t: t                                   # with world=1 inside a larger job)
    else:
      gather = lambda t: all_gather_blocks(t, self._k_total, self._group)
    goal = goal.contiguous()
    B = lidar.shape[0]
    z_local = self.encode_local(lidar, vec)
    z0 = gather(z_local)[0].contiguous()  # [B, 64] of model 0 (one-off, 2 KiB at K = 8)
    x = self._x0_rows.unsqueeze(0).expand(B, -1, -1, -1).contiguous()
    state = (x, torch.zeros_like(x), torch.zeros_like(x), x.clone(), torch.full((B, self._n), 1000.0, device=self._device))
    for step in range(self._num_steps):
      block = self.local_block(z_local, z0, state[0])
      self.update(gather(block).contiguous(), z0, goal, step, state)
    return self.finish(z0, state)

  def finish(self, z0: torch.Tensor, state):
    """plan = F_0(x_best) of the arg-min candidate (rip/agent.py:133-137)."""
    lib, _lib = self._lib.load(), self._lib
    xb, lb = state[3], state[4]
    B = xb.shape[0]
    best = torch.argmin(lb, dim=1)
    xsel = xb[torch.arange(B, device=xb.device), best].contiguous()  # [B,4,2]
    plan = torch.empty_like(xsel)
    _lib.check(lib.rip_flow_forward(self._handle.raw, self._k_fwd, _lib.ptr(xsel), _lib.ptr(z0), B, B, _lib.ptr(plan),
                                    None, self._handle.stream()))
    return plan, best, lb
