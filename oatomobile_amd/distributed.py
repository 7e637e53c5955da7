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
  all-gather per call, never one per model.
"""

from typing import List, Optional, Sequence, Tuple

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


def all_gather_scores(local_scores: torch.Tensor, num_models: int, group=None) -> torch.Tensor:
  """Model-parallel exchange: `local_scores [K_local, B, N]` (this rank's models, in `shard_range` order) ->
  `[K, B, N]` on every rank.  Ranks may own different numbers of models (K % world != 0): blocks are padded to
  the largest share for ONE fixed-size all-gather, then trimmed."""
  rank, world = _world(group)
  if world == 1:
    return local_scores
  shares = [shard_range(num_models, r, world) for r in range(world)]
  kmax = max(e - b for b, e in shares)
  k_local, B, N = local_scores.shape
  if k_local != shares[rank][1] - shares[rank][0]:
    raise ValueError("rank %d should own %d models, got %d" % (rank, shares[rank][1] - shares[rank][0], k_local))
  block = local_scores.new_zeros((kmax, B, N))
  block[:k_local] = local_scores
  flat = local_scores.new_empty((world * kmax, B, N))  # concatenated along dim 0 (the layout gloo and RCCL share)
  dist.all_gather_into_tensor(flat, block.contiguous(), group=group)
  out = flat.view(world, kmax, B, N)
  return torch.cat([out[r, :e - b] for r, (b, e) in enumerate(shares)], dim=0).contiguous()


def gather_rows(local_rows: torch.Tensor, total_rows: int, group=None) -> torch.Tensor:
  """Observation-/candidate-parallel epilogue: concatenates per-rank row blocks (`shard_range(total_rows, r, world)`
  order) of a `[rows_local, ...]` tensor on every rank."""
  rank, world = _world(group)
  if world == 1:
    return local_rows
  shares = [shard_range(total_rows, r, world) for r in range(world)]
  rmax = max(e - b for b, e in shares)
  block = local_rows.new_zeros((rmax,) + tuple(local_rows.shape[1:]))
  block[:local_rows.shape[0]] = local_rows
  flat = local_rows.new_empty((world * rmax,) + tuple(local_rows.shape[1:]))
  dist.all_gather_into_tensor(flat, block.contiguous(), group=group)
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
    st = _lib.current_stream()
    _lib.check(lib.rip_encode_raw(self._handle.raw, _lib.ptr(lidar), 1, _lib.ptr(vec), B, 0, kl, 0, _lib.ptr(z), st))
    _lib.check(lib.rip_score(self._handle.raw, 0, kl, _lib.ptr(z), _lib.ptr(plans.contiguous()), _lib.ptr(goal), B, N,
                             goal.shape[1], self._epsilon, _lib.ptr(S), st))
    return S

  def aggregate(self, scores: torch.Tensor):
    """[K,B,N] -> (loss [B,N], best [B] int32) with the HIP aggregation kernel."""
    lib, _lib = self._lib.load(), self._lib
    K, B, N = scores.shape
    loss = torch.empty(B, N, device=scores.device)
    best = torch.empty(B, device=scores.device, dtype=torch.int32)
    _lib.check(lib.rip_aggregate_scores(_lib.ptr(scores.contiguous()), K, B, N, _lib.ALGORITHMS[self._algorithm],
                                        _lib.ptr(loss), _lib.ptr(best), _lib.current_stream()))
    return loss, best

  def __call__(self, lidar, vec, goal, plans):
    """Returns (best plan [B,4,2], best index [B], loss [B,N]) — identical on every rank."""
    S = all_gather_scores(self.local_scores(lidar, vec, goal, plans), self._k_total, self._group)
    loss, best = self.aggregate(S)
    idx = best.long().view(-1, 1, 1, 1).expand(-1, 1, 4, 2)
    return torch.gather(plans, 1, idx)[:, 0], best, loss
