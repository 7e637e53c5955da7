"""world_size-2 `gloo` tests (CPU) of the multi-GPU plumbing: sharding, the single all-gather of score blocks with
uneven model shares, row gathering.  The scoring / aggregation kernels themselves are covered by the `-m gpu` tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oatomobile_amd import distributed as D


def test_shard_range_partitions():
  for n in (0, 1, 7, 8, 128, 10000):
    for world in (1, 2, 3, 8):
      spans = [D.shard_range(n, r, world) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
      sizes = [e - b for b, e in spans]
      assert max(sizes) - min(sizes) <= 1
  with pytest.raises(ValueError):
    D.shard_range(4, 2, 2)


def test_single_process_passthrough():
  s = torch.arange(24.0).view(2, 3, 4)
  assert D.all_gather_scores(s, 2) is s
  assert D.gather_rows(s, 2) is s


def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def _worker(rank, world, port, K, B, N, rows, out_dir):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    full = torch.from_numpy(np.random.default_rng(0).normal(size=(K, B, N)).astype(np.float32))
    b, e = D.shard_range(K, rank, world)
    gathered = D.all_gather_scores(full[b:e].clone(), K)
    assert gathered.shape == (K, B, N)
    assert torch.equal(gathered, full), "rank %d: gathered scores differ" % rank
    allrows = torch.arange(rows * 8, dtype=torch.float32).view(rows, 4, 2)
    rb, re = D.shard_range(rows, rank, world)
    got = D.gather_rows(allrows[rb:re].clone(), rows)
    assert torch.equal(got, allrows)
    with pytest.raises(ValueError):
      D.all_gather_scores(full[:K].clone(), K)  # wrong share size
    # gradient-mode model-parallel block exchange: [K_local, B, N, 9] per rank -> [K, B, N, 9] (uneven shares too)
    blk = torch.from_numpy(np.random.default_rng(1).normal(size=(K, B, N, 9)).astype(np.float32))
    assert torch.equal(D.all_gather_blocks(blk[b:e].clone(), K), blk)
    # candidate-parallel winner exchange: lowest loss wins, ties go to the lower rank (= lower global index)
    rec = torch.zeros(B, 10)
    rec[:, 0] = torch.tensor([1.0 if rank == 0 else 0.5] + [2.0] * (B - 1))[:B]  # obs 0: rank 1 wins; others tie
    rec[:, 1:9] = float(rank + 1)
    rec[:, 9] = float(100 * rank + 7)
    allrec = D.gather_rank_winners(rec)
    assert allrec.shape == (world, B, 10)
    plan, idx, best = D.reduce_rank_winners(allrec)
    assert float(plan[0, 0, 0]) == 2.0 and int(idx[0]) == 107 and float(best[0]) == 0.5
    if B > 1:
      assert float(plan[1, 0, 0]) == 1.0 and int(idx[1]) == 7  # tie -> rank 0
    # barrier + max-over-ranks timing pattern used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("K,B,N,rows", [(8, 2, 16, 5), (5, 1, 7, 3)])
def test_two_rank_gloo_exchange(tmp_path, K, B, N, rows):
  world = 2
  mp.spawn(_worker, args=(world, _free_port(), K, B, N, rows, str(tmp_path)), nprocs=world, join=True)
  assert all(os.path.exists(os.path.join(tmp_path, "ok%d" % r)) for r in range(world))


def test_eight_rank_gloo_exchange_config4_layout(tmp_path):
  """BASELINE configs[3]'s literal layout on gloo: world 8, K = 8 (ONE model per rank), N = 512 — the score / block
  all-gathers, the row gather with 5 rows over 8 ranks (three ranks own nothing) and the winner exchange with seven
  ranks tied (the lowest rank wins)."""
  world = 8
  mp.spawn(_worker, args=(world, _free_port(), 8, 2, 512, 5, str(tmp_path)), nprocs=world, join=True)
  assert all(os.path.exists(os.path.join(tmp_path, "ok%d" % r)) for r in range(world))
