"""Two-process data-parallel DIM training step (run by tests/test_gpu_parity.py::test_data_parallel_training_two_ranks
under torch.distributed.run; one-GPU hook: both ranks on cuda:0, gloo).  Each rank back-propagates ITS half of a batch,
`DIMTrainer.apply(clip=True)` averages the packed gradient vector over the ranks, clips the AVERAGE to norm 1
(train.py:206-208 under DistributedDataParallel) and steps Adam.  Checks, on every rank:
  * the averaged gradients equal the mean of the two ranks' local gradients (gathered and compared),
  * trainable parameters and Adam moments are identical on both ranks after the step (the BatchNorm running
    statistics are per-rank buffers computed from the local half batch, as under DistributedDataParallel without
    SyncBatchNorm).
Prints one JSON line from rank 0."""
import json, os, sys
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
  rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
  dist.init_process_group(backend=os.environ.get("RIP_BENCH_BACKEND", "gloo"), rank=rank, world_size=world)
  dev = torch.device("cuda", 0 if os.environ.get("RIP_BENCH_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0")))
  torch.cuda.set_device(dev)
  from oatomobile_amd import DIMTrainer, ImitativeModel
  model = ImitativeModel.synthetic(500, max_batch=1).to(dev)   # the same initial weights on every rank
  trainer = DIMTrainer(model, lr=1e-3, max_batch=8, device=dev, group=dist.group.WORLD)
  rng = np.random.default_rng(10 + rank)                        # different data per rank
  B = 6
  batch = dict(visual_features=torch.from_numpy(rng.random((B, 2, 100, 100), dtype=np.float32)).to(dev),
               velocity=torch.from_numpy(rng.normal(0, 3, size=(B, 3)).astype(np.float32)).to(dev),
               is_at_traffic_light=torch.zeros(B, 1, device=dev), traffic_light_state=torch.ones(B, 1, device=dev),
               player_future=torch.from_numpy(np.cumsum(np.abs(rng.normal(size=(B, 4, 3))), axis=1).astype(np.float32)).to(dev))
  gen = torch.Generator(device=dev).manual_seed(3 + rank)
  keep = (torch.rand(B, 1280, device=dev, generator=gen) >= 0.2).float() / 0.8
  loss = trainer.backward(batch, y=batch["player_future"][..., :2].contiguous(), dropout_mask=keep)
  local = trainer.grads.clone()
  gathered = [torch.empty_like(local).cpu() for _ in range(world)]
  dist.all_gather(gathered, local.cpu())
  trainer.apply(clip=True)  # all-reduce -> clip -> Adam: the AVERAGED gradient is what gets clipped (DDP order)
  mean = torch.stack(gathered).mean(0).to(dev)
  mean_norm = float(torch.linalg.vector_norm(mean.double()))
  local_norm = float(torch.linalg.vector_norm(local.double()))
  mean = mean * min(1.0, 1.0 / (mean_norm + 1e-6))
  err = float((trainer.grads - mean).abs().max() / mean.abs().max())
  p = [torch.empty_like(trainer.params).cpu() for _ in range(world)]
  dist.all_gather(p, trainer.params.cpu())
  m = [torch.empty_like(trainer.exp_avg).cpu() for _ in range(world)]
  dist.all_gather(m, trainer.exp_avg.cpu())
  tr = trainer._trainable.cpu().bool()  # BatchNorm running statistics are per-rank buffers (DDP without SyncBatchNorm)
  same = bool(torch.equal(p[0][tr], p[1][tr]) and torch.equal(m[0], m[1]))
  buffers_differ = float((p[0][~tr] - p[1][~tr]).abs().max())
  differ = float((gathered[0] - gathered[1]).abs().max())
  if rank == 0:
    print(json.dumps({"world": world, "loss": float(loss), "avg_grad_rel_err": err, "params_identical": same, "running_stats_differ_by": buffers_differ,
                      "local_grads_differ_by": differ, "mean_grad_norm": mean_norm, "local_grad_norm": local_norm}))
  dist.barrier()
  dist.destroy_process_group()


if __name__ == "__main__":
  main()
