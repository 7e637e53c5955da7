"""The RCCL code path of oatomobile_amd/distributed.py and DIMTrainer on ONE GPU (run by
tests/test_gpu_parity.py::test_rccl_world1_executes_every_collective in a fresh process): a one-rank "nccl" process
group (nccl == RCCL on ROCm) bound to cuda:0, RIP_DIST_ALWAYS_COLLECTIVE=1 so that the one-rank group still goes through
`all_gather_into_tensor` / `all_reduce` on DEVICE tensors (no host staging), and every composition checked against the
collective-free path.  Prints one JSON line."""
import json, os, sys
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
  os.environ["RIP_DIST_ALWAYS_COLLECTIVE"] = "1"
  dev = torch.device("cuda", 0)
  torch.cuda.set_device(dev)
  dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
  from oatomobile_amd import DIMTrainer, ImitativeModel
  from oatomobile_amd import distributed as D
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  from tests.helpers import synth_observation
  calls = {"all_gather_into_tensor": 0, "all_reduce": 0}
  real_ag, real_ar = dist.all_gather_into_tensor, dist.all_reduce

  def ag(out, inp, *a, **k):
    assert out.is_cuda and inp.is_cuda, "host-staged gather on the RCCL path"
    calls["all_gather_into_tensor"] += 1
    return real_ag(out, inp, *a, **k)

  def ar(t, *a, **k):
    assert t.is_cuda
    calls["all_reduce"] += 1
    return real_ar(t, *a, **k)

  dist.all_gather_into_tensor, dist.all_reduce = ag, ar
  K, N, B = 4, 32, 2
  models = [ImitativeModel.synthetic(100 + k, max_batch=1) for k in range(K)]
  obs = [synth_observation(np.random.default_rng(900 + i)) for i in range(B)]
  lidar = torch.stack([torch.from_numpy(o["lidar"]) for o in obs]).to(dev)
  vec = torch.tensor([[*o["velocity"], o["is_at_traffic_light"], o["traffic_light_state"]] for o in obs], device=dev)
  goal = torch.stack([torch.from_numpy(o["goal"][:, :2].copy()) for o in obs]).to(dev)
  rec = {"backend": dist.get_backend(), "world": dist.get_world_size()}
  # candidate-parallel: one all-gather of the winner records
  cp = D.CandidateParallelRIP(models, N, algorithm="WCM", seed=2, max_batch=B, device=dev, group=dist.group.WORLD)
  plan_c, idx_c, loss_c = cp(lidar, vec, goal)
  cp1 = D.CandidateParallelRIP(models, N, algorithm="WCM", seed=2, max_batch=B, device=dev, rank=0, world=1)
  plan_1, idx_1, loss_1 = cp1(lidar, vec, goal)
  rec["candidates_plan_diff"] = float((plan_c - plan_1).abs().max())
  rec["candidates_gathers"] = calls["all_gather_into_tensor"]
  # model-parallel, gradient mode: one all-gather of z + one per Adam step
  before = calls["all_gather_into_tensor"]
  mp = D.ModelParallelRIP(models, K, num_candidates=N, algorithm="WCM", seed=2, max_batch=B, device=dev, group=dist.group.WORLD)
  plan_m = mp(lidar, vec, goal)[0]
  mp1 = D.ModelParallelRIP(models, K, num_candidates=N, algorithm="WCM", seed=2, max_batch=B, device=dev, rank=0, world=1)
  rec["models_plan_diff"] = float((plan_m - mp1(lidar, vec, goal)[0]).abs().max())
  rec["models_gathers"] = calls["all_gather_into_tensor"] - before
  # model-parallel scoring + row gather
  before = calls["all_gather_into_tensor"]
  rows = D.gather_rows(plan_c, B, dist.group.WORLD)
  rec["gather_rows_equal"] = bool(torch.equal(rows, plan_c))
  S = torch.randn(K, B, N, device=dev)
  rec["all_gather_scores_equal"] = bool(torch.equal(D.all_gather_scores(S, K, dist.group.WORLD), S))
  rec["epilogue_gathers"] = calls["all_gather_into_tensor"] - before
  # data-parallel training step: one all-reduce of the packed gradient vector
  tr = DIMTrainer(ImitativeModel.synthetic(7, max_batch=1).to(dev), lr=1e-3, max_batch=4, device=dev, group=dist.group.WORLD)
  rng = np.random.default_rng(5)
  batch = dict(visual_features=torch.from_numpy(rng.random((4, 2, 100, 100), dtype=np.float32)).to(dev),
               velocity=torch.zeros(4, 3, device=dev), is_at_traffic_light=torch.zeros(4, 1, device=dev),
               traffic_light_state=torch.ones(4, 1, device=dev),
               player_future=torch.from_numpy(np.cumsum(np.abs(rng.normal(size=(4, 4, 3))), axis=1).astype(np.float32)).to(dev))
  loss = tr.backward(batch, y=batch["player_future"][..., :2].contiguous(), dropout_mask=torch.ones(4, 1280, device=dev))
  g0 = tr.grads.clone()
  tr.apply()
  rec["train_loss"] = float(loss)
  rec["allreduce_calls"] = calls["all_reduce"]
  rec["allreduce_identity"] = bool(torch.equal(tr.grads, g0))  # the sum over one rank
  with open("/proc/self/maps") as f:
    maps = f.read()
  rec["librccl_mapped"] = "librccl" in maps
  rec["librip_mapped"] = "librip_hip.so" in maps
  print(json.dumps(rec))
  dist.barrier()
  dist.destroy_process_group()


if __name__ == "__main__":
  main()
