"""Development-container only: generate `tests/golden/*.npz` from the REFERENCE's own code.

Imports the reference's hot-path modules from /root/reference through a
`sys.modules` shim (its package `__init__`s need absl/gym/carla, which are not
installed), with

  * `torch.hub.load` replaced by `oracle.mobilenet_v2.mobilenet_v2` (torchvision
    v0.6.0 is absent from the reference tree — encoder parity is unpinned),
  * the two delegates `RIPAgent` expects but `ImitativeModel` lacks
    (`_forward/_inverse`, rip/agent.py:106,111,137) added,
  * stubs for `oatomobile.Env/Agent` and `baselines.base.SetPointAgent`
    (constructor only),
  * models in `.eval()` mode.

Model weights come from `oatomobile_amd.weights.synthetic_state_dict(seed)`
(regenerated bit-identically anywhere), inputs from `numpy.random.default_rng`;
only inputs/outputs are written.  The reference never travels to the GPU box.

Usage:  python tools/make_golden.py            (writes tests/golden/)
"""

import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/oatomobile"

from oatomobile_amd import weights as W  # noqa: E402
from oracle import mobilenet_v2 as mnv2  # noqa: E402


def _install_shim():
  def pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m

  top = pkg("oatomobile", REF)
  for n, p in [("oatomobile.torch", "/torch"), ("oatomobile.torch.networks", "/torch/networks"),
               ("oatomobile.baselines", "/baselines"), ("oatomobile.baselines.torch", "/baselines/torch"),
               ("oatomobile.baselines.torch.dim", "/baselines/torch/dim"),
               ("oatomobile.baselines.torch.rip", "/baselines/torch/rip")]:
    pkg(n, REF + p)

  class Env:  # oatomobile.Env stub
    pass

  class Agent:  # oatomobile.Agent stub
    def __init__(self, environment=None):
      self._environment = environment

  top.Env, top.Agent = Env, Agent
  base = types.ModuleType("oatomobile.baselines.base")

  class SetPointAgent(Agent):
    def __init__(self, environment, **kwargs):
      super().__init__(environment)

  base.SetPointAgent = SetPointAgent
  sys.modules["oatomobile.baselines.base"] = base
  # oatomobile.torch.types only holds typing aliases but imports dm-tree-free code? import lazily.
  torch.hub.load = lambda *a, **kw: mnv2.mobilenet_v2(num_classes=kw["num_classes"])
  dim = importlib.import_module("oatomobile.baselines.torch.dim.model")
  seq = importlib.import_module("oatomobile.torch.networks.sequence")
  tfm = importlib.import_module("oatomobile.torch.transforms")
  rip = importlib.import_module("oatomobile.baselines.torch.rip.agent")
  dim.ImitativeModel._forward = lambda self, x, z: self._decoder._forward(x=x, z=z)
  dim.ImitativeModel._inverse = lambda self, y, z: self._decoder._inverse(y=y, z=z)
  return dim, seq, tfm, rip


def ref_model(dim, seed):
  m = dim.ImitativeModel()
  sd = W.synthetic_state_dict(seed)
  m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
  return m.eval()


def synth_observation(rng, C=2, G=10):
  """SURVEY.md §8(d) config-1 distribution."""
  lidar = (rng.integers(0, 6, size=(200, 200, C)) / 5.0) * (rng.random((200, 200, C)) < 0.12)
  goal = np.cumsum(np.abs(rng.normal(size=(G, 2))) * 2.0, axis=0)
  goal = np.c_[goal, np.zeros((G, 1))]
  return dict(
      lidar=lidar.astype(np.float32),
      velocity=rng.normal(0, 3.0, size=(3,)).astype(np.float32),
      is_at_traffic_light=np.float32(rng.random() < 0.2),
      traffic_light_state=np.float32(rng.integers(0, 4)),
      goal=goal.astype(np.float32),
  )


def main():
  out = os.path.join(ROOT, "tests", "golden")
  os.makedirs(out, exist_ok=True)
  dim, seq, tfm, rip = _install_shim()
  torch.set_num_threads(1)  # deterministic reduction order

  # ---- G1 transform -------------------------------------------------------
  rng = np.random.default_rng(0)
  lidar = rng.random((2, 2, 200, 200)).astype(np.float32)
  m = ref_model(dim, 11)
  sample = m.transform({"lidar": torch.from_numpy(lidar.copy()),
                        "player_future": torch.arange(2 * 40 * 3, dtype=torch.float32).view(2, 40, 3)})
  vis = sample["visual_features"].numpy()
  idx = rng.integers(0, 100, size=(256, 2))
  np.savez_compressed(os.path.join(out, "g1_transform.npz"), seed=0, idx=idx,
                      picked=vis[:, :, idx[:, 0], idx[:, 1]], checksum=np.float64(vis.astype(np.float64).sum()),
                      row7=vis[0, 1, 7, :], col93=vis[1, 0, :, 93],
                      player_future=sample["player_future"].numpy())

  # ---- G2 flow forward / inverse (reference code, unmodified) ---------------
  rng = np.random.default_rng(1)
  m = ref_model(dim, 1)
  z = np.maximum(rng.normal(size=(128, 64)), 0).astype(np.float32)
  x = rng.normal(size=(128, 4, 2)).astype(np.float32)
  with torch.no_grad():
    y, lad_f = m._decoder._forward(torch.from_numpy(x), torch.from_numpy(z))
    xi, logp, lad_i = m._decoder._inverse(y, torch.from_numpy(z))
    yy = torch.from_numpy((rng.normal(size=(128, 4, 2)) * 3).astype(np.float32))
    xi2, logp2, lad_i2 = m._decoder._inverse(yy, torch.from_numpy(z))
  np.savez_compressed(os.path.join(out, "g2_flow.npz"), weight_seed=1, z=z, x=x, y=y.numpy(), lad_f=lad_f.numpy(),
                      x_inv=xi.numpy(), logp=logp.numpy(), lad_i=lad_i.numpy(), y2=yy.numpy(), x_inv2=xi2.numpy(),
                      logp2=logp2.numpy(), lad_i2=lad_i2.numpy())

  # ---- G3 merger with stub encoder features ---------------------------------
  rng = np.random.default_rng(3)
  m = ref_model(dim, 3)
  feats = rng.normal(size=(4, 128)).astype(np.float32)
  vec = np.c_[rng.normal(0, 3, size=(4, 3)), rng.integers(0, 2, size=(4, 1)), rng.integers(0, 4, size=(4, 1))].astype(np.float32)
  with torch.no_grad():
    zz = m._merger(torch.cat([torch.from_numpy(feats), torch.from_numpy(vec)], dim=-1))
  np.savez_compressed(os.path.join(out, "g3_merger.npz"), weight_seed=3, feats=feats, vec=vec, z=zz.numpy())

  # ---- G4 goal likelihood -----------------------------------------------------
  rng = np.random.default_rng(4)
  yg = (rng.normal(size=(128, 4, 2)) * 4).astype(np.float32)
  goal = (rng.normal(size=(128, 10, 2)) * 6).astype(np.float32)
  g4 = dict(y=yg, goal=goal)
  for eps in (0.5, 1.0):
    with torch.no_grad():
      g4["mean_eps%g" % eps] = m._goal_likelihood(torch.from_numpy(yg), torch.from_numpy(goal), epsilon=eps).numpy()
      g4["rows_eps%g" % eps] = np.stack([
          m._goal_likelihood(torch.from_numpy(yg[i:i + 1]), torch.from_numpy(goal[i:i + 1]), epsilon=eps).numpy()
          for i in range(128)])
  np.savez_compressed(os.path.join(out, "g4_goal.npz"), **g4)

  # ---- G5 full _params (eval) -------------------------------------------------
  g5 = {}
  for ws in (5, 6):
    m = ref_model(dim, ws)
    for i in range(2):
      rng = np.random.default_rng(50 + i)
      ob = synth_observation(rng)
      lid = torch.from_numpy(ob["lidar"])[None].permute(0, 3, 1, 2).contiguous()
      s = m.transform({"lidar": lid})
      with torch.no_grad():
        zz = m._params(visual_features=s["visual_features"], velocity=torch.from_numpy(ob["velocity"])[None],
                       is_at_traffic_light=torch.tensor([[float(ob["is_at_traffic_light"])]]),
                       traffic_light_state=torch.tensor([[float(ob["traffic_light_state"])]]))
        feat = m._encoder(s["visual_features"])
      g5["z_w%d_o%d" % (ws, 50 + i)] = zz.numpy()[0]
      g5["feat_w%d_o%d" % (ws, 50 + i)] = feat.numpy()[0]
  np.savez_compressed(os.path.join(out, "g5_params.npz"), **g5)

  # ---- G6 RIP loop: patched reference RIPAgent.__call__, instrumented --------
  # The loop body below is driven through the reference's own objects
  # (model._forward/_inverse/_goal_likelihood, torch.optim.Adam); the final
  # `[30,3]` output comes from calling the reference `RIPAgent.__call__` itself.
  g6 = {}
  K = 4
  models = [ref_model(dim, 100 + k) for k in range(K)]
  for algo in ("WCM", "MA", "BCM"):
    for os_ in (60, 61, 62):
      rng = np.random.default_rng(os_)
      ob = synth_observation(rng)
      agent = rip.RIPAgent(environment=None, algorithm=algo, models=models)
      obs_in = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in ob.items()}
      obs_in["bird_view_camera_cityscapes"] = np.zeros((4, 4, 3), np.float32)
      out30 = agent(obs_in)
      tag = "%s_o%d" % (algo, os_)
      g6["out30_" + tag] = out30
      # instrumented replica of rip/agent.py:85-137 on the reference objects
      lid = torch.from_numpy(ob["lidar"])[None].permute(0, 3, 1, 2).contiguous()
      obs = models[0].transform({"lidar": lid})
      ctx = dict(visual_features=obs["visual_features"], velocity=torch.from_numpy(ob["velocity"])[None],
                 is_at_traffic_light=torch.tensor([[float(ob["is_at_traffic_light"])]]),
                 traffic_light_state=torch.tensor([[float(ob["traffic_light_state"])]]))
      goal_t = torch.from_numpy(ob["goal"][None, :, :2].copy())
      x = torch.zeros(1, 4, 2, requires_grad=True)
      zs = [mm._params(**ctx).detach() for mm in models]
      opt = torch.optim.Adam([x], lr=1e-1)
      x_best, loss_best = x.clone(), torch.tensor(1000.0)
      posts, xs, losses = [], [], []
      for _ in range(10):
        opt.zero_grad()
        yv, _ = models[0]._forward(x=x, z=zs[0])
        ps = []
        for mm, zz in zip(models, zs):
          _, lp, lad = mm._inverse(y=yv, z=zz)
          ps.append(torch.mean(lp - lad) + mm._goal_likelihood(y=yv, goal=goal_t, epsilon=1.0))
        ps = torch.stack(ps, 0)
        loss = {"WCM": lambda: torch.min(-ps, 0)[0], "BCM": lambda: torch.max(-ps, 0)[0],
                "MA": lambda: torch.mean(-ps, 0)}[algo]()
        loss.backward()
        opt.step()
        if loss < loss_best:
          x_best, loss_best = x.clone(), loss.clone()
        posts.append(ps.detach().numpy().copy()); xs.append(x.detach().numpy().copy()); losses.append(float(loss))
      plan, _ = models[0]._forward(x=x_best, z=zs[0])
      g6["post_" + tag] = np.stack(posts)
      g6["x_" + tag] = np.stack(xs)[:, 0]
      g6["loss_" + tag] = np.asarray(losses)
      g6["loss_best_" + tag] = np.float64(loss_best)
      g6["plan_" + tag] = plan.detach().numpy()[0]
      g6["zs_" + tag] = np.stack([z.numpy()[0] for z in zs])
      # cross-check: the instrumented replica and the real __call__ agree
      from oracle.reference_cpu import interpolate_plan
      assert np.allclose(interpolate_plan(g6["plan_" + tag]), out30, atol=1e-5), tag
  np.savez_compressed(os.path.join(out, "g6_rip.npz"), **g6)

  # ---- G7 ImitativeModel.forward (reference, seeded base sample captured) ----
  g7 = {}
  m = ref_model(dim, 7)
  for B, os_ in ((1, 70), (3, 71)):
    obs_list = [synth_observation(np.random.default_rng(os_ + 10 * b)) for b in range(B)]
    lid = torch.stack([torch.from_numpy(o["lidar"]).permute(2, 0, 1) for o in obs_list]).contiguous()
    s = m.transform({"lidar": lid})
    ctx = dict(visual_features=s["visual_features"],
               velocity=torch.stack([torch.from_numpy(o["velocity"]) for o in obs_list]),
               is_at_traffic_light=torch.tensor([[float(o["is_at_traffic_light"])] for o in obs_list]),
               traffic_light_state=torch.tensor([[float(o["traffic_light_state"])] for o in obs_list]))
    goal_t = torch.stack([torch.from_numpy(o["goal"][:, :2].copy()) for o in obs_list])
    for with_goal in (False, True):
      torch.manual_seed(1234)
      x0 = m._decoder._base_dist.sample().clone()  # what dim/model.py:100 will draw
      torch.manual_seed(1234)
      yv = m(num_steps=20, goal=goal_t if with_goal else None, lr=5e-2, epsilon=1.0, **ctx)
      tag = "B%d_goal%d" % (B, int(with_goal))
      g7["x0_" + tag] = x0.numpy()
      g7["y_" + tag] = yv.detach().numpy()
    with torch.no_grad():
      g7["z_B%d" % B] = m._params(**ctx).numpy()
  np.savez_compressed(os.path.join(out, "g7_dim_forward.npz"), **g7)

  # ---- G8 [K=4, N=128] score matrix via K reference _inverse calls -----------
  rng = np.random.default_rng(8)
  ob = synth_observation(rng)
  yk = (np.cumsum(np.abs(rng.normal(size=(128, 4, 2))) * 2, axis=1)).astype(np.float32)
  lid = torch.from_numpy(ob["lidar"])[None].permute(0, 3, 1, 2).contiguous()
  obs = models[0].transform({"lidar": lid})
  ctx = dict(visual_features=obs["visual_features"], velocity=torch.from_numpy(ob["velocity"])[None],
             is_at_traffic_light=torch.tensor([[float(ob["is_at_traffic_light"])]]),
             traffic_light_state=torch.tensor([[float(ob["traffic_light_state"])]]))
  S, SG, zs8 = [], [], []
  with torch.no_grad():
    for mm in models:
      zz = mm._params(**ctx)
      zs8.append(zz.numpy()[0])
      _, lp, lad = mm._inverse(y=torch.from_numpy(yk), z=zz.expand(128, -1))
      S.append((lp - lad).numpy())
      gl = np.stack([mm._goal_likelihood(y=torch.from_numpy(yk[i:i + 1]), goal=torch.from_numpy(ob["goal"][None, :, :2].copy()),
                                         epsilon=1.0).numpy() for i in range(128)])
      SG.append((lp - lad).numpy() + gl)
  np.savez_compressed(os.path.join(out, "g8_scores.npz"), obs_seed=8, y=yk, zs=np.stack(zs8), S=np.stack(S), SG=np.stack(SG))

  # ---- G9 LIDAR point cloud -> BEV histogram (reference function, unmodified) -----
  # oatomobile/utils/carla.py needs `carla`, `absl.logging` and `transforms3d.euler` at import time only (the function
  # under test touches none of them): empty stand-in modules let the reference file itself be imported and run.
  class _Blank(types.ModuleType):  # any attribute (type annotations like carla.ServerSideSensor) resolves to `object`
    def __getattr__(self, name):
      if name.startswith("__"):
        raise AttributeError(name)
      return object

  for name in ("carla", "absl", "absl.logging", "transforms3d", "transforms3d.euler"):
    if name not in sys.modules:
      sys.modules[name] = _Blank(name)
  pkg_utils = types.ModuleType("oatomobile.utils")
  pkg_utils.__path__ = [REF + "/utils"]
  sys.modules["oatomobile.utils"] = pkg_utils
  ref_carla = importlib.import_module("oatomobile.utils.carla")

  class Measurement:  # stands for carla.LidarMeasurement: only `.raw_data` is read (utils/carla.py:212)
    def __init__(self, points):
      self.raw_data = np.ascontiguousarray(points, dtype=np.float32).tobytes()

  rng = np.random.default_rng(9)
  edges = np.linspace(-50, 51, 201)
  clouds = []
  # (a) a CARLA-like frame: dense near the car, heights around the -2.5 m split, some out of range
  a = np.c_[rng.normal(0, 18, size=(20000, 2)), rng.normal(-2.4, 0.6, size=(20000, 1))].astype(np.float32)
  clouds.append(a)
  # (b) edge cases: points exactly on bin edges (as float32), on the outer edges, z exactly -2.5, NaN / inf, far outliers
  eb = edges.astype(np.float32)
  b = np.stack([np.r_[eb, eb[::-1], np.float32(51.0), np.float32(-50.0), np.nextafter(np.float32(51.0), np.float32(60)),
                      np.nextafter(np.float32(-50.0), np.float32(-60)), np.float32(np.nan), np.float32(np.inf), np.float32(1e9)],
                np.r_[eb[::-1], eb, np.float32(51.0), np.float32(-50.0), np.float32(0.0), np.float32(0.0), np.float32(0.0),
                      np.float32(0.0), np.float32(0.0)],
                np.r_[np.full(201, -2.5), np.full(201, -2.5), -2.5, -2.5, -3.0, -1.0, -3.0, -1.0, -3.0].astype(np.float32)], axis=1)
  clouds.append(b.astype(np.float32))
  # (c) one cell hit far more often than the clip, and an empty cloud
  c = np.tile(np.array([[3.3, -7.7, -3.0]], np.float32), (1000, 1))
  c[::2, 2] = -1.0
  clouds.append(c)
  clouds.append(np.zeros((0, 3), np.float32))
  g9 = {}
  for i, pts in enumerate(clouds):
    g9["points%d" % i] = pts
    g9["bev%d" % i] = ref_carla.carla_lidar_measurement_to_ndarray(Measurement(pts))
  np.savez_compressed(os.path.join(out, "g9_lidar.npz"), **g9)

  # ---- G10 conditional imitation learning: BehaviouralModel.forward / CILAgent.__call__ (reference classes) ------
  pkg_cil = types.ModuleType("oatomobile.baselines.torch.cil")
  pkg_cil.__path__ = [REF + "/baselines/torch/cil"]
  sys.modules["oatomobile.baselines.torch.cil"] = pkg_cil
  cil_model = importlib.import_module("oatomobile.baselines.torch.cil.model")
  cil_agent = importlib.import_module("oatomobile.baselines.torch.cil.agent")
  wseed = 10
  bm = cil_model.BehaviouralModel()
  bm.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in W.synthetic_cil_state_dict(wseed).items()}, strict=True)
  bm.eval()
  rng = np.random.default_rng(10)
  Bc = 6
  vis = rng.random((Bc, 2, 100, 100), dtype=np.float32)
  vis[:, :, 30:70, 55:] = 0
  ctx = dict(visual_features=vis, velocity=rng.normal(0, 3, size=(Bc, 3)).astype(np.float32),
             is_at_traffic_light=rng.integers(0, 2, size=(Bc, 1)).astype(np.float32),
             traffic_light_state=rng.integers(0, 4, size=(Bc, 1)).astype(np.float32),
             mode=np.array([[0], [2], [3], [0], [2], [3]], np.float32))
  with torch.no_grad():
    yc = bm(**{k: torch.from_numpy(v) for k, v in ctx.items()}).numpy()
  g10 = dict(weight_seed=wseed, y=yc, **{"ctx_" + k: v for k, v in ctx.items()})
  # agent level: one synthetic observation per command branch (STOP / LEFT / RIGHT)
  agent = cil_agent.CILAgent(None, model=bm)
  agent._device = torch.device("cpu")
  agent._model = bm
  for i, goal_last in enumerate([(1.0, 0.5), (10.0, 12.0), (20.0, 1.0)]):
    ob = synth_observation(np.random.default_rng(100 + i))
    ob["goal"] = np.asarray(ob["goal"], np.float32).copy()
    ob["goal"][-1, :2] = goal_last
    ob["bird_view_camera_cityscapes"] = np.zeros((2, 2, 3), np.float32)
    plan = agent(dict(ob))  # scalars stay numpy scalars: the agent wraps them with np.atleast_1d (cil/agent.py:54-57)
    g10["agent_obs_seed%d" % i] = 100 + i
    g10["agent_goal_last%d" % i] = np.asarray(goal_last, np.float32)
    g10["agent_plan%d" % i] = plan
  np.savez_compressed(os.path.join(out, "g10_cil.npz"), **g10)

  tot = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
  print("wrote", sorted(os.listdir(out)), "total bytes", tot)


if __name__ == "__main__":
  main()
