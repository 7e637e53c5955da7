"""GPU parity tests: the HIP path (through the C ABI, via the product classes) against the CPU
oracle and against the committed golden fixtures from the reference.  Tolerance 1e-4 fp32
(BASELINE.json north_star) unless stated."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oatomobile_amd import weights as W  # noqa: E402
from tests.helpers import synth_observation  # noqa: E402

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
  assert torch.cuda.is_available()
  return torch.device("cuda", 0)


def hip_model(seed, dev, **kw):
  from oatomobile_amd import ImitativeModel
  return ImitativeModel.synthetic(seed, **kw).to(dev)


def oracle_model(seed):
  from oracle import reference_cpu as O
  return O.OracleImitativeModel.from_numpy_state_dict(W.synthetic_state_dict(seed))


def ctx_tensors(obs_list, dev):
  from oatomobile_amd import transform_visual
  lid = torch.stack([torch.from_numpy(o["lidar"]) for o in obs_list]).to(dev)  # [B,200,200,C]
  return dict(
      visual_features=transform_visual(lid, channels_last=True),
      velocity=torch.stack([torch.from_numpy(o["velocity"]) for o in obs_list]).to(dev),
      is_at_traffic_light=torch.tensor([[float(o["is_at_traffic_light"])] for o in obs_list], device=dev),
      traffic_light_state=torch.tensor([[float(o["traffic_light_state"])] for o in obs_list], device=dev),
  )


# Outliers of the statistical end-to-end gates as recorded on the MI355X (profiles/r6/outliers_v1.log: `pytest -s`, the
# "candidates outside" lines): tag -> most candidates that may fall outside 1e-3 + 1e-4 |loss|.  EVERY gate of the suite
# recorded 0 outliers in round 6 (17 kernel x algorithm x size cases, K=8 N=512 over all 512 candidates, 16 observations of
# the bench configuration, the model-parallel compositions; largest |d loss| 9.5e-6), so the table is empty and the
# ceiling of every tag is 0: one candidate outside the tolerance fails the test.
OUTLIER_CEILINGS = {
}


def candidate_gate(tag, lh, lo, plan_h=None, plan_o=None, frac=0.99, x_h=None, x_o=None, ceiling="table"):
  """The N-candidate gate of the end-to-end search tests.  Every candidate's best loss against the oracle's within
  1e-3 + 1e-4 |loss| for >= `frac` of the candidates (N < 100: all of them), the winning plan at 1e-4 when the winner
  is unambiguous (its best loss more than 1e-3 below the runner-up's).  Whatever falls outside is printed: the
  candidate, both losses, the oracle's gap to its runner-up and — when per-step latents `x_h` / `x_o` [steps,N,4,2]
  are given — the first Adam step at which the two trajectories differ by more than 1e-4 (an Adam sign flip of a
  near-zero gradient coordinate moves x by 2 lr there).
  Round 6: the NUMBER of outliers is pinned as well — `OUTLIER_CEILINGS[tag]` is the count recorded on the
  MI355X for that parametrisation (the kernels are deterministic; the ceiling leaves room for one more flip, not for
  a drift from 0.2 % to 0.9 %).  A tag without an entry must have NO outlier."""
  lh, lo = np.asarray(lh, np.float64).reshape(-1), np.asarray(lo, np.float64).reshape(-1)
  d = np.abs(lh - lo)
  out = np.flatnonzero(~(d <= 1e-3 + 1e-4 * np.abs(lo)))
  srt = np.sort(lo)
  gap = float(srt[1] - srt[0]) if lo.size > 1 else float("inf")
  print("%s: %d / %d candidates outside 1e-3 + 1e-4|loss| (max |d loss| %.3g), winner gap %.3g" %
        (tag, out.size, lo.size, float(d.max()), gap))
  for n in out[:8]:
    first = ""
    if x_h is not None and x_o is not None:
      dx = np.abs(np.asarray(x_h)[:, n] - np.asarray(x_o)[:, n]).reshape(len(x_h), -1).max(axis=1)
      bad = np.flatnonzero(dx > 1e-4)
      first = ", trajectories part at Adam step %s (max |dx| %.3g)" % (bad[0] if bad.size else "-", float(dx.max()))
    print("   candidate %d: loss %.6f vs oracle %.6f (|d| %.3g), rank %d of the oracle's ordering%s" %
          (n, lh[n], lo[n], d[n], int(np.sum(lo < lo[n])), first))
  assert 1.0 - out.size / lo.size >= frac, "%s: %d of %d candidates outside the tolerance" % (tag, out.size, lo.size)
  if ceiling == "table":
    ceiling = OUTLIER_CEILINGS.get(tag, 0)
  if ceiling is not None:
    assert out.size <= ceiling, "%s: %d candidates outside the tolerance, recorded ceiling %d" % (tag, out.size, ceiling)
  if plan_h is not None and (gap > 1e-3 or lo.size == 1):
    err = float(np.abs(np.asarray(plan_h) - np.asarray(plan_o)).max())
    print("%s: winner plan max |d| = %.3g m" % (tag, err))
    np.testing.assert_allclose(plan_h, plan_o, atol=TOL)
    return err
  return None


def test_native_library_loaded():
  from oatomobile_amd import _lib
  lib = _lib.load()
  assert lib.rip_abi_version() == _lib.ABI_VERSION
  with open("/proc/self/maps") as f:
    assert "librip_hip.so" in f.read()


def test_g1_transform(golden, dev):
  from oatomobile_amd import ImitativeModel
  g = golden("g1_transform.npz")
  lidar = np.random.default_rng(0).random((2, 2, 200, 200)).astype(np.float32)
  m = hip_model(11, dev)
  sample = m.transform({"lidar": torch.from_numpy(lidar).to(dev),
                        "player_future": torch.arange(2 * 40 * 3, dtype=torch.float32, device=dev).view(2, 40, 3)})
  assert "lidar" not in sample
  vis = sample["visual_features"].cpu().numpy()
  idx = g["idx"]
  np.testing.assert_allclose(vis[:, :, idx[:, 0], idx[:, 1]], g["picked"], atol=2e-6)
  np.testing.assert_allclose(vis[0, 1, 7, :], g["row7"], atol=2e-6)
  np.testing.assert_allclose(vis[1, 0, :, 93], g["col93"], atol=2e-6)
  assert abs(vis.astype(np.float64).sum() - float(g["checksum"])) < 5e-2
  np.testing.assert_array_equal(sample["player_future"].cpu().numpy(), g["player_future"])
  # sensor layout (channels last) gives the same image
  from oatomobile_amd import transform_visual
  vis2 = transform_visual(torch.from_numpy(np.ascontiguousarray(lidar.transpose(0, 2, 3, 1))).to(dev), channels_last=True)
  np.testing.assert_array_equal(vis2.cpu().numpy(), vis)


def test_g2_flow(golden, dev):
  g = golden("g2_flow.npz")
  m = hip_model(int(g["weight_seed"]), dev)
  z, x = torch.from_numpy(g["z"]).to(dev), torch.from_numpy(g["x"]).to(dev)
  y, lad = m._decoder._forward(x, z)
  np.testing.assert_allclose(y.cpu().numpy(), g["y"], atol=TOL)
  np.testing.assert_allclose(lad.cpu().numpy(), g["lad_f"], atol=TOL)
  xi, lp, ladi = m._decoder._inverse(torch.from_numpy(g["y"]).to(dev), z)
  np.testing.assert_allclose(xi.cpu().numpy(), g["x_inv"], atol=TOL)
  np.testing.assert_allclose(lp.cpu().numpy(), g["logp"], atol=TOL)
  np.testing.assert_allclose(ladi.cpu().numpy(), g["lad_i"], atol=TOL)
  xi, lp, ladi = m._inverse(torch.from_numpy(g["y2"]).to(dev), z)
  np.testing.assert_allclose(xi.cpu().numpy(), g["x_inv2"], rtol=1e-5, atol=5e-4)
  np.testing.assert_allclose(lp.cpu().numpy(), g["logp2"], rtol=2e-5, atol=2e-3)  # |logp| ~ 1e3 here
  np.testing.assert_allclose(ladi.cpu().numpy(), g["lad_i2"], atol=TOL)
  # round trip
  xr, _, _ = m._inverse(*[m._forward(x, z)[0], z])
  np.testing.assert_allclose(xr.cpu().numpy(), g["x"], atol=TOL)
  # z broadcast ([1,64]) and ragged sizes
  for n in (1, 3, 65, 257):
    xs = x[:1].repeat(n, 1, 1)
    y1, _ = m._forward(xs, z[:1])
    np.testing.assert_allclose(y1.cpu().numpy(), np.repeat(g["y"][:1], n, 0), atol=TOL)
  ys = m._decoder.forward(z)
  assert ys.shape == (128, 4, 2) and torch.isfinite(ys).all()


def test_g4_goal(golden, dev):
  g = golden("g4_goal.npz")
  m = hip_model(3, dev)
  y, goal = torch.from_numpy(g["y"]).to(dev), torch.from_numpy(g["goal"]).to(dev)
  for eps in (0.5, 1.0):
    rows = m._goal_likelihood_rows(y, goal, epsilon=eps).cpu().numpy()
    np.testing.assert_allclose(rows, g["rows_eps%g" % eps], rtol=1e-5, atol=2e-4)
    np.testing.assert_allclose(m._goal_likelihood(y, goal, epsilon=eps).cpu().numpy(), g["mean_eps%g" % eps],
                               rtol=1e-5, atol=2e-4)


def test_g5_params(golden, dev):
  g = golden("g5_params.npz")
  for ws in (5, 6):
    m = hip_model(ws, dev)
    for os_ in (50, 51):
      ob = synth_observation(np.random.default_rng(os_))
      ctx = ctx_tensors([ob], dev)
      z = m._params(**ctx).cpu().numpy()[0]
      feat = m.encoder_features(ctx["visual_features"]).cpu().numpy()[0]
      np.testing.assert_allclose(feat, g["feat_w%d_o%d" % (ws, os_)], atol=TOL)
      np.testing.assert_allclose(z, g["z_w%d_o%d" % (ws, os_)], atol=TOL)


def test_layerwise_encoder_matches_fused(golden, dev):
  """RIP_OPT_ENCODER_FUSED sweeps: every split between fused blocks and per-layer kernels gives the golden z."""
  g = golden("g5_params.npz")
  m = hip_model(5, dev)
  obs = [synth_observation(np.random.default_rng(50 + i)) for i in range(2)]
  ctx = ctx_tensors(obs, dev)
  for nfused in (0, 4, 17):  # all layer-wise, hybrid, every block fused
    m.fused_encoder = nfused
    z = m._params(**ctx).cpu().numpy()
    for i, os_ in enumerate((50, 51)):
      np.testing.assert_allclose(z[i], g["z_w5_o%d" % os_], atol=TOL, err_msg="fused_blocks=%d" % nfused)


def test_fp32_split_tile_blocks_vs_oracle_and_layerwise(dev):
  """Round 6: features.8-17 of the fp32 encoder as split-f16 tile blocks (encoder_split_tile.hip: fp32 activations,
  two-term binary16 pointwise operands, fp32 accumulation) from 176 (model, observation) pairs per launch; stem + features.1-7 as
  row-streaming kernels at every launch size (small launches cut an observation into row bands), features.8-17 of a small
  launch as expansion + depthwise kernels (one (observation, 64-channel chunk) per workgroup) + layer-wise projections.
  (a) 180 observations of one model against the fp32 ORACLE at the suite's 1e-4 (one observation per workgroup), and the first
      1 / 3 / 7 of them as launches of their own (row bands, expansion + depthwise kernels);
  (b) K = 3 models x B = 601 observations (1803 pairs: 3 / 4 / 2 observations per workgroup, every block's last group
      ragged, persistent workgroups walking two groups) against the layer-wise true-fp32 kernels of the same handle
      (`RIP_OPT_ENCODER_VARIANT` bit 16), z and every block output that reaches memory; B - 1 observations reproduce
      the first B - 1 rows bit for bit (observations are independent, the LDS rows of a ragged group are stale)."""
  from oatomobile_amd import _lib, RIPAgent, arch
  from oracle import reference_cpu as O
  m, mo = hip_model(31, dev, max_batch=180), oracle_model(31)
  obs = [synth_observation(np.random.default_rng(3100 + i)) for i in range(180)]
  ctx = ctx_tensors(obs, dev)
  m._handle().set_option(_lib.OPT_KERNEL_LOG, 1)
  z = m._params(**ctx).cpu().numpy()
  log = m._handle().kernel_log()
  assert sum(l.startswith("irb_split_tile_kernel") for l in log) == 10, log
  assert sum(l.startswith("irb_split_rows_kernel") for l in log) == 6, log  # features.2-7
  assert sum(l.startswith("front_split_kernel<2>") for l in log) == 1 and not any(l.startswith(("stem_kernel", "irb_kernel")) for l in log), log
  assert sum(l.startswith("head_split_kernel") for l in log) == 1 and not any(l.startswith(("pw_kernel", "dw_kernel")) for l in log), log  # features.18 + pool
  zo = O.params(mo, **{k: v.cpu() for k, v in ctx.items()}).numpy()
  print("fp32 encoder with split-f16 tile blocks vs fp32 oracle (180 observations): max|dz| = %.3g of max|z| = %.3g" % (np.abs(z - zo).max(), np.abs(zo).max()))
  np.testing.assert_allclose(z, zo, atol=TOL)
  for nb in (1, 3, 7):  # small launches: row bands, tile blocks / head layer-wise
    zs = m._params(**{k: v[:nb].contiguous() for k, v in ctx.items()}).cpu().numpy()
    logs = m._handle().kernel_log()
    rows = [l for l in logs if l.startswith(("irb_split_rows_kernel", "front_split_kernel"))]
    assert len(rows) == 7 and all(" NB=1 " not in l for l in rows) and not any(l.startswith(("irb_split_tile", "head_split")) for l in logs), logs
    assert sum(l.startswith("irb_split_expdw_kernel") for l in logs) == 10 and not any(l.startswith("dw_kernel") for l in logs), logs
    np.testing.assert_allclose(zs, zo[:nb], atol=TOL)
    print("   %d observation(s) as a launch of its own (%s): max|dz| = %.3g" % (nb, rows[1].split(" ")[1], np.abs(zs - zo[:nb]).max()))
  m._handle().set_option(_lib.OPT_ENCODER_VARIANT, _lib.ENC_VAR_FP32_LAYERWISE)
  z_lw = m._params(**ctx).cpu().numpy()
  assert not any(l.startswith(("irb_split_", "front_split_", "head_split_")) for l in m._handle().kernel_log())
  print("   the layer-wise fp32 kernels on the same observations: max|dz| = %.3g" % np.abs(z_lw - zo).max())

  K, B, C = 3, 601, 2
  models = [hip_model(500 + k, dev, max_batch=1) for k in range(K)]
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=16, max_batch=B, device=dev)
  h, lib = agent._handle, _lib.load()
  h.set_option(_lib.OPT_KERNEL_LOG, 1)
  rng = np.random.default_rng(77)
  vis = torch.from_numpy(rng.random((B, C, 100, 100), dtype=np.float32))
  vis = (vis * (torch.from_numpy(rng.random((B, C, 100, 100), dtype=np.float32)) < 0.3)).to(dev)
  vec = torch.from_numpy(rng.normal(0, 2, size=(B, 5)).astype(np.float32)).to(dev)

  def encode(variant, b):
    h.set_option(_lib.OPT_ENCODER_VARIANT, variant)
    zz = torch.full((K, b, 64), float("nan"), device=dev)
    _lib.check(lib.rip_encode(h.raw, _lib.ptr(vis), _lib.ptr(vec), b, 0, K, _lib.ENC_DTYPES["fp32"], _lib.ptr(zz), None, h.stream()))
    return zz.cpu().numpy(), h.kernel_log()

  z_s, log_s = encode(0, B)
  z_l, log_l = encode(_lib.ENC_VAR_FP32_LAYERWISE, B)
  tiles = [l for l in log_s if l.startswith("irb_split_tile_kernel")]
  assert len(tiles) == 10 and {l.split(" ")[1] for l in tiles} == {"G=3", "G=4", "G=2"}, tiles
  assert sum(l.startswith("irb_split_rows_kernel") for l in log_s) == 6, log_s
  assert not any(l.startswith(("irb_split_", "front_split_", "head_split_")) for l in log_l)
  assert not any(l.startswith("irb_split_expdw") for l in log_s)  # (a large launch: the tile blocks)
  assert np.isfinite(z_s).all()
  d = np.abs(z_s - z_l).max()
  print("K = 3 x B = 601, split-f16 tile blocks vs layer-wise fp32: max|dz| = %.3g of max|z| = %.3g" % (d, np.abs(z_l).max()))
  assert d <= TOL
  z_r, _ = encode(0, B - 1)
  np.testing.assert_array_equal(z_r, z_s[:, :B - 1])
  # a launch that starts in the middle of the handle's models (k_begin = 1: the packed operand blobs are indexed per model)
  zz = torch.full((2, B, 64), float("nan"), device=dev)
  _lib.check(lib.rip_encode(h.raw, _lib.ptr(vis), _lib.ptr(vec), B, 1, 2, _lib.ENC_DTYPES["fp32"], _lib.ptr(zz), None, h.stream()))
  assert sum(l.startswith(("irb_split_", "front_split_", "head_split_")) for l in h.kernel_log()) == 18
  np.testing.assert_array_equal(zz.cpu().numpy(), z_s[1:3])
  # every block output (the selection differs in features.8-17 only; the inputs of a block differ by the blocks before it)
  layers = arch.conv_layers(C)
  worst = 0.0
  for i, spec in enumerate(layers[:-1]):
    outs = []
    for variant in (0, _lib.ENC_VAR_FP32_LAYERWISE):
      h.set_option(_lib.OPT_ENCODER_VARIANT, variant)
      out = torch.empty((K, B, spec.h_out, spec.h_out, spec.cout), device=dev)
      rc = lib.rip_encode_tap_k(h.raw, _lib.ptr(vis), B, 0, K, _lib.ENC_DTYPES["fp32"], i, _lib.ptr(out), out.numel(), h.stream())
      outs.append(None if rc == _lib.RIP_EINVAL else out)
      if rc != _lib.RIP_EINVAL:
        _lib.check(rc)
    if outs[0] is None:
      continue  # interior to a block of the shipped selection
    assert outs[1] is not None
    e = float((outs[0] - outs[1]).abs().max()) / max(1e-6, float(outs[1].abs().max()))
    worst = max(worst, e)
    assert e <= 2e-5, "layer %d: split-f16 block output differs from the layer-wise kernels by %.3g of its range" % (i, e)
  print("   block outputs: largest difference %.3g of a tensor's range" % worst)
  h.set_option(_lib.OPT_ENCODER_VARIANT, 0)


def _status(handle):
  from oatomobile_amd import _lib
  return _lib.load().rip_encoder_status(handle.raw)


def test_mega_encoder_matches_layerwise(golden, dev):
  """RIP_OPT_ENCODER_MEGA: the fp32 encoder of a small batch as ONE persistent launch (model k on XCD k % 8, layer
  barriers in that XCD's L2) against the 55 layer-wise launches: the same layer arithmetic (tile shapes may differ:
  last-bit differences), the golden z of g5 at 1e-4, and a status word that stays 0."""
  from oatomobile_amd import _lib
  g = golden("g5_params.npz")
  m = hip_model(5, dev, max_batch=4)
  m._handle().set_option(_lib.OPT_ENCODER_VARIANT, _lib.ENC_VAR_FP32_LAYERWISE)  # the reference launches: the same true-fp32 arithmetic as the one-launch kernel
  obs = [synth_observation(np.random.default_rng(50 + i)) for i in range(4)]
  for b in (1, 2, 4):
    ctx = ctx_tensors(obs[:b], dev)
    m.mega_encoder = 0
    z0 = m._params(**ctx).cpu().numpy()
    f0 = m.encoder_features(ctx["visual_features"]).cpu().numpy()
    m.mega_encoder = 1
    z1 = m._params(**ctx).cpu().numpy()
    f1 = m.encoder_features(ctx["visual_features"]).cpu().numpy()
    torch.cuda.synchronize()
    assert _status(m._handle()) == 0
    print("mega vs layer-wise, B = %d: max|dz| = %.3g, max|dfeat| = %.3g (max|feat| %.3g)" %
          (b, np.abs(z1 - z0).max(), np.abs(f1 - f0).max(), np.abs(f0).max()))
    np.testing.assert_allclose(z1, z0, atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(f1, f0, atol=2e-5, rtol=2e-5)
    for i, os_ in enumerate((50, 51)[:b]):
      np.testing.assert_allclose(z1[i], g["z_w5_o%d" % os_], atol=TOL)
      np.testing.assert_allclose(f1[i], g["feat_w5_o%d" % os_], atol=TOL)
  # auto (the default) = the launches
  m.mega_encoder = -1
  ctx = ctx_tensors(obs[:1], dev)
  za = m._params(**ctx).cpu().numpy()
  m.mega_encoder = 0
  np.testing.assert_array_equal(za, m._params(**ctx).cpu().numpy())


def test_mega_encoder_repeats_and_ensemble(dev):
  """The barrier counters re-arm themselves at the end of every launch: 200 back-to-back launches (no host sync in
  between) give bit-identical z; a K = 4 ensemble (XCDs 0..3 busy, 4..7 idle) and a K = 8 one (every XCD) agree with
  the layer-wise launches, and with the per-model handles."""
  from oatomobile_amd import ImitativeModel, RIPAgent, _lib
  lib = _lib.load()
  obs = [synth_observation(np.random.default_rng(300 + i)) for i in range(2)]
  for K in (4, 8):
    models = [ImitativeModel.synthetic(700 + k, max_batch=2).to(dev) for k in range(K)]
    agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=16, max_batch=2, device=dev)
    h = agent._handle
    h.set_option(_lib.OPT_ENCODER_VARIANT, _lib.ENC_VAR_FP32_LAYERWISE)  # (the launches the one-launch kernel shares its arithmetic with)
    for mk in models:
      mk._handle().set_option(_lib.OPT_ENCODER_VARIANT, _lib.ENC_VAR_FP32_LAYERWISE)
    for b in (1, 2):
      ctx = ctx_tensors(obs[:b], dev)
      vis = ctx["visual_features"].contiguous()
      vec = torch.cat([ctx["velocity"], ctx["is_at_traffic_light"], ctx["traffic_light_state"]], dim=-1).contiguous()
      zs = {}
      for mega in (0, 1):
        h.set_option(_lib.OPT_ENCODER_MEGA, mega)
        z = torch.empty(K, b, 64, device=dev)
        reps = 200 if (mega == 1 and K == 4 and b == 1) else 3
        outs = []
        for _ in range(reps):
          zi = torch.empty(K, b, 64, device=dev)
          _lib.check(lib.rip_encode(h.raw, _lib.ptr(vis), _lib.ptr(vec), b, 0, K, 0, _lib.ptr(zi), None, h.stream()))
          outs.append(zi)
        torch.cuda.synchronize()
        assert _status(h) == 0
        for zi in outs[1:]:
          assert torch.equal(zi, outs[0])
        zs[mega] = outs[0].cpu().numpy()
      np.testing.assert_allclose(zs[1], zs[0], atol=2e-5, rtol=2e-5)
      for k in (0, K - 1):
        models[k].mega_encoder = 0
        zk = models[k]._params(**ctx).cpu().numpy()
        np.testing.assert_allclose(zs[1][k], zk, atol=2e-5, rtol=2e-5)
    # a sub-range of the ensemble (model-parallel ranks encode their own slice): members 1..2 on XCDs 0..1
    h.set_option(_lib.OPT_ENCODER_MEGA, 1)
    zsub = torch.empty(2, 1, 64, device=dev)
    ctx = ctx_tensors(obs[:1], dev)
    vis = ctx["visual_features"].contiguous()
    vec = torch.cat([ctx["velocity"], ctx["is_at_traffic_light"], ctx["traffic_light_state"]], dim=-1).contiguous()
    _lib.check(lib.rip_encode(h.raw, _lib.ptr(vis), _lib.ptr(vec), 1, 1, 2, 0, _lib.ptr(zsub), None, h.stream()))
    zall = torch.empty(K, 1, 64, device=dev)
    _lib.check(lib.rip_encode(h.raw, _lib.ptr(vis), _lib.ptr(vec), 1, 0, K, 0, _lib.ptr(zall), None, h.stream()))
    torch.cuda.synchronize()
    assert torch.equal(zsub, zall[1:3])
    assert _status(h) == 0


def test_mega_encoder_failure_is_reported_once_and_the_agent_recovers(dev):
  """`rip_encoder_status` is one-shot.  The failure word of the one-launch encoder is raised through the
  test hook (RIP_OPT_DEBUG_ENCODER_FAULT) exactly as its kernel raises it; `agent(observation)` must repeat the call
  ONCE on the layer-wise launches and return the same plan as an agent that never used the one-launch encoder (it
  used to recurse until RecursionError because the sticky word was read again by every retry)."""
  from oatomobile_amd import ImitativeModel, RIPAgent, _lib
  lib = _lib.load()
  ob = synth_observation(np.random.default_rng(77))
  models = [ImitativeModel.synthetic(710 + k, max_batch=1) for k in range(2)]
  ref_agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=16, max_batch=1, device=dev)
  want = ref_agent(dict(ob))
  models2 = [ImitativeModel.synthetic(710 + k, max_batch=1) for k in range(2)]
  agent = RIPAgent(None, algorithm="WCM", models=models2, num_candidates=16, max_batch=1, device=dev)
  h = agent._handle
  h.set_option(_lib.OPT_ENCODER_MEGA, 1)
  if lib.rip_set_option(h.raw, _lib.OPT_DEBUG_ENCODER_FAULT, 2) != 0:
    pytest.skip("one-launch encoder not available on this device (placement probe): nothing to recover from")
  assert lib.rip_encoder_status(h.raw) == 2 and lib.rip_encoder_status(h.raw) == 0  # handed out once
  # a fresh agent: its first call captures the hipGraph WITH the one-launch kernel; the word is then raised as that
  # kernel would raise it during a replay (after the call's entry points ran, visible after the stream sync)
  models3 = [ImitativeModel.synthetic(710 + k, max_batch=1) for k in range(2)]
  agent = RIPAgent(None, algorithm="WCM", models=models3, num_candidates=16, max_batch=1, device=dev)
  h = agent._handle
  h.set_option(_lib.OPT_ENCODER_MEGA, 1)
  first = agent(dict(ob))
  np.testing.assert_allclose(first, want, atol=TOL)
  _lib.check(lib.rip_set_option(h.raw, _lib.OPT_DEBUG_ENCODER_FAULT, 2))
  import sys
  limit = sys.getrecursionlimit()
  sys.setrecursionlimit(200)  # the old behaviour fails fast instead of re-capturing 1000 graphs
  try:
    got = agent(dict(ob))
    again = agent(dict(ob))
  finally:
    sys.setrecursionlimit(limit)
  np.testing.assert_allclose(got, want, atol=TOL)
  np.testing.assert_array_equal(got, again)
  assert lib.rip_encoder_status(h.raw) == 0


BF16_ULP = 2.0 ** -8   # relative spacing of bfloat16 (8 significand bits)


def _bf16_layer_gate(tag, got, want, worst, layers=1):
  """One teacher-forced layer (or fused block) of the bf16 encoder against `oracle/bf16_encoder.py`: both ran the same
  arithmetic on the same input, so they differ only where an fp32 sum that differs in its last places (summation order:
  MFMA K-chunks vs torch's conv; the 16-bit hi + lo depthwise taps of the fused blocks) falls on the other side of a
  bf16 rounding boundary — one bf16 ulp of the value.
  One layer: |d| <= 2^-7 |ref| + 3e-5 max|ref| EVERYWHERE (2 ulp + the fp32 summation noise of cancelling sums).
  A fused block (3 layers, the two inner tensors never leave the chip): a flip of an INNER element moves the outputs it
  feeds by |w| ulp(inner) — an absolute error, so near-zero outputs leave the relative bound; the gate is then >= 99 % of
  the elements inside the one-layer bound and every element within 4 bf16 ulp OF THE TENSOR'S SCALE (2^-6 max|ref|): a
  wrong tap, a dropped residual or a missing ReLU6 moves most elements by O(scale).
  Reported: the fraction of elements that differ at all, and the worst error in ulps of the element."""
  got, want = got.double().cpu(), want.double()
  d = (got - want).abs()
  scale = float(want.abs().max())
  tol = 2.0 ** -7 * want.abs() + 3e-5 * scale
  frac = float((d > 0).double().mean())
  ulps = float((d / (BF16_ULP * want.abs().clamp_min(1e-3 * scale))).max())
  worst.append((ulps, frac, tag))
  inside = float((d <= tol).double().mean())
  msg = "%s: max |d| %.3g at scale %.3g, %.3f %% of the elements differ, %.4f %% beyond the one-layer bound" % (
      tag, float(d.max()), scale, 100 * frac, 100 * (1 - inside))
  if layers == 1:
    assert inside == 1.0, msg
  else:
    assert inside >= 0.99 and float(d.max()) <= 2.0 ** -6 * scale, msg


def test_bf16_encoder_every_layer_teacher_forced_vs_bf16_oracle(dev):
  """a bf16-emulating oracle for the arithmetic the headline runs.  Layer-wise bf16 kernels
  (RIP_OPT_ENCODER_FUSED = 0), all 52 conv layers tapped through `rip_encode_tap`; layer i of the oracle is fed the HIP
  path's own output of layer i-1 (teacher forcing removes error propagation: a wrong tap, a dropped residual or a
  missing ReLU6 in ONE layer cannot hide under the noise of the other 51)."""
  from oracle import bf16_encoder as BE
  from oatomobile_amd import arch
  m, mo = hip_model(21, dev, max_batch=4), oracle_model(21)
  m.encoder_dtype = "bf16"
  m.fused_encoder = 0
  obs = [synth_observation(np.random.default_rng(1000 + i)) for i in range(3)]
  vis = ctx_tensors(obs, dev)["visual_features"]
  L = len(arch.conv_layers())
  taps = {i: m.encoder_layer_output(vis, i).cpu() for i in range(L)}
  pooled_last = taps.pop(L - 1)
  want = BE.teacher_forced(mo, taps, vis.cpu(), [(i, i) for i in range(L)])
  worst = []
  for i in range(L - 1):
    assert torch.equal(BE.bf16_round(taps[i]), taps[i])  # what is stored IS bf16
    _bf16_layer_gate("layer %d (%s)" % (i, arch.conv_layers()[i].name.split("features.")[1]), taps[i], want[i], worst)
  # features.18: fp32 out, 4x4 average pool in its epilogue
  got, ref = pooled_last.double(), want[L - 1].double().mean(dim=(2, 3))
  assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6
  worst.sort(reverse=True)
  print("bf16 layer-wise vs bf16 oracle, teacher-forced: worst layers (ulp, fraction of elements that differ): " +
        "; ".join("%s %.2f ulp %.3f %%" % (t, u, 100 * f) for u, f, t in worst[:4]))
  assert max(f for _, f, _ in worst) < 0.01  # flips are rare: a systematic difference would touch most elements


def _fused_blocks_vs_oracle(dev, B, C, seed, variant=0):
  """The SHIPPED bf16 kernels (RIP_OPT_ENCODER_FUSED = 17: front kernel, row-streaming blocks with the depthwise on the
  matrix cores, tile blocks): every block output that reaches memory, teacher-forced per block against the bf16 oracle
  from the HIP path's own block input, for three rows of the batch (first, middle, last: the last workgroup's partly
  empty observation group / pixel tiles)."""
  from oracle import bf16_encoder as BE
  from oracle import reference_cpu as O
  from oatomobile_amd import _lib, arch
  m = hip_model(seed, dev, max_batch=B, in_channels=C)
  mo = O.OracleImitativeModel.from_numpy_state_dict(W.synthetic_state_dict(seed, C), in_channels=C)
  m.encoder_dtype = "bf16"
  rng = np.random.default_rng(5 + B)
  vis = torch.from_numpy(rng.random((B, C, 100, 100), dtype=np.float32))
  vis = (vis * (torch.from_numpy(rng.random((B, C, 100, 100), dtype=np.float32)) < 0.3)).to(dev)  # sparse, like a BEV
  L = len(arch.conv_layers(C))
  m.fused_encoder = 17
  m._handle().set_option(_lib.OPT_ENCODER_VARIANT, variant)
  m._handle().set_option(_lib.OPT_KERNEL_LOG, 1)
  fused, ranges, first, log = {}, [], 0, []
  for i in range(L):
    try:
      fused[i] = m.encoder_layer_output(vis, i).cpu()
      log += m._handle().kernel_log()
    except _lib.RipError:
      continue  # interior to a fused block
    ranges.append((first, i))
    first = i + 1
  assert len(ranges) >= 18 and sum(b - a == 2 for a, b in ranges) >= 16, ranges  # the blocks really ran fused
  check = sorted({0, B // 2, B - 1})
  pooled = fused.pop(L - 1)
  want = BE.teacher_forced(mo, {i: t[check] for i, t in fused.items()}, vis[check].cpu(), ranges)
  worst = []
  for a, b in ranges:
    if b == L - 1:
      ref = want[b].double().mean(dim=(2, 3))
      assert float((pooled[check].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6
      continue
    _bf16_layer_gate("layers %d..%d" % (a, b), fused[b][check], want[b], worst, layers=b - a + 1)
  worst.sort(key=lambda t: -t[1])
  print("bf16 fused blocks vs bf16 oracle, B = %d, C = %d: most differing blocks (fraction of elements): %s" %
        (B, C, "; ".join("%s %.3f %%" % (t, 100 * f) for _, f, t in worst[:4])))
  assert max(f for _, f, _ in worst) < 0.02  # 16-bit depthwise taps / bias: ~2^-8 of the elements flip per block
  return log


@pytest.mark.parametrize("B", [1, 3, 9, 64, 130])
def test_bf16_fused_blocks_teacher_forced_vs_bf16_oracle(dev, B):
  """Batches that leave the last workgroup's observation group / pixel tiles partly empty (1, 3, 9, 130) and the size
  from which `auto` engages the tile blocks (64).  Replaces round 3's z-level comparisons of fused against layer-wise
  kernels: those held at 2-3 % only while the two were bit-identical — the encoder amplifies a single bf16 flip in
  0.5 % of one early tensor to 4 % of max|z| (measured on the oracle, see test_bf16_encoder_end_to_end), so z cannot
  gate a kernel; a teacher-forced block can."""
  _fused_blocks_vs_oracle(dev, B, 2, 22)


def test_bf16_fused_blocks_four_channel_bev_vs_bf16_oracle(dev):
  """BASELINE configs[1] input (200x200x4 BEV): the fused front (stem + features.1 with C = 4: its LDS input band is
  twice as large, one workgroup per CU) and every other block, 160 observations."""
  _fused_blocks_vs_oracle(dev, 160, 4, 33)


def test_bf16_block_kernels_everywhere(dev):
  """The retained A/B kernels under the teacher-forced block test (the kernel log proves the selection took effect):
  RIP_OPT_ENCODER_VARIANT bit 2 = round 3's front kernel, bit 8 = features.17 layer-wise.  Bits 1 and 4 selected round 1's
  row-streaming kernel until round 5; it is retired (the matrix-core depthwise kernel is faster on all of features.2-7
  since the depthwise taps are bf16 values): the bits are accepted and change nothing."""
  from oatomobile_amd import _lib
  for variant, must in ((_lib.ENC_VAR_FRONT_ROUND3, "front_bf16_kernel<2>"), (_lib.ENC_VAR_F17_LAYERWISE, "dw_"),
                        (_lib.ENC_VAR_IRB_ROUND3 | _lib.ENC_VAR_ROWS_F5_7, "irb2_bf16_kernel<1,32,192,32")):
    for B in (3, 64):
      log = _fused_blocks_vs_oracle(dev, B, 2, 22, variant=variant)
      assert any(l.startswith(must) for l in log), (variant, must, log)
      assert not any(l.startswith("irb_rows") for l in log), log


def _headline_kernel_names(variant):
  """Encoder kernels of the driver-shaped bench (bench.DEFAULT_OBS_BATCH observations x 4 models, bf16) per `RIP_OPT_ENCODER_VARIANT` value:
  the committed manifest tests/golden/headline_kernels.json (recorded from the rocprofv3 kernel traces under profiles/;
  a parity test must not depend on which profiling artefact is newest — VERDICT r5 weak #1)."""
  import json
  with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "headline_kernels.json")) as f:
    return set(json.load(f)[str(variant)])


@pytest.mark.parametrize("variant", [0, 8])
def test_bf16_headline_launch_shape_vs_bf16_oracle(dev, variant):
  """the bf16 kernel selection keys on B * k_count, and the teacher-forced block tests above tap ONE
  model at B <= 160 — they never launch `gemm_pers_bf16_kernel<4,false|true>` / `dw_rows_bf16_kernel<1,4>` (features.17 /
  18 of a large launch), nor the grids the fused blocks take at 2048 (model, observation) pairs.  Here the tap runs the
  headline's own launch (`rip_encode_tap_k`: K = 4 models x B = bench.DEFAULT_OBS_BATCH observations (2048 since round 6), automatic selection), (a) the
  kernel log of a whole encode of that shape must be exactly the encoder kernel set rocprofv3 recorded for the bench
  (the manifest tests/golden/headline_kernels.json), and (b) every output that reaches memory is gated teacher-forced against the bf16 oracle on rows
  {0, 255, 511} of every model (12 images on the oracle side).  variant 0 = what ships (round 5: features.17 is a tile
  block too); variant 8 = features.17 layer-wise, i.e. round 4's selection with `gemm_pers_bf16_kernel<4,false>` x 2 and
  `dw_rows_bf16_kernel<1,4>`, kernels no other launch of the suite reaches."""
  import bench
  from oracle import bf16_encoder as BE
  from oracle import reference_cpu as O
  from oatomobile_amd import _lib, arch, RIPAgent
  K, B, C = 4, bench.DEFAULT_OBS_BATCH, 2  # (round 6: the bench step is 2048 observations)
  seeds = [100 + k for k in range(K)]
  models = [hip_model(sd, dev, max_batch=1) for sd in seeds]
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=16, max_batch=B, device=dev, encoder_dtype="bf16")
  h, lib = agent._handle, _lib.load()
  h.set_option(_lib.OPT_KERNEL_LOG, 1)
  h.set_option(_lib.OPT_ENCODER_VARIANT, variant)
  rng = np.random.default_rng(5 + B)
  vis = torch.from_numpy(rng.random((B, C, 100, 100), dtype=np.float32))
  vis = (vis * (torch.from_numpy(rng.random((B, C, 100, 100), dtype=np.float32)) < 0.3)).to(dev)  # sparse, like a BEV
  vec = torch.zeros(B, 5, device=dev)
  z = torch.empty(K, B, 64, device=dev)
  # (a) the kernels of the whole launch
  _lib.check(lib.rip_encode(h.raw, _lib.ptr(vis), _lib.ptr(vec), B, 0, K, _lib.ENC_DTYPES["bf16"], _lib.ptr(z), None, h.stream()))
  got = {l.split(" ")[0] for l in h.kernel_log()}
  assert got == _headline_kernel_names(variant), (sorted(got ^ _headline_kernel_names(variant)))
  # (b) every block output of that launch, three rows per model
  layers = arch.conv_layers(C)
  L = len(layers)
  check = [0, B // 2 - 1, B - 1]
  fused, ranges, first = {}, [], 0
  seen = set()
  for i in range(L):
    spec, last = layers[i], i + 1 == L
    out = torch.empty((K, B, spec.cout) if last else (K, B, spec.h_out, spec.h_out, spec.cout), device=dev)
    rc = lib.rip_encode_tap_k(h.raw, _lib.ptr(vis), B, 0, K, _lib.ENC_DTYPES["bf16"], i, _lib.ptr(out), out.numel(), h.stream())
    if rc == _lib.RIP_EINVAL:
      continue  # interior to a fused block
    _lib.check(rc)
    seen |= {l.split(" ")[0] for l in h.kernel_log()}
    o = out[:, check]
    fused[i] = (o if last else o.permute(0, 1, 4, 2, 3)).contiguous().cpu()  # [K, 3, C, H, W]
    ranges.append((first, i))
    first = i + 1
  assert seen == got - {"cls_mfma_kernel", "merger_kernel"}, sorted(seen ^ got)  # the taps ran the launch's kernels
  assert len(ranges) >= 18 and sum(b - a == 2 for a, b in ranges) >= (16 if variant else 17), ranges
  worst = []
  for k in range(K):
    mo = O.OracleImitativeModel.from_numpy_state_dict(W.synthetic_state_dict(seeds[k], C), in_channels=C)
    taps = {i: t[k] for i, t in fused.items()}
    pooled = taps.pop(L - 1)
    want = BE.teacher_forced(mo, taps, vis[check].cpu(), ranges)
    for a, b in ranges:
      if b == L - 1:
        ref = want[b].double().mean(dim=(2, 3))
        assert float((pooled.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6
        continue
      _bf16_layer_gate("model %d layers %d..%d" % (k, a, b), taps[b], want[b], worst, layers=b - a + 1)
  worst.sort(key=lambda t: -t[1])
  print("bf16 headline launch (K = 4, B = %d) vs bf16 oracle: most differing blocks (fraction of elements): %s" % (B,
        "; ".join("%s %.3f %%" % (t, 100 * f) for _, f, t in worst[:4])))
  assert max(f for _, f, _ in worst) < 0.02


def test_bf16_model_count_does_not_change_a_model(dev):
  """The persistent GEMM of features.18 walks an XCD-aware work list whose shape depends on the number of models in
  the launch (1, 2, 4, 8 models: 8 / 4 / 2 / 1 XCDs per model; any other count: the plain (tile, slice, model) list), and
  the fused blocks size their grids on B x k_count.  A model's z must not depend on who else is in the launch beyond
  the bf16 noise of a different kernel selection: K = 8 models at B = 192, every prefix count 1 .. 8 and a launch that
  starts in the middle (k_begin = 3), against the model encoded alone."""
  from oatomobile_amd import _lib, RIPAgent
  K, B, C = 8, 192, 2
  models = [hip_model(300 + k, dev, max_batch=1) for k in range(K)]
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=16, max_batch=B, device=dev, encoder_dtype="bf16")
  h, lib = agent._handle, _lib.load()
  h.set_option(_lib.OPT_KERNEL_LOG, 1)
  rng = np.random.default_rng(91)
  vis = torch.from_numpy(rng.random((B, C, 100, 100), dtype=np.float32))
  vis = (vis * (torch.from_numpy(rng.random((B, C, 100, 100), dtype=np.float32)) < 0.3)).to(dev)
  vec = torch.from_numpy(rng.normal(0, 2, size=(B, 5)).astype(np.float32)).to(dev)

  def encode(k0, kc):
    z = torch.full((kc, B, 64), float("nan"), device=dev)
    _lib.check(lib.rip_encode(h.raw, _lib.ptr(vis), _lib.ptr(vec), B, k0, kc, _lib.ENC_DTYPES["bf16"], _lib.ptr(z), None, h.stream()))
    return z.cpu().numpy(), {l.split(" ")[0] for l in h.kernel_log()}

  alone = np.stack([encode(k, 1)[0][0] for k in range(K)])
  assert np.isfinite(alone).all()
  scale = np.abs(alone).max()
  gemm_counts = []
  for k0, kc in [(0, c) for c in range(2, K + 1)] + [(3, 3), (3, 5), (6, 2)]:
    z, kernels = encode(k0, kc)
    assert np.isfinite(z).all(), (k0, kc)
    d = np.abs(z - alone[k0:k0 + kc]).max()
    assert d <= 0.03 * scale, "models %d..%d in one launch differ from the models alone: %.3g of %.3g" % (k0, k0 + kc - 1, d, scale)
    if "gemm_pers_bf16_kernel<4,true>" in kernels:
      gemm_counts.append(kc)
  assert {2, 3, 4, 8} <= set(gemm_counts), gemm_counts  # every branch of the work list ran


def test_bf16_ragged_last_group_of_a_persistent_workgroup(dev):
  """The tile blocks and the row-streaming blocks keep their workgroups resident and let them walk several observation
  groups.  With K = 4 models and B = 510 observations a tile workgroup's SECOND group is the ragged one (2 of 4
  observations; its LDS rows still hold the previous, full group), and the row-streaming workgroups walk 3-4 observations
  each.  Observations are independent: the first 510 rows of the B = 512 launch must come out the same."""
  from oatomobile_amd import _lib, RIPAgent
  K, B, C = 4, 512, 2
  models = [hip_model(400 + k, dev, max_batch=1) for k in range(K)]
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=16, max_batch=B, device=dev, encoder_dtype="bf16")
  h, lib = agent._handle, _lib.load()
  h.set_option(_lib.OPT_KERNEL_LOG, 1)
  rng = np.random.default_rng(123)
  vis = torch.from_numpy(rng.random((B, C, 100, 100), dtype=np.float32))
  vis = (vis * (torch.from_numpy(rng.random((B, C, 100, 100), dtype=np.float32)) < 0.3)).to(dev)
  vec = torch.from_numpy(rng.normal(0, 2, size=(B, 5)).astype(np.float32)).to(dev)

  def encode(b):
    z = torch.full((K, b, 64), float("nan"), device=dev)
    _lib.check(lib.rip_encode(h.raw, _lib.ptr(vis), _lib.ptr(vec), b, 0, K, _lib.ENC_DTYPES["bf16"], _lib.ptr(z), None, h.stream()))
    return z.cpu().numpy(), h.kernel_log()

  z_full, log_full = encode(B)
  z_ragged, log_ragged = encode(B - 2)
  assert np.isfinite(z_ragged).all()
  tiles = [l for l in log_ragged if l.startswith("irb_tile_bf16_kernel")]
  assert tiles and all("grid=(64,1,4)" in l for l in tiles), tiles  # two groups per workgroup, the last one ragged
  d = np.abs(z_ragged - z_full[:, :B - 2]).max()
  same_kernels = [l.split(" ")[0] for l in log_full] == [l.split(" ")[0] for l in log_ragged]
  print("B = 510 vs the first 510 rows of B = 512: max|dz| = %.3g (same kernels: %s)" % (d, same_kernels))
  assert d <= (0.0 if same_kernels else 0.03 * np.abs(z_full).max())


def test_bf16_encoder_end_to_end_vs_bf16_oracle(dev):
  """BASELINE config 3 end to end.  z of the shipped bf16 path against the bf16 oracle, gated by what the oracle ITSELF
  does under the only freedom a correct bf16 implementation has — which way an element on a rounding boundary falls:
  `flip_noise` re-runs the oracle with 0.5 % of the elements of every block output moved by one bf16 ulp (the rate the
  teacher-forced block tests measure for the kernels) and the HIP deviation must stay within twice the largest of three
  such runs.  (A flat 1 % of max|z| is not attainable by any kernel that sums in a different order: the random-weight
  MobileNetV2 amplifies those flips to several per cent of max|z|; the tight gates are the per-layer / per-block ones.)
  Reported next to it: the storage format's own effect (bf16 oracle vs fp32 oracle)."""
  from oracle import bf16_encoder as BE
  from oracle import reference_cpu as O
  m, mo = hip_model(21, dev), oracle_model(21)
  obs = [synth_observation(np.random.default_rng(1000 + i)) for i in range(16)]
  ctx = ctx_tensors(obs, dev)
  z32 = m._params(**ctx).cpu().numpy()
  m.encoder_dtype = "bf16"
  z16 = m._params(**ctx).cpu().numpy()
  z16_small = np.concatenate([m._params(**{k: v[i:i + 2].contiguous() for k, v in ctx.items()}).cpu().numpy()
                              for i in range(0, 16, 2)])
  cpu = {k: v.cpu() for k, v in ctx.items()}
  zo32 = O.params(mo, **cpu).numpy()
  zo16 = BE.params(mo, **cpu).numpy()
  np.testing.assert_allclose(z32, zo32, atol=TOL)
  scale = np.abs(zo16).max()
  floor = max(np.abs(BE.params(mo, **cpu, flip_fraction=0.005, flip_seed=s).numpy() - zo16).max() for s in (1, 2, 3))
  print("oracle under one-ulp flips of 0.5 %% of every block output: max|dz| = %.3g (%.2f %% of max|z| = %.3g)" % (floor, 100 * floor / scale, scale))
  for tag, z in (("auto kernels, B = 16", z16), ("small-batch kernels, B = 2", z16_small)):
    err = np.abs(z - zo16)
    print("bf16 encoder vs bf16 oracle (%s): max|dz| = %.3g, mean|dz| = %.3g" % (tag, err.max(), err.mean()))
    assert err.max() <= 2.0 * floor, tag
  fmt = np.abs(zo16 - zo32)
  print("bf16 storage format itself (bf16 oracle vs fp32 oracle): max|dz| = %.3g, mean|dz| = %.3g" % (fmt.max(), fmt.mean()))
  assert not np.array_equal(z16, z32)  # really a different arithmetic


def test_bf16_encoder_large_batch_kernels_match_small_batch(dev):
  """The row-streaming depthwise kernels (buffer-descriptor padding, packed fp32 math) only engage for launches of
  >= 1M (pixel, 8-channel) items: 640 observations put every one of the 17 depthwise layers on that path.  Same
  arithmetic as the small-batch kernels except bf16 rounding via v_cvt_pk_bf16_f32 and fma pairing."""
  B = 640
  m = hip_model(23, dev, max_batch=B)
  m.encoder_dtype = "bf16"
  rng = np.random.default_rng(77)
  vis = torch.from_numpy(rng.random((B, 2, 100, 100), dtype=np.float32)).to(dev)
  vis[:, :, 40:60, 30:70] = 0  # empty patches, like a BEV
  ctx = dict(visual_features=vis,
             velocity=torch.from_numpy(rng.normal(0, 3, size=(B, 3)).astype(np.float32)).to(dev),
             is_at_traffic_light=torch.from_numpy(rng.integers(0, 2, size=(B, 1)).astype(np.float32)).to(dev),
             traffic_light_state=torch.from_numpy(rng.integers(0, 4, size=(B, 1)).astype(np.float32)).to(dev))
  z_big = m._params(**ctx).cpu().numpy()
  z_small = np.concatenate([m._params(**{k: v[i:i + 4].contiguous() for k, v in ctx.items()}).cpu().numpy()
                            for i in range(0, 64, 4)])
  assert np.isfinite(z_big).all()
  d = np.abs(z_big[:64] - z_small)
  print("large vs small batch kernels: max|dz| = %.3g of max|z| = %.3g" % (d.max(), np.abs(z_small).max()))
  assert d.max() <= 0.03 * np.abs(z_small).max()
  # rows are independent: the tail of the batch equals the same observations computed alone
  z_tail = m._params(**{k: v[B - 4:].contiguous() for k, v in ctx.items()}).cpu().numpy()
  assert np.abs(z_big[B - 4:] - z_tail).max() <= 0.03 * np.abs(z_small).max()


def test_params_missing_key_raises(dev):
  m = hip_model(5, dev)
  with pytest.raises(ValueError, match="Missing `velocity`"):
    m._params(visual_features=torch.zeros(1, 2, 100, 100, device=dev))
  with pytest.raises(ValueError, match="Missing `visual_features`"):
    m(num_steps=1)


def test_config2_batch64_vs_oracle(dev):
  """BASELINE config 2: batch 64, fp32, z and log_prob - logabsdet vs the oracle."""
  from oracle import reference_cpu as O
  m, mo = hip_model(21, dev), oracle_model(21)
  obs = [synth_observation(np.random.default_rng(1000 + i)) for i in range(64)]
  ctx = ctx_tensors(obs, dev)
  z = m._params(**ctx)
  ctx_cpu = {k: v.cpu() for k, v in ctx.items()}
  np.testing.assert_allclose(ctx_cpu["visual_features"].numpy(),
                             O.transform_visual(torch.stack([torch.from_numpy(o["lidar"]).permute(2, 0, 1) for o in obs])).numpy(),
                             atol=2e-6)
  zo = O.params(mo, **ctx_cpu)
  np.testing.assert_allclose(z.cpu().numpy(), zo.numpy(), atol=TOL)
  y = torch.from_numpy(np.cumsum(np.abs(np.random.default_rng(1).normal(size=(64, 4, 2))), axis=1).astype(np.float32))
  _, lp, lad = m._inverse(y.to(dev), z)
  _, lpo, lado = O.flow_inverse(mo, y, zo)
  np.testing.assert_allclose((lp - lad).cpu().numpy(), (lpo - lado).numpy(), rtol=1e-5, atol=TOL)


@pytest.mark.parametrize("algo", ["WCM", "MA", "BCM"])
def test_g6_rip_reference_recipe(golden, dev, algo):
  """N=1: the reference's RIPAgent.__call__ output [30,3] (golden, from the reference itself)."""
  from oatomobile_amd import RIPAgent
  g = golden("g6_rip.npz")
  models = [hip_model(100 + k, dev) for k in range(4)]
  agent = RIPAgent(None, algorithm=algo, models=models)
  for os_ in (60, 61, 62):
    tag = "%s_o%d" % (algo, os_)
    ob = synth_observation(np.random.default_rng(os_))
    out = agent(dict(ob))
    assert out.shape == (30, 3) and out.dtype == np.float64
    np.testing.assert_allclose(out, g["out30_" + tag], atol=TOL)


@pytest.mark.parametrize("kernel", ["chain", "phase", "split", "pair"])
@pytest.mark.parametrize("algo", ["WCM", "MA", "BCM"])
def test_g6_search_traces(golden, dev, algo, kernel):
  """Per-step posteriors, latents, best loss and plan of BOTH search kernels vs the instrumented reference loop
  (golden G6, from the reference's own objects).  The matrix-core kernels search 32 candidates; candidate 0 starts at zeros = the reference start, so its row must follow the reference step for
  step at 1e-4."""
  from oatomobile_amd import _lib, RIPAgent
  g = golden("g6_rip.npz")
  models = [hip_model(100 + k, dev) for k in range(4)]
  N = 1 if kernel == "chain" else 32
  agent = RIPAgent(None, algorithm=algo, models=models, num_candidates=N, search_kernel=kernel, seed=3)
  lib = _lib.load()
  for os_ in (60, 62):
    tag = "%s_o%d" % (algo, os_)
    ob = synth_observation(np.random.default_rng(os_))
    z = torch.from_numpy(g["zs_" + tag]).to(dev).reshape(4, 1, 64).contiguous()
    goal = torch.from_numpy(ob["goal"][None, :, :2].copy()).to(dev)
    x0 = agent._x0(1)
    assert float(x0[0, 0].abs().max()) == 0.0
    plans = torch.empty(1, N, 4, 2, device=dev)
    lb = torch.empty(1, N, device=dev)
    tp = torch.empty(10, 4, 1, N, device=dev)
    tx = torch.empty(10, 1, N, 4, 2, device=dev)
    tg = torch.empty(10, 1, N, 4, 2, device=dev)
    _lib.check(lib.rip_search(agent._handle.raw, _lib.ptr(z), _lib.ptr(goal), _lib.ptr(x0), 1, N, 10,
                              _lib.ALGORITHMS[algo], 10, 0.1, 1.0, None, _lib.ptr(plans), _lib.ptr(lb), None,
                              _lib.ptr(tp), _lib.ptr(tx), _lib.ptr(tg), agent._handle.stream()))
    np.testing.assert_allclose(tp.cpu().numpy()[:, :, 0, 0], g["post_" + tag], rtol=1e-5, atol=3e-4)
    np.testing.assert_allclose(tx.cpu().numpy()[:, 0, 0], g["x_" + tag], atol=TOL)
    np.testing.assert_allclose(float(lb.cpu()[0, 0]), float(g["loss_best_" + tag]), rtol=1e-5, atol=3e-4)
    np.testing.assert_allclose(plans.cpu().numpy()[0, 0], g["plan_" + tag], atol=TOL)
    assert torch.isfinite(tg).all()


@pytest.mark.parametrize("kernel", ["chain", "phase", "split", "pair"])
@pytest.mark.parametrize("algo,K", [("WCM", 4), ("MA", 3), ("BCM", 2)])
def test_teacher_forced_steps_vs_oracle(dev, kernel, algo, K):
  """Removes trajectory amplification from the kernel-vs-oracle comparison: every Adam step of the ORACLE's
  N = 128 trajectories (its pre-step latents x_t) is fed to ONE Adam step of each kernel (10 steps = batch of 10);
  per-candidate posteriors of all K models, loss and dLoss/dx must match the oracle's step at 1e-4."""
  from oatomobile_amd import _lib, RIPAgent
  from oracle import reference_cpu as O
  N, S = 128, 10
  models = [hip_model(200 + k, dev) for k in range(K)]
  refs = [oracle_model(200 + k) for k in range(K)]
  agent = RIPAgent(None, algorithm=algo, models=models, num_candidates=N, seed=5, search_kernel=kernel, max_batch=S)
  ob = synth_observation(np.random.default_rng(77))
  ctx = ctx_tensors([ob], dev)
  with torch.no_grad():
    zs = [O.params(m, **{k: v.cpu() for k, v in ctx.items()}) for m in refs]
  goal = torch.from_numpy(ob["goal"][None, :, :2].copy())
  res = O.rip_search(refs, zs, goal, agent._x0_rows.cpu(), algorithm=algo, num_steps=S)
  xpre = res["trace_x_pre"].contiguous().to(dev)                       # [S, N, 4, 2] -> batch of S "observations"
  z = torch.stack([zz[0] for zz in zs])[:, None, :].repeat(1, S, 1).contiguous().to(dev)  # [K, S, 64]
  goal_d = goal.repeat(S, 1, 1).contiguous().to(dev)
  lb = torch.empty(S, N, device=dev)
  tp = torch.empty(1, K, S, N, device=dev)
  tg = torch.empty(1, S, N, 4, 2, device=dev)
  lib = _lib.load()
  _lib.check(lib.rip_search(agent._handle.raw, _lib.ptr(z), _lib.ptr(goal_d), _lib.ptr(xpre), S, N, 10,
                            _lib.ALGORITHMS[algo], 1, 0.1, 1.0, None, None, _lib.ptr(lb), None, _lib.ptr(tp), None,
                            _lib.ptr(tg), agent._handle.stream()))
  post_h = tp.cpu().numpy()[0].transpose(1, 0, 2)     # [S, K, N]
  post_o = res["trace_post"].numpy()                  # [S, K, N]
  loss_o, grad_o = res["trace_loss"].numpy(), res["trace_grad"].numpy()
  grad_h = tg.cpu().numpy()[0]
  loss_h = lb.cpu().numpy()
  e_post = np.abs(post_h - post_o) / (1.0 + 1e-1 * np.abs(post_o))
  e_grad = np.abs(grad_h - grad_o) / (1.0 + np.abs(grad_o))
  print("%s/%s teacher-forced: max |d post| %.3g (scaled %.3g), max |d loss| %.3g, max |d grad| %.3g (scaled %.3g), "
        "max|grad| %.3g" % (kernel, algo, np.abs(post_h - post_o).max(), e_post.max(), np.abs(loss_h - loss_o).max(),
                            np.abs(grad_h - grad_o).max(), e_grad.max(), np.abs(grad_o).max()))
  np.testing.assert_allclose(post_h, post_o, rtol=1e-5, atol=TOL)
  np.testing.assert_allclose(loss_h, np.minimum(loss_o, 1000.0), rtol=1e-5, atol=TOL)
  # WCM / BCM: a candidate whose two best models tie within rounding may back-propagate the other one; such rows
  # are identified on the ORACLE's side (gap of the selected posterior < 1e-4) and skipped
  if algo == "MA":
    ok = np.ones((S, N), bool)
  else:
    srt = np.sort(post_o, axis=1)
    gap = (srt[:, -1] - srt[:, -2]) if algo == "WCM" else (srt[:, 1] - srt[:, 0])
    ok = gap > 1e-3
  assert ok.mean() > 0.99
  np.testing.assert_allclose(grad_h[ok], grad_o[ok], rtol=1e-4, atol=TOL)


def _scaled_flow_models(seeds, wscale, dev):
  """HIP and oracle models from the synthetic weights with the flow (GRU + head) weights multiplied by `wscale`."""
  from oatomobile_amd import ImitativeModel
  from oracle import reference_cpu as O
  hips, refs = [], []
  for s in seeds:
    sd = W.synthetic_state_dict(s)
    for key in sd:
      if key.startswith("_decoder.") and key.endswith(("weight_ih", "weight_hh", "0.weight", "2.weight")):
        sd[key] = (sd[key] * np.float32(wscale)).astype(np.float32)
    hips.append(ImitativeModel().load_numpy_state_dict(sd).to(dev))
    refs.append(O.OracleImitativeModel.from_numpy_state_dict(sd))
  return hips, refs


def test_split_kernel_weight_range_guard(dev):
  """Round 5 packs the adjoint's operand rows as w * 2^8 in binary16 (max 65504): a model with a flow weight of magnitude
  >= 200 must not reach the split-f16 kernel.  `rip_load_model` records the largest flow weight; with one such model in
  the handle `auto` takes the fp32-MFMA kernel (`rip_search_plan`), an explicit request for the split kernel is
  RIP_EINVAL, and the search is the fp32 kernel's, finite."""
  from oatomobile_amd import _lib, RIPAgent
  K, N, B = 2, 128, 24
  hips, _ = _scaled_flow_models([300, 301], 1.0, dev)
  big, _ = _scaled_flow_models([302], 2000.0, dev)  # U(-1/8, 1/8) x 2000: weights up to 250
  lib = _lib.load()
  rng = np.random.default_rng(3)
  z = torch.from_numpy(np.abs(rng.normal(size=(K, B, 64))).astype(np.float32)).to(dev)
  goal = torch.from_numpy(np.cumsum(np.abs(rng.normal(size=(B, 10, 2))) * 2, axis=1).astype(np.float32)).to(dev)
  outs = {}
  for tag, models, kern in (("ok", hips, "auto"), ("guarded", [hips[0], big[0]], "auto"), ("phase", [hips[0], big[0]], "phase")):
    agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, seed=9, search_kernel=kern, max_batch=B)
    import ctypes
    info = (ctypes.c_int32 * 10)()
    _lib.check(lib.rip_search_plan(agent._handle.raw, B, N, info, 10))
    plan = torch.empty(B, 4, 2, device=dev)
    lb = torch.empty(B, N, device=dev)
    _lib.check(lib.rip_search(agent._handle.raw, _lib.ptr(z), _lib.ptr(goal), _lib.ptr(agent._x0(B)), B, N, 10, 0, 3, 0.1, 1.0,
                              _lib.ptr(plan), None, _lib.ptr(lb), None, None, None, None, agent._handle.stream()))
    outs[tag] = (int(info[0]), plan.cpu(), lb.cpu())
  assert outs["ok"][0] == 4 and outs["guarded"][0] == 3 and outs["phase"][0] == 3
  assert torch.equal(outs["guarded"][1], outs["phase"][1]) and torch.equal(outs["guarded"][2], outs["phase"][2])
  assert torch.isfinite(outs["guarded"][1]).all()
  forced = RIPAgent(None, algorithm="WCM", models=[hips[0], big[0]], num_candidates=N, seed=9, search_kernel="split", max_batch=B)
  rc = lib.rip_search(forced._handle.raw, _lib.ptr(z), _lib.ptr(goal), _lib.ptr(forced._x0(B)), B, N, 10, 0, 3, 0.1, 1.0,
                      _lib.ptr(plan), None, _lib.ptr(lb), None, None, None, None, forced._handle.stream())
  assert rc == _lib.RIP_EINVAL and b"w * 2^8" in lib.rip_last_error()


def _geomean(xs):
  return float(np.exp(np.mean(np.log(np.maximum(np.asarray(xs, np.float64), 1e-30)))))


@pytest.mark.parametrize("zscale,wscale", [(1e-3, 1.0), (1.0, 1.0), (1e2, 1.0), (1e3, 1.0), (1.0, 0.1), (1.0, 10.0), (5e3, 1.0)])
def test_split_kernel_operand_ranges(dev, zscale, wscale):
  """the split-f16 kernel carries hidden states as two binary16 terms UNSCALED (|h| <= max(1, |z|)) —
  every other test drives it with O(1) z and weights.  One teacher-forced Adam step (algorithm MA: every model's adjoint
  reaches the gradient, no arg-best ties) against the oracle with z scaled by 1e-3 .. 1e3 (low terms near the subnormal
  range / hi terms in the thousands), flow weights x0.1 / x10 (saturated gates) and goals 100 m away.  Bar: 1e-4 relative
  to max(1, |posterior|) (gradients: to the candidate's largest entry) — or, where the conditioning of the inputs puts
  that out of reach of fp32 arithmetic itself, no worse than twice what the fp32-MFMA kernel (`phase`) reaches on the
  same launch.  z x 5e3 (max |z| ~ 2e4) is past the limit the split kernel accepts (2^14; binary16 ends at 65504): its
  prefix kernel raises the operand-range word, the split kernel returns and the fp32-MFMA kernel queued behind it runs
  the step — bit for bit the fp32 kernel's result, no silent inf.
  Round 5: in the ill-conditioned cases (the fp32 kernel itself beyond 1e-4) the statistic — a MAXIMUM over 3 x 128
  candidates of an error the search amplifies ~1e5-fold — moves by an order of magnitude from one input seed to the
  next for ANY kernel: round 4's kernel, which passed on seed 11, is at 3.5x / 7.8x the fp32 kernel's posterior /
  gradient error on seed 12 (profiles/r5/range_seeds_v1.log; the logarithm of the ratio scatters with sigma ~ 1).  Those
  cases therefore run SIX seeds and gate the geometric mean of (split error / fp32-kernel error) at 2 (measured over
  seeds 11..16 at weights x 10: 1.02 / 1.15 for round 4's kernel, 1.34 / 1.06 for round 5's), every single seed at 16x
  (round 6; the recorded per-seed ratios are next to the assertion): the defect this gate exists for — round 5's first forward step left the WEIGHTS' low terms unscaled,
  2^-24-quantised — was 780x on the gradients at z x 100 and fails both.
  Found with this test and fixed: `pow2_scale` overflowed to inf for candidates whose gate gradients had all but
  vanished (NaN gradients at z x 1e3), and `goal_ll` returned -inf at |y| ~ 5e4 (all kernels; flow_math.h)."""
  first = _operand_range_case(dev, zscale, wscale, 11)
  e_post, e_grad = first
  if max(e_post["phase"], e_grad["phase"]) <= 1e-4 or zscale >= 5e3:  # well conditioned (or the guard case): one seed, strict
    assert e_post["split"] <= max(1e-4, 2.0 * e_post["phase"])
    assert e_grad["split"] <= max(1e-4, 2.0 * e_grad["phase"])
    return
  runs = [first] + [_operand_range_case(dev, zscale, wscale, sd) for sd in (12, 13, 14, 15, 16)]
  for name, idx in (("posteriors", 0), ("gradients", 1)):
    ratios = [max(r[idx]["split"], 1e-4) / max(r[idx]["phase"], 1e-4) for r in runs]
    print("  %s: split / fp32-kernel error over seeds 11..16: %s, geometric mean %.2f" %
          (name, " ".join("%.2f" % v for v in ratios), _geomean(ratios)))
    # Round 6: the per-seed cap follows the recorded data instead of "3.4 sigma".  Per-seed ratios, seeds 11..16.
    # Weights x 10 (profiles/r5/range_seeds_v1.log) — round 4's kernel: posteriors 0.86 3.46 2.56 0.43 0.46 0.75, gradients
    # 0.94 7.76 2.60 0.56 0.40 0.53.  The shipped kernel (profiles/r6/outliers_v1.log) — weights x 10: posteriors 1.96 3.86
    # 3.52 0.51 0.68 0.63, gradients 2.02 4.61 0.75 1.00 0.35 0.57; z x 100: posteriors 0.19 0.54 0.92 0.94 1.00 4.00,
    # gradients 0.26 1.01 2.57 12.66 0.93 0.79; z x 1000: posteriors 0.72 1.00 0.83 0.30 1.07 0.43, gradients 1.96 2.27 0.94
    # 1.41 3.28 1.25.  The largest ratio a correct build has shown is 12.7 (the advisor's 10 would fail what ships): the
    # cap is 16 (it was 30); the defect the gate exists for was 780x.
    assert _geomean(ratios) <= 2.0 and max(ratios) <= 16.0, (name, ratios)


def _operand_range_case(dev, zscale, wscale, seed):
  """One teacher-forced MA step of the split and the fp32-MFMA kernel against the oracle; returns the two kernels' largest
  relative posterior / gradient errors over three observations."""
  from oatomobile_amd import _lib, RIPAgent
  from oracle import reference_cpu as O
  K, N, S, algo = 3, 128, 24, "MA"  # S x N = 3072 >= 1280: what `auto` would give the split kernel as well
  hips, refs = _scaled_flow_models([300 + k for k in range(K)], wscale, dev)
  rng = np.random.default_rng(seed)
  z_np = (np.abs(rng.normal(size=(K, S, 64))) * zscale).astype(np.float32)  # a ReLU output: non-negative
  z_np[:, :, ::7] = 0.0
  goal_np = (np.cumsum(np.abs(rng.normal(size=(S, 10, 2))) * 2.0, axis=1) + 100.0).astype(np.float32)
  x_np = rng.normal(size=(S, N, 4, 2)).astype(np.float32)
  z, goal, x = (torch.from_numpy(a).to(dev) for a in (z_np, goal_np, x_np))
  lib = _lib.load()

  def one_step(kernel):
    agent = RIPAgent(None, algorithm=algo, models=hips, num_candidates=N, seed=9, search_kernel=kernel, max_batch=S)
    handle = agent._handle
    lb = torch.empty(S, N, device=dev)
    tp = torch.empty(1, K, S, N, device=dev)
    tg = torch.empty(1, S, N, 4, 2, device=dev)
    _lib.check(lib.rip_search(handle.raw, _lib.ptr(z), _lib.ptr(goal), _lib.ptr(x), S, N, 10, _lib.ALGORITHMS[algo], 1, 0.1,
                              1.0, None, None, _lib.ptr(lb), None, _lib.ptr(tp), None, _lib.ptr(tg), handle.stream()))
    return tp.cpu().numpy()[0].transpose(1, 0, 2), lb.cpu().numpy(), tg.cpu().numpy()[0]   # [S,K,N], [S,N], [S,N,4,2]

  post_h, loss_h, grad_h = one_step("split")
  post_p, loss_p, grad_p = one_step("phase")
  bad = ~np.isfinite(post_h)
  assert not bad.any() and np.isfinite(loss_h).all() and np.isfinite(grad_h).all(), (
      "non-finite results: %d posteriors (first at [s,k,n] = %s), %d losses, %d gradient entries" %
      (int(bad.sum()), np.argwhere(bad)[:4].tolist(), int((~np.isfinite(loss_h)).sum()), int((~np.isfinite(grad_h)).sum())))
  if float(z_np.max()) >= 16384.0:  # past the split kernel's range: the fp32-MFMA kernel ran, bit for bit
    assert np.array_equal(post_h, post_p) and np.array_equal(loss_h, loss_p) and np.array_equal(grad_h, grad_p)
  # the oracle, one observation (= one z row per model) at a time
  e_post, e_grad = {"split": 0.0, "phase": 0.0}, {"split": 0.0, "phase": 0.0}
  for b in (0, S // 2, S - 1):
    zs = [torch.from_numpy(z_np[k, b:b + 1]) for k in range(K)]
    res = O.rip_search(refs, zs, torch.from_numpy(goal_np[b:b + 1]), torch.from_numpy(x_np[b]), algorithm=algo, num_steps=1)
    post_o, grad_o = res["trace_post"].numpy()[0], res["trace_grad"].numpy()[0]   # [K,N], [N,4,2]
    gmax = np.maximum(1.0, np.abs(grad_o).max(axis=(1, 2), keepdims=True))
    for name, post, grad in (("split", post_h, grad_h), ("phase", post_p, grad_p)):
      e_post[name] = max(e_post[name], float((np.abs(post[b] - post_o) / np.maximum(1.0, np.abs(post_o))).max()))
      e_grad[name] = max(e_grad[name], float((np.abs(grad[b] - grad_o) / gmax).max()))
  print("z x %g, flow weights x %g, seed %d: relative error of the posteriors split %.3g / fp32 kernel %.3g, of the gradients %.3g / %.3g" %
        (zscale, wscale, seed, e_post["split"], e_post["phase"], e_grad["split"], e_grad["phase"]))
  return e_post, e_grad


def test_g7_dim_forward(golden, dev):
  g = golden("g7_dim_forward.npz")
  m = hip_model(7, dev)
  for B, os_ in ((1, 70), (3, 71)):
    obs_list = [synth_observation(np.random.default_rng(os_ + 10 * b)) for b in range(B)]
    ctx = ctx_tensors(obs_list, dev)
    goal = torch.stack([torch.from_numpy(o["goal"][:, :2].copy()) for o in obs_list]).to(dev)
    for with_goal in (0, 1):
      tag = "B%d_goal%d" % (B, with_goal)
      y = m(num_steps=20, goal=goal if with_goal else None, lr=5e-2, epsilon=1.0,
            x0=torch.from_numpy(g["x0_" + tag]), **ctx)
      err = float(np.abs(y.cpu().numpy() - g["y_" + tag]).max())
      print("g7 %s: max|dy| = %.3g" % (tag, err))
      np.testing.assert_allclose(y.cpu().numpy(), g["y_" + tag], atol=TOL)


def test_g8_scores(golden, dev):
  from oatomobile_amd import _lib, RIPAgent
  g = golden("g8_scores.npz")
  models = [hip_model(100 + k, dev) for k in range(4)]
  agent = RIPAgent(None, algorithm="WCM", models=models)
  ob = synth_observation(np.random.default_rng(int(g["obs_seed"])))
  z = torch.from_numpy(g["zs"]).to(dev).reshape(4, 1, 64).contiguous()
  y = torch.from_numpy(g["y"]).to(dev).reshape(1, 128, 4, 2).contiguous()
  goal = torch.from_numpy(ob["goal"][None, :, :2].copy()).to(dev)
  S = torch.empty(4, 1, 128, device=dev)
  lib = _lib.load()
  _lib.check(lib.rip_score(agent._handle.raw, 0, 4, _lib.ptr(z), _lib.ptr(y), None, 1, 128, 0, 1.0, _lib.ptr(S),
                           _lib.current_stream()))
  np.testing.assert_allclose(S.cpu().numpy()[:, 0], g["S"], rtol=2e-5, atol=2e-3)
  _lib.check(lib.rip_score(agent._handle.raw, 0, 4, _lib.ptr(z), _lib.ptr(y), _lib.ptr(goal), 1, 128, 10, 1.0,
                           _lib.ptr(S), _lib.current_stream()))
  np.testing.assert_allclose(S.cpu().numpy()[:, 0], g["SG"], rtol=2e-5, atol=2e-3)


@pytest.mark.parametrize("kernel,algo,K,N", [("chain", "WCM", 4, 128), ("chain", "MA", 3, 16), ("chain", "BCM", 2, 5),
                                             ("chain", "WCM", 1, 7), ("chain", "WCM", 8, 8),
                                             ("phase", "WCM", 4, 128), ("phase", "MA", 3, 16), ("phase", "BCM", 2, 48),
                                             ("phase", "WCM", 1, 16), ("phase", "WCM", 8, 32), ("phase", "MA", 5, 16),
                                             ("split", "WCM", 4, 128), ("split", "MA", 3, 16), ("split", "BCM", 2, 48),
                                             ("split", "WCM", 1, 16), ("split", "WCM", 8, 32), ("split", "MA", 5, 16),
                                             ("pair", "WCM", 4, 128), ("pair", "MA", 3, 16), ("pair", "BCM", 2, 48),
                                             ("pair", "WCM", 1, 16), ("pair", "WCM", 8, 32), ("pair", "MA", 5, 16)])
def test_search_candidates_vs_oracle(dev, kernel, algo, K, N):
  """N candidates (BASELINE config 3 = K4/N128): every candidate's best loss and plan vs the oracle."""
  from oatomobile_amd import RIPAgent
  from oracle import reference_cpu as O
  models = [hip_model(200 + k, dev) for k in range(K)]
  refs = [oracle_model(200 + k) for k in range(K)]
  agent = RIPAgent(None, algorithm=algo, models=models, num_candidates=N, seed=5, search_kernel=kernel)
  ob = synth_observation(np.random.default_rng(77))
  lidar = torch.from_numpy(ob["lidar"]).to(dev)[None]
  vec = torch.tensor([[*ob["velocity"], ob["is_at_traffic_light"], ob["traffic_light_state"]]], device=dev)
  goal = torch.from_numpy(ob["goal"][None, :, :2].copy()).to(dev)
  plan, loss = agent.plan_batch(lidar, vec, goal, return_loss=True)
  _, res = O.rip_call(refs, ob["lidar"], ob["velocity"], ob["is_at_traffic_light"], ob["traffic_light_state"],
                      ob["goal"], x0=agent._x0_rows.cpu(), algorithm=algo)
  # the kernel's own per-step latents next to the oracle's, so that an outlier can be dated
  x_h = None
  if True:
    from oatomobile_amd import _lib
    zz = torch.empty(K, 1, 64, device=dev)
    lib = _lib.load()
    _lib.check(lib.rip_encode_raw(agent._handle.raw, _lib.ptr(lidar), 1, 200, 200, _lib.ptr(vec), 1, 0, K, 0, _lib.ptr(zz),
                                  agent._handle.stream()))
    tx = torch.empty(10, 1, N, 4, 2, device=dev)
    _lib.check(lib.rip_search(agent._handle.raw, _lib.ptr(zz), _lib.ptr(goal), _lib.ptr(agent._x0(1)), 1, N, goal.shape[1],
                              _lib.ALGORITHMS[algo], 10, 0.1, 1.0, None, None, None, None, None, _lib.ptr(tx), None,
                              agent._handle.stream()))
    x_h = tx.cpu().numpy()[:, 0]
  candidate_gate("%s %s K=%d N=%d" % (kernel, algo, K, N), loss.cpu().numpy()[0], res["loss_best"].numpy(),
                 plan.cpu().numpy()[0], res["plan"].numpy(), x_h=x_h, x_o=res["trace_x"].numpy())


def test_mfma_kernels_match_chain_kernel(dev):
  """The two search kernels are the same algorithm: per-candidate best losses and plans agree (B=3, N=32)."""
  from oatomobile_amd import RIPAgent
  models = [hip_model(300 + k, dev) for k in range(4)]
  obs = [synth_observation(np.random.default_rng(700 + i)) for i in range(3)]
  lidar = torch.stack([torch.from_numpy(o["lidar"]) for o in obs]).to(dev)
  vec = torch.tensor([[*o["velocity"], o["is_at_traffic_light"], o["traffic_light_state"]] for o in obs], device=dev)
  goal = torch.stack([torch.from_numpy(o["goal"][:, :2].copy()) for o in obs]).to(dev)
  out = {}
  for kern in ("chain", "phase", "split"):
    agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=32, max_batch=3, seed=9, search_kernel=kern)
    plan, loss = agent.plan_batch(lidar, vec, goal, return_loss=True)
    out[kern] = (plan.cpu().numpy(), loss.cpu().numpy())
  for kern in ("phase", "split"):
    close = np.abs(out["chain"][1] - out[kern][1]) <= 1e-3 + 1e-4 * np.abs(out["chain"][1])
    print("%s vs chain: %.4f of %d candidates within tolerance" % (kern, close.mean(), close.size))
    assert close.mean() >= 0.99, kern
    np.testing.assert_allclose(out["chain"][0], out[kern][0], atol=TOL, err_msg=kern)
  with pytest.raises(Exception):
    RIPAgent(None, algorithm="WCM", models=models, num_candidates=5, search_kernel="phase").plan_batch(
        lidar[:1].contiguous(), vec[:1].contiguous(), goal[:1].contiguous())


@pytest.mark.parametrize("algo", ["WCM", "MA", "BCM"])
def test_model_parallel_scoring_layout(dev, algo):
  """BASELINE config 4 layout on one GPU: two K_local=2 scorers (as two ranks would hold) + concatenation in
  shard order == one K=4 scorer; aggregation kernel vs numpy; arg-best plan vs the oracle's scores."""
  from oatomobile_amd import distributed as D
  from oracle import reference_cpu as O
  K, N = 4, 96
  models = [hip_model(400 + k, dev) for k in range(K)]
  refs = [oracle_model(400 + k) for k in range(K)]
  obs = [synth_observation(np.random.default_rng(900 + i)) for i in range(2)]
  lidar = torch.stack([torch.from_numpy(o["lidar"]) for o in obs]).to(dev)
  vec = torch.tensor([[*o["velocity"], o["is_at_traffic_light"], o["traffic_light_state"]] for o in obs], device=dev)
  goal = torch.stack([torch.from_numpy(o["goal"][:, :2].copy()) for o in obs]).to(dev)
  plans = torch.from_numpy(np.cumsum(np.abs(np.random.default_rng(5).normal(size=(2, N, 4, 2))) * 2, axis=2).astype(np.float32)).to(dev)
  full = D.ModelParallelScorer(models, K, algorithm=algo, max_batch=2, device=dev)
  S_full = full.local_scores(lidar, vec, goal, plans)
  halves = []
  for r in range(2):
    b, e = D.shard_range(K, r, 2)
    sc = D.ModelParallelScorer.__new__(D.ModelParallelScorer)  # a rank of a 2-rank job, without a process group
    from oatomobile_amd import _lib
    sc._lib, sc._group, sc._algorithm, sc._epsilon, sc._k_total, sc._device = _lib, None, algo, 1.0, K, dev
    sc._models = models[b:e]
    sc._handle = _lib.Handle(e - b, 2, 2, 0)
    for k, m in enumerate(sc._models):
      sc._handle.load_model(k, m.packed_weights())
    halves.append(sc.local_scores(lidar, vec, goal, plans))
  S_cat = torch.cat(halves, 0)
  np.testing.assert_allclose(S_cat.cpu().numpy(), S_full.cpu().numpy(), rtol=1e-6, atol=1e-5)
  loss, best = full.aggregate(S_full)
  Sn = -S_full.cpu().numpy()
  ref_loss = {"WCM": Sn.min(0), "BCM": Sn.max(0), "MA": Sn.mean(0)}[algo]
  np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, rtol=1e-6, atol=1e-5)
  np.testing.assert_array_equal(best.cpu().numpy(), ref_loss.argmin(1))
  plan, best2, _ = full(lidar, vec, goal, plans)
  np.testing.assert_array_equal(plan.cpu().numpy(), plans.cpu().numpy()[np.arange(2), ref_loss.argmin(1)])
  # scores against the oracle (observation 0)
  ob = obs[0]
  ctx = ctx_tensors([ob], dev)
  zs = [O.params(m, **{k: v.cpu() for k, v in ctx.items()}) for m in refs]
  So = O.rip_scores(refs, zs, plans[0].cpu(), torch.from_numpy(ob["goal"][None, :, :2].copy())).numpy()
  np.testing.assert_allclose(S_full.cpu().numpy()[:, 0], So, rtol=2e-5, atol=2e-3)


def test_full_size_properties(dev):
  """Size-independent properties at BASELINE's full sizes (config 4: K=8, N=512; config 5-like row counts)."""
  from oatomobile_amd import RIPAgent, _lib
  from oracle import reference_cpu as O
  # (1) flow round trip and determinant consistency on 10 000 rows
  m = hip_model(1, dev)
  rng = np.random.default_rng(11)
  n = 10000
  z = torch.from_numpy(np.maximum(rng.normal(size=(n, 64)), 0).astype(np.float32)).to(dev)
  x = torch.from_numpy(rng.normal(size=(n, 4, 2)).astype(np.float32)).to(dev)
  y, lad_f = m._forward(x, z)
  xr, lp, lad_i = m._inverse(y, z)
  np.testing.assert_allclose(xr.cpu().numpy(), x.cpu().numpy(), atol=TOL)
  np.testing.assert_allclose(lad_f.cpu().numpy(), lad_i.cpu().numpy(), atol=TOL)
  np.testing.assert_allclose(lp.cpu().numpy(), -0.5 * (x.cpu().numpy().reshape(n, -1)**2).sum(1) - 4 * np.log(2 * np.pi),
                             rtol=1e-5, atol=1e-3)
  # (2) config 4 size: K=8 models, N=512 candidates, against the oracle and under a permutation of the candidates
  K, N = 8, 512
  models = [hip_model(500 + k, dev) for k in range(K)]
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, seed=3)
  ob = synth_observation(np.random.default_rng(123))
  lidar = torch.from_numpy(ob["lidar"]).to(dev)[None]
  vec = torch.tensor([[*ob["velocity"], ob["is_at_traffic_light"], ob["traffic_light_state"]]], device=dev)
  goal = torch.from_numpy(ob["goal"][None, :, :2].copy()).to(dev)
  plan, loss = agent.plan_batch(lidar, vec, goal, return_loss=True)
  assert torch.isfinite(plan).all() and torch.isfinite(loss).all()
  assert float(loss.max()) < 1000.0  # every candidate improved on the sentinel (rip/agent.py:100)
  refs = [oracle_model(500 + k) for k in range(K)]
  _, res = O.rip_call(refs, ob["lidar"], ob["velocity"], ob["is_at_traffic_light"], ob["traffic_light_state"], ob["goal"],
                      x0=agent._x0_rows.cpu(), algorithm="WCM")  # all 512 candidates (round 6; the oracle side is seconds)
  candidate_gate("full size K=8 N=512", loss.cpu().numpy()[0], res["loss_best"].numpy(), plan.cpu().numpy()[0], res["plan"].numpy())
  perm = torch.randperm(N, generator=torch.Generator().manual_seed(0))
  agent._x0_rows = agent._x0_rows[perm.to(dev)].contiguous()
  agent._x0_cache = {}
  plan_p, loss_p = agent.plan_batch(lidar, vec, goal, return_loss=True)
  np.testing.assert_allclose(loss_p.cpu().numpy()[0], loss.cpu().numpy()[0][perm.numpy()], rtol=1e-6, atol=1e-6)
  np.testing.assert_allclose(plan_p.cpu().numpy(), plan.cpu().numpy(), atol=1e-6)


def test_batched_act_matches_single(dev):
  """plan_batch over B observations == B single calls (observation-parallel replay)."""
  from oatomobile_amd import RIPAgent
  models = [hip_model(300 + k, dev) for k in range(4)]
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=8, max_batch=5)
  obs = [synth_observation(np.random.default_rng(500 + i)) for i in range(5)]
  lidar = torch.stack([torch.from_numpy(o["lidar"]) for o in obs]).to(dev)
  vec = torch.tensor([[*o["velocity"], o["is_at_traffic_light"], o["traffic_light_state"]] for o in obs], device=dev)
  goal = torch.stack([torch.from_numpy(o["goal"][:, :2].copy()) for o in obs]).to(dev)
  plans = agent.plan_batch(lidar, vec, goal).cpu().numpy()
  for i, o in enumerate(obs):
    single = agent.plan_batch(lidar[i:i + 1].contiguous(), vec[i:i + 1].contiguous(), goal[i:i + 1].contiguous()).cpu().numpy()[0]
    np.testing.assert_allclose(plans[i], single, atol=1e-6)


def test_replay_cached_observations(dev, tmp_path):
  """BASELINE config 5 in miniature: .npz episode -> batched replay == per-observation __call__."""
  from oatomobile_amd import RIPAgent, replay
  from oatomobile_amd.agents import interpolate_plan
  models = [hip_model(300 + k, dev) for k in range(2)]
  agent = RIPAgent(None, algorithm="MA", models=models, num_candidates=4, max_batch=4)
  rng = np.random.default_rng(3)
  ep = replay.Episode(str(tmp_path), "ep")
  obs = []
  for i in range(6):
    o = synth_observation(np.random.default_rng(800 + i))
    fut = np.cumsum(np.abs(rng.normal(size=(80, 3))), axis=0).astype(np.float32)
    ep.append("t%d" % i, lidar=o["lidar"], velocity=o["velocity"], is_at_traffic_light=o["is_at_traffic_light"],
              traffic_light_state=o["traffic_light_state"], player_future=fut)
    obs.append((o, fut))
  plans = replay.replay(agent, ep.files(), batch_size=4)
  assert plans.shape == (6, 4, 2)
  for i, (o, fut) in enumerate(obs):
    single = dict(o)
    single["goal"] = np.c_[replay.goal_from_future(fut), np.zeros((10, 1), np.float32)]
    np.testing.assert_allclose(interpolate_plan(plans[i]), agent(single), atol=1e-5)


def test_four_channel_bev(dev):
  """BASELINE.json quotes a 200x200x4 BEV; the reference sensor has 2 channels.  C=4 parity vs the oracle."""
  from oracle import reference_cpu as O
  m = hip_model(31, dev, in_channels=4)
  mo = O.OracleImitativeModel.from_numpy_state_dict(W.synthetic_state_dict(31, 4), in_channels=4)
  obs = [synth_observation(np.random.default_rng(40 + i), C=4) for i in range(2)]
  ctx = ctx_tensors(obs, dev)
  z = m._params(**ctx).cpu().numpy()
  zo = O.params(mo, **{k: v.cpu() for k, v in ctx.items()}).numpy()
  np.testing.assert_allclose(z, zo, atol=TOL)


def test_abi_error_paths(dev):
  import ctypes
  from oatomobile_amd import _lib
  lib = _lib.load()
  h = ctypes.c_void_p(0)
  assert lib.rip_create(ctypes.byref(h), 0, 2, 1, 1, 0) == -1 and b"K=0" in lib.rip_last_error()
  assert lib.rip_create(ctypes.byref(h), 2, 2, 1, 1, 99) == -1
  assert lib.rip_create(ctypes.byref(h), 2, 2, 1, 0, 0) == -1 and b"max_candidates" in lib.rip_last_error()
  hd = _lib.Handle(2, 2, 1, 0)
  z = torch.zeros(2, 1, 64, device=dev)
  x0 = torch.zeros(1, 1, 4, 2, device=dev)
  rc = lib.rip_search(hd.raw, _lib.ptr(z), None, _lib.ptr(x0), 1, 1, 0, 0, 10, 0.1, 1.0, None, None, None, None, None,
                      None, None, None)
  assert rc == -3 and b"no weights" in lib.rip_last_error()
  with pytest.raises(_lib.RipError):
    hd.load_model(0, np.zeros(10, np.float32))
  hd.close()


def test_g9_lidar_bev_bit_exact(golden, dev):
  """rip_lidar_bev (carla_lidar_measurement_to_ndarray, utils/carla.py:165-233) against the golden
  fixtures from the reference function and against the oracle on a ragged batch: integer work, bit-exact."""
  from oatomobile_amd import lidar_to_bev
  from oracle import lidar as L
  g = golden("g9_lidar.npz")
  clouds = [g["points%d" % i] for i in range(4)]
  batch = lidar_to_bev(clouds, device=dev).cpu().numpy()
  assert batch.shape == (4, 200, 200, 2) and batch.dtype == np.float32
  for i in range(4):
    np.testing.assert_array_equal(batch[i], g["bev%d" % i], err_msg="cloud %d" % i)
  np.testing.assert_array_equal(lidar_to_bev(clouds[0], device=dev).cpu().numpy(), g["bev0"])  # single cloud
  # larger ragged batch, sizes 0 .. 120k points, values straddling every edge
  rng = np.random.default_rng(90)
  big = []
  for i, n in enumerate([0, 1, 63, 1024, 1025, 40000, 120000, 7]):
    pts = np.c_[rng.uniform(-55, 56, size=(n, 2)), rng.uniform(-4, -1, size=(n, 1))].astype(np.float32)
    pts[::5, 2] = -2.5
    pts[1::7, 0] = L.bev_edges().astype(np.float32)[rng.integers(0, 201, size=len(pts[1::7]))]
    big.append(pts)
  out = lidar_to_bev(big, device=dev).cpu().numpy()
  for i, pts in enumerate(big):
    np.testing.assert_array_equal(out[i], L.lidar_to_bev(pts), err_msg="ragged cloud %d (%d points)" % (i, len(pts)))
  # property at full size: the histogram of a concatenation is the clipped sum (counts <= 5 per cell and channel)
  assert out.min() >= 0.0 and out.max() <= 1.0 and set(np.unique(out)).issubset({np.float32(k / 5) for k in range(6)})


def test_g10_cil_forward_and_agent(golden, dev):
  """BehaviouralModel.forward / CILAgent.__call__ through rip_encode + rip_cil_decode against the
  golden fixtures from the reference classes and against the oracle on a fresh batch.  1e-4 fp32 relative to the plan
  scale (40 residual steps reach tens of metres)."""
  from oatomobile_amd import BehaviouralModel, CILAgent, weights
  from oracle import cil as C
  g = golden("g10_cil.npz")
  ws = int(g["weight_seed"])
  m = BehaviouralModel.synthetic(ws).to(dev)
  ctx = {k[4:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("ctx_")}
  y = m(**ctx).cpu().numpy()
  assert y.shape == (6, 40, 2)
  np.testing.assert_allclose(y, g["y"], rtol=1e-4, atol=1e-4 * np.abs(g["y"]).max())
  # agent level, one observation per command branch
  agent = CILAgent(None, model=m, device=dev)
  for i in range(3):
    ob = synth_observation(np.random.default_rng(int(g["agent_obs_seed%d" % i])))
    ob["goal"] = np.asarray(ob["goal"], np.float32).copy()
    ob["goal"][-1, :2] = g["agent_goal_last%d" % i]
    plan = agent(dict(ob))
    assert plan.shape == (39, 3)
    np.testing.assert_allclose(plan, g["agent_plan%d" % i], rtol=1e-4, atol=1e-4 * np.abs(g["agent_plan%d" % i]).max())
  # a larger batch against the oracle (fresh weights), including the transform with the STOP -> 0 rewrite
  mo = C.OracleBehaviouralModel.from_numpy_state_dict(weights.synthetic_cil_state_dict(31))
  mh = BehaviouralModel.synthetic(31).to(dev)
  rng = np.random.default_rng(310)
  B = 37
  raw = dict(lidar=rng.random((B, 2, 200, 200), dtype=np.float32) * (rng.random((B, 2, 200, 200)) < 0.1),
             velocity=rng.normal(0, 3, size=(B, 3)).astype(np.float32),
             is_at_traffic_light=rng.integers(0, 2, size=(B, 1)).astype(np.float32),
             traffic_light_state=rng.integers(0, 4, size=(B, 1)).astype(np.float32),
             mode=rng.integers(0, 4, size=(B, 1)).astype(np.float32))
  so = mo.transform({k: torch.from_numpy(v.astype(np.float32).copy()) for k, v in raw.items()})
  sh = mh.transform({k: torch.from_numpy(v.astype(np.float32).copy()).to(dev) for k, v in raw.items()})
  np.testing.assert_array_equal(sh["mode"].cpu().numpy(), so["mode"].numpy())
  with torch.no_grad():
    yo = mo(**so).numpy()
  yh = mh(**sh).cpu().numpy()
  np.testing.assert_allclose(yh, yo, rtol=1e-4, atol=1e-4 * np.abs(yo).max())


# ---------------------------------------------------------------------------------------------------------
# round 2: the bench configuration as a whole, multi-GPU compositions on one GPU, ABI contract, online path
# ---------------------------------------------------------------------------------------------------------
def test_bench_configuration_parity(dev):
  """EXACTLY what bench.py times (BASELINE configs[2]: B = bench.DEFAULT_OBS_BATCH observations per step, K = 4 WCM, N = 128, bf16
  encoder, auto-selected fused encoder blocks and search kernel) against the oracle: the search is exact given z, so
  the oracle is fed the HIP bf16 z of 16 sampled observations; the plan-level effect of bf16 is REPORTED against
  the fp32 encoder on the same observations."""
  import bench
  from oatomobile_amd import RIPAgent, _lib
  from oracle import reference_cpu as O
  B, K, N = bench.DEFAULT_OBS_BATCH, 4, 128
  seeds = [100 + k for k in range(K)]
  models = [hip_model(s_, dev, max_batch=1) for s_ in seeds]
  refs = [oracle_model(s_) for s_ in seeds]
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, max_batch=B, seed=0, device=dev,
                   encoder_dtype="bf16")
  lidar, vec, goal = (torch.from_numpy(a).to(dev) for a in bench.synth_batch(np.random.default_rng(1000), B, 2))
  plan, loss = agent.plan_batch(lidar, vec, goal, return_loss=True)
  assert torch.isfinite(plan).all() and float(loss.max()) < 1000.0
  lib = _lib.load()
  z = torch.empty(K, B, 64, device=dev)
  _lib.check(lib.rip_encode_raw(agent._handle.raw, _lib.ptr(lidar), 1, 200, 200, _lib.ptr(vec), B, 0, K, 1, _lib.ptr(z),
                                agent._handle.stream()))
  z = z.cpu()
  idx = np.random.default_rng(7).choice(B, size=16, replace=False)
  plan_h, loss_h = plan.cpu().numpy(), loss.cpu().numpy()
  x0 = agent._x0_rows.cpu()
  worst_plan, frac = 0.0, []
  for b in idx:
    res = O.rip_search(refs, [z[k, b:b + 1] for k in range(K)], goal[b:b + 1].cpu(), x0, algorithm="WCM")
    lo = res["loss_best"].numpy()
    close = np.abs(loss_h[b] - lo) <= 1e-3 + 1e-4 * np.abs(lo)
    frac.append(close.mean())
    err = candidate_gate("bench configuration, observation %d" % b, loss_h[b], lo, plan_h[b], res["plan"].numpy())
    worst_plan = max(worst_plan, err or 0.0)
  print("bench config vs oracle (given the HIP bf16 z): candidates within tolerance %.4f (min over obs %.4f), "
        "worst winner-plan error %.3g m" % (np.mean(frac), np.min(frac), worst_plan))
  assert np.min(frac) >= 0.99
  # plan-level effect of the bf16 encoder (reported; the bf16 z differs from the fp32 z by up to ~6 % of max|z|)
  agent32 = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, max_batch=16, seed=0, device=dev)
  sel = torch.from_numpy(idx).to(dev)
  plan32 = agent32.plan_batch(lidar[sel].contiguous(), vec[sel].contiguous(), goal[sel].contiguous()).cpu().numpy()
  d = np.abs(plan_h[idx] - plan32)
  print("bf16-encoder vs fp32-encoder plans (16 observations): max |d| = %.3g m, mean |d| = %.3g m, plan scale %.3g m"
        % (d.max(), d.mean(), np.abs(plan32).max()))
  assert np.isfinite(d).all() and d.mean() < 0.05 * max(1.0, np.abs(plan32).max())


def test_candidate_parallel_halves_equal_whole(dev):
  """SURVEY §8e candidate-parallel mode on one GPU: two ranks' shares (N/2 candidates each, as a 2-rank job would
  hold them) reduced with `reduce_rank_winners` == one GPU searching all N candidates."""
  from oatomobile_amd import RIPAgent
  from oatomobile_amd import distributed as D
  K, N, B = 3, 64, 2
  models = [hip_model(600 + k, dev) for k in range(K)]
  obs = [synth_observation(np.random.default_rng(910 + i)) for i in range(B)]
  lidar = torch.stack([torch.from_numpy(o["lidar"]) for o in obs]).to(dev)
  vec = torch.tensor([[*o["velocity"], o["is_at_traffic_light"], o["traffic_light_state"]] for o in obs], device=dev)
  goal = torch.stack([torch.from_numpy(o["goal"][:, :2].copy()) for o in obs]).to(dev)
  whole = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, max_batch=B, seed=11, search_kernel="phase")
  plan_w, loss_w = whole.plan_batch(lidar, vec, goal, return_loss=True)
  recs, losses = [], []
  for r in range(2):
    cp = D.CandidateParallelRIP(models, N, algorithm="WCM", seed=11, max_batch=B, device=dev, search_kernel="phase",
                                rank=r, world=2)
    loss, plans = cp.local_search(lidar, vec, goal)
    losses.append(loss)
    plan_l, idx_l = D.select_best_plan(loss, plans)
    recs.append(torch.cat([loss.gather(1, idx_l[:, None]), plan_l.reshape(B, 8), (idx_l + cp._begin).float()[:, None]], 1))
    # a single "rank" object called alone is the world-1 composition of its own share
    p1, i1, l1 = cp(lidar, vec, goal)
    assert torch.equal(i1, idx_l + cp._begin)
  np.testing.assert_allclose(torch.cat(losses, 1).cpu().numpy(), loss_w.cpu().numpy(), rtol=1e-6, atol=1e-6)
  plan_c, idx_c, best_c = D.reduce_rank_winners(torch.stack(recs))
  np.testing.assert_array_equal(idx_c.cpu().numpy(), loss_w.argmin(1).cpu().numpy())
  np.testing.assert_allclose(plan_c.cpu().numpy(), plan_w.cpu().numpy(), atol=1e-6)
  np.testing.assert_allclose(best_c.cpu().numpy(), loss_w.min(1).values.cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("algo", ["WCM", "MA", "BCM"])
def test_model_parallel_gradient_mode(dev, algo):
  """SURVEY §8e gradient-mode model parallelism (BASELINE configs[3] layout) on one GPU: K = 4 models split over two
  emulated ranks (2 + 2; rank 1 also holds model 0's flow), blocks concatenated in shard order where a 2-rank job
  all-gathers, rank-replicated update — equal to (a) ONE rank holding all models and (b) `rip_search`'s
  wave-per-chain kernel, and within 1e-4 of the oracle per candidate."""
  from oatomobile_amd import RIPAgent
  from oatomobile_amd import distributed as D
  from oracle import reference_cpu as O
  K, N, B, S = 4, 16, 2, 10
  models = [hip_model(700 + k, dev) for k in range(K)]
  refs = [oracle_model(700 + k) for k in range(K)]
  obs = [synth_observation(np.random.default_rng(920 + i)) for i in range(B)]
  lidar = torch.stack([torch.from_numpy(o["lidar"]) for o in obs]).to(dev)
  vec = torch.tensor([[*o["velocity"], o["is_at_traffic_light"], o["traffic_light_state"]] for o in obs], device=dev)
  goal = torch.stack([torch.from_numpy(o["goal"][:, :2].copy()) for o in obs]).to(dev)
  # (a) one rank with everything
  one = D.ModelParallelRIP(models, K, num_candidates=N, algorithm=algo, seed=4, max_batch=B, device=dev, rank=0, world=1)
  plan1, best1, lb1 = one(lidar, vec, goal)
  # two emulated ranks, stepped in lock step
  ranks = [D.ModelParallelRIP(models[0:2], K, num_candidates=N, algorithm=algo, seed=4, max_batch=B, device=dev, rank=0, world=2),
           D.ModelParallelRIP(models[2:4], K, flow0=models[0], num_candidates=N, algorithm=algo, seed=4, max_batch=B,
                              device=dev, rank=1, world=2)]
  zl = [r.encode_local(lidar, vec) for r in ranks]
  z0 = torch.cat(zl, 0)[0].contiguous()
  states = []
  for r in ranks:
    x = r._x0_rows.unsqueeze(0).expand(B, -1, -1, -1).contiguous()
    states.append((x, torch.zeros_like(x), torch.zeros_like(x), x.clone(), torch.full((B, N), 1000.0, device=dev)))
  for step in range(S):
    blocks = [r.local_block(zl[i], z0, states[i][0]) for i, r in enumerate(ranks)]
    gathered = torch.cat(blocks, 0).contiguous()  # what all_gather_blocks returns on every rank
    for i, r in enumerate(ranks):
      r.update(gathered, z0, goal, step, states[i])
  outs = [r.finish(z0, states[i]) for i, r in enumerate(ranks)]
  for plan_r, best_r, lb_r in outs:  # replicated state: every rank ends bitwise where the single rank does
    assert torch.equal(lb_r, lb1) and torch.equal(best_r, best1) and torch.equal(plan_r, plan1)
  # (b) the fused single-GPU kernel
  agent = RIPAgent(None, algorithm=algo, models=models, num_candidates=N, max_batch=B, seed=4, search_kernel="chain")
  plan_s, loss_s = agent.plan_batch(lidar, vec, goal, return_loss=True)
  l1, ls = lb1.cpu().numpy(), loss_s.cpu().numpy()
  # same algorithm, another kernel's rounding; 32 candidates: the count is pinned by OUTLIER_CEILINGS (round 6), not by `frac`
  candidate_gate("model-parallel %s vs chain kernel" % algo, l1, ls, frac=0.9)
  srt = np.sort(ls, axis=1)
  for b in range(B):
    if srt[b, 1] - srt[b, 0] > 1e-3:
      np.testing.assert_allclose(plan1.cpu().numpy()[b], plan_s.cpu().numpy()[b], atol=5e-4)
  # (c) the oracle, observation 0
  ob = obs[0]
  _, res = O.rip_call(refs, ob["lidar"], ob["velocity"], ob["is_at_traffic_light"], ob["traffic_light_state"], ob["goal"],
                      x0=one._x0_rows.cpu(), algorithm=algo)
  lo = res["loss_best"].numpy()
  candidate_gate("model-parallel %s vs oracle, observation 0" % algo, lb1.cpu().numpy()[0], lo, frac=0.9)


def test_abi_contract_no_growth_and_validation(dev):
  """include/rip_hip.h: scratch is sized by rip_create (RIP_ESTATE beyond it, never a reallocation); dtype / shape /
  device of raw-pointer inputs are rejected in Python; a BEV that is not 200 x 200 is resized like
  F.interpolate; the caller's current device is left untouched."""
  from oatomobile_amd import RIPAgent, _lib
  from oracle import reference_cpu as O
  models = [hip_model(800 + k, dev) for k in range(2)]
  agent = RIPAgent(None, algorithm="MA", models=models, num_candidates=8, max_batch=2, seed=1)
  ob = synth_observation(np.random.default_rng(5))
  lidar = torch.from_numpy(ob["lidar"]).to(dev)[None]
  vec = torch.tensor([[*ob["velocity"], ob["is_at_traffic_light"], ob["traffic_light_state"]]], device=dev)
  goal = torch.from_numpy(ob["goal"][None, :, :2].copy()).to(dev)
  cur = torch.cuda.current_device()
  agent.plan_batch(lidar, vec, goal)
  assert torch.cuda.current_device() == cur
  with pytest.raises(ValueError, match="float32"):
    agent.plan_batch(lidar.double(), vec, goal)
  with pytest.raises(ValueError, match="float32"):
    agent.plan_batch(lidar, vec.half(), goal)
  with pytest.raises(ValueError, match="shape"):
    agent.plan_batch(lidar, vec[:, :4].contiguous(), goal)
  with pytest.raises(ValueError, match="shape"):
    agent.plan_batch(lidar, vec, goal[..., :1].contiguous())
  with pytest.raises(ValueError):
    agent.plan_batch(lidar.repeat(3, 1, 1, 1), vec.repeat(3, 1), goal.repeat(3, 1, 1))  # > max_batch
  with pytest.raises(RuntimeError):
    agent.plan_batch(lidar.cpu(), vec, goal)
  with pytest.raises(ValueError, match="lidar"):
    agent(dict(ob, lidar=ob["lidar"][..., :1]))
  # beyond the scratch rip_create sized: RIP_ESTATE, not a hipMalloc
  lib = _lib.load()
  z = torch.zeros(2, 1, 64, device=dev)
  x0 = torch.zeros(1, 32, 4, 2, device=dev)
  plan = torch.empty(1, 4, 2, device=dev)
  rc = lib.rip_search(agent._handle.raw, _lib.ptr(z), _lib.ptr(goal), _lib.ptr(x0), 1, 32, 10, 1, 10, 0.1, 1.0,
                      _lib.ptr(plan), None, None, None, None, None, None, agent._handle.stream())
  assert rc == -3 and b"max_candidates" in lib.rip_last_error()
  # any BEV size: 160 x 240 sensor grid vs the oracle (F.interpolate semantics, torch/transforms.py:39-44)
  rng = np.random.default_rng(6)
  ob2 = dict(ob, lidar=((rng.integers(0, 6, size=(160, 240, 2)) / 5.0) * (rng.random((160, 240, 2)) < 0.12)).astype(np.float32))
  out = agent(dict(ob2))
  refs = [oracle_model(800 + k) for k in range(2)]
  ref, _ = O.rip_call(refs, ob2["lidar"], ob2["velocity"], ob2["is_at_traffic_light"], ob2["traffic_light_state"],
                      ob2["goal"], x0=agent._x0_rows.cpu(), algorithm="MA")
  np.testing.assert_allclose(out, ref, atol=5e-4)


def test_agent_sees_weight_updates_and_graph_equals_eager(dev):
  """the agent's weight snapshot follows `load_state_dict()` / `refresh()`; `__call__` through the
  captured hipGraph == eager launches == plan_batch, call after call, with changing observations."""
  from oatomobile_amd import RIPAgent
  from oatomobile_amd.agents import interpolate_plan
  models = [hip_model(810 + k, dev) for k in range(2)]
  g_agent = RIPAgent(None, algorithm="MA", models=models, num_candidates=16, seed=2, graph=True)
  e_agent = RIPAgent(None, algorithm="MA", models=models, num_candidates=16, seed=2, graph=False)
  for i in range(4):
    ob = synth_observation(np.random.default_rng(30 + i))
    a, b = g_agent(dict(ob)), e_agent(dict(ob))
    np.testing.assert_array_equal(a, b)
    lidar = torch.from_numpy(ob["lidar"]).to(dev)[None]
    vec = torch.tensor([[*ob["velocity"], ob["is_at_traffic_light"], ob["traffic_light_state"]]], device=dev)
    goal = torch.from_numpy(ob["goal"][None, :, :2].copy()).to(dev)
    np.testing.assert_array_equal(a, interpolate_plan(g_agent.plan_batch(lidar, vec, goal).cpu().numpy()[0]))
  st = next(iter(g_agent._online.values()))
  print("online path: hipGraph captured = %s" % (st["graph"] is not None))
  # new weights for model 1: both agents must follow without being rebuilt
  before = g_agent(dict(ob))
  models[1].load_numpy_state_dict(W.synthetic_state_dict(999))
  after_g, after_e = g_agent(dict(ob)), e_agent(dict(ob))
  np.testing.assert_array_equal(after_g, after_e)
  assert np.abs(after_g - before).max() > 1e-6
  fresh = RIPAgent(None, algorithm="MA", models=models, num_candidates=16, seed=2, graph=False)
  np.testing.assert_array_equal(fresh(dict(ob)), after_g)


def test_g14_dim_agent_reference_recipe(golden, dev):
  """DIMAgent.__call__ (dim/agent.py:45-84) vs the reference agent's own [30,3] output (base sample pinned)."""
  from oatomobile_amd import DIMAgent
  g = golden("g14_dim_agent.npz")
  m = hip_model(int(g["weight_seed"]), dev)
  agent = DIMAgent(None, model=m, device=dev)
  for i in range(2):
    ob = synth_observation(np.random.default_rng(int(g["obs_seed%d" % i])))
    out = agent(dict(ob), x0=torch.from_numpy(g["x0_%d" % i]))
    assert out.shape == (30, 3) and out.dtype == np.float64
    print("g14 obs %d: max|d plan| = %.3g" % (i, np.abs(out - g["plan%d" % i]).max()))
    np.testing.assert_allclose(out, g["plan%d" % i], atol=TOL)


# ---------------------------------------------------------------------------------------------------------
# N3: the DIM training step (dim/train.py:175-213)
# ---------------------------------------------------------------------------------------------------------
def _rel_l2(a, b):
  return float(np.sqrt(((a.astype(np.float64) - b)**2).sum() / max((b.astype(np.float64)**2).sum(), 1e-30)))


def test_g15_train_step_vs_reference(golden, dev):
  """Two consecutive training steps (train mode: BatchNorm batch statistics + running-stat update, dropout mask and
  target perturbation replayed from the recording) against the reference's own model run through train_step.
  Held tightly: loss, z, BatchNorm running statistics, the whole-model gradient norm, and every recorded tensor that
  has no ReLU6 between it and the loss.  The encoder's gradients are defined up to the decisions at the ReLU6 kinks (an
  element within rounding of a kink is decided differently by any two implementations and moves a per-channel
  gradient by ~1/(B*H*W)): against this recording they are held in relative L2; the sharp per-element check of the
  backward kernels is test_train_backward_vs_oracle_same_kinks."""
  from oatomobile_amd import DIMTrainer
  g = golden("g15_train_step.npz")
  m = hip_model(int(g["weight_seed"]), dev)
  tr = DIMTrainer(m, lr=float(g["lr"]), max_batch=8, device=dev)
  for step in range(2):
    t = "s%d_" % step
    batch = {k: torch.from_numpy(g[t + k]).to(dev) for k in ("visual_features", "velocity", "is_at_traffic_light",
                                                               "traffic_light_state", "player_future")}
    loss = tr.backward(batch, y=torch.from_numpy(g[t + "y"]), dropout_mask=torch.from_numpy(g[t + "dropout_mask"]))
    print("g15 step %d: loss %.6f (reference %.6f)" % (step, float(loss), float(g[t + "loss"])))
    # (step 1 starts from step 0's parameters, which already carry the kink-decision and zero-gradient noise)
    np.testing.assert_allclose(float(loss), float(g[t + "loss"]), rtol=2e-5 if step == 0 else 2e-3)
    np.testing.assert_allclose(tr.z.cpu().numpy(), g[t + "z"], rtol=1e-4 if step == 0 else 5e-2, atol=2e-5 if step == 0 else 5e-2)
    grads = {k: v.cpu().numpy().copy() for k, v in tr.named_gradients().items()}
    gn = float(torch.linalg.vector_norm(tr.grads.double()))
    np.testing.assert_allclose(gn, float(g[t + "grad_norm"]), rtol=5e-3 if step == 0 else 0.15)
    tr.apply()
    params = {k: v.cpu().numpy() for k, v in tr.state_dict().items()}
    worst = 0.0
    for k in map(str, g["keys"]):
      smooth = not k.startswith("_encoder._model.features")  # classifier, merger, flow: no ReLU6 towards the loss
      if t + "grad:" + k in g.files:
        gref, gact = g[t + "grad:" + k].reshape(-1), grads[k].reshape(-1)
        pref, pact = g[t + "param:" + k].reshape(-1), params[k].reshape(-1)
      else:
        idx = g[t + "grad:" + k + ":idx"]
        gref, gact = g[t + "grad:" + k + ":val"], grads[k].reshape(-1)[idx]
        pref, pact = g[t + "param:" + k + ":val"], params[k].reshape(-1)[idx]
      if np.abs(gref).max() < 1e-6:
        continue  # mathematically zero gradient (a BN bias in front of another batch-statistics BN): rounding noise
      err = _rel_l2(gact, gref)
      worst = max(worst, err)
      if smooth and step == 0:
        np.testing.assert_allclose(gact, gref, rtol=1e-3, atol=1e-6 + 1e-4 * np.abs(gref).max(), err_msg=k)
        solid = np.abs(gref) > 1e-5 + 1e-3 * np.abs(gref).max()
        np.testing.assert_allclose(pact[solid], pref[solid], rtol=1e-4, atol=2e-5, err_msg="param:" + k)
      elif step == 0:
        assert err < 0.02, (k, err)
      # step 1 is not held per tensor: the FIRST Adam step moves every parameter by exactly +-lr according to the sign of
      # its gradient, so the ~0.5 % kink-decision deviations of step 0 flip the direction of the near-zero gradient
      # entries and the two models then differ by 2 lr in those coordinates (the CPU oracle against this recording
      # shows the same effect, test_oracle_golden.py)
    print("g15 step %d: worst relative L2 gradient deviation over %d recorded tensors: %.3g" % (step, len(g["keys"]), worst))
    for key in g.files:
      if key.startswith(t + "buffer:"):
        name = key[len(t + "buffer:"):]
        if name.endswith("num_batches_tracked"):
          assert int(params[name]) == int(g[key])
        else:
          np.testing.assert_allclose(params[name], g[key], rtol=1e-4 if step == 0 else 2e-2, atol=2e-6 if step == 0 else 1e-3)


def test_train_batch_statistics_are_the_same_bits_on_every_run(dev):
  """The BatchNorm reductions of the training step are two-stage (per-block partial columns, then a fixed-order sum in
  double: csrc/train.hip `colstats_kernel` / `stat_reduce_kernel`), not float atomics: two trainers stepping the same
  batch from the same weights must leave bit-identical running statistics for every layer in front of the first
  split-K convolution (at this batch size features.12's projection: K = 576 against 20 output tiles; those GEMMs add
  their K chunks with float atomics, so from there on the INPUTS of the reductions differ in the last place) and
  statistics / z / loss equal to rounding behind it."""
  from oatomobile_amd import DIMTrainer
  B = 12
  rng = np.random.default_rng(441)
  obs = [synth_observation(rng) for _ in range(B)]
  ctx = ctx_tensors(obs, dev)
  future = torch.from_numpy(np.cumsum(np.abs(rng.normal(size=(B, 4, 3))) * 2.0, axis=1).astype(np.float32))
  y = future[..., :2] + 1e-2 * torch.from_numpy(rng.normal(size=(B, 4, 2)).astype(np.float32))
  mask = torch.from_numpy(((rng.random((B, 1280)) >= 0.2) / 0.8).astype(np.float32))
  results = []
  for _ in range(2):
    tr = DIMTrainer(hip_model(34, dev), lr=1e-3, max_batch=16, device=dev)
    loss = tr.backward(dict(ctx, player_future=future.to(dev)), y=y, dropout_mask=mask, train=True)
    sd = tr.state_dict()
    stats = {k: v.cpu() for k, v in sd.items() if k.endswith("running_mean") or k.endswith("running_var")}
    results.append((float(loss), tr.z.cpu().clone(), stats))
    tr.close()
  (l0, z0, s0), (l1, z1, s1) = results
  assert len(s0) == 104  # 52 BatchNorm layers
  np.testing.assert_allclose(l0, l1, rtol=1e-6)
  np.testing.assert_allclose(z0.numpy(), z1.numpy(), rtol=1e-4, atol=1e-5)
  exact = 0
  for k in s0:
    block = int(k.split("features.")[1].split(".")[0])
    if block <= 11:
      assert torch.equal(s0[k], s1[k]), k
      exact += 1
    else:
      np.testing.assert_allclose(s0[k].numpy(), s1[k].numpy(), rtol=1e-4, atol=1e-6, err_msg=k)
  assert exact >= 60


@pytest.mark.parametrize("train", [True, False])
def test_train_backward_vs_oracle_same_kinks(dev, train):
  """The backward kernels, per element: the CPU oracle back-propagates with the ReLU6 kink decisions of the HIP forward
  (read back with rip_train_peek), so both differentiate the same piecewise-linear function; then every one of the 158
  parameter tensors' gradients must agree (batch-statistics BatchNorm and frozen BatchNorm).  Also reports how many of
  the ~13 M activations the two forwards decide differently."""
  from oatomobile_amd import DIMTrainer, arch
  from oracle import train_cpu as TC
  B = 9
  sd = W.synthetic_state_dict(33)
  m = hip_model(33, dev)
  mo = TC.trainable_model(sd)
  if not train:
    mo.eval()
  tr = DIMTrainer(m, lr=1e-3, max_batch=16, device=dev)
  rng = np.random.default_rng(330)
  obs = [synth_observation(rng) for _ in range(B)]
  ctx = ctx_tensors(obs, dev)
  future = torch.from_numpy(np.cumsum(np.abs(rng.normal(size=(B, 4, 3))) * 2.0, axis=1).astype(np.float32))
  y = future[..., :2] + 1e-2 * torch.from_numpy(rng.normal(size=(B, 4, 2)).astype(np.float32))
  mask = torch.from_numpy(((rng.random((B, 1280)) >= 0.2) / 0.8).astype(np.float32))
  loss = tr.backward(dict(ctx, player_future=future.to(dev)), y=y, dropout_mask=mask, train=train)
  layers = arch.conv_layers(2)
  posts = [tr.peek(i, "post").cpu() for i, l in enumerate(layers) if l.relu6]
  cpu = {k: v.cpu() for k, v in ctx.items()}
  args = (cpu["visual_features"], cpu["velocity"], cpu["is_at_traffic_light"], cpu["traffic_light_state"], y, mask)
  # the oracle's own kink decisions first: count the disagreements
  captured = []
  hooks = [mod.register_forward_hook(lambda md, inp, out: captured.append(out.detach())) for mod in TC.kink_modules(mo)]
  loss_o, z_o = TC.loss_and_grads(mo, *args)
  for hk in hooks:
    hk.remove()
  flips = sum(int((((a > 0) & (a < 6)) != ((b > 0) & (b < 6))).sum()) for a, b in zip(posts, captured))
  fwd = max(float((a - b).abs().max()) for a, b in zip(posts, captured))
  print("train=%s: %d of %d ReLU6 decisions differ between the HIP and the CPU forward (max |activation difference| %.2g)"
        % (train, flips, sum(a.numel() for a in posts), fwd))
  assert flips < 100 and fwd < 1e-3
  np.testing.assert_allclose(float(loss), float(loss_o), rtol=2e-5)
  np.testing.assert_allclose(tr.z.cpu().numpy(), z_o.numpy(), rtol=1e-4, atol=2e-5)
  # same kinks: per-element agreement of every gradient
  if train:
    mo.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})  # undo the running-stat update
  TC.set_kink_masks(mo, posts)
  TC.loss_and_grads(mo, *args)
  hg = tr.named_gradients()
  worst = 0.0
  gmax = max(float(p.grad.abs().max()) for p in mo.parameters())
  for k, p in mo.named_parameters():
    go, gh = p.grad.numpy(), hg[k].cpu().numpy()
    scale = np.abs(go).max()
    if scale < 1e-6 * gmax:
      continue  # mathematically zero (a BN bias in front of another batch-statistics BN): rounding noise on both sides
    worst = max(worst, np.abs(gh - go).max() / scale)
    np.testing.assert_allclose(gh, go, rtol=2e-3, atol=1e-6 + 3e-4 * scale, err_msg=k)
  print("train=%s: worst max|dgrad| / max|grad| over the parameter tensors with the same kinks: %.3g" % (train, worst))


def test_train_step_hand_back_to_inference(dev):
  """Three Adam steps (losses follow the oracle's), evaluate_step (dim/train.py:229-249) and the hand-back: the
  trained weights, written into the ImitativeModel, drive the INFERENCE kernels to what the oracle computes in eval
  mode from the same trained state_dict.  (The eval-mode function of two independently trained copies is not
  comparable beyond a few steps: the parameters with mathematically zero gradient — BN biases in front of another
  batch-statistics BN — random-walk by +-lr per step under Adam, which train mode cannot see and eval mode can.)"""
  from oatomobile_amd import DIMTrainer
  from oracle import reference_cpu as O
  from oracle import train_cpu as TC
  B = 9
  sd = W.synthetic_state_dict(34)
  m = hip_model(34, dev)
  mo = TC.trainable_model(sd)
  opt = TC.make_adam(mo, lr=1e-3)
  tr = DIMTrainer(m, lr=1e-3, max_batch=16, device=dev)
  nbt0 = tr.num_batches_tracked
  rng = np.random.default_rng(340)
  obs = [synth_observation(rng) for _ in range(B)]
  ctx = ctx_tensors(obs, dev)
  future = torch.from_numpy(np.cumsum(np.abs(rng.normal(size=(B, 4, 3))) * 2.0, axis=1).astype(np.float32))
  y = future[..., :2] + 1e-2 * torch.from_numpy(rng.normal(size=(B, 4, 2)).astype(np.float32))
  mask = torch.from_numpy(((rng.random((B, 1280)) >= 0.2) / 0.8).astype(np.float32))
  batch = dict(ctx, player_future=future.to(dev))
  cpu = {k: v.cpu() for k, v in ctx.items()}
  losses = []
  for _ in range(3):
    loss = tr.train_step(batch, y=y, dropout_mask=mask)
    loss_o, _ = TC.loss_and_grads(mo, cpu["visual_features"], cpu["velocity"], cpu["is_at_traffic_light"],
                                  cpu["traffic_light_state"], y, mask)
    opt.step()
    losses.append((float(loss), float(loss_o)))
    np.testing.assert_allclose(float(loss), float(loss_o), rtol=2e-3)
  print("3 Adam steps, loss HIP / oracle:", losses)
  assert losses[2][0] < losses[0][0]  # it trains
  assert tr.step_count == 3 and tr.num_batches_tracked == nbt0 + 3
  trained = {k: v.cpu().numpy() for k, v in tr.state_dict().items()}
  m2 = O.OracleImitativeModel.from_numpy_state_dict(trained)  # eval mode, the weights the HIP trainer produced
  with torch.no_grad():
    zo = O.params(m2, **cpu)
    _, lp, lad = O.flow_inverse(m2, future[..., :2], zo)
    ev_o = float(-(lp - lad).mean())
  ev = float(tr.evaluate_step(batch))
  print("evaluate_step loss %.5f (oracle on the same weights %.5f)" % (ev, ev_o))
  np.testing.assert_allclose(ev, ev_o, rtol=1e-4)
  np.testing.assert_allclose(tr.z.cpu().numpy(), zo.numpy(), rtol=1e-4, atol=1e-4)
  tr.sync_to_model()
  z_inf = m._params(**ctx).cpu().numpy()
  np.testing.assert_allclose(z_inf, zo.numpy(), rtol=1e-4, atol=1e-4)


def test_config4_k8_n512_on_the_mfma_kernel(dev):
  """BASELINE configs[3] on ONE GPU: K = 8 models, N = 512 candidates through the phase-sequential MFMA kernel (the
  wave-per-model pipeline stops at K = 4): candidates vs the oracle, scoring mode vs the search's own posteriors."""
  from oatomobile_amd import RIPAgent
  from oracle import reference_cpu as O
  K, N, B = 8, 512, 4
  models = [hip_model(500 + k, dev) for k in range(K)]
  refs = [oracle_model(500 + k) for k in range(K)]
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, max_batch=B, seed=3, search_kernel="split")
  obs = [synth_observation(np.random.default_rng(120 + i)) for i in range(B)]
  lidar = torch.stack([torch.from_numpy(o["lidar"]) for o in obs]).to(dev)
  vec = torch.tensor([[*o["velocity"], o["is_at_traffic_light"], o["traffic_light_state"]] for o in obs], device=dev)
  goal = torch.stack([torch.from_numpy(o["goal"][:, :2].copy()) for o in obs]).to(dev)
  plan, loss = agent.plan_batch(lidar, vec, goal, return_loss=True)
  assert torch.isfinite(plan).all() and float(loss.max()) < 1000.0
  for b in (0, 3):
    ob = obs[b]
    _, res = O.rip_call(refs, ob["lidar"], ob["velocity"], ob["is_at_traffic_light"], ob["traffic_light_state"], ob["goal"],
                        x0=agent._x0_rows.cpu()[:96], algorithm="WCM")
    candidate_gate("configs[3] K=8 N=512, observation %d (first 96 candidates)" % b, loss.cpu().numpy()[b, :96],
                   res["loss_best"].numpy(), x_o=None)
  # the same launch through the wave-per-chain kernel (any K): per-candidate best losses agree
  chain = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, max_batch=B, seed=3, search_kernel="chain")
  plan_c, loss_c = chain.plan_batch(lidar, vec, goal, return_loss=True)
  close = np.abs(loss.cpu().numpy() - loss_c.cpu().numpy()) <= 1e-3 + 1e-4 * np.abs(loss_c.cpu().numpy())
  print("phase kernel vs wave-per-chain kernel, %d candidates: %.5f within tolerance" % (close.size, close.mean()))
  assert close.mean() >= 0.99


def test_replay_512_cached_observations(dev, tmp_path):
  """BASELINE configs[4] at a size that still runs in seconds: 512 cached `.npz` datums (the reference's on-disk
  schema, datasets/carla.py:107-164) replayed in batches of 256 through the bench configuration (K = 4, N = 128, MFMA
  search), sampled observations against the oracle, batch-order independence, and the decode-inclusive rate."""
  import time
  from oatomobile_amd import RIPAgent, replay
  from oracle import reference_cpu as O
  K, N, F = 4, 128, 512
  models = [hip_model(100 + k, dev) for k in range(K)]
  refs = [oracle_model(100 + k) for k in range(K)]
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, max_batch=256, seed=0)
  ep = replay.Episode(str(tmp_path), "ep")
  rng = np.random.default_rng(44)
  futures = []
  for i in range(F):
    o = synth_observation(np.random.default_rng(4400 + i))
    fut = np.cumsum(np.abs(rng.normal(size=(80, 3))) * 0.4, axis=0).astype(np.float32)
    ep.append(lidar=o["lidar"], velocity=o["velocity"], is_at_traffic_light=o["is_at_traffic_light"],
              traffic_light_state=o["traffic_light_state"], player_future=fut)
    futures.append((o, fut))
  files = ep.files()
  t0 = time.perf_counter()
  plans = replay.replay(agent, files, batch_size=256)
  dt = time.perf_counter() - t0
  print("replay of %d cached datums: %.0f observations/s including np.load decode (batch 256)" % (F, F / dt))
  assert plans.shape == (F, 4, 2) and np.isfinite(plans).all()
  for i in (0, 255, 256, 511):
    o, fut = futures[i]
    goal = np.c_[replay.goal_from_future(fut), np.zeros((10, 1), np.float32)]
    _, res = O.rip_call(refs, o["lidar"], o["velocity"], o["is_at_traffic_light"], o["traffic_light_state"], goal,
                        x0=agent._x0_rows.cpu(), algorithm="WCM")
    srt = np.sort(res["loss_best"].numpy())
    if srt[1] - srt[0] > 1e-3:
      np.testing.assert_allclose(plans[i], res["plan"].numpy(), atol=TOL)
  # another batching of the same files gives the same plans (observations are independent)
  plans2 = replay.replay(agent, files[::-1], batch_size=128)[::-1]
  np.testing.assert_allclose(plans2, plans, atol=1e-5)
  # decode in worker processes (shared-memory batch buffers): same plans; the rate includes the workers' start-up
  # (spawn + import, seconds) on a job this small — bench.py's `replay` line measures the steady state
  nw = max(1, min(32, replay.effective_cpus() - 1))
  t0 = time.perf_counter()
  plans3 = replay.replay(agent, files, batch_size=256, workers=nw)
  dt = time.perf_counter() - t0
  print("same replay with %d decode processes: %.0f observations/s including their start-up" % (nw, F / dt))
  np.testing.assert_array_equal(plans3, plans)


def _run_bench_ranks(n, extra, steps=3, timeout=600):
  import json, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, RIP_BENCH_SHARE_GPU="1", RIP_BENCH_BACKEND="gloo")
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
    env.pop(k, None)
  out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", "1",
                        "--no-cpu-baseline", "--no-extras"] + extra, cwd=root, env=env, capture_output=True, text=True,
                       timeout=timeout)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, out.stdout[-2000:]  # rank 0 only
  return json.loads(lines[0])


def _run_bench_two_ranks(extra):
  return _run_bench_ranks(2, extra)


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu():
  """`python bench.py --gpus 2` (self-spawn under torch.distributed.run, barrier + max-over-ranks timing, rank 0
  prints the line) on a ONE-GPU box: both ranks on cuda:0 and gloo instead of RCCL (RCCL refuses two ranks on one
  device).  Checks the world > 1 code path and the JSON contract, not the numbers."""
  rec = _run_bench_two_ranks(["--obs-batch", "32"])
  assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
  assert rec["value"] > 0 and abs(rec["value"] - 2 * 32 * 3 / (rec["ms_per_step"] * 3e-3)) < 1e-6 * rec["value"]
  assert rec["roofline"]["frac"] is None or 0 < rec["roofline"]["frac"] < 1
  # N > 1 without --mode (what the driver runs): the same invocation also runs the compositions that need a collective
  assert rec["rccl"]["ranks_seen"] == 2 and rec["rccl"]["all_reduce_of_ones"] == 2.0
  for key, collectives in (("candidate_parallel", 1), ("model_parallel", 11)):  # K = 4 over 2 ranks: 10 Adam steps + z_0
    line = rec[key]
    assert "error" not in line, line
    assert line["calls_per_s"] > 0 and line["collectives_per_step"] == collectives, line
    assert line["max_abs_plan_diff_vs_single_gpu"] <= 1e-4, line
  assert rec["candidate_parallel"]["candidates_total"] == 2 * 128


@pytest.mark.gpu
def test_bench_two_gpus_over_rccl():
  """The same default invocation on TWO GPUs over RCCL (backend nccl, one rank per device): skipped — not passed — on
  a one-GPU box; no scaling figure exists until the driver has a multi-GPU node (multi-GPU)."""
  import json, subprocess, sys
  if torch.cuda.device_count() < 2:
    pytest.skip("needs two GPUs (this box has %d)" % torch.cuda.device_count())
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RIP_BENCH_SHARE_GPU", "RIP_BENCH_BACKEND")}
  out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--obs-batch", "64",
                        "--no-cpu-baseline", "--no-extras"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-3000:]
  rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
  assert rec["n_gpus"] == 2 and rec["backend"] == "nccl" and rec["world_size_seen"] == 2
  assert rec["rccl"] == {"backend": "nccl", "ranks_seen": 2, "all_reduce_of_ones": 2.0, "librccl_mapped": True}
  for key in ("candidate_parallel", "model_parallel"):
    assert "error" not in rec[key] and rec[key]["max_abs_plan_diff_vs_single_gpu"] <= 1e-4, rec[key]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["candidates", "models"])
def test_bench_parallel_modes_two_ranks(mode):
  """`bench.py --mode candidates|models --gpus 2`: CandidateParallelRIP / ModelParallelRIP across two PROCESSES
  (one-GPU hook: shared device, gloo, device all-gathers staged through the host).  The mode's own check — the
  plan of the distributed search against the single-GPU search of the same observations — is part of the line."""
  rec = _run_bench_two_ranks(["--mode", mode, "--obs-batch", "8"])
  assert rec["n_gpus"] == 2 and rec["value"] > 0
  assert rec["config"]["mode"] == mode
  assert rec["check"]["max_abs_plan_diff_vs_single_gpu"] <= 1e-4, rec["check"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["models", "candidates"])
def test_bench_parallel_modes_eight_ranks(mode):
  """BASELINE configs[3]'s LITERAL layout as a dry run on one GPU — world 8 (eight gloo processes sharing
  the device), K = 8 models, N = 512 candidates.  `models`: ONE model per rank, every rank but rank 0 holding a copy of
  model 0's flow (`flow0`) for y = F_0(x; z_0), one all-gather of z_0 and one of the [1,B,512,9] block per Adam step,
  the K-aggregation of rip/agent.py:109-127 on every rank after the gather.  `candidates`: all 8 models on every rank,
  64 of the 512 latent starts each, one all-gather of (best loss, plan).  Two Adam steps; the plans must equal those of
  ONE rank holding all 8 models and all 512 candidates (bench.py's own check).  Makes the first real SCALE record an
  execution of tested code, not a first run."""
  per_rank = {"models": "512", "candidates": "64"}[mode]
  rec = _run_bench_ranks(8, ["--mode", mode, "--models", "8", "--candidates", per_rank, "--search-steps", "2", "--obs-batch", "2"],
                         steps=2, timeout=1200)
  assert rec["n_gpus"] == 8 and rec["world_size_seen"] == 8 and rec["value"] > 0
  assert rec["config"]["mode"] == mode and rec["config"]["models"] == 8 and rec["config"]["candidates"] == 512
  assert rec["collectives_per_step"] == (3 if mode == "models" else 1)
  assert rec["check"]["max_abs_plan_diff_vs_single_gpu"] <= 1e-4, rec["check"]


@pytest.mark.gpu
def test_data_parallel_training_two_ranks():
  """`DIMTrainer(group=...)` as two PROCESSES (the 9.7 MB gradient all-reduce): each rank
  back-propagates its own half batch, `apply()` averages the packed gradients and steps Adam; the averaged vector
  equals the mean of the two local ones and both ranks end with identical trainable parameters and moments (running
  statistics stay per-rank, as under DistributedDataParallel without SyncBatchNorm)."""
  import json, socket, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  env = dict(os.environ, RIP_BENCH_SHARE_GPU="1", RIP_BENCH_BACKEND="gloo")
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
    env.pop(k, None)
  out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "tests", "mp", "train_two_ranks.py")],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-3000:]
  rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
  assert rec["world"] == 2 and np.isfinite(rec["loss"])
  assert rec["local_grads_differ_by"] > 0        # the ranks really saw different data
  assert rec["avg_grad_rel_err"] <= 1e-5, rec     # == clip(mean of the local gradients), not mean(clip(local))
  assert rec["mean_grad_norm"] > 1.0, rec         # the clip was active
  assert rec["params_identical"], rec


@pytest.mark.gpu
def test_training_from_datum_files_end_to_end(dev, tmp_path):
  """The reference's training data path with this package's pieces (dim/train.py:122-160, 175-213): datum files ->
  `replay.as_torch` -> `torch.utils.data.DataLoader` -> `.to(device)` + `ImitativeModel.transform` (the `transform`
  closure) -> `DIMTrainer.train_step`.  The loss of the first step equals the oracle's on the same batch and draws."""
  from oatomobile_amd import DIMTrainer, replay
  from oracle import reference_cpu as O
  ep = replay.Episode(str(tmp_path), "train")
  rng = np.random.default_rng(77)
  for i in range(6):
    o = synth_observation(np.random.default_rng(7700 + i))
    fut = np.cumsum(np.abs(rng.normal(size=(80, 3))) * 0.3, axis=0).astype(np.float32)
    ep.append("d%d" % i, lidar=o["lidar"], velocity=o["velocity"], is_at_traffic_light=o["is_at_traffic_light"],
              traffic_light_state=o["traffic_light_state"], player_future=fut)
  ds = replay.as_torch(ep._episode_dir, modalities=("lidar", "is_at_traffic_light", "traffic_light_state", "player_future",
                                                    "velocity"))
  loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, num_workers=0)
  model = hip_model(41, dev)
  trainer = DIMTrainer(model, lr=1e-3, max_batch=4, device=dev)
  losses = []
  for batch in loader:
    batch = {k: v.to(dev) for k, v in batch.items()}
    batch = model.transform(batch)
    assert tuple(batch["visual_features"].shape) == (3, 2, 100, 100) and tuple(batch["player_future"].shape) == (3, 4, 3)
    y = batch["player_future"][..., :2].contiguous()
    keep = torch.ones(3, 1280, device=dev)
    if not losses:  # first batch: the oracle's loss on the same inputs (train-mode BatchNorm, no dropout / noise)
      ref = oracle_model(41)
      ref.train()
      for mod in ref.modules():
        if isinstance(mod, torch.nn.Dropout):
          mod.eval()  # the step below runs with an all-ones keep mask
      with torch.no_grad():
        z = O.params(ref, batch["visual_features"].cpu(), batch["velocity"].cpu().reshape(3, 3),
                     batch["is_at_traffic_light"].cpu().reshape(3, 1), batch["traffic_light_state"].cpu().reshape(3, 1))
        _, lp, lad = O.flow_inverse(ref, y.cpu(), z)
        loss_ref = float(-(lp - lad).mean())
    losses.append(float(trainer.train_step(batch, y=y, dropout_mask=keep)))
  assert len(losses) == 2 and all(np.isfinite(losses))
  assert abs(losses[0] - loss_ref) <= 1e-4 * max(1.0, abs(loss_ref)), (losses[0], loss_ref)


# ---------------------------------------------------------------------------------------------------------
# round 3: R11 on the device, RCCL executed, gradient clipping order
# ---------------------------------------------------------------------------------------------------------
def test_r11_device_interpolation_bit_identical(golden, dev):
  """R11 (rip/agent.py:141-151) on the device: `rip_interpolate_plans` and the epilogue fused into the candidate
  selection (`plan_batch(interpolate=True)`, `agent(observation)`) against the host restatement of scipy's interp1d
  (`interpolate_plan`, itself pinned by the reference's `out30_*` recordings) — float64, bit for bit."""
  from oatomobile_amd import RIPAgent, _lib
  from oatomobile_amd.agents import interpolate_plan, interpolate_plans
  rng = np.random.default_rng(12)
  plans = (rng.normal(size=(257, 4, 2)) * np.array([1.0, 30.0, 1e-3, 1e4]).reshape(4, 1)).astype(np.float32)
  plans[0] = 0.0
  plans[1, 1] = plans[1, 0]  # a zero-slope segment
  out = interpolate_plans(torch.from_numpy(plans).to(dev)).cpu().numpy()
  assert out.shape == (257, 30, 3) and out.dtype == np.float64
  for i in range(257):
    np.testing.assert_array_equal(out[i], interpolate_plan(plans[i]))
  assert interpolate_plans(torch.empty(0, 4, 2, device=dev)).shape == (0, 30, 3)
  # the fused epilogue of the whole act(): [B,30,3] float64 == R11 of the same call's [B,4,2]
  models = [hip_model(100 + k, dev) for k in range(4)]
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=16, max_batch=5, seed=1)
  obs = [synth_observation(np.random.default_rng(60 + i)) for i in range(5)]
  lidar = torch.stack([torch.from_numpy(o["lidar"]) for o in obs]).to(dev)
  vec = torch.tensor([[*o["velocity"], o["is_at_traffic_light"], o["traffic_light_state"]] for o in obs], device=dev)
  goal = torch.stack([torch.from_numpy(o["goal"][:, :2].copy()) for o in obs]).to(dev)
  p4 = agent.plan_batch(lidar, vec, goal).cpu().numpy()
  p30, loss = agent.plan_batch(lidar, vec, goal, interpolate=True, return_loss=True)
  assert p30.dtype == torch.float64 and tuple(p30.shape) == (5, 30, 3) and tuple(loss.shape) == (5, 16)
  for i in range(5):
    np.testing.assert_array_equal(p30[i].cpu().numpy(), interpolate_plan(p4[i]))
    np.testing.assert_array_equal(agent(dict(obs[i])), interpolate_plan(p4[i]))
  buf = torch.empty(5, 30, 3, device=dev, dtype=torch.float64)
  assert agent.plan_batch(lidar, vec, goal, interpolate=True, out=buf) is buf
  with pytest.raises(ValueError):
    agent.plan_batch(lidar, vec, goal, interpolate=True, out=torch.empty(5, 4, 2, device=dev))
  # the reference's own [30,3] output (N = 1 = its algorithm) through the fused epilogue
  g = golden("g6_rip.npz")
  ref_agent = RIPAgent(None, algorithm="WCM", models=models)
  ob = synth_observation(np.random.default_rng(60))
  one = ref_agent.plan_batch(torch.from_numpy(ob["lidar"]).to(dev)[None],
                             torch.tensor([[*ob["velocity"], ob["is_at_traffic_light"], ob["traffic_light_state"]]], device=dev),
                             torch.from_numpy(ob["goal"][None, :, :2].copy()).to(dev), interpolate=True)
  np.testing.assert_allclose(one.cpu().numpy()[0], g["out30_WCM_o60"], atol=TOL)


def test_rccl_world1_executes_every_collective():
  """RCCL itself runs (one-rank `nccl` process group on this GPU, device tensors, no host staging): candidate-parallel,
  gradient-mode model-parallel, the row / score gathers and the training all-reduce, each equal to its collective-free
  composition; `librccl` is mapped into that process."""
  import json, socket, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RIP_BENCH_BACKEND", "RIP_BENCH_SHARE_GPU"):
    env.pop(k, None)
  out = subprocess.run([sys.executable, os.path.join(root, "tests", "mp", "rccl_world1.py")], cwd=root, env=env,
                       capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-3000:]
  rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
  print(rec)
  assert rec["backend"] == "nccl" and rec["world"] == 1
  assert rec["librccl_mapped"] and rec["librip_mapped"]
  assert rec["candidates_gathers"] == 1 and rec["candidates_plan_diff"] <= 1e-6
  assert rec["models_gathers"] == 11 and rec["models_plan_diff"] <= 1e-5  # z_0 + one block per Adam step
  assert rec["gather_rows_equal"] and rec["all_gather_scores_equal"] and rec["epilogue_gathers"] == 2
  assert rec["allreduce_calls"] == 1 and rec["allreduce_identity"] and np.isfinite(rec["train_loss"])


def test_bench_candidates_mode_one_rank_under_rccl():
  """`bench.py --gpus 1 --mode candidates`: a one-rank nccl process group, the all-gather of the winner records goes
  through RCCL; the line names the backend and the world size torch.distributed reports."""
  import json, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RIP_BENCH_BACKEND", "RIP_BENCH_SHARE_GPU", "MASTER_PORT"):
    env.pop(k, None)
  out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--mode", "candidates", "--obs-batch", "8",
                        "--steps", "3", "--warmup", "1"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-3000:]
  rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
  assert rec["backend"] == "nccl" and rec["world_size_seen"] == 1 and rec["n_gpus"] == 1 and rec["value"] > 0
  assert rec["check"]["max_abs_plan_diff_vs_single_gpu"] <= 1e-6


def test_train_clip_is_applied_after_the_reduction(dev):
  """train.py:206-208 `clip_grad_norm(model.parameters(), 1.0)`: the packed vector's clipped gradient equals
  torch.nn.utils.clip_grad_norm_ on the same gradients as separate parameters, and `apply(clip=True)` steps Adam
  with it; `evaluate_step` leaves gradients alone and runs any batch size (B = 1 included)."""
  from oatomobile_amd import DIMTrainer
  m = hip_model(9, dev)
  tr = DIMTrainer(m, lr=1e-3, max_batch=4, device=dev)
  rng = np.random.default_rng(8)

  def batch_of(B):
    return dict(visual_features=torch.from_numpy(rng.random((B, 2, 100, 100), dtype=np.float32)).to(dev),
                velocity=torch.from_numpy(rng.normal(0, 3, size=(B, 3)).astype(np.float32)).to(dev),
                is_at_traffic_light=torch.zeros(B, 1, device=dev), traffic_light_state=torch.ones(B, 1, device=dev),
                player_future=torch.from_numpy((np.cumsum(np.abs(rng.normal(size=(B, 4, 3))), axis=1) * 5).astype(np.float32)).to(dev))

  batch = batch_of(3)
  y = batch["player_future"][..., :2].contiguous()
  keep = torch.ones(3, 1280, device=dev)
  tr.backward(batch, y=y, dropout_mask=keep)
  g = tr.grads.clone()
  norm = float(torch.linalg.vector_norm(g.double()))
  assert norm > 1.0, "the clip must bite for this test to mean anything (norm %g)" % norm
  # torch's own clipping on the same gradients, held as separate parameter tensors
  ps = []
  for k, v in tr.named_gradients().items():
    p_ = torch.nn.Parameter(torch.zeros_like(v))
    p_.grad = v.clone()
    ps.append(p_)
  total = torch.nn.utils.clip_grad_norm_(ps, 1.0)
  np.testing.assert_allclose(float(total), norm, rtol=1e-5)
  p0, m0 = tr.params.clone(), tr.exp_avg.clone()
  tr.apply(clip=True)
  clipped = torch.cat([p_.grad.reshape(-1) for p_ in ps])
  np.testing.assert_allclose(tr.grads.cpu().numpy(), clipped.cpu().numpy(), rtol=1e-5, atol=1e-9)
  np.testing.assert_allclose(float(torch.linalg.vector_norm(tr.grads.double())), 1.0, rtol=1e-4)
  np.testing.assert_allclose(tr.exp_avg.cpu().numpy(), (0.1 * clipped).cpu().numpy(), rtol=1e-5, atol=1e-10)  # Adam's first moment
  assert not torch.equal(tr.params, p0) and torch.equal(m0, torch.zeros_like(m0))
  # evaluate_step: forward only, gradients untouched, B = 1 .. max_batch, equal to the frozen-statistics backward's loss
  gkeep = tr.grads.clone()
  for B in (1, 4):
    b = batch_of(B)
    le = float(tr.evaluate_step(b))
    assert torch.equal(tr.grads, gkeep)
    lb = float(tr.backward(b, train=False))
    np.testing.assert_allclose(le, lb, rtol=1e-6)
    tr.grads.copy_(gkeep)
  # train mode with one observation (the reference's last DataLoader batch may hold one: drop_last=False)
  l1 = float(tr.train_step(batch_of(1)))
  assert np.isfinite(l1) and torch.isfinite(tr.params).all()
  tr.close()


def test_packed_cache_replay_bit_identical(dev, tmp_path):
  """SURVEY §8f N1 / BASELINE configs[4]: the packed replay cache (uint8 codes + a 256-entry float table, expanded inside
  the transform kernel: `rip_encode_raw_u8`) against the `.npz` replay of the same datums — z, [4,2] plans and the
  [30,3] float64 plans are BIT-identical, for batches that do and do not divide the file count."""
  from oatomobile_amd import RIPAgent, _lib, replay
  models = [hip_model(100 + k, dev) for k in range(3)]
  agent = RIPAgent(None, algorithm="WCM", models=models, num_candidates=32, max_batch=16, seed=4, encoder_dtype="bf16")
  ep = replay.Episode(str(tmp_path), "ep")
  rng = np.random.default_rng(5)
  for i in range(37):
    o = synth_observation(np.random.default_rng(2200 + i))
    fut = np.cumsum(np.abs(rng.normal(size=(80, 3))) * 0.4, axis=0).astype(np.float32)
    ep.append("f%02d" % i, lidar=o["lidar"], velocity=o["velocity"], is_at_traffic_light=o["is_at_traffic_light"],
              traffic_light_state=o["traffic_light_state"], player_future=fut)
  files = ep.files()
  cache = replay.pack_cache(files, str(tmp_path / "cache"))
  ref4 = replay.replay(agent, files, batch_size=16)
  ref30 = replay.replay(agent, files, batch_size=16, interpolate=True)
  np.testing.assert_array_equal(replay.replay_cache(agent, cache, 16), ref4)
  # (the bf16 encoder picks its kernels by launch size, so bit-identity is per batch partition: 5 against 5)
  np.testing.assert_array_equal(replay.replay_cache(agent, cache, 5), replay.replay(agent, files, batch_size=5))
  np.testing.assert_allclose(replay.replay_cache(agent, cache, 5), ref4, atol=1e-4)
  np.testing.assert_array_equal(replay.replay_cache(agent, cache, 16, interpolate=True), ref30)
  np.testing.assert_array_equal(replay.replay_cache(agent, cache, 16, begin=16, end=37), ref4[16:37])  # a rank's share
  # round 6: two handles on two streams (even / odd batches) — the same bits, whole and ragged batch counts
  np.testing.assert_array_equal(replay.replay_cache(agent, cache, 16, streams=2), ref4)
  np.testing.assert_array_equal(replay.replay_cache(agent, cache, 5, interpolate=True, streams=2), replay.replay_cache(agent, cache, 5, interpolate=True))
  # the encoder output itself, coded vs float32 BEV (fp32 encoder)
  lib, h = _lib.load(), agent._handle.raw
  codes = torch.from_numpy(np.array(cache.codes[:8])).to(dev)
  lut = torch.from_numpy(cache.lut).to(dev)
  lidar = torch.from_numpy(cache.lut[np.asarray(cache.codes[:8])]).to(dev)
  vec = torch.from_numpy(cache.vec[:8].copy()).to(dev)
  za, zb = torch.empty(3, 8, 64, device=dev), torch.empty(3, 8, 64, device=dev)
  _lib.check(lib.rip_encode_raw(h, _lib.ptr(lidar), 1, 200, 200, _lib.ptr(vec), 8, 0, 3, 0, _lib.ptr(za), agent._handle.stream()))
  _lib.check(lib.rip_encode_raw_u8(h, _lib.ptr(codes, torch.uint8), _lib.ptr(lut), 200, 200, _lib.ptr(vec), 8, 0, 3, 0, _lib.ptr(zb),
                                   agent._handle.stream()))
  assert torch.equal(za, zb)
  with pytest.raises(ValueError):
    agent.plan_batch_coded(codes.float(), lut, vec, torch.zeros(8, 10, 2, device=dev))


def test_roctx_ranges_are_opt_in():
  """SURVEY §5 tracing hook: with RIP_ROCTX=1 the entry points open rocTX ranges (a whole act() runs through them and
  `rip_trace_push / _pop` report an open range); without it they are no-ops."""
  import subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  code = ("import numpy as np, torch, sys\n"
          "sys.path.insert(0, %r)\n"
          "from oatomobile_amd import ImitativeModel, RIPAgent, _lib\n"
          "from tests.helpers import synth_observation\n"
          "a = RIPAgent(None, algorithm='WCM', models=[ImitativeModel.synthetic(100), ImitativeModel.synthetic(101)], num_candidates=4)\n"
          "with _lib.trace_range('test range'):\n"
          "  out = a(dict(synth_observation(np.random.default_rng(60))))\n"
          "assert out.shape == (30, 3)\n"
          "lib = _lib.load()\n"
          "print('TRACING', lib.rip_trace_push(b'probe'), lib.rip_trace_pop())\n") % root
  for flag, want in (("1", "TRACING 1 1"), ("0", "TRACING 0 0")):
    env = dict(os.environ, RIP_ROCTX=flag)
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert want in out.stdout, (flag, out.stdout[-500:])


def test_search_and_encoder_parity_on_trained_weights(dev):
  """VERDICT r5 weak #1c: every other search / encoder gate runs on `weights.synthetic_state_dict` — U(-1/8, 1/8)-like
  flow weights, BatchNorm at its initial statistics — and no real checkpoint is available here.  This test makes weights
  with a TRAINED distribution on the box: K = 3 models, each `DIMTrainer`-trained for 200 Adam steps at lr 3e-3 on its own
  synthetic batches (the flow's GRU / head weights and every BatchNorm's running statistics move away from their
  initial values), written back into the ImitativeModels.  Then, on those weights: (a) the fp32 encoder's z against the
  oracle built from the trained state_dict, 1e-4; (b) the operand-range bookkeeping (`auto` must still pick the split-f16
  kernel: largest flow weight far below SPLIT_W_LIMIT); (c) one teacher-forced Adam step of the split-f16 and the paired
  kernel (algorithm MA: every model's adjoint reaches the gradient) against the oracle at 1e-4, like
  `test_teacher_forced_steps_vs_oracle`; (d) the whole 10-step WCM search of N = 64 candidates through `candidate_gate`
  (0 outliers)."""
  import ctypes
  from oatomobile_amd import DIMTrainer, RIPAgent, _lib
  from oracle import reference_cpu as O
  K, N, Bt = 3, 64, 16
  torch.manual_seed(20260601)  # the trainer's dropout masks come from torch's generator (its atomic reductions still differ run to run)
  torch.cuda.manual_seed_all(20260601)
  models, refs, wmax = [], [], 0.0
  for k in range(K):
    m = hip_model(900 + k, dev)
    tr = DIMTrainer(m, lr=3e-3, max_batch=Bt, device=dev)
    rng = np.random.default_rng(9000 + k)
    first = last = None
    for it in range(200):
      if it % 20 == 0:  # a new batch every 20 steps
        obs = [synth_observation(rng) for _ in range(Bt)]
        future = torch.from_numpy(np.cumsum(np.abs(rng.normal(size=(Bt, 4, 3))) * 2.0, axis=1).astype(np.float32)).to(dev)
        batch = dict(ctx_tensors(obs, dev), player_future=future)
      loss = float(tr.train_step(batch))
      first = loss if first is None else first
      last = loss
    assert np.isfinite(last) and last < first, (first, last)
    trained = {name: v.cpu().numpy() for name, v in tr.state_dict().items()}
    tr.sync_to_model()
    tr.close()
    flow_w = np.concatenate([np.abs(v).ravel() for name, v in trained.items() if name.startswith("_decoder.") and "weight" in name])
    wmax = max(wmax, float(flow_w.max()))
    print("model %d: loss %.3f -> %.3f over 200 steps, largest |flow weight| %.3f (synthetic init: 0.125)" % (k, first, last, float(flow_w.max())))
    models.append(m)
    refs.append(O.OracleImitativeModel.from_numpy_state_dict(trained))
  assert 0.126 < wmax < 200.0  # the weights moved, and stay inside the split kernel's operand range
  ob = synth_observation(np.random.default_rng(77))
  ctx = ctx_tensors([ob], dev)
  with torch.no_grad():
    zs = [O.params(r, **{name: v.cpu() for name, v in ctx.items()}) for r in refs]
  # (a) fp32 encoder + merger on trained BatchNorm statistics
  for k in range(K):
    np.testing.assert_allclose(models[k]._params(**ctx).cpu().numpy(), zs[k].numpy(), rtol=1e-4, atol=TOL)
  goal = torch.from_numpy(ob["goal"][None, :, :2].copy())
  lib = _lib.load()
  for kernel in ("split", "pair"):
    agent = RIPAgent(None, algorithm="MA", models=models, num_candidates=N, seed=5, search_kernel=kernel, max_batch=10)
    if kernel == "split":  # (b) `auto` on this handle still selects the split-f16 kernel for a large launch
      out = (ctypes.c_int32 * 10)()
      _lib.check(lib.rip_set_option(agent._handle.raw, _lib.OPT_SEARCH_KERNEL, 0))
      _lib.check(lib.rip_search_plan(agent._handle.raw, 64, N, ctypes.cast(out, ctypes.c_void_p), 10))
      assert out[0] == 4, list(out)
      _lib.check(lib.rip_set_option(agent._handle.raw, _lib.OPT_SEARCH_KERNEL, _lib.SEARCH_KERNELS["split"]))
    # (c) teacher-forced: the oracle's ten pre-step latents as a batch of ten one-step searches
    res = O.rip_search(refs, zs, goal, agent._x0_rows.cpu(), algorithm="MA", num_steps=10)
    S = 10
    xpre = res["trace_x_pre"].contiguous().to(dev)
    z = torch.stack([zz[0] for zz in zs])[:, None, :].repeat(1, S, 1).contiguous().to(dev)
    goal_d = goal.repeat(S, 1, 1).contiguous().to(dev)
    lb = torch.empty(S, N, device=dev)
    tp = torch.empty(1, K, S, N, device=dev)
    tg = torch.empty(1, S, N, 4, 2, device=dev)
    _lib.check(lib.rip_search(agent._handle.raw, _lib.ptr(z), _lib.ptr(goal_d), _lib.ptr(xpre), S, N, 10, _lib.ALGORITHMS["MA"], 1,
                              0.1, 1.0, None, None, _lib.ptr(lb), None, _lib.ptr(tp), None, _lib.ptr(tg), agent._handle.stream()))
    post_h, post_o = tp.cpu().numpy()[0].transpose(1, 0, 2), res["trace_post"].numpy()
    grad_h, grad_o = tg.cpu().numpy()[0], res["trace_grad"].numpy()
    print("%s on trained weights, teacher-forced: max |d post| %.3g, max |d grad| %.3g (max |grad| %.3g)" %
          (kernel, np.abs(post_h - post_o).max(), np.abs(grad_h - grad_o).max(), np.abs(grad_o).max()))
    np.testing.assert_allclose(post_h, post_o, rtol=1e-5, atol=TOL)
    # (the trained weights differ from run to run — the dropout masks are seeded above, the training step's atomic
    # reductions are not ordered: 200 steps end at loss 7.7-9.5 — and a head unit within rounding of its ReLU kink flips a
    # gradient coordinate by O(1 %): one run of ~20 showed 4 of 5120 coordinates off by up to 0.023 at |grad| 6.6, every
    # other run 2-9e-6.  The gate allows 8 such coordinates.)
    bad = ~np.isclose(grad_h, grad_o, rtol=1e-4, atol=TOL)
    assert bad.sum() <= 8 and np.abs(grad_h - grad_o)[bad].max(initial=0.0) <= 0.01 * np.abs(grad_o).max(), (int(bad.sum()), float(np.abs(grad_h - grad_o).max()))
    # (d) the whole search
    wcm = RIPAgent(None, algorithm="WCM", models=models, num_candidates=N, seed=5, search_kernel=kernel)
    lidar = torch.from_numpy(ob["lidar"]).to(dev)[None]
    vec = torch.tensor([[*ob["velocity"], ob["is_at_traffic_light"], ob["traffic_light_state"]]], device=dev)
    plan, loss = wcm.plan_batch(lidar, vec, goal.to(dev), return_loss=True)
    _, full = O.rip_call(refs, ob["lidar"], ob["velocity"], ob["is_at_traffic_light"], ob["traffic_light_state"], ob["goal"],
                         x0=wcm._x0_rows.cpu(), algorithm="WCM")
    candidate_gate("%s WCM on trained weights K=%d N=%d" % (kernel, K, N), loss.cpu().numpy()[0], full["loss_best"].numpy(),
                   plan.cpu().numpy()[0], full["plan"].numpy())


@pytest.mark.parametrize("kernel", ["split", "pair"])
@pytest.mark.parametrize("algo,K,N,B", [("WCM", 4, 128, 24), ("BCM", 3, 64, 40), ("WCM", 4, 48, 50)])
def test_split_kernel_is_deterministic_and_counts_its_adjoints(dev, algo, K, N, B, kernel):
  """Two launches of the split-f16 search give the same bits (selected plan, every candidate's plan and best loss,
  selected index), and `rip_search_stats` counts the inverse-pass adjoints that executed: between one (some member is
  always selected) and K - 1 per 16-candidate block and Adam step.  (Rounds 3 / 4 tested RIP_OPT_SEARCH_REGROUP here; the
  option is retired — accepted, no effect.)"""
  import ctypes
  from oatomobile_amd import RIPAgent, _lib
  models = [hip_model(100 + k, dev) for k in range(K)]
  agent = RIPAgent(None, algorithm=algo, models=models, num_candidates=N, max_batch=B, seed=7, search_kernel=kernel)
  lib, h = _lib.load(), agent._handle.raw
  rng = np.random.default_rng(31)
  z = torch.from_numpy(np.maximum(rng.normal(size=(K, B, 64)), 0).astype(np.float32)).to(dev)
  goal = torch.from_numpy(np.cumsum(np.abs(rng.normal(size=(B, 10, 2))) * 2, axis=1).astype(np.float32)).to(dev)
  x0 = agent._x0(B)
  out = {}
  for run in (0, 1):
    _lib.check(lib.rip_set_option(h, _lib.OPT_SEARCH_REGROUP, run))  # retired: must be accepted and change nothing
    assert lib.rip_set_option(h, _lib.OPT_SEARCH_KERNEL, 2) == _lib.RIP_EINVAL and b"removed in round 5" in lib.rip_last_error()
    plan = torch.empty(B, 4, 2, device=dev)
    plans = torch.empty(B, N, 4, 2, device=dev)
    lb = torch.empty(B, N, device=dev)
    best = torch.empty(B, device=dev, dtype=torch.int32)
    _lib.check(lib.rip_search_stats(h, None, 1))
    _lib.check(lib.rip_search(h, _lib.ptr(z), _lib.ptr(goal), _lib.ptr(x0), B, N, 10, _lib.ALGORITHMS[algo], 10, 0.1, 1.0,
                              _lib.ptr(plan), _lib.ptr(plans), _lib.ptr(lb), _lib.ptr(best, torch.int32), None, None, None,
                              agent._handle.stream()))
    cnt = ctypes.c_uint64(0)
    _lib.check(lib.rip_search_stats(h, ctypes.byref(cnt), 1))
    out[run] = (plan.cpu(), plans.cpu(), lb.cpu(), best.cpu(), cnt.value)
  for i in range(4):
    assert torch.equal(out[0][i], out[1][i]), ("plan", "plans", "loss_best", "best index")[i]
  blocks = B * N // 16 * 10
  print("%s K=%d N=%d B=%d: inverse adjoints per block-step %.2f" % (algo, K, N, B, out[0][4] / blocks))
  assert out[0][4] == out[1][4] and 0 < out[0][4] <= blocks * (K - 1)
