"""CPU tests of the host logic: state_dict compatibility, packing, C-ABI symbol table, interpolation,
frame transforms, loud failure without a device."""
import ctypes
import os
import sys
import re

import numpy as np
import pytest
import torch

from oatomobile_amd import arch
from oatomobile_amd import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_keys_match_oracle_tree():
  """The product's parameter container and the oracle's torchvision-shaped tree expose identical keys/shapes
  (328 tensors, 2 419 716 parameters — SURVEY.md §8b)."""
  from oatomobile_amd import ImitativeModel
  from oracle import reference_cpu as O
  a = ImitativeModel().state_dict()
  b = O.OracleImitativeModel().state_dict()
  assert list(a.keys()) == list(b.keys()) and len(a) == 328
  for k in a:
    assert tuple(a[k].shape) == tuple(b[k].shape), k
  assert sum(p.numel() for p in ImitativeModel().parameters()) == 2419716


def test_checkpoint_roundtrip_through_product_container(tmp_path):
  """A reference-style checkpoint (bare state_dict, torch/savers.py:45) loads strict and re-packs identically."""
  from oatomobile_amd import ImitativeModel
  from oracle import reference_cpu as O
  sd = W.synthetic_state_dict(9)
  ref = O.OracleImitativeModel.from_numpy_state_dict(sd)
  path = os.path.join(tmp_path, "model-1.pt")
  torch.save(ref.state_dict(), path)
  m = ImitativeModel()
  m.load_state_dict(torch.load(path), strict=True)
  np.testing.assert_array_equal(m.packed_weights(), W.pack_state_dict(sd))
  assert m.packed_weights().size == arch.packed_numel(2)
  with pytest.raises(RuntimeError):
    bad = dict(ref.state_dict())
    bad.pop("_merger._model.0.weight")
    ImitativeModel().load_state_dict(bad, strict=True)


def test_pack_rejects_bad_shapes():
  sd = W.synthetic_state_dict(1)
  sd["_decoder._decoder.weight_hh"] = np.zeros((192, 63), np.float32)
  with pytest.raises(ValueError, match="size mismatch"):
    W.pack_state_dict(sd)


def test_synthetic_weights_are_deterministic():
  a, b = W.synthetic_state_dict(5), W.synthetic_state_dict(5)
  for k in a:
    np.testing.assert_array_equal(a[k], b[k])
  assert float(np.abs(W.pack_state_dict(a)).sum()) > 0


def test_library_exports_every_declared_symbol():
  """include/rip_hip.h <-> librip_hip.so <-> _lib.SIGNATURES agree (no compute calls: no GPU here)."""
  from oatomobile_amd import _lib
  header = open(os.path.join(ROOT, "include", "rip_hip.h")).read()
  declared = set(re.findall(r"\b(rip_[a-z0-9_]+)\s*\(", header))
  lib = _lib.load()
  bound = {name for name, _, _ in _lib.SIGNATURES}
  assert declared == bound, declared ^ bound
  for name in declared:
    assert hasattr(lib, name)
  assert lib.rip_abi_version() == _lib.ABI_VERSION == 4


def test_search_kernel_names_are_checked():
  """`search_kernel=` of the agents goes through `_lib.search_kernel_id`: an unknown name is a `ValueError` that lists the
  valid kernels ("mfma", removed in round 5, used to surface as a bare KeyError), and every name maps to the
  option value include/rip_hip.h documents."""
  from oatomobile_amd import _lib
  assert _lib.SEARCH_KERNELS == {"auto": 0, "chain": 1, "phase": 3, "split": 4, "pair": 5}
  for name, value in _lib.SEARCH_KERNELS.items():
    assert _lib.search_kernel_id(name) == value
  with pytest.raises(ValueError, match="removed") as e:
    _lib.search_kernel_id("mfma")
  assert all(name in str(e.value) for name in _lib.SEARCH_KERNELS)
  with pytest.raises(ValueError, match="unknown search_kernel"):
    _lib.search_kernel_id("Split")
  header = open(os.path.join(ROOT, "include", "rip_hip.h")).read()
  for value in (1, 3, 4, 5):
    assert re.search(r"\*\s+%d = " % value, header), value


def test_docs_quote_the_header_and_the_tests_as_they_are():
  """The documents that describe the boundary quote the number of C-ABI entry points and cite GPU tests by name: both
  are checked against include/rip_hip.h and tests/test_gpu_parity.py (doc drift was a finding of two reviews)."""
  header = open(os.path.join(ROOT, "include", "rip_hip.h")).read()
  n = len(set(re.findall(r"\b(rip_[a-z0-9_]+)\s*\(", header)))
  for doc in ("README.md", "INTEGRATION.md", "DESIGN.md"):
    text = open(os.path.join(ROOT, doc)).read()
    quoted = {int(m) for m in re.findall(r"(\d+) (?:`extern \"C\"` )?entry points", text)}
    quoted -= {34, 39, 41, 42}  # earlier rounds' counts, quoted as history
    assert quoted and quoted <= {n}, (doc, quoted, n)
  tests = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read() + open(os.path.join(ROOT, "tests", "test_host_cpu.py")).read()
  tests += open(os.path.join(ROOT, "tests", "test_distributed_cpu.py")).read()
  tests += open(os.path.join(ROOT, "tests", "test_oracle_golden.py")).read()
  defined = set(re.findall(r"def (test_[a-z0-9_]+)\(", tests))
  for doc in ("README.md", "INTEGRATION.md", "DESIGN.md"):
    text = open(os.path.join(ROOT, doc)).read()
    for name in set(re.findall(r"`(test_[a-z0-9_]+)`", text)):
      assert name in defined or any(d.startswith(name.rstrip("_")) for d in defined), (doc, name)


def test_hot_kernels_keep_their_loads_in_flight_and_use_no_scratch():
  """Static check of the gfx950 assembly of the kernels the headline and the training step spend their time in
  (tools/dev/scan_waits.py; DESIGN §4.2's two compiler findings): (a) no private-segment (scratch) use — a ternary on
  HIP's float4 / uint4 STRUCT can put an operand array there —, (b) at most a handful of vector-memory loads that are
  followed by `s_waitcnt vmcnt(0)` within eight instructions, i.e. waited for on their own: a load behind a per-lane
  branch is, and a prologue or row loop built from such loads is that many memory round trips in sequence."""
  sys.path.insert(0, os.path.join(ROOT, "tools", "dev"))
  import scan_waits
  res = scan_waits.scan({"flow_split.hip", "encoder_bf16_front2.hip", "encoder_bf16_irb2.hip", "encoder_bf16_tile.hip",
                         "train.hip"})
  assert res, "hipcc produced no assembly"
  hot = {"search_split_kernelILb0ELi4E": (8, 0), "front2_bf16_kernel": (2, 0), "irb2_bf16_kernel": (2, 0),
         "irb_tile_bf16_kernel": (6, 32),  # (the 96 -> 96 block spills 24 bytes: 250+ registers)
         "dw_fwd_kernel": (2, 0), "dw_dgrad_kernel": (2, 0), "dw_wgrad_kernel": (3, 0), "act_bwd_stats_kernel": (2, 0),
         "colstats_kernel": (5, 0), "gemm_f32_kernelILb0ELb1ELi64ELi64ELi32ELi2ELb1E": (2, 0),
         "gemm_f32_kernelILb1ELb0ELi64ELi64ELi32ELi2ELb1E": (2, 0), "gemm_f32_kernelILb0ELb0ELi64ELi64ELi32ELi2ELb1E": (2, 0)}
  seen = set()
  for (src, kern), (alone, loads, scratch) in res.items():
    for tag, (max_alone, max_scratch) in hot.items():
      if tag in kern:
        seen.add(tag)
        assert scratch <= max_scratch, (kern, "scratch bytes", scratch)
        assert alone <= max_alone, (kern, "%d of %d loads are waited for on their own" % (alone, loads))
  assert seen == set(hot), set(hot) - seen


def test_no_cpu_fallback():
  from oatomobile_amd import ImitativeModel
  m = ImitativeModel()
  with pytest.raises(RuntimeError, match="no CPU path"):
    m._forward(torch.zeros(1, 4, 2), torch.zeros(1, 64))
  with pytest.raises(RuntimeError, match="no CPU path"):
    m.transform({"lidar": torch.zeros(1, 2, 200, 200)})


def test_product_does_not_import_oracle():
  for dirpath, _, files in os.walk(os.path.join(ROOT, "oatomobile_amd")):
    for f in files:
      if f.endswith((".py", ".hip", ".h", ".cpp")):
        src = open(os.path.join(dirpath, f)).read()
        assert "import oracle" not in src and "from oracle" not in src, f


def test_interpolate_plan_matches_scipy():
  import scipy.interpolate
  from oatomobile_amd.agents import interpolate_plan
  plan = np.random.default_rng(3).normal(size=(4, 2)).astype(np.float32)
  t = list(range(0, 40, 10))
  ref = scipy.interpolate.interp1d(x=t, y=plan, axis=0)(np.arange(0, t[-1]))
  out = interpolate_plan(plan)
  assert out.shape == (30, 3) and out.dtype == np.float64
  np.testing.assert_allclose(out[:, :2], ref, atol=1e-12)


def test_frame_transforms_roundtrip():
  from oatomobile_amd.agents import local2world, rot2mat, world2local
  rng = np.random.default_rng(0)
  loc, rot = rng.normal(size=3) * 50, np.array([3.0, 47.0, -2.0])
  pts = rng.normal(size=(30, 3)) * 10
  back = world2local(current_location=loc, current_rotation=rot,
                     world_locations=local2world(current_location=loc, current_rotation=rot, local_locations=pts))
  np.testing.assert_allclose(back, pts, atol=1e-9)
  R = rot2mat(np.array([0.0, 90.0, 0.0]))  # yaw 90deg: world x-axis maps to local -y
  np.testing.assert_allclose(R @ np.array([1.0, 0, 0]), [0, -1, 0], atol=1e-12)
  np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)


def test_setpoint_agent_replan_logic():
  from oatomobile_amd.agents import SetPointAgent

  class Fixed(SetPointAgent):
    calls = 0

    def __call__(self, observation):
      Fixed.calls += 1
      return np.c_[np.arange(30.0), np.zeros(30), np.zeros(30)]

  a = Fixed(None, replan_every_steps=3)
  ob = dict(location=np.zeros(3), rotation=np.zeros(3))
  outs = [a.act(ob) for _ in range(4)]
  assert Fixed.calls == 2  # steps 0 and 3
  assert outs[0]["target_speed"] == pytest.approx(20 / 3.6)  # forced for the first 100 steps (base.py:166-167)
  np.testing.assert_allclose(outs[0]["setpoint"], [5, 0, 0])
  np.testing.assert_allclose(outs[1]["setpoint"], [6, 0, 0])  # buffer popped
  a._steps_counter = 200
  assert a.act(ob)["target_speed"] == pytest.approx(1.0 / 0.05)


def test_encoder_plan_matches_arch_tables():
  blocks = arch.blocks()
  assert len(blocks) == 17 and blocks[-1].h_out == 4 and blocks[-1].oup == 320
  assert [b.index for b in blocks if b.residual] == [3, 5, 6, 8, 9, 10, 12, 13, 15, 16]
  assert arch.packed_numel(2) + 52 == sum(int(np.prod(s)) if s else 1 for _, s in arch.state_dict_spec(2))


def test_replay_datum_and_episode_format(tmp_path):
  """N1: the reference's on-disk layout (core/dataset.py:32-109) and `load_datum` semantics (datasets/carla.py:107-164)."""
  from oatomobile_amd import replay
  rng = np.random.default_rng(0)
  ep = replay.Episode(str(tmp_path), "episode0")
  frames = []
  for i in range(3):
    fr = dict(lidar=rng.random((200, 200, 2)), velocity=rng.normal(size=3), is_at_traffic_light=np.int64(i % 2),
              traffic_light_state=np.int64(i), player_future=np.cumsum(np.abs(rng.normal(size=(80, 3))), axis=0))
    frames.append(fr)
    ep.append("tok%d" % i, **fr)
  assert ep.fetch() == ["tok0", "tok1", "tok2"]
  ep.append(**frames[0])  # the reference's own call form: a random token (core/dataset.py:53-70)
  assert len(ep.fetch()) == 4 and len(ep.fetch()[3]) == 32
  np.testing.assert_array_equal(ep.read_sample(ep.fetch()[3], attr="velocity"), frames[0]["velocity"])
  assert set(ep.read_sample("tok1")) == set(frames[1])
  d = replay.load_datum(ep.files()[1], mode=True)
  assert d["lidar"].dtype == np.float32 and d["lidar"].shape == (200, 200, 2)
  assert d["is_at_traffic_light"].shape == (1,) and d["is_at_traffic_light"][0] == 1.0
  np.testing.assert_allclose(d["velocity"], frames[1]["velocity"].astype(np.float32))
  assert d["mode"].shape == (1,) and d["name"].endswith("tok1.npz")
  chw = replay.load_datum(ep.files()[0], modalities=("lidar",), dataformat="CHW")
  assert chw["lidar"].shape == (2, 200, 200)
  g = replay.goal_from_future(frames[0]["player_future"])
  np.testing.assert_allclose(g, frames[0]["player_future"][7::8][:10, :2].astype(np.float32))
  assert replay.goal_from_future(frames[0]["player_future"][:20]).shape == (10, 2)  # padded


def test_cil_model_state_dict_and_command_logic():
  """BehaviouralModel mirrors the reference's state_dict (cil/model.py:34-66) and CILAgent's command rule
  (cil/agent.py:66-77); no device needed."""
  import numpy as np
  from oatomobile_amd import BehaviouralModel, arch, weights
  from oatomobile_amd.cil import command_from_goal
  m = BehaviouralModel.synthetic(4)
  sd = weights.synthetic_cil_state_dict(4)
  assert list(m.state_dict().keys()) == list(sd.keys()) == [k for k, _ in arch.cil_state_dict_spec()]
  for k, v in m.state_dict().items():
    np.testing.assert_array_equal(v.numpy(), sd[k])
  assert weights.pack_cil_decoder(sd).size == 30146
  assert weights.encoder_only_packed(sd).size == arch.packed_numel()
  assert [command_from_goal(g) for g in ((1.0, 0.5), (10.0, 12.0), (20.0, 1.0), (0.0, 2.9), (-5.0, 0.0))] == [1, 2, 3, 1, 2]
  import pytest
  with pytest.raises(ValueError, match="Missing `mode`"):
    m(visual_features=None, velocity=None, is_at_traffic_light=None, traffic_light_state=None)


# ---------------------------------------------------------------------------------------------------------
# host rows pinned to the reference (fixtures from tools/make_golden_host.py, which runs the reference's code)
# ---------------------------------------------------------------------------------------------------------
def test_g11_frame_transforms_vs_reference(golden):
  """utils/carla.py:642-700 (`rot2mat`, `world2local`, `local2world`) incl. pitch / roll and the squeeze of a
  single point."""
  from oatomobile_amd.agents import local2world, rot2mat, world2local
  g = golden("g11_frames.npz")
  for i in range(int(g["num_cases"])):
    loc, rot, pts = g["loc%d" % i], g["rot%d" % i], g["pts%d" % i]
    np.testing.assert_allclose(rot2mat(rot), g["R%d" % i], atol=1e-14)
    w2l = world2local(current_location=loc, current_rotation=rot, world_locations=pts)
    l2w = local2world(current_location=loc, current_rotation=rot, local_locations=pts)
    np.testing.assert_allclose(w2l, g["w2l%d" % i], rtol=1e-13, atol=1e-11)
    np.testing.assert_allclose(l2w, g["l2w%d" % i], rtol=1e-13, atol=1e-11)
    one_w = world2local(current_location=loc, current_rotation=rot, world_locations=pts[0])
    one_l = local2world(current_location=loc, current_rotation=rot, local_locations=pts[0])
    assert one_w.shape == g["w2l_single%d" % i].shape == (3,)      # squeezed (utils/carla.py:674)
    assert one_l.shape == g["l2w_single%d" % i].shape == (1, 3)    # not squeezed (:700)
    np.testing.assert_allclose(one_w, g["w2l_single%d" % i], rtol=1e-13, atol=1e-11)
    np.testing.assert_allclose(one_l, g["l2w_single%d" % i], rtol=1e-13, atol=1e-11)


def test_g12_setpoint_agent_act_vs_reference(golden):
  """base.py:116-176: replan cadence, buffer pop, ego->world transform, rendered predictions, target speed and the
  spawn window — against what the reference's `act` handed its PID controller on the same 20-tick episodes."""
  from oatomobile_amd.agents import SetPointAgent
  g = golden("g12_setpoint.npz")
  for case in range(int(g["num_cases"])):
    plans = list(g["plans%d" % case])
    calls = []

    class Fixed(SetPointAgent):

      def __call__(self, observation, *a, **k):
        calls.append(1)
        return plans[len(calls) - 1]

    agent = Fixed(None, replan_every_steps=int(g["replan%d" % case]), setpoint_index=int(g["setpoint_index%d" % case]))
    for t in range(len(g["locs%d" % case])):
      if t == 12:
        agent._steps_counter = 150
      out = agent.act(dict(location=g["locs%d" % case][t], rotation=g["rots%d" % case][t]))
      np.testing.assert_allclose(out["target_speed"] * 3.6, g["target_kmh%d" % case][t], rtol=1e-12)
      np.testing.assert_allclose(out["setpoint"], g["waypoint%d" % case][t], rtol=1e-12, atol=1e-10)
      np.testing.assert_allclose(np.atleast_2d(out["predictions"])[0], g["pred0_%d" % case][t], rtol=1e-10, atol=1e-9)
    assert len(calls) == len(plans)  # same number of model calls as the reference made


def test_g13_episode_written_by_reference(golden, tmp_path):
  """An episode written by the reference's `Episode.append` (its file bytes are the fixture) read with
  `replay.Episode` / `replay.load_datum`: token order, every modality, dtype / shape rules, the `mode` ladder
  (datasets/carla.py:107-164) and `read_sample` (core/dataset.py:79-109)."""
  from oatomobile_amd import replay
  g = golden("g13_episode.npz")
  d = tmp_path / "ep0"
  d.mkdir()
  (d / "metadata").write_bytes(g["metadata"].tobytes())
  tokens = [str(t) for t in g["tokens"]]
  for i, tok in enumerate(tokens):
    (d / (tok + ".npz")).write_bytes(g["file%d" % i].tobytes())
  ep = replay.Episode(str(tmp_path), "ep0")
  assert ep.fetch() == tokens
  modes = []
  for i, f in enumerate(ep.files()):
    datum = replay.load_datum(f, mode=True)
    for k in replay.MODALITIES + ("mode",):
      ref = g["datum%d_%s" % (i, k)]
      assert datum[k].dtype == ref.dtype == np.float32 and datum[k].shape == ref.shape, k
      np.testing.assert_array_equal(datum[k], ref)
    assert datum["name"] == f
    modes.append(int(datum["mode"][0]))
    chw = replay.load_datum(f, modalities=("lidar",), dataformat="CHW")["lidar"]
    assert tuple(chw.shape) == tuple(g["datum%d_lidar_chw_shape" % i])
    assert float(chw.astype(np.float64).sum()) == float(g["datum%d_lidar_chw_sum" % i])
    np.testing.assert_array_equal(ep.read_sample(tokens[i], attr="control"), g["sample%d_control" % i])
  assert modes == [1, 2, 0, 2]  # STOP, LEFT, FORWARD, and theta = arccos(.) >= 0 never reaches RIGHT (:152, as coded)
  # and the other direction: what replay.Episode writes, np.load (the reference's reader) reads back
  ep2 = replay.Episode(str(tmp_path), "ep1")
  ep2.append(lidar=g["datum0_lidar"], velocity=g["datum0_velocity"])
  with np.load(ep2.files()[0]) as z:
    np.testing.assert_array_equal(z["lidar"], g["datum0_lidar"])
  # the unbatched dataset (`CARLADataset.as_torch`, datasets/carla.py:617-695) over the reference-written files:
  # same length, keys, CHW shapes and values (a doubling transform was recorded)
  ds = replay.as_torch(ep._episode_dir, modalities=("lidar", "velocity", "is_at_traffic_light", "traffic_light_state",
                                                    "player_future"), transform=lambda v: v * 2.0, mode=True)
  assert len(ds) == int(g["torch_len"])
  by_name = {os.path.basename(f)[:-4]: j for j, f in enumerate(ds._npz_files)}
  for i, tok in enumerate(tokens):
    item = ds[by_name[tok]]
    assert sorted(item.keys()) == list(g["torch%d_keys" % i])
    for k in item:
      assert tuple(np.asarray(item[k]).shape) == tuple(g["torch%d_%s_shape" % (i, k)]), k
      assert float(np.asarray(item[k], dtype=np.float64).sum()) == float(g["torch%d_%s_sum" % (i, k)]), k


def test_validation_of_raw_pointer_inputs():
  """dtype / contiguity / residency are enforced before a raw pointer crosses the C ABI."""
  from oatomobile_amd import _lib
  with pytest.raises(RuntimeError, match="no CPU path"):
    _lib.ptr(torch.zeros(3))
  with pytest.raises(ValueError, match="shape"):
    _lib.expect_shape(torch.zeros(2, 3), (2, 4), "t")
  _lib.expect_shape(torch.zeros(2, 3), (None, 3), "t")
  assert _lib.ptr(None).value in (None, 0)


def test_datum_batches_worker_processes_equal_inline(tmp_path):
  """`replay.DatumBatches`: batches decoded by worker processes into shared-memory buffers (ragged last batch, more
  workers than rows in it) equal the inline decode, in file order."""
  from oatomobile_amd import replay
  ep = replay.Episode(str(tmp_path), "ep")
  rng = np.random.default_rng(7)
  for i in range(21):
    ep.append("t%02d" % i, lidar=(rng.random((200, 200, 2)) < 0.1).astype(np.float32) * (1 + i % 5) / 5.0,
              velocity=rng.normal(size=3).astype(np.float32), is_at_traffic_light=np.float32(i % 2),
              traffic_light_state=np.float32(i % 4),
              player_future=np.cumsum(np.abs(rng.normal(size=(80, 3))), axis=0).astype(np.float32))
  files = ep.files()
  inline = [tuple(t.clone() for t in b) for b in replay.DatumBatches(files, 8)]
  multi = [tuple(t.clone() for t in b) for b in replay.DatumBatches(files, 8, workers=3, prefetch=2)]
  assert [b[0].shape[0] for b in inline] == [8, 8, 5] == [b[0].shape[0] for b in multi]
  for x, y in zip(inline, multi):
    for u, v in zip(x, y):
      assert u.dtype == torch.float32 and torch.equal(u, v)
  d0 = replay.load_datum(files[9])
  np.testing.assert_array_equal(inline[1][0][1].numpy(), d0["lidar"])
  np.testing.assert_array_equal(inline[1][2][1].numpy(), replay.goal_from_future(d0["player_future"]))
  assert list(replay.DatumBatches([], 8)) == []
  with pytest.raises(ValueError, match="BEV channels"):
    replay.DatumBatches(files, 8, channels=4)


def test_datum_batches_falls_back_without_shared_memory(tmp_path, monkeypatch):
  """A `/dev/shm` that cannot hold the batch ring (small container default) must not fail the replay: the loader warns
  and decodes inline."""
  from multiprocessing import shared_memory
  from oatomobile_amd import replay
  ep = replay.Episode(str(tmp_path), "ep")
  rng = np.random.default_rng(9)
  for i in range(5):
    ep.append("t%d" % i, lidar=rng.random((200, 200, 2)).astype(np.float32), velocity=np.zeros(3, np.float32),
              is_at_traffic_light=np.float32(0), traffic_light_state=np.float32(1), player_future=np.ones((80, 3), np.float32))
  real = shared_memory.SharedMemory

  class NoSpace(real):
    def __init__(self, *a, **k):
      if k.get("create"):
        raise OSError(28, "No space left on device")
      super().__init__(*a, **k)

  monkeypatch.setattr(shared_memory, "SharedMemory", NoSpace)
  with pytest.warns(UserWarning, match="no shared memory"):
    sizes = [b[0].shape[0] for b in replay.DatumBatches(ep.files(), 2, workers=2)]
  assert sizes == [2, 2, 1]


def test_effective_cpus_is_positive_and_bounded():
  from oatomobile_amd import replay
  n = replay.effective_cpus()
  assert 1 <= n <= (os.cpu_count() or 1)


def test_packed_replay_cache_round_trip(tmp_path):
  """`replay.pack_cache`: `lut[codes]` reproduces `load_datum(...)["lidar"]` bit for bit, goals / vectors as the
  `.npz` replay derives them, a value that first appears in a LATER chunk re-codes the rows packed before it, and data
  with more than 256 distinct values is refused (the `.npz` files stay the source of truth)."""
  from oatomobile_amd import replay
  from tests.helpers import synth_observation
  ep = replay.Episode(str(tmp_path), "ep")
  rng = np.random.default_rng(0)
  for i in range(7):
    o = synth_observation(np.random.default_rng(100 + i))
    lidar = o["lidar"].copy()
    if i < 3:
      lidar[lidar > 0.9] = 0.8  # the level 1.0 first appears in the second chunk
    fut = np.cumsum(np.abs(rng.normal(size=(80, 3))), axis=0).astype(np.float32)
    ep.append("t%d" % i, lidar=lidar, velocity=o["velocity"], is_at_traffic_light=o["is_at_traffic_light"],
              traffic_light_state=o["traffic_light_state"], player_future=fut)
  files = ep.files()
  cache = replay.pack_cache(files, str(tmp_path / "cache"), chunk=3)
  assert len(cache) == 7 and cache.channels == 2 and cache.codes.dtype == np.uint8
  assert np.isnan(cache.lut[6:]).all() and np.array_equal(cache.lut[:6], np.float32(np.arange(6) / 5.0))
  for i, f in enumerate(files):
    d = replay.load_datum(f)
    np.testing.assert_array_equal(cache.lidar(i), d["lidar"])
    np.testing.assert_array_equal(cache.goal[i], replay.goal_from_future(d["player_future"]))
    np.testing.assert_array_equal(cache.vec[i, :3], d["velocity"].reshape(3))
  got = [(c.shape[0], tuple(v.shape), tuple(g.shape)) for c, v, g in cache.batches(4)]
  assert got == [(4, (4, 5), (4, 10, 2)), (3, (3, 5), (3, 10, 2))]
  reopened = replay.PackedCache(str(tmp_path / "cache"))
  np.testing.assert_array_equal(np.asarray(reopened.codes), np.asarray(cache.codes))
  # the worker processes and the in-process packer write the same cache (codes, table, vectors, goals)
  serial = replay.pack_cache(files, str(tmp_path / "cache_serial"), chunk=3, workers=1)
  for name in ("codes", "lut", "vec", "goal"):
    np.testing.assert_array_equal(np.asarray(getattr(serial, name)), np.asarray(getattr(cache, name)))
  # -0.0 is a bit pattern of its own (float comparison would merge it with +0.0: ADVICE r3)
  ep3 = replay.Episode(str(tmp_path), "negzero")
  lid = synth_observation(np.random.default_rng(5))["lidar"].copy()
  lid[0, :7, 0] = -0.0
  ep3.append("z0", lidar=lid, velocity=np.zeros(3, np.float32), is_at_traffic_light=np.zeros(1, np.float32),
             traffic_light_state=np.zeros(1, np.float32), player_future=np.ones((80, 3), np.float32))
  cz = replay.pack_cache(ep3.files(), str(tmp_path / "cache3"))
  back = cz.lidar(0)
  np.testing.assert_array_equal(back.view(np.uint32), replay.load_datum(ep3.files()[0])["lidar"].view(np.uint32))
  assert np.signbit(back[0, :7, 0]).all() and not np.signbit(back[0, 7:, 0]).any()
  ep2 = replay.Episode(str(tmp_path), "noisy")
  ep2.append("n0", lidar=rng.random((200, 200, 2)).astype(np.float32), velocity=np.zeros(3, np.float32),
             is_at_traffic_light=np.zeros(1, np.float32), traffic_light_state=np.zeros(1, np.float32),
             player_future=np.zeros((80, 3), np.float32))
  with pytest.raises(ValueError, match="256 distinct"):
    replay.pack_cache(ep2.files(), str(tmp_path / "cache2"))
