"""Offline replay of cached observations (BASELINE config 5).

Reads the reference's on-disk formats —
  * datum files: one compressed `.npz` per frame with the sensor keys (`lidar`, `velocity`,
    `is_at_traffic_light`, `traffic_light_state`, `player_future`, ...), loaded like
    `CARLADataset.load_datum` (oatomobile/datasets/carla.py:107-164);
  * episodes: `<parent>/<token>/<sample>.npz` + a `metadata` file listing sample tokens in order
    (oatomobile/core/dataset.py:32-109)
— and pushes them through `RIPAgent.plan_batch` in device-resident batches (observation-parallel: with several
ranks each replays `distributed.shard_range(len(files), rank, world)`).

`goal` is not among the collected sensors (datasets/carla.py:175-182); like SURVEY §8d config 5 it is derived
from the recorded future: every `stride`-th waypoint of `player_future`, first `num_goals`, xy only.
"""

import os
import uuid
from typing import Iterable, List, Mapping, Optional, Sequence

import numpy as np
import torch

MODALITIES = ("lidar", "velocity", "is_at_traffic_light", "traffic_light_state", "player_future")


def load_datum(fname: str, modalities: Sequence[str] = MODALITIES, mode: bool = False,
               dataformat: str = "HWC") -> Mapping[str, np.ndarray]:
  """datasets/carla.py:107-164: float32 casts, scalars -> 1-D, optional HWC->CHW, optional driving-mode label
  ({0 FORWARD, 1 STOP, 2 LEFT, 3 RIGHT} from the last future waypoint), `name` = path."""
  assert dataformat in ("HWC", "CHW")
  sample = {}
  with np.load(fname) as datum:
    for attr in modalities:
      v = np.atleast_1d(datum[attr]).astype(np.float32)
      if v.ndim == 3 and dataformat == "CHW":
        v = np.transpose(v, (2, 0, 1))
      sample[attr] = v
  if mode and "player_future" in sample:
    x_T, y_T = sample["player_future"][-1, :2]
    norm = np.linalg.norm([x_T, y_T])
    theta = np.degrees(np.arccos(x_T / (norm + 1e-3)))
    label = 1 if norm < 3 else (2 if theta > 15 else (3 if theta <= -15 else 0))
    sample["mode"] = np.atleast_1d(label).astype(np.float32)
  sample["name"] = fname
  return sample


class Episode:
  """core/dataset.py:32-109: a directory of `.npz` samples + `metadata` token list."""

  def __init__(self, parent_dir: str, token: str) -> None:
    self._parent_dir, self._token = parent_dir, token
    self._episode_dir = os.path.join(parent_dir, token)
    os.makedirs(self._episode_dir, exist_ok=True)  # core/dataset.py:46-48
    self._metadata_fname = os.path.join(self._episode_dir, "metadata")

  def append(self, *sample_token: str, **observations: np.ndarray) -> None:
    """core/dataset.py:53-70: one compressed `.npz` per sample under a fresh random token (uuid4 hex, like
    utils/uuid.py); an explicit token may be passed positionally (deterministic tests)."""
    if len(sample_token) > 1:
      raise TypeError("append() takes at most one positional argument (the sample token)")
    token = sample_token[0] if sample_token else uuid.uuid4().hex
    np.savez_compressed(os.path.join(self._episode_dir, "%s.npz" % token), **observations)
    with open(self._metadata_fname, "a") as f:
      f.write("%s\n" % token)

  def read_sample(self, sample_token: str, attr: Optional[str] = None):
    """core/dataset.py:79-109: the whole observation of a sample, or one attribute of it."""
    with np.load(self.sample_path(sample_token), allow_pickle=True) as npz_file:
      if attr is not None:
        return npz_file[attr]
      return {k: npz_file[k] for k in npz_file}

  def fetch(self) -> List[str]:
    with open(self._metadata_fname) as f:
      return [t for t in f.read().split("\n") if t]

  def sample_path(self, sample_token: str) -> str:
    return os.path.join(self._episode_dir, "%s.npz" % sample_token)

  def files(self) -> List[str]:
    return [self.sample_path(t) for t in self.fetch()]


def goal_from_future(player_future: np.ndarray, num_goals: int = 10, stride: int = 8) -> np.ndarray:
  """`player_future[stride-1::stride][:num_goals, :2]`, padded by repeating the last waypoint."""
  g = np.asarray(player_future, dtype=np.float32)[stride - 1::stride][:num_goals, :2]
  if g.shape[0] < num_goals:
    g = np.concatenate([g, np.repeat(g[-1:], num_goals - g.shape[0], axis=0)], axis=0)
  return g


def replay(agent, files: Sequence[str], batch_size: int, num_goals: int = 10, goal_stride: int = 8) -> np.ndarray:
  """Plans for every datum in `files` -> [len(files), 4, 2] (host).  `agent` is a `RIPAgent` built with
  `max_batch >= batch_size`.  Decode (np.load) runs on the host; upload is one pinned copy per batch.

  The decode is what bounds this loop (measured: 1.4 k observations/s for compressed 200x200x2 datums against
  70 k/s of act() on the device): ~0.7 ms of zipfile + zlib + dtype conversion per frame, mostly under the GIL — a
  16-thread pool was SLOWER (0.8 k/s).  The reference spreads it over 50 DataLoader worker processes
  (dim/train.py:150-155); sharding `files` over ranks / processes (`distributed.shard_range`) is the same lever here."""
  dev = agent._device
  out = np.empty((len(files), 4, 2), np.float32)
  C = agent._in_channels
  if len(files) == 0:
    return out
  H, W = load_datum(files[0], modalities=("lidar",))["lidar"].shape[:2]
  lidar_h = torch.empty(batch_size, H, W, C).pin_memory()
  vec_h = torch.empty(batch_size, 5).pin_memory()
  goal_h = torch.empty(batch_size, num_goals, 2).pin_memory()
  for i0 in range(0, len(files), batch_size):
    chunk = files[i0:i0 + batch_size]
    for j, f in enumerate(chunk):
      d = load_datum(f)
      lidar_h[j] = torch.from_numpy(d["lidar"])
      vec_h[j, :3] = torch.from_numpy(d["velocity"].reshape(3))
      vec_h[j, 3] = float(d["is_at_traffic_light"].reshape(-1)[0])
      vec_h[j, 4] = float(d["traffic_light_state"].reshape(-1)[0])
      goal_h[j] = torch.from_numpy(goal_from_future(d["player_future"], num_goals, goal_stride))
    n = len(chunk)
    plan = agent.plan_batch(lidar_h[:n].to(dev, non_blocking=True), vec_h[:n].to(dev, non_blocking=True),
                            goal_h[:n].to(dev, non_blocking=True))
    out[i0:i0 + n] = plan.cpu().numpy()
  return out
