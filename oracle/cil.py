"""CPU restatement of the conditional-imitation-learning path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/ and tools/ as the checker.

Reference:
  * `BehaviouralModel` (oatomobile/baselines/torch/cil/model.py:34-66): MobileNetV2(num_classes=128, in_channels=2)
    encoder, MLP merger 134 -> 64 -> 64 -> 64 (ReLU after every layer), `nn.GRUCell(2, 64)`, `nn.Linear(64, 2)`;
  * `BehaviouralModel.forward` (:68-127): z = merger(cat(encoder(visual), velocity, is_at_traffic_light,
    traffic_light_state, mode)); x = 0; T times: z = GRUCell(x, z); x = x + Linear(z); stack -> [B, T, 2];
  * `BehaviouralModel.transform` (:129-170): like the DIM transform plus "STOP" (mode 1) -> 0;
  * `CILAgent.__call__` (cil/agent.py:45-97): mode from the last goal way-point (norm < 3 -> STOP, angle > 15 deg ->
    LEFT, else RIGHT), forward, linear interpolation of the plan onto 40 steps, append z = 0.
Pinned against the reference classes themselves in tests/golden/g10_cil.npz (tools/make_golden.py); the encoder is the
same torchvision-v0.6.0 restatement as the DIM oracle's (encoder parity unpinned, see oracle/mobilenet_v2.py).
"""
from typing import Mapping, Tuple

import numpy as np
import scipy.interpolate
import torch
import torch.nn as nn

from oracle.reference_cpu import _Encoder, _MLP, transform_visual, downsample_target


class OracleBehaviouralModel(nn.Module):
  """Same children / state_dict keys as the reference `BehaviouralModel` (cil/model.py:34-66)."""

  def __init__(self, output_shape: Tuple[int, int] = (40, 2), in_channels: int = 2) -> None:
    super().__init__()
    self._output_shape = tuple(output_shape)
    self._encoder = _Encoder(num_classes=128, in_channels=in_channels)
    self._merger = _MLP(128 + 3 + 1 + 1 + 1, [64, 64, 64], activate_final=True)
    self._decoder = nn.GRUCell(input_size=2, hidden_size=64)
    self._output = nn.Linear(in_features=64, out_features=self._output_shape[-1])
    self.eval()
    for p in self.parameters():
      p.requires_grad_(False)

  @classmethod
  def from_numpy_state_dict(cls, sd: Mapping[str, np.ndarray], in_channels: int = 2) -> "OracleBehaviouralModel":
    m = cls(in_channels=in_channels)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    return m.eval()

  def forward(self, **context: torch.Tensor) -> torch.Tensor:
    """cil/model.py:68-127."""
    for key in ("visual_features", "velocity", "is_at_traffic_light", "traffic_light_state", "mode"):
      if key not in context:
        raise ValueError("Missing `%s` keyword argument." % key)
    feats = self._encoder(context["visual_features"])
    z = torch.cat([feats, context["velocity"], context["is_at_traffic_light"], context["traffic_light_state"],
                   context["mode"]], dim=-1)
    z = self._merger(z)
    x = torch.zeros(z.shape[0], self._output_shape[-1], dtype=z.dtype)
    y = []
    for _ in range(self._output_shape[0]):
      z = self._decoder(x, z)
      x = self._output(z) + x
      y.append(x)
    return torch.stack(y, dim=1)

  def transform(self, sample):
    """cil/model.py:129-170."""
    if "player_future" in sample:
      sample["player_future"] = downsample_target(sample["player_future"], self._output_shape[-2])
    if "lidar" in sample:
      sample["visual_features"] = sample.pop("lidar")
    if "visual_features" in sample:
      sample["visual_features"] = transform_visual(sample["visual_features"])
    if "mode" in sample:
      sample["mode"][sample["mode"] == 1.0] = 0.0
    return sample


def command_from_goal(goal_xy_last: np.ndarray) -> int:
  """cil/agent.py:66-77 (note: the FORWARD branch is unreachable as written; restated as coded)."""
  x_t, y_t = float(goal_xy_last[0]), float(goal_xy_last[1])
  norm = np.linalg.norm([x_t, y_t])
  theta = np.degrees(np.arccos(x_t / (norm + 1e-3)))
  if norm < 3:
    return 1
  elif theta > 15:
    return 2
  elif theta <= 15:
    return 3
  return 0


def cil_call(model: OracleBehaviouralModel, observation: Mapping[str, np.ndarray]) -> np.ndarray:
  """cil/agent.py:45-97 on a single observation dict (lidar [200,200,2], goal [G,3], velocity [3], ...)."""
  goal = np.asarray(observation["goal"], dtype=np.float32)[..., :2]
  mode = np.atleast_2d(command_from_goal(goal[-1])).astype(np.float32)
  sample = dict(
      lidar=torch.from_numpy(np.transpose(np.asarray(observation["lidar"], np.float32)[None], (0, 3, 1, 2)).copy()),
      velocity=torch.from_numpy(np.asarray(observation["velocity"], np.float32)[None]),
      is_at_traffic_light=torch.from_numpy(np.atleast_1d(np.asarray(observation["is_at_traffic_light"], np.float32))[None]),
      traffic_light_state=torch.from_numpy(np.atleast_1d(np.asarray(observation["traffic_light_state"], np.float32))[None]),
      mode=torch.from_numpy(mode),
  )
  sample = model.transform(sample)
  with torch.no_grad():
    plan = model(**sample).numpy()[0]
  length = 40
  inc = length // plan.shape[0]
  time_index = list(range(0, length, inc))
  xy = scipy.interpolate.interp1d(x=time_index, y=plan, axis=0)(np.arange(0, time_index[-1]))
  return np.c_[xy, np.zeros((xy.shape[0], 1))]
