"""Conditional imitation learning on the GPU: `BehaviouralModel` and `CILAgent`.

Drop-in mirrors of `oatomobile.baselines.torch.cil.BehaviouralModel` (cil/model.py:31-170) and `CILAgent`
(cil/agent.py:28-97): same constructor arguments, attribute names (`_encoder`, `_merger`, `_decoder`, `_output`),
`state_dict` keys (reference checkpoints load strict), `forward(**context)` and `transform`.  The `nn.Module` tree is a
parameter container; the MobileNetV2 encoder runs through `rip_encode` (a one-model handle that carries this model's
encoder tensors) and everything after it -- merger MLP, GRUCell roll-out, residual output head -- is the single
`rip_cil_decode` kernel.  `.eval()` semantics; there is no CPU path.
"""
from typing import Any, Mapping, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from oatomobile_amd import _lib
from oatomobile_amd import arch
from oatomobile_amd import weights as _weights
from oatomobile_amd.agents import SetPointAgent, interpolate_plan
from oatomobile_amd.model import _Tree, _f32c, _require_device, transform_visual


class BehaviouralModel(nn.Module):
  """A HIP/MI355X implementation of the reference's behavioural cloning model (cil/model.py:31)."""

  def __init__(self, output_shape: Tuple[int, int] = (40, 2), in_channels: int = 2, max_batch: int = 64) -> None:
    super().__init__()
    if len(output_shape) != 2 or output_shape[-1] != 2 or output_shape[0] < 1:
      raise ValueError("output_shape must be (T, 2) (got %r)" % (tuple(output_shape),))
    self._output_shape = tuple(output_shape)
    self._in_channels = int(in_channels)
    self._max_batch = int(max_batch)
    self._encoder = _Tree()
    self._merger = _Tree()
    self._decoder = _Tree()
    self._output = _Tree()
    roots = {"_encoder": self._encoder, "_merger": self._merger, "_decoder": self._decoder, "_output": self._output}
    for key, shape in arch.cil_state_dict_spec(self._in_channels):
      head, _, rest = key.partition(".")
      roots[head].add(rest, shape)
    self._hip = None      # (encoder handle, device index)
    self._blob = None     # decoder weights on the device
    self._dirty = True
    self.eval()

  # -- weights -------------------------------------------------------------------------------
  def load_state_dict(self, state_dict, strict: bool = True, **kw):
    out = super().load_state_dict(state_dict, strict=strict, **kw)
    self._dirty = True
    return out

  def load_numpy_state_dict(self, sd: Mapping[str, np.ndarray]) -> "BehaviouralModel":
    self.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    return self

  @classmethod
  def synthetic(cls, seed: int, in_channels: int = 2, **kw) -> "BehaviouralModel":
    return cls(in_channels=in_channels, **kw).load_numpy_state_dict(_weights.synthetic_cil_state_dict(seed, in_channels))

  def to(self, *args, **kwargs):
    self = super().to(*args, **kwargs)
    self._dirty = True
    return self

  @property
  def device(self) -> torch.device:
    return self._output.weight.device

  def _sync(self):
    dev = self.device
    if dev.type != "cuda":
      raise RuntimeError("oatomobile_amd.BehaviouralModel is on %s — this build has no CPU path; call `.to('cuda')` "
                         "on a ROCm machine." % dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if self._hip is None or self._hip[1] != idx:
      if self._hip is not None:
        self._hip[0].close()
      self._hip = (_lib.Handle(1, self._in_channels, self._max_batch, idx), idx)
      self._dirty = True
    if self._dirty:
      sd = self.state_dict()
      self._hip[0].load_model(0, _weights.encoder_only_packed(sd, self._in_channels))
      blob = _weights.pack_cil_decoder(sd)
      assert blob.size == _lib.load().rip_cil_blob_floats()
      self._blob = torch.from_numpy(blob).to(dev)
      self._dirty = False
    return self._hip[0]

  # -- reference API -------------------------------------------------------------------------
  def forward(self, **context: torch.Tensor) -> torch.Tensor:
    """Returns the expert plan [B, T, 2] (cil/model.py:68-127 -> rip_encode + rip_cil_decode)."""
    for key in ("visual_features", "velocity", "is_at_traffic_light", "traffic_light_state", "mode"):
      if key not in context:
        raise ValueError("Missing `%s` keyword argument." % key)
    vis = context["visual_features"]
    _require_device(vis, "visual_features")
    vis = _f32c(vis)
    if vis.dim() != 4 or vis.shape[1] != self._in_channels or vis.shape[2] != arch.INPUT_HW or vis.shape[3] != arch.INPUT_HW:
      raise ValueError("visual_features must be [B,%d,%d,%d] (output of `transform`), got %s" %
                       (self._in_channels, arch.INPUT_HW, arch.INPUT_HW, tuple(vis.shape)))
    h = self._sync()
    b = vis.shape[0]
    vec = torch.cat([_f32c(context["velocity"]).reshape(b, 3), _f32c(context["is_at_traffic_light"]).reshape(b, 1),
                     _f32c(context["traffic_light_state"]).reshape(b, 1), _f32c(context["mode"]).reshape(b, 1)],
                    dim=-1).contiguous()  # cil/model.py:88-98
    feat = torch.empty(b, arch.NUM_FEATURES, device=vis.device, dtype=torch.float32)
    zdummy = torch.empty(b, arch.HIDDEN_SIZE, device=vis.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.rip_encode(h.raw, _lib.ptr(vis), _lib.ptr(vec[:, :5].contiguous()), b, 0, 1,
                              _lib.ENC_DTYPES[getattr(self, "encoder_dtype", "fp32")], _lib.ptr(zdummy), _lib.ptr(feat),
                              h.stream()))
    T = self._output_shape[0]
    y = torch.empty(b, T, 2, device=vis.device, dtype=torch.float32)
    with torch.cuda.device(vis.device):  # stateless entry point: launches on the current device
      _lib.check(lib.rip_cil_decode(_lib.ptr(feat), _lib.ptr(vec), _lib.ptr(self._blob), b, T, _lib.ptr(y),
                                    h.stream()))
    return y

  def transform(self, sample: Mapping[str, torch.Tensor]) -> Mapping[str, torch.Tensor]:
    """cil/model.py:129-170: mutates and returns `sample`."""
    if "player_future" in sample:
      pf = sample["player_future"]
      inc = pf.shape[1] // self._output_shape[-2]
      sample["player_future"] = pf[:, 0::inc, :]
    if "lidar" in sample:
      sample["visual_features"] = sample.pop("lidar")
    if "visual_features" in sample:
      sample["visual_features"] = transform_visual(sample["visual_features"])
    if "mode" in sample:  # removes the "STOP" command (cil/model.py:166-168)
      sample["mode"][sample["mode"] == 1.0] = 0.0
    return sample


def command_from_goal(goal_xy_last) -> int:
  """cil/agent.py:66-77, restated as coded (STOP = 1 when the last goal way-point is closer than 3 m, LEFT = 2 when it
  is more than 15 degrees off the heading, RIGHT = 3 otherwise; the FORWARD branch of the reference is unreachable)."""
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


class CILAgent(SetPointAgent):
  """The conditional imitation learning agent (cil/agent.py:28)."""

  def __init__(self, environment: Any = None, *, model: BehaviouralModel, device: Optional[torch.device] = None,
               **kwargs) -> None:
    super().__init__(environment=environment, **kwargs)
    self._device = torch.device(device) if device is not None else torch.device("cuda")
    self._model = model.to(self._device)

  def __call__(self, observation: Mapping[str, np.ndarray], *args, **kwargs) -> np.ndarray:
    """Returns the imitative prior: ego-frame plan [39, 3] (cil/agent.py:45-97)."""
    goal = np.asarray(observation["goal"], dtype=np.float32)[..., :2]
    mode = np.atleast_2d(command_from_goal(goal[-1])).astype(np.float32)
    lidar = np.asarray(observation["lidar"], dtype=np.float32)[None]  # [1, 200, 200, C]
    sample = dict(
        lidar=torch.from_numpy(np.ascontiguousarray(np.transpose(lidar, (0, 3, 1, 2)))).to(self._device),
        velocity=torch.from_numpy(np.asarray(observation["velocity"], np.float32).reshape(1, 3)).to(self._device),
        is_at_traffic_light=torch.tensor([[float(observation["is_at_traffic_light"])]], device=self._device),
        traffic_light_state=torch.tensor([[float(observation["traffic_light_state"])]], device=self._device),
        mode=torch.from_numpy(mode).to(self._device),
    )
    sample = self._model.transform(sample)
    plan = self._model(**sample).detach().cpu().numpy()[0]  # [T, 2]
    return interpolate_plan(plan)
