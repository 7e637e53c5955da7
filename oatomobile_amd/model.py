"""`ImitativeModel` — drop-in for `oatomobile.baselines.torch.ImitativeModel`, computed by HIP kernels.

Same constructor, attribute names (`_encoder`, `_merger`, `_decoder`,
`_output_shape`), methods (`to`, `forward`, `_params`, `_goal_likelihood`,
`transform`) and `state_dict` keys as the reference
(oatomobile/baselines/torch/dim/model.py:36-253), plus the `_forward/_inverse`
delegates `RIPAgent` calls but the reference forgot (rip/agent.py:106,111,137).

The `nn.Module` tree here is a *parameter container only* (so reference
checkpoints `load_state_dict(strict=True)`); every tensor operation of the path
runs in librip_hip.so (oatomobile_amd/csrc).  Inference semantics are `.eval()`
semantics: BatchNorm running statistics, Dropout off.  There is no CPU path:
tensors must live on a ROCm device.
"""

from typing import Mapping, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.distributions as D
import torch.nn as nn

from oatomobile_amd import _lib
from oatomobile_amd import arch
from oatomobile_amd import weights as _weights


class _Tree(nn.Module):
  """Nested parameter container addressed by dotted state_dict keys."""

  def add(self, dotted: str, shape: Tuple[int, ...]) -> None:
    head, _, rest = dotted.partition(".")
    if rest:
      if head not in self._modules:
        self.add_module(head, _Tree())
      self._modules[head].add(rest, shape)
    elif head == "num_batches_tracked":
      self.register_buffer(head, torch.zeros((), dtype=torch.long))
    elif head in ("running_mean", "running_var"):
      self.register_buffer(head, torch.zeros(shape) if head == "running_mean" else torch.ones(shape))
    else:
      self.register_parameter(head, nn.Parameter(torch.zeros(shape), requires_grad=False))

  def forward(self, *args, **kwargs):
    raise RuntimeError("parameter container: computation happens in librip_hip.so (use ImitativeModel's methods)")


def _require_device(t: torch.Tensor, what: str) -> None:
  if not t.is_cuda:
    raise RuntimeError("oatomobile_amd: `%s` is on %s — this build has no CPU path; move the model and its inputs "
                       "to a ROCm device (`.to('cuda')`)." % (what, t.device))


def _f32c(t: torch.Tensor) -> torch.Tensor:
  return t.detach().to(torch.float32).contiguous()


class AutoregressiveFlow(_Tree):
  """Mirror of `oatomobile.torch.networks.sequence.AutoregressiveFlow` (sequence.py:28-216):
  children `_decoder` (GRUCell weights) and `_locscale._model.{0,2}`; `_base_dist`, `forward`,
  `_forward`, `_inverse`."""

  def __init__(self, owner: "ImitativeModel", output_shape: Tuple[int, int], hidden_size: int) -> None:
    super().__init__()
    object.__setattr__(self, "_owner", owner)
    self._output_shape = tuple(output_shape)
    d = self._output_shape[-2] * self._output_shape[-1]
    self._base_dist = D.MultivariateNormal(loc=torch.zeros(d), scale_tril=torch.eye(d))  # sequence.py:47-50

  def to(self, *args, **kwargs):
    """sequence.py:67-74: the base distribution is not a buffer; rebuild it on the new device."""
    self = super().to(*args, **kwargs)
    self._base_dist = D.MultivariateNormal(
        loc=self._base_dist.mean.to(*args, **kwargs),
        scale_tril=self._base_dist.scale_tril.to(*args, **kwargs),
    )
    return self

  def forward(self, z: torch.Tensor) -> torch.Tensor:
    """sequence.py:76-93: sample the base distribution and push it forward."""
    x = self._base_dist.sample((z.shape[0],)).reshape(-1, *self._output_shape)
    return self._forward(x, z)[0]

  def _forward(self, x: torch.Tensor, z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """sequence.py:95-151 -> rip_flow_forward.  x [N,T,2], z [N,64] (or [1,64]) -> y [N,T,2], logabsdet [N]."""
    _require_device(x, "x")
    h = self._owner._handle()
    x, z = _f32c(x), _f32c(z)
    n = x.shape[0]
    y = torch.empty_like(x)
    lad = torch.empty(n, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.rip_flow_forward(h.raw, 0, _lib.ptr(x), _lib.ptr(z), n, z.shape[0], _lib.ptr(y), _lib.ptr(lad),
                                    h.stream()))
    return y, lad

  def _inverse(self, y: torch.Tensor, z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """sequence.py:153-216 -> rip_flow_inverse.  Returns x [N,T,2], log_prob [N], logabsdet [N]."""
    _require_device(y, "y")
    h = self._owner._handle()
    y, z = _f32c(y), _f32c(z)
    n = y.shape[0]
    x = torch.empty_like(y)
    lp = torch.empty(n, device=y.device, dtype=torch.float32)
    lad = torch.empty(n, device=y.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.rip_flow_inverse(h.raw, 0, _lib.ptr(y), _lib.ptr(z), n, z.shape[0], _lib.ptr(x), _lib.ptr(lp),
                                    _lib.ptr(lad), h.stream()))
    return x, lp, lad


class ImitativeModel(nn.Module):
  """A HIP/MI355X implementation of the reference's imitative model (dim/model.py:36)."""

  def __init__(self, output_shape: Tuple[int, int] = (4, 2), in_channels: int = 2, max_batch: int = 64) -> None:
    """Args:
      output_shape: event shape of the base/data distribution; only (4, 2) is built (dim/model.py:41).
      in_channels: BEV channels (2 in the reference: dim/model.py:53; BASELINE.json quotes 4).
      max_batch: largest observation batch one call may carry (sizes the encoder workspace).
    """
    super().__init__()
    if tuple(output_shape) != (arch.T, 2):
      raise ValueError("only output_shape=(4, 2) is implemented (got %r)" % (tuple(output_shape),))
    self._output_shape = tuple(output_shape)
    self._in_channels = int(in_channels)
    self._max_batch = int(max_batch)
    self._encoder = _Tree()
    self._merger = _Tree()
    self._decoder = AutoregressiveFlow(self, self._output_shape, hidden_size=arch.HIDDEN_SIZE)
    roots = {"_encoder": self._encoder, "_merger": self._merger, "_decoder": self._decoder}
    for key, shape in arch.state_dict_spec(self._in_channels):
      head, _, rest = key.partition(".")
      roots[head].add(rest, shape)
    self._hip = None  # (handle, device_index)
    self._dirty = True
    self._version = 0  # bumped whenever the weights may have changed (agents re-upload their snapshot on a mismatch)
    self._options_applied = None
    self.eval()

  # -- weights -------------------------------------------------------------------------------
  def load_state_dict(self, state_dict, strict: bool = True, **kw):
    out = super().load_state_dict(state_dict, strict=strict, **kw)
    self._touch()
    return out

  def _touch(self) -> None:
    self._dirty = True
    self._version += 1

  def load_numpy_state_dict(self, sd: Mapping[str, np.ndarray]) -> "ImitativeModel":
    self.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    return self

  @classmethod
  def synthetic(cls, seed: int, in_channels: int = 2, **kw) -> "ImitativeModel":
    """Random-but-deterministic weights (no checkpoints exist offline)."""
    return cls(in_channels=in_channels, **kw).load_numpy_state_dict(_weights.synthetic_state_dict(seed, in_channels))

  def packed_weights(self) -> np.ndarray:
    return _weights.pack_state_dict(self.state_dict(), self._in_channels)

  def refresh(self) -> None:
    """Re-upload weights after in-place parameter edits (also seen by every agent that holds this model)."""
    self._touch()

  def to(self, *args, **kwargs):
    """dim/model.py:70-74: also rebuilds the decoder's base distribution on the device."""
    self = super().to(*args, **kwargs)
    self._decoder = self._decoder.to(*args, **kwargs)
    self._dirty = True  # same values: the agents' snapshots stay valid (no version bump)
    return self

  @property
  def device(self) -> torch.device:
    return self._merger._model._modules["0"].weight.device

  def _handle(self) -> "_lib.Handle":
    dev = self.device
    if dev.type != "cuda":
      raise RuntimeError("oatomobile_amd.ImitativeModel is on %s — this build has no CPU path; call "
                         "`.to('cuda')` on a ROCm machine." % dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if self._hip is None or self._hip[1] != idx:
      if self._hip is not None:
        self._hip[0].close()
      self._hip = (_lib.Handle(1, self._in_channels, self._max_batch, idx), idx)
      self._dirty = True
    if self._dirty:
      self._hip[0].load_model(0, self.packed_weights())
      self._dirty = False
    fused, mega = getattr(self, "fused_encoder", None), getattr(self, "mega_encoder", None)
    if (fused is not None or mega is not None) and self._options_applied != (id(self._hip[0]), fused, mega):
      if fused is not None:
        self._hip[0].set_option(_lib.OPT_ENCODER_FUSED, int(fused))
      if mega is not None:  # experimental one-launch fp32 encoder: 1 = up to 4 observations, 0 / -1 never
        self._hip[0].set_option(_lib.OPT_ENCODER_MEGA, int(mega))
      self._options_applied = (id(self._hip[0]), fused, mega)
    return self._hip[0]

  # -- reference API -------------------------------------------------------------------------
  def forward(self, num_steps: int, goal: Optional[torch.Tensor] = None, lr: float = 1e-1, epsilon: float = 1.0,
              x0: Optional[torch.Tensor] = None, **context: torch.Tensor) -> torch.Tensor:
    """Returns a local mode of the posterior, [B, T, 2] (dim/model.py:76-141 -> rip_dim_forward).

    `x0` (not in the reference) pins the base sample for reproducible tests; by default one
    `_base_dist.sample()` is drawn and repeated over the batch exactly like dim/model.py:100-104.
    """
    if "visual_features" not in context:
      raise ValueError("Missing `visual_features` keyword argument.")
    batch_size = context["visual_features"].shape[0]
    z = self._params(**context)
    if x0 is None:
      x0 = self._decoder._base_dist.sample()
    x0 = _f32c(x0.to(z.device)).reshape(1, -1).repeat(batch_size, 1) if x0.numel() == 8 else _f32c(x0.to(z.device))
    x0 = x0.reshape(batch_size, *self._output_shape).contiguous()
    g = None
    G = 0
    if goal is not None:
      g = _f32c(goal.to(z.device))
      if g.shape[0] == 1 and batch_size > 1:
        g = g.expand(batch_size, -1, -1).contiguous()
      G = g.shape[1]
    y = torch.empty(batch_size, *self._output_shape, device=z.device, dtype=torch.float32)
    lib = _lib.load()
    h = self._handle()
    _lib.check(lib.rip_dim_forward(h.raw, 0, _lib.ptr(z), _lib.ptr(g), _lib.ptr(x0), batch_size, G,
                                   int(num_steps), float(lr), float(epsilon), _lib.ptr(y), None, h.stream()))
    return y

  def _goal_likelihood(self, y: torch.Tensor, goal: torch.Tensor, **hyperparams) -> torch.Tensor:
    """dim/model.py:143-171: log-likelihood of the plans' last waypoint under the goal mixture, batch mean."""
    return self._goal_likelihood_rows(y, goal, **hyperparams).mean(dim=0)

  def _goal_likelihood_rows(self, y: torch.Tensor, goal: torch.Tensor, **hyperparams) -> torch.Tensor:
    _require_device(y, "y")
    epsilon = float(hyperparams.get("epsilon", 1.0))
    y, goal = _f32c(y), _f32c(goal.to(y.device))
    n = y.shape[0]
    rows = torch.empty(n, device=y.device, dtype=torch.float32)
    _lib.expect_shape(y, (None, arch.T, 2), "y")
    _lib.expect_shape(goal, (None, None, 2), "goal")
    lib = _lib.load()
    with torch.cuda.device(y.device):  # stateless entry point: launches on the current device
      _lib.check(lib.rip_goal_likelihood(_lib.ptr(y), _lib.ptr(goal), n, goal.shape[0], goal.shape[1], epsilon,
                                         _lib.ptr(rows), _lib.current_stream(y.device)))
    return rows

  def _params(self, **context: torch.Tensor) -> torch.Tensor:
    """Contextual parameters z [B, 64] (dim/model.py:173-219 -> rip_encode)."""
    for key in ("visual_features", "velocity", "is_at_traffic_light", "traffic_light_state"):
      if key not in context:
        raise ValueError("Missing `%s` keyword argument." % key)
    vis = context["visual_features"]
    _require_device(vis, "visual_features")
    vis = _f32c(vis)
    if vis.dim() != 4 or vis.shape[1] != self._in_channels or vis.shape[2] != arch.INPUT_HW or vis.shape[3] != arch.INPUT_HW:
      raise ValueError("visual_features must be [B,%d,%d,%d] (output of `transform`), got %s" %
                       (self._in_channels, arch.INPUT_HW, arch.INPUT_HW, tuple(vis.shape)))
    b = vis.shape[0]
    vec = torch.cat([
        _f32c(context["velocity"]).reshape(b, 3),
        _f32c(context["is_at_traffic_light"]).reshape(b, 1),
        _f32c(context["traffic_light_state"]).reshape(b, 1),
    ], dim=-1).contiguous()  # dim/model.py:206-214 (the cat is 5 floats per row: plumbing)
    z = torch.empty(b, arch.HIDDEN_SIZE, device=vis.device, dtype=torch.float32)
    lib = _lib.load()
    h = self._handle()
    _lib.check(lib.rip_encode(h.raw, _lib.ptr(vis), _lib.ptr(vec), b, 0, 1,
                              _lib.ENC_DTYPES[getattr(self, "encoder_dtype", "fp32")], _lib.ptr(z), None, h.stream()))
    return z

  def encoder_features(self, visual_features: torch.Tensor) -> torch.Tensor:
    """MobileNetV2 logits [B,128] (what `self._encoder(visual_features)` returns in the reference)."""
    vis = _f32c(visual_features)
    _require_device(vis, "visual_features")
    b = vis.shape[0]
    vec = torch.zeros(b, 5, device=vis.device)
    z = torch.empty(b, 64, device=vis.device)
    feat = torch.empty(b, arch.NUM_FEATURES, device=vis.device)
    h = self._handle()
    _lib.check(_lib.load().rip_encode(h.raw, _lib.ptr(vis), _lib.ptr(vec), b, 0, 1, 0, _lib.ptr(z), _lib.ptr(feat),
                                      h.stream()))
    return feat

  def encoder_layer_output(self, visual_features: torch.Tensor, layer: int) -> torch.Tensor:
    """Diagnostics (rip_encode_tap): the output of conv layer `layer` (index into `arch.conv_layers()`) under the
    model's current `encoder_dtype` / `fused_encoder` selection, as fp32 NCHW [B,C,H,W] ([B,1280] for the last layer,
    whose average pool is fused into it).  `_lib.RipError` (RIP_EINVAL) when that layer is interior to a fused block."""
    vis = _f32c(visual_features)
    _require_device(vis, "visual_features")
    b = vis.shape[0]
    spec = arch.conv_layers(self._in_channels)[layer]
    last = layer + 1 == len(arch.conv_layers(self._in_channels))
    out = torch.empty((b, spec.cout) if last else (b, spec.h_out, spec.h_out, spec.cout), device=vis.device)
    h = self._handle()
    _lib.check(_lib.load().rip_encode_tap(h.raw, _lib.ptr(vis), b, 0, _lib.ENC_DTYPES[getattr(self, "encoder_dtype", "fp32")],
                                          int(layer), _lib.ptr(out), out.numel(), h.stream()))
    return out if last else out.permute(0, 3, 1, 2).contiguous()

  def transform(self, sample: Mapping[str, torch.Tensor]) -> Mapping[str, torch.Tensor]:
    """dim/model.py:221-253: mutates and returns `sample` (lidar -> visual_features, 200->100 + H/W swap;
    player_future subsampled to T steps)."""
    if "player_future" in sample:
      pf = sample["player_future"]
      inc = pf.shape[1] // self._output_shape[-2]  # transforms.py:23-31
      sample["player_future"] = pf[:, 0::inc, :]
    if "lidar" in sample:
      sample["visual_features"] = sample.pop("lidar")
    if "visual_features" in sample:
      sample["visual_features"] = transform_visual(sample["visual_features"])
    return sample

  def _forward(self, x: torch.Tensor, z: torch.Tensor):
    return self._decoder._forward(x=x, z=z)

  def _inverse(self, y: torch.Tensor, z: torch.Tensor):
    return self._decoder._inverse(y=y, z=z)


def transform_visual(visual_features: torch.Tensor, output_hw: int = arch.INPUT_HW,
                     channels_last: bool = False) -> torch.Tensor:
  """torch/transforms.py:34-49 (bilinear, align_corners=True, then H/W swap) -> rip_transform.
  in: [B,C,H,W] (or [B,H,W,C] with channels_last=True); out: [B,C,output_hw,output_hw]."""
  _require_device(visual_features, "visual_features")
  v = _f32c(visual_features)
  if v.dim() != 4:
    raise ValueError("visual_features must be 4-D, got shape %s" % (tuple(v.shape),))
  if channels_last:
    b, h, w, c = v.shape
  else:
    b, c, h, w = v.shape
  out = torch.empty(b, c, output_hw, output_hw, device=v.device, dtype=torch.float32)
  with torch.cuda.device(v.device):  # stateless entry point: launches on the current device
    _lib.check(_lib.load().rip_transform(_lib.ptr(v), b, c, h, w, int(channels_last), output_hw, _lib.ptr(out),
                                         _lib.current_stream(v.device)))
  return out
