"""DIM training step on MI355X — `oatomobile/baselines/torch/dim/train.py:175-213` behind the
`rip_train_*` entry points of librip_hip.so (csrc/train.hip, csrc/flow.hip).

    trainer = DIMTrainer(model, lr=1e-3)                 # optim.Adam(model.parameters(), lr) (train.py:112-116)
    loss = trainer.train_step(batch)                     # train_step(model, optimizer, batch) (train.py:175-213)
    trainer.sync_to_model()                              # updated weights (and BN buffers) back into `model`

`batch` is what the reference's `transform` closure produces (train.py:122-134): `visual_features [B,C,100,100]`,
`velocity [B,3]`, `is_at_traffic_light [B,1]`, `traffic_light_state [B,1]`, `player_future [B,4,>=2]`, on the device.

Semantics are the reference's train mode: BatchNorm on batch statistics (running statistics updated with momentum
0.1), Dropout(0.2) in front of the MobileNetV2 classifier, the target perturbed with N(0, noise_level^2) noise
(train.py:184-189).  The two random draws come from torch's device generator; tests pass them in (`y=`,
`dropout_mask=`) to replay a step recorded from the reference.

Parameters, gradients and the Adam moments are single packed fp32 device tensors in the reference's state_dict order
(`arch.packed_spec`), so data-parallel training is one `all_reduce` of `trainer.grads` between `backward()` and
`apply()` (9.7 MB: the first bandwidth-relevant collective of this code base; only a trainer built with `group=`
reduces, and the averaged gradient — not the local one — is what `clip=True` clips).
"""

import ctypes
from typing import Mapping, Optional

import numpy as np
import torch

from oatomobile_amd import _lib
from oatomobile_amd import arch
from oatomobile_amd.model import ImitativeModel

DROPOUT_P = 0.2  # torchvision MobileNetV2.classifier[0]


class DIMTrainer:
  """One model, one device; owns the packed parameter / gradient / Adam-moment tensors and the HIP workspace."""

  def __init__(self, model: ImitativeModel, lr: float = 1e-3, weight_decay: float = 0.0, noise_level: float = 1e-2,
               max_batch: int = 512, device: Optional[torch.device] = None, betas=(0.9, 0.999), eps: float = 1e-8,
               group=None) -> None:
    if not torch.cuda.is_available():
      raise RuntimeError("oatomobile_amd.DIMTrainer needs a ROCm device; there is no CPU path.")
    self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if self._device.index is None:
      self._device = torch.device("cuda", torch.cuda.current_device())
    self._model = model
    self._C = model._in_channels
    self._lr, self._wd, self._noise = float(lr), float(weight_decay), float(noise_level)
    self._betas, self._eps = (float(betas[0]), float(betas[1])), float(eps)
    self._group = group
    if group is None and torch.distributed.is_available() and torch.distributed.is_initialized() \
        and torch.distributed.get_world_size() > 1:
      # torch's convention is group=None == the default WORLD group; here None means "do not reduce" (ranks that train
      # independent ensemble members).  Say so once instead of silently training unsynchronised replicas.
      import warnings
      warnings.warn("DIMTrainer(group=None) in a %d-rank job: gradients are NOT all-reduced (independent replicas).  "
                    "Pass group=torch.distributed.group.WORLD for data-parallel training." %
                    torch.distributed.get_world_size(), stacklevel=2)
    self._max_batch = int(max_batch)
    self._lib = _lib.load()
    n = int(self._lib.rip_train_numel(self._C))
    if n != arch.packed_numel(self._C):
      raise RuntimeError("packed layout mismatch: library %d, arch.packed_spec %d" % (n, arch.packed_numel(self._C)))
    self._h = ctypes.c_void_p(0)
    _lib.check(self._lib.rip_train_create(ctypes.byref(self._h), self._C, self._max_batch, self._device.index))
    mask = np.empty(n, np.uint8)
    _lib.check(self._lib.rip_train_trainable_mask(self._h, mask.ctypes.data_as(ctypes.c_void_p), n))
    self.params = torch.from_numpy(model.packed_weights()).to(self._device)
    self.grads = torch.zeros_like(self.params)
    self.exp_avg = torch.zeros_like(self.params)
    self.exp_avg_sq = torch.zeros_like(self.params)
    self._trainable = torch.from_numpy(mask).to(self._device)
    self._loss = torch.zeros((), device=self._device)
    self.step_count = 0
    nbt = [v for k, v in model.state_dict().items() if k.endswith("num_batches_tracked")]
    self.num_batches_tracked = int(nbt[0]) if nbt else 0  # nn.BatchNorm2d counts its train-mode forward passes

  # ---- the reference's train_step, in its two halves ----
  def backward(self, batch: Mapping[str, torch.Tensor], *, y: Optional[torch.Tensor] = None,
               dropout_mask: Optional[torch.Tensor] = None, train: bool = True, gradients: bool = True) -> torch.Tensor:
    """train.py:181-204: perturbs the target, runs the forward pass in train mode and back-propagates
    `-mean(log_prob - logabsdet)`; gradients land in `self.grads`.  Returns the loss (device scalar).
    `train=False`: running statistics, no dropout, no perturbation ("frozen" BatchNorm); `gradients=False`: forward
    only — loss and `self.z`, `self.grads` is left alone (`evaluate_step`)."""
    vis = batch["visual_features"]
    if not vis.is_cuda:
      raise RuntimeError("oatomobile_amd.DIMTrainer: the batch is on %s — no CPU path" % (vis.device,))
    vis = vis.detach().to(torch.float32).contiguous()
    B = vis.shape[0]
    _lib.expect_shape(vis, (None, self._C, arch.INPUT_HW, arch.INPUT_HW), "visual_features")
    if B > self._max_batch:
      raise ValueError("batch of %d exceeds max_batch=%d" % (B, self._max_batch))
    vec = torch.cat([batch["velocity"].reshape(B, 3), batch["is_at_traffic_light"].reshape(B, 1),
                     batch["traffic_light_state"].reshape(B, 1)], dim=-1).to(torch.float32).contiguous()
    target = batch["player_future"][..., :2].to(torch.float32)
    _lib.expect_shape(target, (B, arch.T, 2), "player_future[..., :2]")
    if y is None:
      y = torch.normal(mean=target, std=float(self._noise)) if train else target  # train.py:184-189 (one launch)
    y = y.to(self._device, torch.float32).contiguous()
    if train and dropout_mask is None:
      # keep with probability 1 - p, scaled by 1 / (1 - p) (nn.Dropout): two launches
      dropout_mask = torch.empty(B, arch.LAST_CHANNELS, device=self._device).bernoulli_(1.0 - DROPOUT_P).mul_(1.0 / (1.0 - DROPOUT_P))
    if dropout_mask is not None:
      dropout_mask = dropout_mask.to(self._device, torch.float32).contiguous()
      _lib.expect_shape(dropout_mask, (B, arch.LAST_CHANNELS), "dropout_mask")
    self.z = torch.empty(B, arch.HIDDEN_SIZE, device=self._device)
    _lib.check(self._lib.rip_train_forward_backward(
        self._h, _lib.ptr(self.params), _lib.ptr(self.grads if gradients else None), _lib.ptr(vis), _lib.ptr(vec), _lib.ptr(y),
        _lib.ptr(dropout_mask), B, int(train), _lib.ptr(self._loss.view(1)), _lib.ptr(self.z),
        _lib.current_stream(self._device)))
    if train:
      self.num_batches_tracked += 1
    return self._loss.clone()

  def allreduce(self) -> None:
    """Data-parallel exchange: averages the packed gradient vector over the ranks of the `group` this trainer was
    built with (DistributedDataParallel semantics; BatchNorm running statistics stay per-rank buffers, there is no
    SyncBatchNorm).  Only a trainer that was GIVEN a group reduces: a job whose ranks train independent models
    (replay-sharded ensembles) keeps `group=None` and its gradients are never touched.  ONE all-reduce of 9.7 MB."""
    if self._group is None:
      return
    dist = torch.distributed
    with _lib.trace_range("rip all_reduce gradients (%d B)" % (self.grads.numel() * 4)):
      dist.all_reduce(self.grads, op=dist.ReduceOp.SUM, group=self._group)
    world = dist.get_world_size(self._group)
    if world > 1:
      self.grads /= world

  def clip_grad_norm(self, max_norm: float = 1.0) -> torch.Tensor:
    """train.py:207-208: `torch.nn.utils.clip_grad_norm(model.parameters(), 1.0)` on the packed gradient vector (the
    running-statistic slots hold zeros, so its 2-norm is the norm over the parameters).  Returns the norm."""
    norm = torch.linalg.vector_norm(self.grads)
    self.grads *= torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    return norm

  def apply(self, clip: bool = False) -> None:
    """train.py:206-211: [all-reduce ->] [clip ->] `optimizer.step()` (torch.optim.Adam defaults).  The order is
    DistributedDataParallel's: the AVERAGED gradient is clipped, not each rank's local one."""
    self.allreduce()
    if clip:
      self.clip_grad_norm(1.0)
    self.step_count += 1
    _lib.check(self._lib.rip_train_adam(
        _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
        _lib.ptr(self._trainable, torch.uint8), self.params.numel(), self.step_count, self._lr, self._betas[0],
        self._betas[1], self._eps, self._wd, _lib.current_stream(self._device)))

  def train_step(self, batch: Mapping[str, torch.Tensor], *, y: Optional[torch.Tensor] = None,
                 dropout_mask: Optional[torch.Tensor] = None, clip: bool = False) -> torch.Tensor:
    """train.py:175-213."""
    loss = self.backward(batch, y=y, dropout_mask=dropout_mask)
    self.apply(clip=clip)
    return loss

  def evaluate_step(self, batch: Mapping[str, torch.Tensor]) -> torch.Tensor:
    """train.py:229-249 (model.eval(): running statistics, no dropout, the unperturbed target)."""
    return self.backward(batch, train=False, gradients=False)

  def peek(self, layer: int, what: str = "post") -> torch.Tensor:
    """What the last `backward` saved for conv layer `layer` (0 = features.0, ...), as NCHW: "pre" (conv output
    before BatchNorm), "post" (after BatchNorm / ReLU6 / residual) or "grad" (dLoss/dpost)."""
    spec = arch.conv_layers(self._C)[layer]
    B = self.z.shape[0]
    buf = torch.empty(B, spec.h_out, spec.h_out, spec.cout, device=self._device)
    _lib.check(self._lib.rip_train_peek(self._h, layer, {"pre": 0, "post": 1, "grad": 2}[what], B, _lib.ptr(buf), buf.numel(),
                                        _lib.current_stream(self._device)))
    return buf.permute(0, 3, 1, 2)

  # ---- views of the packed vectors in the reference's state_dict terms ----
  def _unpack(self, vector: torch.Tensor):
    out, pos = {}, 0
    for key, shape in arch.packed_spec(self._C):
      n = int(np.prod(shape)) if len(shape) else 1
      out[key] = vector[pos:pos + n].view(*shape)
      pos += n
    return out

  def named_gradients(self):
    """`{state_dict key: gradient view}` (running-statistic keys hold zeros)."""
    return self._unpack(self.grads)

  def state_dict(self):
    """The trained weights as a reference-compatible `state_dict` (device tensors; `num_batches_tracked` counters are
    the number of train-mode forward passes, like nn.BatchNorm2d keeps them)."""
    sd = self._unpack(self.params)
    full = {}
    for key, _ in arch.state_dict_spec(self._C):
      if key.endswith("num_batches_tracked"):
        full[key] = torch.tensor(self.num_batches_tracked, dtype=torch.long)
      else:
        full[key] = sd[key].detach().clone()
    return full

  def sync_to_model(self) -> ImitativeModel:
    """Writes the trained weights into the wrapped `ImitativeModel` (its inference handle and every agent holding
    the model re-upload on their next call)."""
    self._model.load_state_dict({k: v.to(self._model.device) for k, v in self.state_dict().items()}, strict=True)
    return self._model

  def close(self) -> None:
    if self._h:
      self._lib.rip_train_destroy(self._h)
      self._h = ctypes.c_void_p(0)

  def __del__(self):
    try:
      self.close()
    except Exception:  # interpreter shutdown
      pass
