"""TEST INFRASTRUCTURE — CPU oracle of the DIM training step.

Restates `train_step` of oatomobile/baselines/torch/dim/train.py:175-213 on `OracleImitativeModel` (PyTorch CPU
autograd): train mode (BatchNorm batch statistics + running-stat update, momentum 0.1, unbiased running variance;
Dropout(0.2) of the MobileNetV2 classifier), loss = -mean(log_prob - logabsdet) of the teacher-forced flow inverse on
the perturbed target, Adam (torch defaults, lr 1e-3, :112-116).  The two random draws of the reference — the
perturbation of the target (:184-189) and the dropout mask — are INPUTS here, so a step is reproducible.  Pinned to
the reference by tests/golden/g15_train_step.npz (tools/make_golden_host.py runs the reference's own model through the
same lines).  Only tests/ import this file.
"""

from typing import Mapping, Optional

import numpy as np
import torch

from oracle import reference_cpu as O


class _MaskedDropout(torch.nn.Module):
  """Dropout with the keep/scale mask supplied by the caller (values 0 or 1/(1-p))."""

  def __init__(self) -> None:
    super().__init__()
    self.mask: Optional[torch.Tensor] = None

  def forward(self, x: torch.Tensor) -> torch.Tensor:
    return x if self.mask is None else x * self.mask


class _ReLU6WithMask(torch.autograd.Function):
  """y = clamp(x, 0, 6) whose backward uses a GIVEN pass-through mask instead of the one x implies."""

  @staticmethod
  def forward(ctx, x, mask):
    ctx.save_for_backward(mask)
    return x.clamp(0.0, 6.0)

  @staticmethod
  def backward(ctx, grad):
    (mask,) = ctx.saved_tensors
    return grad * mask, None


class _KinkReLU6(torch.nn.Module):
  """ReLU6; with `mask` set, the derivative is that mask (1 where the gradient passes).  A gradient through ReLU6 is
  defined up to the decisions at the kinks: an implementation whose forward differs in the 7th digit decides the
  ~1e-7 fraction of activations that sit ON a kink differently, and every such element moves a per-channel gradient by
  ~1/(B*H*W).  To compare backward passes like with like the checker can take the kink decisions of the implementation
  under test (`set_kink_masks`)."""

  def __init__(self) -> None:
    super().__init__()
    self.mask: Optional[torch.Tensor] = None

  def forward(self, x: torch.Tensor) -> torch.Tensor:
    return torch.nn.functional.relu6(x) if self.mask is None else _ReLU6WithMask.apply(x, self.mask)


def trainable_model(state_dict: Mapping[str, np.ndarray], in_channels: int = 2) -> O.OracleImitativeModel:
  m = O.OracleImitativeModel.from_numpy_state_dict(state_dict, in_channels)
  for p in m.parameters():
    p.requires_grad_(True)
  m._encoder._model.classifier[0] = _MaskedDropout()

  def swap(mod):
    for name, child in mod.named_children():
      if isinstance(child, torch.nn.ReLU6):
        setattr(mod, name, _KinkReLU6())
      else:
        swap(child)

  swap(m._encoder)
  m.train()
  return m


def kink_modules(model: O.OracleImitativeModel):
  """The ReLU6 modules in network order (one per conv layer that is followed by ReLU6)."""
  return [mod for mod in model._encoder.modules() if isinstance(mod, _KinkReLU6)]


def set_kink_masks(model: O.OracleImitativeModel, post_activations) -> None:
  """`post_activations`: the post-ReLU6 outputs [B,C,H,W] of the implementation under test, network order (or None
  to go back to the oracle's own decisions)."""
  mods = kink_modules(model)
  if post_activations is None:
    for mod in mods:
      mod.mask = None
    return
  assert len(mods) == len(post_activations), (len(mods), len(post_activations))
  for mod, post in zip(mods, post_activations):
    mod.mask = ((post > 0) & (post < 6)).to(torch.float32)


def loss_and_grads(model: O.OracleImitativeModel, visual_features: torch.Tensor, velocity: torch.Tensor,
                   is_at_traffic_light: torch.Tensor, traffic_light_state: torch.Tensor, y: torch.Tensor,
                   dropout_mask: Optional[torch.Tensor]):
  """dim/train.py:182-204: returns (loss, z); gradients are left in `p.grad`."""
  for p in model.parameters():
    p.grad = None
  model._encoder._model.classifier[0].mask = dropout_mask
  z = O.params(model, visual_features, velocity, is_at_traffic_light, traffic_light_state)  # :192-197
  _, log_prob, logabsdet = O.flow_inverse(model, y, z)  # :198
  loss = -torch.mean(log_prob - logabsdet, dim=0)  # :201
  loss.backward()  # :204
  return loss.detach(), z.detach()


def make_adam(model: O.OracleImitativeModel, lr: float = 1e-3, weight_decay: float = 0.0) -> torch.optim.Adam:
  return torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)  # dim/train.py:112-116
