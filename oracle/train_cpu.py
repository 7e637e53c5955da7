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


def trainable_model(state_dict: Mapping[str, np.ndarray], in_channels: int = 2) -> O.OracleImitativeModel:
  m = O.OracleImitativeModel.from_numpy_state_dict(state_dict, in_channels)
  for p in m.parameters():
    p.requires_grad_(True)
  m._encoder._model.classifier[0] = _MaskedDropout()
  m.train()
  return m


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
