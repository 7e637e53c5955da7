"""TEST INFRASTRUCTURE — stand-in for torchvision v0.6.0 `mobilenet_v2`.

The reference builds its encoder with
`torch.hub.load(github="pytorch/vision:v0.6.0", model="mobilenet_v2", num_classes=...)`
(oatomobile/torch/networks/perception.py:36-40).  torchvision is neither
vendored in /root/reference nor installed here and there is no network, so the
encoder arithmetic is **parity-unpinned** by the reference itself.  This module
re-creates the published v0.6.0 module tree from its architecture constants
(SURVEY.md §8c): same child names => same `state_dict` keys, same PyTorch
layer definitions (Conv2d/BatchNorm2d(eps=1e-5)/ReLU6/Linear).  It is what
`tools/make_golden.py` injects in place of `torch.hub.load`, and what the CPU
oracle runs.
"""

import torch
import torch.nn as nn


def _make_divisible(v: float, divisor: int = 8) -> int:
  new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
  if new_v < 0.9 * v:
    new_v += divisor
  return new_v


def conv_bn_relu6(inp: int, oup: int, kernel: int = 3, stride: int = 1, groups: int = 1) -> nn.Sequential:
  """children '0' conv (bias-free), '1' BN, '2' ReLU6 — torchvision's ConvBNReLU."""
  return nn.Sequential(
      nn.Conv2d(inp, oup, kernel, stride, (kernel - 1) // 2, groups=groups, bias=False),
      nn.BatchNorm2d(oup),
      nn.ReLU6(inplace=True),
  )


class InvertedResidual(nn.Module):

  def __init__(self, inp: int, oup: int, stride: int, expand_ratio: int) -> None:
    super().__init__()
    hidden = int(round(inp * expand_ratio))
    self.use_res_connect = stride == 1 and inp == oup
    layers = []
    if expand_ratio != 1:
      layers.append(conv_bn_relu6(inp, hidden, kernel=1))
    layers += [
        conv_bn_relu6(hidden, hidden, stride=stride, groups=hidden),
        nn.Conv2d(hidden, oup, 1, 1, 0, bias=False),
        nn.BatchNorm2d(oup),
    ]
    self.conv = nn.Sequential(*layers)

  def forward(self, x: torch.Tensor) -> torch.Tensor:
    y = self.conv(x)
    return x + y if self.use_res_connect else y


class MobileNetV2(nn.Module):
  SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
             (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))

  def __init__(self, num_classes: int = 1000, width_mult: float = 1.0) -> None:
    super().__init__()
    inp = _make_divisible(32 * width_mult)
    last = _make_divisible(1280 * max(1.0, width_mult))
    features = [conv_bn_relu6(3, inp, stride=2)]
    for t, c, n, s in self.SETTING:
      oup = _make_divisible(c * width_mult)
      for i in range(n):
        features.append(InvertedResidual(inp, oup, s if i == 0 else 1, t))
        inp = oup
    features.append(conv_bn_relu6(inp, last, kernel=1))
    self.features = nn.Sequential(*features)
    self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(last, num_classes))
    for m in self.modules():
      if isinstance(m, nn.Conv2d):
        nn.init.kaiming_normal_(m.weight, mode="fan_out")
      elif isinstance(m, nn.BatchNorm2d):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)
      elif isinstance(m, nn.Linear):
        nn.init.normal_(m.weight, 0, 0.01)
        nn.init.zeros_(m.bias)

  def forward(self, x: torch.Tensor) -> torch.Tensor:
    x = self.features(x)
    x = nn.functional.adaptive_avg_pool2d(x, 1).reshape(x.shape[0], -1)
    return self.classifier(x)


def mobilenet_v2(num_classes: int = 1000, **_unused) -> MobileNetV2:
  return MobileNetV2(num_classes=num_classes)
