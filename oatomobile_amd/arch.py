"""Static description of the deep-imitative-model network (host-side tables).

Everything here is plain Python data: the MobileNetV2 layer list the encoder
kernels walk, the `state_dict` key order of the reference `ImitativeModel`
(328 tensors) and the flat packing order the C ABI (`rip_load_model`) consumes.

Reference facts this table restates:
  * encoder = torchvision v0.6.0 `mobilenet_v2(num_classes=128)` with the first
    conv swapped to `in_channels`  (oatomobile/torch/networks/perception.py:36-51)
  * merger  = MLP 133 -> 64 -> 64 -> 64, ReLU after every layer
    (oatomobile/baselines/torch/dim/model.py:56-62, torch/networks/mlp.py:51-68)
  * decoder = GRUCell(2, 64) + MLP 64 -> 32 -> 4
    (oatomobile/torch/networks/sequence.py:53-65)
"""

from typing import List, NamedTuple, Tuple

# (expand ratio t, output channels c, repeats n, first stride s)
INVERTED_RESIDUAL_SETTING: Tuple[Tuple[int, int, int, int], ...] = (
    (1, 16, 1, 1),
    (6, 24, 2, 2),
    (6, 32, 3, 2),
    (6, 64, 4, 2),
    (6, 96, 3, 1),
    (6, 160, 3, 2),
    (6, 320, 1, 1),
)
STEM_CHANNELS = 32
LAST_CHANNELS = 1280
NUM_FEATURES = 128  # classifier width == MobileNetV2(num_classes=128)
VECTOR_INPUTS = 5  # velocity[3] + is_at_traffic_light[1] + traffic_light_state[1]
MERGER_SIZES = (64, 64, 64)
HIDDEN_SIZE = 64
T = 4  # trajectory steps: ImitativeModel(output_shape=(4, 2)) (dim/model.py:41)
HEAD_HIDDEN = 32
BN_EPS = 1e-5
INPUT_HW = 100  # after the 200 -> 100 bilinear down-sample


class Block(NamedTuple):
  """One inverted-residual block (torchvision `features.{index}`)."""
  index: int
  inp: int
  oup: int
  hidden: int
  stride: int
  expand: bool  # False only for the t == 1 block
  residual: bool
  h_in: int
  h_out: int


def conv_out(h: int, stride: int) -> int:
  """3x3, padding 1."""
  return (h + 2 - 3) // stride + 1


def blocks(input_hw: int = INPUT_HW) -> List[Block]:
  out = []
  h = conv_out(input_hw, 2)  # stem
  inp = STEM_CHANNELS
  index = 1
  for t, c, n, s in INVERTED_RESIDUAL_SETTING:
    for i in range(n):
      stride = s if i == 0 else 1
      h_out = conv_out(h, stride)
      out.append(
          Block(index=index, inp=inp, oup=c, hidden=inp * t, stride=stride,
                expand=(t != 1), residual=(stride == 1 and inp == c), h_in=h,
                h_out=h_out))
      inp, h, index = c, h_out, index + 1
  return out


class ConvLayer(NamedTuple):
  """One conv + BatchNorm of the MobileNetV2 stack, in network order (what rip_train_peek indexes)."""
  name: str    # state_dict prefix of the conv weight's module, e.g. "_encoder._model.features.2.conv.0.0"
  cout: int
  h_out: int
  relu6: bool  # followed by ReLU6 (every conv but the linear-bottleneck projections)


def conv_layers(in_channels: int = 2, input_hw: int = INPUT_HW) -> List[ConvLayer]:
  f = "_encoder._model.features."
  out = [ConvLayer(f + "0.0", STEM_CHANNELS, conv_out(input_hw, 2), True)]
  for b in blocks(input_hw):
    p = f + "%d.conv." % b.index
    j = 0
    if b.expand:
      out.append(ConvLayer(p + "0.0", b.hidden, b.h_in, True))
      j = 1
    out.append(ConvLayer(p + "%d.0" % j, b.hidden, b.h_out, True))
    out.append(ConvLayer(p + "%d" % (j + 1), b.oup, b.h_out, False))
  out.append(ConvLayer(f + "18.0", LAST_CHANNELS, out[-1].h_out, True))
  return out


def _bn(prefix: str, c: int):
  return [
      (prefix + ".weight", (c,)),
      (prefix + ".bias", (c,)),
      (prefix + ".running_mean", (c,)),
      (prefix + ".running_var", (c,)),
      (prefix + ".num_batches_tracked", ()),
  ]


def state_dict_spec(in_channels: int = 2):
  """Ordered `(key, shape)` list of the reference `ImitativeModel.state_dict()`."""
  spec = []
  f = "_encoder._model.features."
  spec.append((f + "0.0.weight", (STEM_CHANNELS, in_channels, 3, 3)))
  spec += _bn(f + "0.1", STEM_CHANNELS)
  for b in blocks():
    p = f + "%d.conv." % b.index
    j = 0
    if b.expand:
      spec.append((p + "0.0.weight", (b.hidden, b.inp, 1, 1)))
      spec += _bn(p + "0.1", b.hidden)
      j = 1
    spec.append((p + "%d.0.weight" % j, (b.hidden, 1, 3, 3)))
    spec += _bn(p + "%d.1" % j, b.hidden)
    spec.append((p + "%d.weight" % (j + 1), (b.oup, b.hidden, 1, 1)))
    spec += _bn(p + "%d" % (j + 2), b.oup)
  last_in = INVERTED_RESIDUAL_SETTING[-1][1]
  spec.append((f + "18.0.weight", (LAST_CHANNELS, last_in, 1, 1)))
  spec += _bn(f + "18.1", LAST_CHANNELS)
  spec.append(("_encoder._model.classifier.1.weight", (NUM_FEATURES, LAST_CHANNELS)))
  spec.append(("_encoder._model.classifier.1.bias", (NUM_FEATURES,)))
  sizes = (NUM_FEATURES + VECTOR_INPUTS,) + MERGER_SIZES
  for i in range(3):
    spec.append(("_merger._model.%d.weight" % (2 * i), (sizes[i + 1], sizes[i])))
    spec.append(("_merger._model.%d.bias" % (2 * i), (sizes[i + 1],)))
  g = 3 * HIDDEN_SIZE
  spec.append(("_decoder._decoder.weight_ih", (g, 2)))
  spec.append(("_decoder._decoder.weight_hh", (g, HIDDEN_SIZE)))
  spec.append(("_decoder._decoder.bias_ih", (g,)))
  spec.append(("_decoder._decoder.bias_hh", (g,)))
  spec.append(("_decoder._locscale._model.0.weight", (HEAD_HIDDEN, HIDDEN_SIZE)))
  spec.append(("_decoder._locscale._model.0.bias", (HEAD_HIDDEN,)))
  spec.append(("_decoder._locscale._model.2.weight", (4, HEAD_HIDDEN)))
  spec.append(("_decoder._locscale._model.2.bias", (4,)))
  return spec


def packed_spec(in_channels: int = 2):
  """The fp32 tensors `rip_load_model` consumes, in order: `state_dict_spec`
  minus the int64 `num_batches_tracked` counters."""
  return [(k, s) for (k, s) in state_dict_spec(in_channels)
          if not k.endswith("num_batches_tracked")]


def packed_numel(in_channels: int = 2) -> int:
  n = 0
  for _, shape in packed_spec(in_channels):
    m = 1
    for d in shape:
      m *= d
    n += m
  return n


# ------------------------------------------------------------------------------------------------
# BehaviouralModel (conditional imitation learning; oatomobile/baselines/torch/cil/model.py:34-66):
# the same MobileNetV2 encoder, a merger over 128 + 3 + 1 + 1 + 1 inputs (the extra scalar is `mode`), a bare
# GRUCell(2, 64) and a Linear(64, 2) head unrolled for T = 40 steps.
# ------------------------------------------------------------------------------------------------
CIL_VECTOR_INPUTS = 6
CIL_TIMESTEPS = 40


def cil_decoder_spec():
  """Ordered `(key, shape)` list of the non-encoder tensors of `BehaviouralModel.state_dict()`; also the layout of the
  fp32 blob `rip_cil_decode` consumes."""
  spec = []
  sizes = (NUM_FEATURES + CIL_VECTOR_INPUTS,) + MERGER_SIZES
  for i in range(3):
    spec.append(("_merger._model.%d.weight" % (2 * i), (sizes[i + 1], sizes[i])))
    spec.append(("_merger._model.%d.bias" % (2 * i), (sizes[i + 1],)))
  g = 3 * HIDDEN_SIZE
  spec.append(("_decoder.weight_ih", (g, 2)))
  spec.append(("_decoder.weight_hh", (g, HIDDEN_SIZE)))
  spec.append(("_decoder.bias_ih", (g,)))
  spec.append(("_decoder.bias_hh", (g,)))
  spec.append(("_output.weight", (2, HIDDEN_SIZE)))
  spec.append(("_output.bias", (2,)))
  return spec


def cil_state_dict_spec(in_channels: int = 2):
  """Ordered `(key, shape)` list of the reference `BehaviouralModel.state_dict()`."""
  enc = [(k, s) for (k, s) in state_dict_spec(in_channels) if k.startswith("_encoder.")]
  return enc + cil_decoder_spec()
