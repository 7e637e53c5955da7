"""ctypes binding of librip_hip.so (C ABI: include/rip_hip.h).

The product path has no CPU fallback: if the shared library is missing or does
not load, importing a symbol from here raises — run `python -c "import
__graft_entry__ as g; g.build()"` (hipcc, gfx950) first.
"""

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librip_hip.so")

# (name, restype, argtypes) — must list every symbol include/rip_hip.h declares.
SIGNATURES = [
    ("rip_abi_version", c_int, []),
    ("rip_last_error", c_char_p, []),
    ("rip_create", c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int]),
    ("rip_destroy", c_int, [c_void_p]),
    ("rip_load_model", c_int, [c_void_p, c_int, c_void_p, c_size_t]),
    ("rip_transform", c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    ("rip_encode", c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    ("rip_encode_tap", c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    ("rip_encode_tap_k", c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    ("rip_kernel_log", c_int, [c_void_p, c_char_p, c_size_t]),
    ("rip_encode_raw", c_int,
     [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    ("rip_encode_raw_u8", c_int,
     [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    ("rip_flow_forward", c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    ("rip_flow_inverse", c_int,
     [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("rip_goal_likelihood", c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    ("rip_score", c_int,
     [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    ("rip_aggregate_scores", c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    ("rip_lidar_bev", c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    ("rip_cil_decode", c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    ("rip_cil_blob_floats", c_int, []),
    ("rip_search", c_int, [
        c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p,
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p
    ]),
    ("rip_mp_local", c_int,
     [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    ("rip_mp_update", c_int, [
        c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p
    ]),
    ("rip_dim_forward", c_int,
     [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p,
      c_void_p]),
    ("rip_act", c_int, [
        c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
        c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p
    ]),
    ("rip_interpolate_plans", c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    ("rip_search_plan", c_int, [c_void_p, c_int, c_int, c_void_p, c_int]),
    ("rip_search_stats", c_int, [c_void_p, c_void_p, c_int]),
    ("rip_encoder_status", c_int, [c_void_p]),
    ("rip_trace_push", c_int, [c_char_p]),
    ("rip_trace_pop", c_int, []),
    ("rip_set_option", c_int, [c_void_p, c_int, c_int]),
    ("rip_num_models", c_int, [c_void_p]),
    ("rip_in_channels", c_int, [c_void_p]),
    ("rip_max_batch", c_int, [c_void_p]),
    ("rip_max_candidates", c_int, [c_void_p]),
    ("rip_train_numel", c_size_t, [c_int]),
    ("rip_train_create", c_int, [POINTER(c_void_p), c_int, c_int, c_int]),
    ("rip_train_destroy", c_int, [c_void_p]),
    ("rip_train_trainable_mask", c_int, [c_void_p, c_void_p, c_size_t]),
    ("rip_train_forward_backward", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    ("rip_train_peek", c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    ("rip_train_num_layers", c_int, [c_void_p]),
    ("rip_train_adam", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_float, c_float, c_float, c_float, c_float,
      c_void_p]),
]
ABI_VERSION = 4

ALGORITHMS = {"WCM": 0, "MA": 1, "BCM": 2}
ENC_DTYPES = {"fp32": 0, "bf16": 1}
OPT_SEARCH_KERNEL, OPT_ENCODER_FUSED, OPT_SEARCH_REGROUP, OPT_ENCODER_MEGA, OPT_DEBUG_ENCODER_FAULT = 0, 1, 2, 3, 4
OPT_ENCODER_VARIANT, OPT_KERNEL_LOG = 5, 6
ENC_VAR_IRB_ROUND3, ENC_VAR_FRONT_ROUND3, ENC_VAR_ROWS_F5_7, ENC_VAR_F17_LAYERWISE = 1, 2, 4, 8
ENC_VAR_FP32_LAYERWISE = 16  # fp32 encoder without the split-f16 tile blocks (encoder_split_tile.hip)
SEARCH_KERNELS = {"auto": 0, "chain": 1, "phase": 3, "split": 4, "pair": 5}  # "pair": split-f16, paired workgroup shape forced


def search_kernel_id(name: str) -> int:
  """`search_kernel=` of the agents -> the value of `OPT_SEARCH_KERNEL`; a `ValueError` names the valid kernels
  (round 1's "mfma" kernel, option 2, was removed in round 5: the library answers RIP_EINVAL for it)."""
  try:
    return SEARCH_KERNELS[name]
  except KeyError:
    hint = " (\"mfma\", round 1's wave-per-model kernel, was removed: use \"split\" or \"phase\")" if name == "mfma" else ""
    raise ValueError("unknown search_kernel %r%s; valid: %s" % (name, hint, ", ".join(sorted(SEARCH_KERNELS)))) from None

_lib = None


RIP_OK, RIP_EINVAL, RIP_EHIP, RIP_ESTATE = 0, -1, -2, -3  # include/rip_hip.h


class RipError(RuntimeError):
  """A librip_hip.so entry point returned a negative code."""


def load() -> ctypes.CDLL:
  """Loads librip_hip.so once and types every entry point; raises if it is absent."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise ImportError(
        "oatomobile_amd: %s is missing — the HIP extension is not built "
        "(run `python -c \"import __graft_entry__ as g; g.build()\"`). There is no CPU fallback." % LIB_PATH)
  lib = ctypes.CDLL(LIB_PATH)
  for name, restype, argtypes in SIGNATURES:
    fn = getattr(lib, name)  # AttributeError if the .so is stale
    fn.restype = restype
    fn.argtypes = argtypes
  _lib = lib
  return lib


def check(rc: int) -> None:
  if rc != 0:
    msg = load().rip_last_error()
    raise RipError("librip_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def ptr(t, dtype=None) -> c_void_p:
  """Device pointer of a contiguous fp32 (or `dtype`) HIP tensor, or NULL for None.  Raw pointers cross the C ABI
  unchecked on the other side, so dtype / residency / contiguity are enforced here."""
  if t is None:
    return c_void_p(0)
  import torch
  want = torch.float32 if dtype is None else dtype
  if not t.is_cuda:
    raise RuntimeError("oatomobile_amd: expected a ROCm device tensor, got one on %s (no CPU path)" % (t.device,))
  if t.dtype != want:
    raise ValueError("oatomobile_amd: expected a %s tensor, got %s" % (want, t.dtype))
  if not t.is_contiguous():
    raise ValueError("oatomobile_amd: expected a contiguous tensor (shape %s, strides %s)" % (tuple(t.shape), t.stride()))
  return c_void_p(t.data_ptr())


def current_stream(device=None) -> c_void_p:
  """torch's current stream ON `device` (a torch.device, an index or a tensor) — not on torch's current device:
  a handle on cuda:1 must never be handed cuda:0's stream."""
  import torch
  if device is not None and hasattr(device, "device"):
    device = device.device
  return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def expect_shape(t, shape, what: str) -> None:
  """ValueError unless `t.shape` matches `shape` (None = any extent)."""
  got = tuple(t.shape)
  if len(got) != len(shape) or any(w is not None and w != g for w, g in zip(shape, got)):
    raise ValueError("%s must have shape %s, got %s" % (what, "[" + ",".join("*" if w is None else str(w) for w in shape) + "]", got))


class Handle:
  """Owns one `rip_handle*` (K models on one device)."""

  def __init__(self, num_models: int, in_channels: int, max_batch: int, device_index: int,
               max_candidates: int = 1) -> None:
    self._lib = load()
    self._h = c_void_p(0)
    check(self._lib.rip_create(ctypes.byref(self._h), num_models, in_channels, max_batch, max_candidates, device_index))
    self.num_models, self.in_channels, self.max_batch, self.device_index = num_models, in_channels, max_batch, device_index
    self.max_candidates = max_candidates

  def stream(self) -> c_void_p:
    """torch's current stream on THIS handle's device."""
    return current_stream(self.device_index)

  @property
  def raw(self) -> c_void_p:
    return self._h

  def load_model(self, k: int, packed) -> None:
    import numpy as np
    packed = np.ascontiguousarray(packed, dtype=np.float32)
    check(self._lib.rip_load_model(self._h, k, packed.ctypes.data_as(c_void_p), packed.size))

  def set_option(self, option: int, value: int) -> None:
    check(self._lib.rip_set_option(self._h, option, value))

  def kernel_log(self) -> list:
    """The encoder kernels the handle's last encode / tap call launched (after `set_option(OPT_KERNEL_LOG, 1)`), one
    "kernel<template arguments> grid=(x,y,z) block=n" string per launch."""
    n = self._lib.rip_kernel_log(self._h, None, 0)
    buf = ctypes.create_string_buffer(n + 1)
    self._lib.rip_kernel_log(self._h, buf, n + 1)
    return [l for l in buf.value.decode().split("\n") if l]

  def close(self) -> None:
    if self._h:
      self._lib.rip_destroy(self._h)
      self._h = c_void_p(0)

  def __del__(self):
    try:
      self.close()
    except Exception:  # interpreter shutdown
      pass


class trace_range:
  """`with trace_range("name"):` — a rocTX range through librip_hip.so's tracing hook (RIP_ROCTX=1; a no-op
  otherwise): the host-side counterpart of the ranges rip_encode / rip_search open."""

  def __init__(self, name: str) -> None:
    self._name = name.encode()

  def __enter__(self):
    load().rip_trace_push(self._name)
    return self

  def __exit__(self, *exc):
    load().rip_trace_pop()
    return False
