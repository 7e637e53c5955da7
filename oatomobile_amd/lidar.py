"""LIDAR point cloud -> bird's-eye-view histogram on the GPU (`rip_lidar_bev`).

Mirror of `carla_lidar_measurement_to_ndarray` (oatomobile/utils/carla.py:165-233), which turns a CARLA
`LidarMeasurement` into the float32 [200, 200, 2] `lidar` observation the imitative models consume.  The CARLA object
itself never reaches this package: pass the parsed points ([P, 3] float32: x, y, z) or the raw buffer
(`LidarMeasurement.raw_data`) through `points_from_raw`.  Batches are ragged lists of clouds.
"""
from typing import Sequence, Union

import numpy as np
import torch

from . import _lib

BEV_SHAPE = (200, 200, 2)


def points_from_raw(raw_data: Union[bytes, bytearray, memoryview]) -> np.ndarray:
  """utils/carla.py:212-213: the raw float32 buffer viewed as [P, 3]."""
  points = np.frombuffer(raw_data, dtype=np.dtype("f4"))
  return np.reshape(points, (int(points.shape[0] / 3), 3))


def lidar_to_bev(points: Union[np.ndarray, torch.Tensor, Sequence[Union[np.ndarray, torch.Tensor]]],
                 device: Union[str, torch.device, None] = None) -> torch.Tensor:
  """Returns the BEV histogram(s) as a CUDA tensor: [200, 200, 2] for one cloud, [B, 200, 200, 2] for a sequence.

  Same values, bit for bit, as the reference function on each cloud (pixels_per_meter=2, hist_max_per_pixel=5,
  meters_max=50: the reference's defaults, the only configuration its callers use)."""
  single = not isinstance(points, (list, tuple))
  clouds = [points] if single else list(points)
  if device is None:
    device = next((c.device for c in clouds if isinstance(c, torch.Tensor) and c.is_cuda), torch.device("cuda"))
  device = torch.device(device)
  if device.type != "cuda":
    raise RuntimeError("oatomobile_amd.lidar.lidar_to_bev needs a ROCm device (no CPU path in this build)")
  parts, counts = [], []
  for c in clouds:
    t = c if isinstance(c, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32))
    if t.dtype != torch.float32:
      raise TypeError("point clouds must be float32, got %s" % t.dtype)
    t = t.reshape(-1, 3)
    parts.append(t.to(device, non_blocking=True))
    counts.append(t.shape[0])
  B = len(parts)
  pts = torch.cat(parts, dim=0).contiguous() if B else torch.zeros(0, 3, device=device)
  offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), device=device)
  bev = torch.empty((B,) + BEV_SHAPE, dtype=torch.float32, device=device)
  if B:
    with torch.cuda.device(device):
      lib = _lib.load()
      _lib.check(lib.rip_lidar_bev(_lib.ptr(pts), _lib.ptr(offsets, torch.int32), B, _lib.ptr(bev),
                                   _lib.current_stream(device)))
  return bev[0] if single else bev
