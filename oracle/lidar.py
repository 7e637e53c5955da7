"""CPU restatement of the LIDAR point-cloud -> bird's-eye-view histogram that produces the
`lidar` observation the hot path starts from.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/ and tools/ as the checker.

Reference: `carla_lidar_measurement_to_ndarray` (oatomobile/utils/carla.py:165-233):
  * the raw float32 buffer is viewed as [P, 3] points (utils/carla.py:212-213);
  * points with z <= -2.5 form the `below` cloud, z >= -2.5 the `above` cloud (:216-217; z == -2.5 is in both);
  * each cloud is histogrammed over (x, y) with `np.histogramdd` on the edges
    `np.linspace(-50, 51, 201)` (:190-201; 200 bins of 0.505 m, NOT 0.5 m: the upper edge is 51);
  * counts are clipped at 5 and divided by 5 (:203-205), stacked on the last axis -> float32 [200, 200, 2].
`np.histogramdd` itself is the third-party algorithm (numpy, version pinned by the image: 2.2): half-open bins
[e_i, e_{i+1}) found with `searchsorted(edges, x, side="right")`, the last bin closed on the right, outliers and NaNs
dropped.  Pinned against the reference function itself in tests/golden/g9_lidar.npz (tools/make_golden.py).
"""
import numpy as np

PIXELS_PER_METER = 2
HIST_MAX_PER_PIXEL = 5
METERS_MAX = 50


def bev_edges(pixels_per_meter: int = PIXELS_PER_METER, meters_max: int = METERS_MAX) -> np.ndarray:
  """utils/carla.py:190-194 (x) and :195-199 (y): identical edge vectors, float64."""
  return np.linspace(-meters_max, meters_max + 1, meters_max * 2 * pixels_per_meter + 1)


def splat_points(point_cloud: np.ndarray, pixels_per_meter: int = PIXELS_PER_METER,
                 hist_max_per_pixel: int = HIST_MAX_PER_PIXEL, meters_max: int = METERS_MAX) -> np.ndarray:
  """utils/carla.py:182-208."""
  edges = bev_edges(pixels_per_meter, meters_max)
  hist = np.histogramdd(point_cloud[..., :2], bins=(edges, edges))[0]
  hist[hist > hist_max_per_pixel] = hist_max_per_pixel
  return hist / hist_max_per_pixel


def lidar_to_bev(points: np.ndarray) -> np.ndarray:
  """utils/carla.py:211-233 on an already parsed [P, 3] float32 point array."""
  points = np.asarray(points, dtype=np.float32).reshape(-1, 3)
  below = points[points[..., 2] <= -2.5]
  above = points[points[..., 2] >= -2.5]
  return np.stack([splat_points(below), splat_points(above)], axis=-1).astype(np.float32)


def bin_index(values: np.ndarray) -> np.ndarray:
  """Bin of each float32 coordinate under numpy's rule, -1 for outliers / NaN (what the HIP kernel restates with a
  guess + fix-up against the same float64 edge table)."""
  e = bev_edges()
  idx = np.searchsorted(e, values.astype(np.float64), side="right") - 1
  idx[values.astype(np.float64) == e[-1]] = len(e) - 2
  idx[(idx < 0) | (idx > len(e) - 2) | np.isnan(values)] = -1
  return idx
