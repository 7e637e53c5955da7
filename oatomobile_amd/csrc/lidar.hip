// LIDAR point cloud -> bird's-eye-view occupancy histogram for gfx950 (the step that produces the
// `lidar` observation the hot path starts from).
//
// Reference: carla_lidar_measurement_to_ndarray (oatomobile/utils/carla.py:165-233): the [P,3] float32 points are split at
// z = -2.5 m into a `below` (z <= -2.5) and an `above` (z >= -2.5) cloud, each is histogrammed over (x, y) with
// np.histogramdd on the float64 edges np.linspace(-50, 51, 201) (200 bins of 0.505 m), counts are clipped at 5 and
// divided by 5; result float32 [200, 200, 2].  Integer work, bit-exact against the reference (tests/golden/g9_lidar.npz).
//
// Mapping: one workgroup of 1024 threads per observation; the 200x200 counters of ONE height channel live in LDS as
// 16-bit halves of 32-bit words (80 KB), the clipped result of the first channel is parked as bytes (40 KB) while the
// second channel is counted (the point list is read twice, the second time from L2), then both channels are written as
// one coalesced float2 per cell.  HBM traffic = 12 B per point + 320 KB per observation, no global atomics.
//
// Bin rule = numpy's: searchsorted(edges, x, side="right") - 1 on the float64 edge table, last bin closed on the right,
// outliers / NaN dropped.  The table is built on the host with numpy's own formula (arange * step + start, last = stop;
// two roundings, no FMA) and the kernel fixes up an arithmetic guess against it, so the result does not depend on how
// the device rounds the guess.
#include <hip/hip_runtime.h>

#include "flow.h"

namespace rip {

namespace {

constexpr int BEV = 200;               // bins per axis
constexpr int CELLS = BEV * BEV;       // 40 000
constexpr int BEV_THREADS = 1024;
constexpr int HIST_MAX = 5;

__constant__ double c_edges[BEV + 1];

// `e`: the edge table in LDS (per-lane indices: from constant memory this is a divergent vector load per look-up)
__device__ __forceinline__ int bev_bin(float v, const double* e) {
  const double x = (double)v;
  if (!(x >= -50.0) || !(x <= 51.0)) return -1;  // outliers and NaN (both comparisons false); e[0], e[200] are exact
  if (x == 51.0) return BEV - 1;                 // the last bin is closed on the right
  int g = (int)((x + 50.0) * (200.0 / 101.0));
  g = g < 0 ? 0 : (g > BEV - 1 ? BEV - 1 : g);
  // the guess is off by at most one: a single conditional step each way, checked against the table
  g -= (g > 0 && x < e[g]) ? 1 : 0;
  g += (g < BEV - 1 && x >= e[g + 1]) ? 1 : 0;
  g -= (g > 0 && x < e[g]) ? 1 : 0;
  return g;
}

__global__ __launch_bounds__(BEV_THREADS) void lidar_bev_kernel(const float* __restrict__ points,
                                                                 const int* __restrict__ offsets,
                                                                 float* __restrict__ bev) {
  __shared__ unsigned cnt[CELLS / 2];       // two 16-bit counters per word
  __shared__ unsigned char first[CELLS];    // clipped counts of channel 0
  __shared__ double edges[BEV + 1];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid <= BEV) edges[tid] = c_edges[tid];
  const int p0 = offsets[b], p1 = offsets[b + 1];
  for (int ch = 0; ch < 2; ++ch) {
    for (int i = tid; i < CELLS / 2; i += BEV_THREADS) cnt[i] = 0u;
    __syncthreads();
    for (int p = p0 + tid; p < p1; p += BEV_THREADS) {
      const float x = points[(size_t)p * 3], y = points[(size_t)p * 3 + 1], z = points[(size_t)p * 3 + 2];
      const bool take = ch == 0 ? (z <= -2.5f) : (z >= -2.5f);  // utils/carla.py:216-217 (z == -2.5 counts in both)
      if (!take) continue;
      const int bx = bev_bin(x, edges), by = bev_bin(y, edges);
      if (bx < 0 || by < 0) continue;
      const int cell = bx * BEV + by;
      const unsigned sh = 16u * (cell & 1);
      // counts saturate at the clip: a cell is left alone once it shows >= 5, so a 16-bit half can never carry
      // (at most 5 + one increment per thread already past the check)
      if (((cnt[cell >> 1] >> sh) & 0xffffu) < (unsigned)HIST_MAX) atomicAdd(&cnt[cell >> 1], 1u << sh);
    }
    __syncthreads();
    if (ch == 0) {
      for (int c = tid; c < CELLS; c += BEV_THREADS) {
        const unsigned v = (cnt[c >> 1] >> (16u * (c & 1))) & 0xffffu;
        first[c] = (unsigned char)(v > HIST_MAX ? HIST_MAX : v);
      }
      __syncthreads();
    }
  }
  // hist / 5 in float64, then float32 (utils/carla.py:205, :233): k / 5 for k = 0..5
  const float lut[HIST_MAX + 1] = {0.0f, (float)(1.0 / 5.0), (float)(2.0 / 5.0), (float)(3.0 / 5.0), (float)(4.0 / 5.0), 1.0f};
  float2* out = reinterpret_cast<float2*>(bev) + (size_t)b * CELLS;
  for (int c = tid; c < CELLS; c += BEV_THREADS) {
    const unsigned v = (cnt[c >> 1] >> (16u * (c & 1))) & 0xffffu;
    out[c] = make_float2(lut[first[c]], lut[v > HIST_MAX ? HIST_MAX : v]);
  }
}

}  // namespace

hipError_t launch_lidar_bev(const float* points, const int* offsets, int B, float* bev, hipStream_t s) {
  static bool edges_ready[64] = {false};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!edges_ready[dev]) {
#pragma clang fp contract(off)
    double edges[BEV + 1];
    const double start = -50.0, stop = 51.0;
    const double step = (stop - start) / (double)BEV;  // numpy.linspace: delta / div
    for (int i = 0; i <= BEV; ++i) {
      const double m = (double)i * step;  // arange(0, num) * step
      edges[i] = m + start;               // + start
    }
    edges[BEV] = stop;                     // endpoint
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_edges), edges, sizeof(edges));
    if (e != hipSuccess) return e;
    edges_ready[dev] = true;
  }
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(lidar_bev_kernel, dim3(B), dim3(BEV_THREADS), 0, s, points, offsets, bev);
  return hipGetLastError();
}

}  // namespace rip
