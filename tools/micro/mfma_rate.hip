// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 from ONE wave per SIMD (the plan-search kernel's situation:
// 512 registers per wave) and from two.  Prints shader-clock cycles per MFMA for 1..8 independent accumulator chains.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CHAINS, int BIGREG>
__global__ __launch_bounds__(256) void rate_kernel(float* out, long long* cyc, int iters) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  // optional register ballast so that only one wave fits a SIMD (like the 512-register search kernel)
  float ballast[BIGREG ? 300 : 1];
  for (int i = 0; i < (BIGREG ? 300 : 1); ++i) ballast[i] = a + i;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int i = 0; i < (BIGREG ? 300 : 1); ++i) s += ballast[i] * 1e-30f;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CHAINS, int BIGREG>
void run(int waves_per_simd, float* out, long long* cyc) {
  const int iters = 2000;
  // 256 threads = 4 waves = one per SIMD; grid = 256 CUs * waves_per_simd
  hipLaunchKernelGGL((rate_kernel<CHAINS, BIGREG>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  printf("chains %d  waves/SIMD %d  ballast %d : %.1f cycles per MFMA (per wave)\n", CHAINS, waves_per_simd, BIGREG,
         (double)c / (iters * 8.0 * CHAINS));
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 2 * 256 * sizeof(float));
  hipMalloc(&cyc, sizeof(long long));
  run<1, 0>(1, out, cyc);
  run<2, 0>(1, out, cyc);
  run<3, 0>(1, out, cyc);
  run<4, 0>(1, out, cyc);
  run<8, 0>(1, out, cyc);
  run<4, 0>(2, out, cyc);
  run<4, 1>(1, out, cyc);
  return 0;
}
