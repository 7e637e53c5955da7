// Micro-benchmark 2: what a single wave can overlap with v_mfma_f32_16x16x4_f32 (32 cycles each).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_mix.hip -o tools/micro/mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

// MODE 0: 48 distinct VGPR A operands.  1: A operands pinned to AGPRs.  2: + 2 independent v_fma per MFMA.
// 3: + 6 v_fma per MFMA.  4: + 1 v_exp_f32 + 1 v_rcp_f32 per MFMA.  5: + VALU that READS the previous MFMA result.
template <int MODE>
__global__ __launch_bounds__(256) void mix_kernel(float* out, long long* cyc, int iters) {
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[48];
  for (int i = 0; i < 48; ++i) {
    a[i] = threadIdx.x * 1e-3f + i;
    if (MODE == 1) asm volatile("" : "+a"(a[i]));
    else asm volatile("" : "+v"(a[i]));
  }
  float b = 1.0f + threadIdx.x * 1e-4f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.5f + i;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 48; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b, acc[u & 3], 0, 0, 0);
      if (MODE == 2) { v[0] = fmaf(v[0], 1.0001f, 0.5f); v[1] = fmaf(v[1], 1.0001f, 0.5f); }
      if (MODE == 3) {
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = fmaf(v[j], 1.0001f, 0.5f);
      }
      if (MODE == 4) { v[0] = __expf(v[0] * 1e-3f); v[1] = __frcp_rn(v[1] + 2.f); }
      if (MODE == 5) { v[u & 7] += acc[(u + 2) & 3][0]; }
      if (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* what, float* out, long long* cyc, int waves_per_simd = 1) {
  const int iters = 400;
  // 256 threads = one wave per SIMD; 256 CUs x waves_per_simd workgroups
  hipLaunchKernelGGL((mix_kernel<MODE>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  printf("%-58s %d wave(s)/SIMD: %.1f cycles per MFMA per wave = %.1f per MFMA on the pipe\n", what, waves_per_simd,
         (double)c / (iters * 48.0), (double)c / (iters * 48.0) / waves_per_simd);
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 4 * 256 * 256 * sizeof(float));
  hipMalloc(&cyc, sizeof(long long));
  run<0>("48 distinct VGPR A operands", out, cyc);
  run<1>("A operands pinned to AGPRs", out, cyc);
  run<2>("+ 2 independent v_fma per MFMA", out, cyc);
  run<3>("+ 6 independent v_fma per MFMA", out, cyc);
  run<4>("+ v_exp + v_rcp per MFMA", out, cyc);
  run<5>("+ VALU reading the MFMA result issued 2 MFMAs earlier", out, cyc);
  // the same mixes with two / four waves sharing a SIMD: does another wave's MFMA fill the gaps?
  run<0>("48 distinct VGPR A operands", out, cyc, 2);
  run<3>("+ 6 independent v_fma per MFMA", out, cyc, 2);
  run<4>("+ v_exp + v_rcp per MFMA", out, cyc, 2);
  run<3>("+ 6 independent v_fma per MFMA", out, cyc, 4);
  run<4>("+ v_exp + v_rcp per MFMA", out, cyc, 4);
  return 0;
}
