// Micro-benchmark 3: one GRU + head step of the plan search (the instruction mix of flow_phase.hip:fwd_step_lds,
// operands from LDS, gate math on the VALU) for NB = 1 or 2 candidate blocks per wave, with 1 or 2 waves per SIMD.
// Question: does ONE wave that interleaves two independent blocks (shared operand reads, static schedule, the whole
// register file) keep the matrix pipe busier than two waves of one block each (what the search kernel does)?
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/micro/dual_block.hip -o /tmp/dual_block && /tmp/dual_block
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }

constexpr int F_ROWS = 63;

template <int NB>
__device__ __forceinline__ void step(const float4* wl, float (&H)[NB][16], float (&yp)[NB][2], int q, float (&o)[NB][4]) {
  float Hn[NB][16];
  float bin[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) bin[b] = q == 0 ? yp[b][0] : (q == 1 ? yp[b][1] : (q == 2 ? 1.f : 0.f));
  const float4 wxr = wl[48 * 64], wxz = wl[49 * 64], wxg = wl[50 * 64], wxh = wl[51 * 64];
  const float wxra[4] = {wxr.x, wxr.y, wxr.z, wxr.w}, wxza[4] = {wxz.x, wxz.y, wxz.z, wxz.w};
  const float wxga[4] = {wxg.x, wxg.y, wxg.z, wxg.w}, wxha[4] = {wxh.x, wxh.y, wxh.z, wxh.w};
  constexpr float L2E = 1.4426950408889634f;
#pragma unroll
  for (int up = 0; up < 4; ++up) {
    f32x4 ar[NB], az[NB], agn[NB], ahn[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) ar[b] = az[b] = agn[b] = ahn[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 wr = wl[((0 * 4 + up) * 4 + j) * 64];
      const float4 wz = wl[((1 * 4 + up) * 4 + j) * 64];
      const float4 wh = wl[((2 * 4 + up) * 4 + j) * 64];
      const float wra[4] = {wr.x, wr.y, wr.z, wr.w}, wza[4] = {wz.x, wz.y, wz.z, wz.w}, wha[4] = {wh.x, wh.y, wh.z, wh.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          ar[b] = mfma(wra[e], H[b][4 * j + e], ar[b]);
          az[b] = mfma(wza[e], H[b][4 * j + e], az[b]);
          ahn[b] = mfma(wha[e], H[b][4 * j + e], ahn[b]);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      ar[b] = mfma(wxra[up], bin[b], ar[b]);
      az[b] = mfma(wxza[up], bin[b], az[b]);
      agn[b] = mfma(wxga[up], bin[b], agn[b]);
      ahn[b] = mfma(wxha[up], bin[b], ahn[b]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float rr = rcpf_(__builtin_amdgcn_exp2f(-L2E * ar[b][r]) + 1.0f);
        const float zz = rcpf_(__builtin_amdgcn_exp2f(-L2E * az[b][r]) + 1.0f);
        const float pre = fmaf(rr, ahn[b][r], agn[b][r]);
        const float nn = 1.0f - 2.0f * rcpf_(__builtin_amdgcn_exp2f(2.0f * L2E * pre) + 1.0f);
        Hn[b][up * 4 + r] = fmaf(zz, H[b][up * 4 + r] - nn, nn);
      }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) H[b][i] = Hn[b][i];
  const float bone = q == 2 ? 1.f : 0.f;
  f32x4 a0[NB], a1[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) a0[b] = a1[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 wa = wl[(52 + j) * 64], wb = wl[(56 + j) * 64];
    const float waa[4] = {wa.x, wa.y, wa.z, wa.w}, wba[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        a0[b] = mfma(waa[e], H[b][4 * j + e], a0[b]);
        a1[b] = mfma(wba[e], H[b][4 * j + e], a1[b]);
      }
  }
  const float4 t60 = wl[60 * 64], t61 = wl[61 * 64], t62 = wl[62 * 64];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    a0[b] = mfma(t60.x, bone, a0[b]);
    a1[b] = mfma(t60.y, bone, a1[b]);
    f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = oa;
    oa = mfma(t60.z, fmaxf(a0[b][0], 0.f), oa);
    ob = mfma(t61.z, fmaxf(a1[b][0], 0.f), ob);
    oa = mfma(t60.w, fmaxf(a0[b][1], 0.f), oa);
    ob = mfma(t61.w, fmaxf(a1[b][1], 0.f), ob);
    oa = mfma(t61.x, fmaxf(a0[b][2], 0.f), oa);
    ob = mfma(t62.x, fmaxf(a1[b][2], 0.f), ob);
    oa = mfma(t61.y, fmaxf(a0[b][3], 0.f), oa);
    ob = mfma(t62.y, fmaxf(a1[b][3], 0.f), ob);
    oa = mfma(t62.z, bone, oa);
#pragma unroll
    for (int r = 0; r < 4; ++r) o[b][r] = oa[r] + ob[r];
  }
}

template <int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void bench_kernel(const float4* w, float* out, long long* cyc, int steps) {
  extern __shared__ float4 fbuf[];
  for (int i = threadIdx.x; i < F_ROWS * 64; i += WAVES * 64) fbuf[i] = w[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, q = lane >> 4;
  const float4* wl = fbuf + lane;
  float H[NB][16], yp[NB][2], o[NB][4];
  for (int b = 0; b < NB; ++b) {
    for (int i = 0; i < 16; ++i) H[b][i] = 0.01f * (i + b + lane % 7);
    yp[b][0] = 0.1f * b;
    yp[b][1] = 0.2f;
  }
  const long long t0 = clock64();
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
    int zero = 0;
    asm volatile("" : "+v"(zero));
    step<NB>(wl + zero, H, yp, q, o);
    for (int b = 0; b < NB; ++b) {  // coupling: the next step's input depends on the head's output
      const float s0 = __logf(1.0f + __expf(o[b][2])) + 1e-3f, s1 = __logf(1.0f + __expf(o[b][3])) + 1e-3f;
      yp[b][0] = (yp[b][0] + o[b][0]) + s0 * 0.3f;
      yp[b][1] = (yp[b][1] + o[b][1]) + s1 * 0.3f;
    }
  }
  const long long t1 = clock64();
  float acc = 0.f;
  for (int b = 0; b < NB; ++b) {
    for (int i = 0; i < 16; ++i) acc += H[b][i];
    acc += yp[b][0] + yp[b][1];
  }
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = acc;
  if (lane == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NB, int WAVES>
void run(const float4* w, float* out, long long* cyc, const char* name) {
  const int steps = 300, blocks = 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(bench_kernel<NB, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      150 * 1024);
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((bench_kernel<NB, WAVES>), dim3(blocks), dim3(WAVES * 64), 150 * 1024, 0, w, out, cyc, steps);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * WAVES);
  hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double mean = 0;
  for (long long v : h) mean += (double)v;
  mean /= h.size();
  const double mfma_per_step = 251.0 * NB;  // per wave
  const double waves_per_simd = WAVES / 4.0;
  const double cyc_per_step = mean / steps;
  printf("%-44s %8.0f cycles/step/wave  -> MFMA pipe busy %.1f %% (%.0f MFMAs x 32 cycles x %.0f wave(s) per SIMD)\n", name,
         cyc_per_step, 100.0 * mfma_per_step * 32.0 * waves_per_simd / cyc_per_step, mfma_per_step, waves_per_simd);
}

int main() {
  float4* w;
  float* out;
  long long* cyc;
  hipMalloc(&w, F_ROWS * 64 * sizeof(float4));
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipMalloc(&cyc, 256 * 8 * sizeof(long long));
  std::vector<float> hw(F_ROWS * 64 * 4);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.02f * (float)((i * 2654435761u >> 16) % 100) - 1.0f;
  hipMemcpy(w, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice);
  run<1, 4>(w, out, cyc, "1 block per wave, 1 wave per SIMD");
  run<1, 8>(w, out, cyc, "1 block per wave, 2 waves per SIMD");
  run<2, 4>(w, out, cyc, "2 blocks per wave, 1 wave per SIMD");
  run<1, 12>(w, out, cyc, "1 block per wave, 3 waves per SIMD");
  run<2, 8>(w, out, cyc, "2 blocks per wave, 2 waves per SIMD");
  return 0;
}
