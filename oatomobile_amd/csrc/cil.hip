// Decoder of the conditional-imitation-learning model for gfx950.
//
// Reference: BehaviouralModel.forward (oatomobile/baselines/torch/cil/model.py:68-127) after the encoder:
//   z = merger(cat(features[128], velocity[3], is_at_traffic_light, traffic_light_state, mode))   (:88-101; MLP
//       134 -> 64 -> 64 -> 64, ReLU after every layer, :50-56)
//   x = 0; for t in range(T): z = GRUCell(x, z); x = x + Linear(z); y[t] = x                       (:106-125)
// One wavefront per sample, lane j = hidden unit j (the wave-per-chain layout of flow.hip): W_hh rows live in
// registers (192 per lane), the hidden state is broadcast through LDS, the 2-wide head is a wave reduction.  The
// whole decode is ~0.5 MFLOP per sample; it is launch latency, not throughput, that matters here, so everything after
// the encoder is this ONE kernel.
//
// Weight blob (fp32, arch.py:cil_decoder_spec order): W0[64][134] b0[64] W1[64][64] b1[64] W2[64][64] b2[64]
// W_ih[192][2] W_hh[192][64] b_ih[192] b_hh[192] W_out[2][64] b_out[2].  torch.nn.GRUCell gate order (r, z, n).
#include <hip/hip_runtime.h>

#include "flow.h"
#include "flow_math.h"

namespace rip {

namespace {

constexpr int NF = 128, NV = 6, NIN = NF + NV, H = 64;
constexpr int OFF_W0 = 0, OFF_B0 = OFF_W0 + H * NIN, OFF_W1 = OFF_B0 + H, OFF_B1 = OFF_W1 + H * H;
constexpr int OFF_W2 = OFF_B1 + H, OFF_B2 = OFF_W2 + H * H, OFF_WIH = OFF_B2 + H, OFF_WHH = OFF_WIH + 3 * H * 2;
constexpr int OFF_BIH = OFF_WHH + 3 * H * H, OFF_BHH = OFF_BIH + 3 * H, OFF_WO = OFF_BHH + 3 * H, OFF_BO = OFF_WO + 2 * H;
constexpr int CIL_BLOB = OFF_BO + 2;  // 30146 floats
constexpr int WAVES = 4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__global__ __launch_bounds__(WAVES * 64) void cil_decode_kernel(const float* __restrict__ feat,
                                                                const float* __restrict__ vec,
                                                                const float* __restrict__ w, int B, int T,
                                                                float* __restrict__ y) {
  __shared__ float act[WAVES][NIN + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * WAVES + wave;
  if (b >= B) return;
  float* a = act[wave];
  // ---- merger input: cat(features, vector inputs)  (cil/model.py:88-98)
  a[lane] = feat[(size_t)b * NF + lane];
  a[64 + lane] = feat[(size_t)b * NF + 64 + lane];
  if (lane < NV) a[NF + lane] = vec[(size_t)b * NV + lane];
  __builtin_amdgcn_wave_barrier();
  // ---- merger: three Linear + ReLU layers, lane j = output unit j
  float h;
  {
    const float* r0 = w + OFF_W0 + lane * NIN;
    float s = w[OFF_B0 + lane];
    for (int i = 0; i < NIN; ++i) s = fmaf(r0[i], a[i], s);
    h = fmaxf(s, 0.f);
  }
  __builtin_amdgcn_wave_barrier();
  a[lane] = h;
  __builtin_amdgcn_wave_barrier();
  {
    const float* r1 = w + OFF_W1 + lane * H;
    float s = w[OFF_B1 + lane];
    for (int i = 0; i < H; ++i) s = fmaf(r1[i], a[i], s);
    h = fmaxf(s, 0.f);
  }
  __builtin_amdgcn_wave_barrier();
  a[lane] = h;
  __builtin_amdgcn_wave_barrier();
  {
    const float* r2 = w + OFF_W2 + lane * H;
    float s = w[OFF_B2 + lane];
    for (int i = 0; i < H; ++i) s = fmaf(r2[i], a[i], s);
    h = fmaxf(s, 0.f);
  }
  // ---- GRU rollout (cil/model.py:106-125); per-lane rows of W_hh / W_ih of gates r, z, n
  float whr[H], whz[H], whn[H];
#pragma unroll
  for (int i = 0; i < H; ++i) {
    whr[i] = w[OFF_WHH + (0 * H + lane) * H + i];
    whz[i] = w[OFF_WHH + (1 * H + lane) * H + i];
    whn[i] = w[OFF_WHH + (2 * H + lane) * H + i];
  }
  const float wir0 = w[OFF_WIH + (0 * H + lane) * 2], wir1 = w[OFF_WIH + (0 * H + lane) * 2 + 1];
  const float wiz0 = w[OFF_WIH + (1 * H + lane) * 2], wiz1 = w[OFF_WIH + (1 * H + lane) * 2 + 1];
  const float win0 = w[OFF_WIH + (2 * H + lane) * 2], win1 = w[OFF_WIH + (2 * H + lane) * 2 + 1];
  const float bir = w[OFF_BIH + lane], biz = w[OFF_BIH + H + lane], bin = w[OFF_BIH + 2 * H + lane];
  const float bhr = w[OFF_BHH + lane], bhz = w[OFF_BHH + H + lane], bhn = w[OFF_BHH + 2 * H + lane];
  const float wo0 = w[OFF_WO + lane], wo1 = w[OFF_WO + H + lane];
  const float bo0 = w[OFF_BO], bo1 = w[OFF_BO + 1];
  float x0 = 0.f, x1 = 0.f;
  for (int t = 0; t < T; ++t) {
    __builtin_amdgcn_wave_barrier();
    a[lane] = h;
    __builtin_amdgcn_wave_barrier();
    float gr = bhr, gz = bhz, gn = bhn;
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const float hi = a[i];
      gr = fmaf(whr[i], hi, gr);
      gz = fmaf(whz[i], hi, gz);
      gn = fmaf(whn[i], hi, gn);
    }
    const float ir = fmaf(wir1, x1, fmaf(wir0, x0, bir));
    const float iz = fmaf(wiz1, x1, fmaf(wiz0, x0, biz));
    const float in = fmaf(win1, x1, fmaf(win0, x0, bin));
    const float r = sigmoidf_(ir + gr);
    const float z = sigmoidf_(iz + gz);
    const float n = tanhf_(fmaf(r, gn, in));
    h = fmaf(z, h - n, n);  // (1 - z) * n + z * h
    // dx = Linear(h) (cil/model.py:118), x = dx + x (:119)
    const float d0 = wave_sum(wo0 * h) + bo0;
    const float d1 = wave_sum(wo1 * h) + bo1;
    x0 += d0;
    x1 += d1;
    if (lane == 0) {
      y[((size_t)b * T + t) * 2] = x0;
      y[((size_t)b * T + t) * 2 + 1] = x1;
    }
  }
}

}  // namespace

int cil_blob_floats() { return CIL_BLOB; }

hipError_t launch_cil_decode(const float* feat, const float* vec, const float* w, int B, int T, float* y, hipStream_t s) {
  if (B <= 0 || T <= 0) return hipSuccess;
  hipLaunchKernelGGL(cil_decode_kernel, dim3((B + WAVES - 1) / WAVES), dim3(WAVES * 64), 0, s, feat, vec, w, B, T, y);
  return hipGetLastError();
}

}  // namespace rip
