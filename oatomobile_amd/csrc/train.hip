// DIM training step for gfx950: forward in train mode, backward and Adam of
//   oatomobile/baselines/torch/dim/train.py:175-213
//     z = model._params(...)                      dim/model.py:173-219, MobileNetV2 in TRAIN mode: BatchNorm batch
//                                                 statistics (+ running-stat update), Dropout(0.2) before the classifier
//     _, log_prob, logabsdet = decoder._inverse   torch/networks/sequence.py:153-216
//     loss = -mean(log_prob - logabsdet); loss.backward(); Adam(lr).step()
// on the reference's own parameter layout: ONE fp32 vector in state_dict order (arch.py:packed_spec — conv weight,
// BN weight, BN bias, running_mean, running_var per conv; classifier; merger; GRUCell; head), with the gradient and the
// two Adam moments as vectors of the same layout, all owned by the caller (torch tensors: the data-parallel all-reduce
// of the gradients is one collective on one tensor).
//
// fp32 throughout, activations NHWC [B*H*W, C] (what the inference encoder uses), every conv layer keeps its pre-BN
// output and its post-activation output for the backward pass (11.7 MB per image).  Kernels:
//   gemm_f32_kernel     all dense contractions (pointwise convs fwd / dgrad / wgrad, classifier, merger, the flow's
//                       weight gradients): 64x64x16 LDS tiles on v_mfma_f32_16x16x4_f32, split-K with atomics for the
//                       wgrad reductions over B*H*W
//   stem / depthwise    direct 3x3 kernels (fwd, dgrad, wgrad)
//   BatchNorm           per-channel (sum, sum of squares) -> batch mean / biased variance, running-stat update with the
//                       unbiased variance (momentum 0.1), normalise + ReLU6 (+ residual); backward as the two
//                       per-channel reductions (sum g, sum g x^) + one elementwise pass
//   flow_train_kernel   (flow.hip) teacher-forced inverse, its adjoint through all 4 steps into z, and the per-step gate
//                       gradients whose outer products with the saved inputs are the GRU / head weight gradients (GEMMs)
//   adam_kernel         torch.optim.Adam defaults, skipping the running statistics
// Parity with the reference's step: tests/golden/g15.  The kernels are one tuning pass beyond the first correct path
// (48.7 -> 11 ms per 128-observation step); the step is still one launch per layer and operation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "encoder.h"
#include "flow.h"
#include "train.h"

namespace rip {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int FEAT = 128, LAST_C = 1280, VEC = 5, HID = 64;
// a buffer descriptor over [base, base + bytes): loads at a byte offset beyond it return zeros, with no branch
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gemm_srd(const void* base, size_t bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffffull ? 0x7fffffffull : bytes)), 0x00020000);
}
constexpr int GEMM_OOB = 0x7ffffff0;
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;
// x[c] + x[c + 16] + x[c + 32] + x[c + 48] in every one of the four lanes (two swaps on the VALU, no LDS)
__device__ __forceinline__ float q4_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
constexpr float BN_EPS = 1e-5f, BN_MOMENTUM = 0.1f;

// ------------------------------------------------------------------------------------------------------------
// GEMM: C[M,N] (ldc) (+)= op(A)[M,K] op(B)[K,N];  TA: A is stored [K,M]; TB: B is stored [N,K].
// 256 threads = 4 waves as WR x WC, each computing TM x TN tiles of v_mfma_f32_16x16x4_f32 out of a BM x BN block
// (64x64, 128x32 or 256x16: the pointwise convs have 16..96 channels on one side and B*H*W on the other).  K runs in
// chunks of BK through double-buffered LDS tiles [k][m] / [k][n]; the next chunk's global loads (16-byte when the
// operand's contiguous dimension allows: VEC) are in flight while the current one is multiplied, one barrier per chunk.
// gridDim.z > 1: split-K, partial sums added with atomics (C zeroed by the caller); accumulate: C += (no split).
// ------------------------------------------------------------------------------------------------------------
template <bool TA, bool TB, int BM, int BN, int BK, int WR, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                       int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                                       int kchunk, int accumulate) {
  constexpr int WC = 4 / WR, TM = BM / WR / 16, TN = BN / WC / 16;
  constexpr int LDA_S = BM + 4, LDB_S = BN + 4;
  constexpr int A4 = BM * BK / 4 / 256, B4 = (BN * BK / 4 + 255) / 256;  // float4 groups per thread and chunk
  static_assert(BM * BK / 4 % 256 == 0, "A tile");
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDA_S];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB_S];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kb = blockIdx.z * kchunk, ke = min(K, kb + kchunk);
  const int wm = (wave / WC) * (TM * 16), wn = (wave % WC) * (TN * 16);
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // a group = 4 consecutive elements along the stored matrix's contiguous dimension
  const __amdgpu_buffer_rsrc_t asrd = gemm_srd(A, ((size_t)((TA ? K : M) - 1) * lda + (TA ? M : K)) * sizeof(float));
  const __amdgpu_buffer_rsrc_t bsrd = gemm_srd(B, ((size_t)((TB ? N : K) - 1) * ldb + (TB ? K : N)) * sizeof(float));
  float4 ra[A4], rb[B4];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A4; ++i) {
      const int e = tid + i * 256;
      const int m = TA ? (e % (BM / 4)) * 4 : e / (BK / 4), k = TA ? e / (BM / 4) : (e % (BK / 4)) * 4;
      const int gm = m0 + m, gk = k0 + k;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* p = TA ? A + (size_t)gk * lda + gm : A + (size_t)gm * lda + gk;
      if (VEC) {
        // buffer load with an out-of-range offset for groups outside the matrix / the K chunk: zeros, and no branch
        // (a load behind a branch is waited for at the merge: this prefetch was one memory round trip per group)
        const bool ok = gm < M && gk < ke;
        const int off = ok ? (int)(((size_t)(TA ? gk : gm) * lda + (TA ? gm : gk)) * sizeof(float)) : GEMM_OOB;
        const u32x4_t w4 = __builtin_amdgcn_raw_buffer_load_b128(asrd, off, 0, 0);
        v = make_float4(__uint_as_float(w4.x), __uint_as_float(w4.y), __uint_as_float(w4.z), __uint_as_float(w4.w));
      } else {
        const int st = TA ? 1 : 1;  // both walk the contiguous dimension
        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool ok = TA ? (gk < ke && gm + c < M) : (gm < M && gk + c < ke);
          if (ok) t[c] = p[c * st];
        }
        v = make_float4(t[0], t[1], t[2], t[3]);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B4; ++i) {
      const int e = tid + i * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < BN * BK / 4) {
        const int nn = TB ? e / (BK / 4) : (e % (BN / 4)) * 4, k = TB ? (e % (BK / 4)) * 4 : e / (BN / 4);
        const int gn = n0 + nn, gk = k0 + k;
        const float* p = TB ? B + (size_t)gn * ldb + gk : B + (size_t)gk * ldb + gn;
        if (VEC) {
          const bool ok = gn < N && gk < ke;
          const int off = ok ? (int)(((size_t)(TB ? gn : gk) * ldb + (TB ? gk : gn)) * sizeof(float)) : GEMM_OOB;
          const u32x4_t w4 = __builtin_amdgcn_raw_buffer_load_b128(bsrd, off, 0, 0);
          v = make_float4(__uint_as_float(w4.x), __uint_as_float(w4.y), __uint_as_float(w4.z), __uint_as_float(w4.w));
        } else {
          float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const bool ok = TB ? (gn < N && gk + c < ke) : (gk < ke && gn + c < N);
            if (ok) t[c] = p[c];
          }
          v = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
      rb[i] = v;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A4; ++i) {
      const int e = tid + i * 256;
      if (TA) {
        const int m = (e % (BM / 4)) * 4, k = e / (BM / 4);
        *reinterpret_cast<float4*>(&As[buf][k][m]) = ra[i];
      } else {
        const int m = e / (BK / 4), k = (e % (BK / 4)) * 4;
        As[buf][k][m] = ra[i].x;
        As[buf][k + 1][m] = ra[i].y;
        As[buf][k + 2][m] = ra[i].z;
        As[buf][k + 3][m] = ra[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < B4; ++i) {
      const int e = tid + i * 256;
      if (e < BN * BK / 4) {
        if (TB) {
          const int nn = e / (BK / 4), k = (e % (BK / 4)) * 4;
          Bs[buf][k][nn] = rb[i].x;
          Bs[buf][k + 1][nn] = rb[i].y;
          Bs[buf][k + 2][nn] = rb[i].z;
          Bs[buf][k + 3][nn] = rb[i].w;
        } else {
          const int nn = (e % (BN / 4)) * 4, k = e / (BN / 4);
          *reinterpret_cast<float4*>(&Bs[buf][k][nn]) = rb[i];
        }
      }
    }
  };

  if (kb < ke) {
    load_tiles(kb);
    store_tiles(0);
  }
  __syncthreads();
  int buf = 0;
  for (int k0 = kb; k0 < ke; k0 += BK) {
    const bool more = k0 + BK < ke;
    if (more) load_tiles(k0 + BK);  // in flight during this chunk's MFMAs
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int k = kk * 4 + q;
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = As[buf][k][wm + 16 * i + n];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bs[buf][k][wn + 16 * j + n];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tiles(buf ^ 1);  // the other buffer: its last readers passed the previous barrier
    __syncthreads();
    buf ^= 1;
  }
  // result tile: lane (n, q), register r <-> row 4 q + r, column n.  `accumulate` (no split): the old values are added
  // in a first pass of unconditional loads from clamped addresses — inside the per-element branch below each load was
  // waited for on its own (16 memory round trips in sequence per thread)
  if (accumulate && gridDim.z == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gm = m0 + wm + i * 16 + 4 * q + r, gn = n0 + wn + j * 16 + n;
          acc[i][j][r] += C[(size_t)(gm < M ? gm : M - 1) * ldc + (gn < N ? gn : N - 1)];
        }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm + i * 16 + 4 * q + r, gn = n0 + wn + j * 16 + n;
        if (gm < M && gn < N) {
          float* p = C + (size_t)gm * ldc + gn;
          if (gridDim.z > 1)
            atomicAdd(p, acc[i][j][r]);
          else
            *p = acc[i][j][r];
        }
      }
}

template <bool TA, bool TB, int BM, int BN, int BK, int WR>
void gemm_launch(bool vec, dim3 grid, hipStream_t s, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                 int M, int N, int K, int kchunk, int accumulate) {
  if (vec)
    hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, BM, BN, BK, WR, true>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N,
                       K, kchunk, accumulate);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, BM, BN, BK, WR, false>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N,
                       K, kchunk, accumulate);
}

template <bool TA, bool TB>
void gemm_shape(bool vec, int bn_sel, dim3 grid, hipStream_t s, const float* A, int lda, const float* B, int ldb, float* C,
                int ldc, int M, int N, int K, int kchunk, int accumulate) {
  if (bn_sel == 16)
    gemm_launch<TA, TB, 256, 16, 16, 4>(vec, grid, s, A, lda, B, ldb, C, ldc, M, N, K, kchunk, accumulate);
  else if (bn_sel == 32)
    gemm_launch<TA, TB, 128, 32, 32, 4>(vec, grid, s, A, lda, B, ldb, C, ldc, M, N, K, kchunk, accumulate);
  else
    gemm_launch<TA, TB, 64, 64, 32, 2>(vec, grid, s, A, lda, B, ldb, C, ldc, M, N, K, kchunk, accumulate);
}

hipError_t gemm(bool ta, bool tb, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N,
                int K, int accumulate, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return hipSuccess;
  // block shape by the narrow side; wide-and-short outputs (M small, N large) keep the square block
  const int bn = (N <= 16 && M >= 256) ? 16 : ((N <= 32 && M >= 128) ? 32 : 64);
  const int bm = bn == 16 ? 256 : (bn == 32 ? 128 : 64), bk = bn == 16 ? 16 : 32;
  dim3 grid((N + bn - 1) / bn, (M + bm - 1) / bm, 1);
  int kchunk = (K + bk - 1) / bk * bk;
  // reductions over B*H*W with few output tiles: split K so that the chip has work (C must then be pre-zeroed or
  // hold the value to add to: atomics add into it)
  const long tiles = (long)grid.x * grid.y;
  // (weight gradients, `ta`: from K = 1024 — the 4x4 stage's K = 2048 with 45-100 output tiles ran as that many
  // workgroups of 64 serial chunks, 68 us a layer)
  // forward / input gradients of the same stage (M = 2048 rows, 96 output tiles, K = 960: 30 serial chunks, 35-45 us):
  // from K = 512 when fewer than 128 tiles
  if ((K >= (ta ? 1024 : 4096) && tiles < 512) || (!ta && K >= 512 && tiles < 128)) {
    // ~1024 workgroups (4 per CU), chunks of at least 256: with 1024-long chunks the 13x13 / 7x7 weight gradients
    // (K = 21632 / 6272, a handful of output tiles) ran as 60-120 workgroups of 30+ serial iterations each (68 us)
    int splits = (int)std::min<long>((1024 + tiles - 1) / tiles, (K + 255) / 256);
    if (splits > 1) {
      kchunk = ((K + splits - 1) / splits + bk - 1) / bk * bk;
      grid.z = (K + kchunk - 1) / kchunk;
      if (!accumulate) {
        hipError_t e = hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s);
        if (e != hipSuccess) return e;
      }
    }
  }
  // 16-byte loads need the contiguous dimension of both stored operands to be whole groups of four floats
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  // ... and the buffer loads address an operand through a 32-bit byte offset with GEMM_OOB as the "outside" value: an
  // operand of 2 GiB or more (features.2's [B*2500, 96] activations from B ~ 2237) would wrap in-range offsets past the
  // descriptor's clamped size (silent zeros) or put GEMM_OOB inside it — such a call takes the predicated 64-bit
  // pointer path instead (correct at any size, slower)
  const size_t a_bytes = ((size_t)((ta ? K : M) - 1) * lda + (ta ? M : K)) * sizeof(float);
  const size_t b_bytes = ((size_t)((tb ? N : K) - 1) * ldb + (tb ? K : N)) * sizeof(float);
  const bool vec = lda % 4 == 0 && ldb % 4 == 0 && al16(A) && al16(B) && (ta ? M % 4 == 0 : K % 4 == 0) &&
                   (tb ? K % 4 == 0 : N % 4 == 0) && a_bytes < (size_t)GEMM_OOB && b_bytes < (size_t)GEMM_OOB;
  if (ta && tb)
    gemm_shape<true, true>(vec, bn, grid, s, A, lda, B, ldb, C, ldc, M, N, K, kchunk, accumulate);
  else if (ta)
    gemm_shape<true, false>(vec, bn, grid, s, A, lda, B, ldb, C, ldc, M, N, K, kchunk, accumulate);
  else if (tb)
    gemm_shape<false, true>(vec, bn, grid, s, A, lda, B, ldb, C, ldc, M, N, K, kchunk, accumulate);
  else
    gemm_shape<false, false>(vec, bn, grid, s, A, lda, B, ldb, C, ldc, M, N, K, kchunk, accumulate);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// direct convolutions (3x3, pad 1)
// ------------------------------------------------------------------------------------------------------------
// stem: in NCHW [B,C,Hin,Hin] -> out NHWC [B,Ho,Ho,Co]; w [Co][C][3][3] (reference layout)
// Branch-free like the depthwise kernels below: the C*9 window values are loaded first from clamped addresses (the 32
// `oc` lanes of a pixel read the same words), a tap off the image enters the chain as +0; CT = C when it is 2 or 4
// (fully unrolled), 0 = any C.
template <int CT>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                       float* __restrict__ out, int B, int C_, int Hin, int Ho, int Co) {
  const int C = CT > 0 ? CT : C_;
  // the weights [Co][C*9] as ws[tap][oc] (CT > 0, Co <= 64): a lane reads its 18 / 36 words at consecutive addresses
  // across the oc lanes instead of 72-byte-strided global loads
  __shared__ float ws[(CT > 0 ? CT : 1) * 9 * 64];
  if (CT > 0) {
    for (int i = threadIdx.x; i < Co * CT * 9; i += 256) ws[(i % (CT * 9)) * 64 + i / (CT * 9)] = w[i];
    __syncthreads();
  }
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * Ho * Ho * Co;
  if (idx >= total) return;
  const int oc = idx % Co;
  const size_t p = idx / Co;
  const int ox = p % Ho, oy = (p / Ho) % Ho, b = p / ((size_t)Ho * Ho);
  int off[9];
  bool ok[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * oy - 1 + ky;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = 2 * ox - 1 + kx;
      ok[ky * 3 + kx] = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Hin;
      off[ky * 3 + kx] = min(max(iy, 0), Hin - 1) * Hin + min(max(ix, 0), Hin - 1);
    }
  }
  const float* ib = in + (size_t)b * C * Hin * Hin;
  const float* wb = w + (size_t)oc * C * 9;
  float acc = 0.f;
  if (CT > 0) {
    float v[(CT > 0 ? CT : 1) * 9];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) v[c * 9 + t] = ib[(size_t)c * Hin * Hin + off[t]];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc = fmaf(ok[t] ? v[c * 9 + t] : 0.f, ws[(c * 9 + t) * 64 + oc], acc);
  } else {
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc = fmaf(ok[t] ? ib[(size_t)c * Hin * Hin + off[t]] : 0.f, wb[c * 9 + t], acc);
  }
  out[idx] = acc;
}

// dw[oc][c][ky][kx] = sum_{b,oy,ox} dpre[b,oy,ox,oc] in[b,c,2oy-1+ky,2ox-1+kx].  Blocks own chunks of output pixels;
// thread (oc = t % Co, sub = t / Co) walks every (256 / Co)-th pixel of the chunk with its C*9 partial sums in
// registers (the input taps of a pixel are the same for all oc: broadcast loads), then LDS and one global atomic per
// (block, output).  C <= 4 (TAPS registers); dw zeroed by the caller.
template <int CMAX>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ dpre,
                                                         float* __restrict__ dw, int B, int C, int Hin, int Ho, int Co,
                                                         int pix_per_block) {
  __shared__ float sm[64 * CMAX * 9];
  const int taps = C * 9;
  for (int i = threadIdx.x; i < Co * taps; i += 256) sm[i] = 0.f;
  __syncthreads();
  const size_t npix = (size_t)B * Ho * Ho;
  const size_t p0 = (size_t)blockIdx.x * pix_per_block;
  const size_t p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
  const int groups = 256 / Co;
  const int oc = threadIdx.x % Co, sub = threadIdx.x / Co;
  if (sub < groups) {
    float acc[CMAX * 9];
#pragma unroll
    for (int t = 0; t < CMAX * 9; ++t) acc[t] = 0.f;
    for (size_t p = p0 + sub; p < p1; p += groups) {
      const int ox = p % Ho, oy = (p / Ho) % Ho, b = p / ((size_t)Ho * Ho);
      const float g = dpre[p * Co + oc];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        if (c >= C) break;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = 2 * oy - 1 + ky;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox - 1 + kx;
            const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Hin;  // clamped load + select: no branch
            const float v = in[(((size_t)b * C + c) * Hin + min(max(iy, 0), Hin - 1)) * Hin + min(max(ix, 0), Hin - 1)];
            acc[c * 9 + ky * 3 + kx] = fmaf(g, ok ? v : 0.f, acc[c * 9 + ky * 3 + kx]);
          }
        }
      }
    }
    // Co == 32: lanes oc and oc + 32 of a wave hold the same output channel — added with one swap before the LDS
    // atomics (same-address float atomics in LDS serialise)
    const bool pair = Co == 32 && groups == 8;
#pragma unroll
    for (int t = 0; t < CMAX * 9; ++t) {
      float v = acc[t];
      if (pair) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      }
      if (t < taps && (!pair || (threadIdx.x & 32) == 0)) atomicAdd(&sm[oc * taps + t], v);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Co * taps; i += 256) atomicAdd(&dw[i], sm[i]);
}

// depthwise: x NHWC [B,Hi,Hi,C] -> out [B,Ho,Ho,C]; w [C][3][3]
// thread = (pixel, 4 channels): 16-byte activation accesses; per element the FMA chain runs over (ky, kx) ascending
__device__ __forceinline__ float4 dw_taps4(const float* __restrict__ w, int c4, int t) {
  const float* p = w + (size_t)(4 * c4) * 9 + t;
  return make_float4(p[0], p[9], p[18], p[27]);
}
// Branch-free: a branch around a tap off the image makes the compiler wait for every outstanding load at the merge
// (`s_waitcnt vmcnt(0)`: nine memory round trips in sequence), so all nine loads go out first, from clamped addresses,
// and a tap off the image enters the FMA chain as +0 (acc + 0 * w == acc: same sums as skipping it).
__global__ void dw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int B,
                              int C, int Hi, int Ho, int stride) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int C4 = C >> 2;
  const size_t total = (size_t)B * Ho * Ho * C4;
  if (idx >= total) return;
  const int c4 = idx % C4;
  const size_t p = idx / C4;
  const int ox = p % Ho, oy = (p / Ho) % Ho, b = p / ((size_t)Ho * Ho);
  const float4* x4 = reinterpret_cast<const float4*>(x) + (size_t)b * Hi * Hi * C4 + c4;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = stride * oy - 1 + ky;
    const int iyc = min(max(iy, 0), Hi - 1);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = stride * ox - 1 + kx;
      const int ixc = min(max(ix, 0), Hi - 1);
      v[ky * 3 + kx] = x4[((size_t)iyc * Hi + ixc) * C4];
    }
  }
  float4 acc = zero;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = stride * oy - 1 + ky;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = stride * ox - 1 + kx;
      const bool ok = (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Hi;
      const float4 vv = ok ? v[ky * 3 + kx] : zero;
      const float4 wv = dw_taps4(w, c4, ky * 3 + kx);
      acc.x = fmaf(vv.x, wv.x, acc.x);
      acc.y = fmaf(vv.y, wv.y, acc.y);
      acc.z = fmaf(vv.z, wv.z, acc.z);
      acc.w = fmaf(vv.w, wv.w, acc.w);
    }
  }
  reinterpret_cast<float4*>(out)[idx] = acc;
}

// dx[b,iy,ix,c] += sum_{ky,kx} dpre[b,oy,ox,c] w[c,ky,kx] with stride*oy - 1 + ky == iy  (branch-free like the forward)
__global__ void dw_dgrad_kernel(const float* __restrict__ dpre, const float* __restrict__ w, float* __restrict__ dx,
                                int B, int C, int Hi, int Ho, int stride, int accumulate) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int C4 = C >> 2;
  const size_t total = (size_t)B * Hi * Hi * C4;
  if (idx >= total) return;
  const int c4 = idx % C4;
  const size_t p = idx / C4;
  const int ix = p % Hi, iy = (p / Hi) % Hi, b = p / ((size_t)Hi * Hi);
  const float4* g4 = reinterpret_cast<const float4*>(dpre) + (size_t)b * Ho * Ho * C4 + c4;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4* d4 = reinterpret_cast<float4*>(dx) + idx;
  float4 o = zero;
  if (accumulate) o = *d4;
  float4 g[9];
  bool ok[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;
    const int oy = stride == 1 ? ty : ty >> 1;
    const bool oky = ty >= 0 && (stride == 1 || (ty & 1) == 0) && oy < Ho;
    const int oyc = min(max(oy, 0), Ho - 1);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = ix + 1 - kx;
      const int ox = stride == 1 ? tx : tx >> 1;
      ok[ky * 3 + kx] = oky && tx >= 0 && (stride == 1 || (tx & 1) == 0) && ox < Ho;
      const int oxc = min(max(ox, 0), Ho - 1);
      g[ky * 3 + kx] = g4[((size_t)oyc * Ho + oxc) * C4];
    }
  }
  float4 acc = zero;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float4 gv = ok[t] ? g[t] : zero;
    const float4 wv = dw_taps4(w, c4, t);
    acc.x = fmaf(gv.x, wv.x, acc.x);
    acc.y = fmaf(gv.y, wv.y, acc.y);
    acc.z = fmaf(gv.z, wv.z, acc.z);
    acc.w = fmaf(gv.w, wv.w, acc.w);
  }
  o.x += acc.x;
  o.y += acc.y;
  o.z += acc.z;
  o.w += acc.w;
  *d4 = o;
}

// dw[c][tap] = sum_{b,oy,ox} dpre[b,oy,ox,c] x[b, s*oy-1+ky, s*ox-1+kx, c].  A block owns 64 channels (blockIdx.x) of G
// observations x one band of output rows (blockIdx.y); thread = (4-channel group, slot): the 16 slots walk the
// (observation, column) pairs, every pair DOWN the band with a 3x3 register window of 16-byte x values (three new
// loads per row at stride 1, six at stride 2; clamped addresses and selects, no branch around a load) and one 16-byte
// dpre value per row: 36 FMAs per 4-7 loads.  The 36 sums stay in registers across pairs; at the end the four lanes
// of a wave that share a channel group are added with permlane swaps (same-address float atomics in LDS serialise:
// they were 27 of 38 us), one lane in four adds into LDS, and the block ends in 64 * 9 global atomics (dw zeroed by
// the caller).  ~500-1000 blocks for every layer shape: the 4x4 / 7x7 maps take G > 1 observations per block, the
// 50x50 / 25x25 maps several bands per observation.  (History: a first version walked pixel chunks with scalar loads
// and a div / mod per pixel — 289 us per layer, 4.9 ms of a 26 ms step; an (observation, band) x all-channels version
// ended every block in C * 9 atomics after 36 LDS atomics per item.)
template <int STRIDE>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dpre,
                                                       float* __restrict__ dw, int B, int C, int Hi, int Ho, int G, int bands) {
  __shared__ float sm[4][64 * 9];  // one row per wave: added in a fixed order at the end, no LDS atomics
  const int C4 = C >> 2;
  const int c4l = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int c4 = (int)blockIdx.x * 16 + c4l;
  const int grp = (int)blockIdx.y / bands, band = (int)blockIdx.y - grp * bands;
  const int b0 = grp * G, nb = min(G, B - b0);
  const int rows = (Ho + bands - 1) / bands;
  const int oy0 = band * rows, oy1 = min(Ho, oy0 + rows);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = zero;
  if (c4 < C4) {
    for (int pair = slot; pair < nb * Ho; pair += 16) {
      const int bl = pair / Ho, ox = pair - bl * Ho;
      const float4* x4 = reinterpret_cast<const float4*>(x) + (size_t)(b0 + bl) * Hi * Hi * C4;
      const float4* g4 = reinterpret_cast<const float4*>(dpre) + (size_t)(b0 + bl) * Ho * Ho * C4;
      const int ix0 = STRIDE * ox - 1;
      const bool okl = ix0 >= 0, okr = ix0 + 2 < Hi;
      const int xl = okl ? 0 : 1, xr = okr ? 2 : 1;
      auto load_row = [&](int iy, float4(&r)[3]) {
        const bool oky = iy >= 0 && iy < Hi;
        const float4* p = x4 + ((size_t)min(max(iy, 0), Hi - 1) * Hi + ix0) * C4 + c4;
        const float4 a = p[xl * C4], m = p[C4], c = p[xr * C4];
        r[0] = (oky && okl) ? a : zero;
        r[1] = oky ? m : zero;
        r[2] = (oky && okr) ? c : zero;
      };
      float4 win[3][3];
      load_row(STRIDE * oy0 - 1, win[0]);
      if (STRIDE == 1) load_row(oy0, win[1]);
      for (int oy = oy0; oy < oy1; ++oy) {
        if (STRIDE == 1) {
          load_row(oy + 1, win[2]);
        } else {
          load_row(2 * oy, win[1]);
          load_row(2 * oy + 1, win[2]);
        }
        const float4 g = g4[((size_t)oy * Ho + ox) * C4 + c4];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            float4& a = acc[ky * 3 + kx];
            const float4 v = win[ky][kx];
            a.x = fmaf(g.x, v.x, a.x);
            a.y = fmaf(g.y, v.y, a.y);
            a.z = fmaf(g.z, v.z, a.z);
            a.w = fmaf(g.w, v.w, a.w);
          }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          if (STRIDE == 1) {
            win[0][kx] = win[1][kx];
            win[1][kx] = win[2][kx];
          } else {
            win[0][kx] = win[2][kx];
          }
        }
      }
    }
  }
  // the four slots of a wave that share a channel group (lanes c, c + 16, c + 32, c + 48) are added on the VALU first:
  // same-address float atomics in LDS serialise (36 of them per thread were 27 of this kernel's 38 us)
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    acc[t].x = q4_sum(acc[t].x);
    acc[t].y = q4_sum(acc[t].y);
    acc[t].z = q4_sum(acc[t].z);
    acc[t].w = q4_sum(acc[t].w);
  }
  if (slot % 4 == 0) {  // lanes 0..15 of each wave (zeros for a group past the tensor)
    float* row = sm[threadIdx.x >> 6];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      row[(4 * c4l + 0) * 9 + t] = acc[t].x;
      row[(4 * c4l + 1) * 9 + t] = acc[t].y;
      row[(4 * c4l + 2) * 9 + t] = acc[t].z;
      row[(4 * c4l + 3) * 9 + t] = acc[t].w;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 9; i += 256) {
    const int ch = (int)blockIdx.x * 64 + i / 9;
    if (ch < C) atomicAdd(&dw[(size_t)ch * 9 + i % 9], ((sm[0][i] + sm[1][i]) + sm[2][i]) + sm[3][i]);
  }
}

// ------------------------------------------------------------------------------------------------------------
// BatchNorm (train mode), NHWC [M, C]
// ------------------------------------------------------------------------------------------------------------
// per-channel reductions over the rows of an NHWC [M, C] tensor (blocks own row chunks, the channel runs fastest over
// the threads; per-block partial sums, then one float atomic per channel and block):
//   STAT_SHIFTED  out[c] += sum (x - k[c]),  out[C + c] += sum (x - k[c])^2  with the shift k[c] = x[0][c] (`mean` arg):
//                 mean = k + S1/M, var = S2/M - (S1/M)^2 in double.  One pass, and no E[x^2] - m^2 cancellation: the
//                 shift is a sample of the channel, so |S1/M| is of the order of its spread even when the channel's mean
//                 is 100 x that (the residual branches produce such channels)
//   STAT_BWD      out[c] += sum g,  out[C + c] += sum g * xhat,  xhat = (y - mean[c]) * invstd[c]   (x = g, y = pre)
constexpr int STAT_GROUPS = 16;  // 4-channel groups per block of the reduction kernels (64 channels)
// block-level sums of the reduction kernels: sm[0 .. 4 cgl) = S1, sm[4 cgl .. 8 cgl) = S2 of the block's channel chunk
// Block-level sums of the reduction kernels, in a FIXED order: every thread parks its eight sums in LDS, then thread
// i < 8 cgl adds the 256 / cgl row lanes of its (sum, channel) one after the other and stores the total to the block's
// column of the partial table part[2 C][ld] (plain stores: no atomics anywhere — LDS float atomics on one address
// serialise, global ones on a few cache lines were the whole kernel — and, given the same input, the same bits on
// every run; the second stage adds the columns in a fixed order as well).
__device__ __forceinline__ void stat_block_reduce(float* sall, int cgl, int gl, int rl, float4 s1, float4 s2, int chunk,
                                                  int C, float* __restrict__ part, int ld) {
  float4* mine = reinterpret_cast<float4*>(sall + 8 * (rl * cgl + gl));
  const bool in_block = rl * cgl + gl < 256 && rl < 256 / cgl;  // (cgl = 6: the last four threads have no row lane)
  if (in_block) {
    mine[0] = s1;  // zeros for threads outside the tensor
    mine[1] = s2;
  }
  __syncthreads();
  const int w = 4 * cgl, RL = 256 / cgl;
  for (int i = threadIdx.x; i < 2 * w; i += 256) {
    const int ci = i < w ? i : i - w;  // channel inside the chunk
    const float* col = sall + 8 * (ci >> 2) + (i < w ? 0 : 4) + (ci & 3);
    float t = 0.f;
    for (int r = 0; r < RL; ++r) t += col[8 * r * cgl];
    const int c = chunk * w + ci;
    if (c < C) part[(size_t)((i < w ? 0 : C) + c) * ld + blockIdx.x] = t;
  }
}
enum { STAT_SHIFTED = 0, STAT_BWD = 2 };
template <int MODE>
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       float* __restrict__ part, int ld, size_t M, int C,
                                                       int rows_per_block) {
  extern __shared__ float sm[];  // [2*C] (scalar fallback path)
  __shared__ __attribute__((aligned(16))) float sall[256 * 8];
  for (int i = threadIdx.x; i < 2 * C; i += 256) sm[i] = 0.f;
  __syncthreads();
  const size_t r0 = (size_t)blockIdx.x * rows_per_block;
  const size_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  // thread = (4 channels, row lane): 16-byte loads.  A block owns up to 16 such groups (64 channels: blockIdx.y picks
  // the chunk) and `rows_per_block` rows, so that the wide, short layers (7x7 x 960, 4x4 x 1280: a few thousand rows)
  // still make hundreds of blocks with a handful of loads per thread, instead of tens of blocks walking 64 rows each
  if ((C & 3) == 0) {
    const int C4 = C >> 2;
    const int cgl = C4 < STAT_GROUPS ? C4 : STAT_GROUPS;
    const int RL = 256 / cgl;
    const int gl = (int)threadIdx.x % cgl, rl = (int)threadIdx.x / cgl;
    const int c4 = (int)blockIdx.y * cgl + gl;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* y4 = reinterpret_cast<const float4*>(y);
    const bool active = c4 < C4 && rl < RL;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (active) {
      const float4 mu = *reinterpret_cast<const float4*>(mean + 4 * c4);  // STAT_SHIFTED: the shift k[c]
      const float4 is = MODE == STAT_BWD ? *reinterpret_cast<const float4*>(invstd + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      // Rows in batches of four with ALL the batch's loads issued before the first use (`#pragma unroll 4` on the plain
      // loop kept a bounds test, i.e. a branch, between the iterations: every load was waited for on its own — one row
      // in flight per thread); the sums are taken in the same row order as before.
      auto add_row = [&](const float4& v, const float4& yy) {
        if (MODE == STAT_SHIFTED) {
          const float4 d = make_float4(v.x - mu.x, v.y - mu.y, v.z - mu.z, v.w - mu.w);
          s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
          s2.x = fmaf(d.x, d.x, s2.x); s2.y = fmaf(d.y, d.y, s2.y); s2.z = fmaf(d.z, d.z, s2.z); s2.w = fmaf(d.w, d.w, s2.w);
        } else {
          s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
          s2.x = fmaf(v.x, (yy.x - mu.x) * is.x, s2.x);
          s2.y = fmaf(v.y, (yy.y - mu.y) * is.y, s2.y);
          s2.z = fmaf(v.z, (yy.z - mu.z) * is.z, s2.z);
          s2.w = fmaf(v.w, (yy.w - mu.w) * is.w, s2.w);
        }
      };
      size_t r = r0 + rl;
      for (; r + 3 * (size_t)RL < r1; r += 4 * (size_t)RL) {
        float4 v[4], yy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[u] = x4[(r + u * (size_t)RL) * C4 + c4];
          if (MODE == STAT_BWD) yy[u] = y4[(r + u * (size_t)RL) * C4 + c4];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) add_row(v[u], yy[u]);
      }
      for (; r < r1; r += RL) {
        const float4 v = x4[r * C4 + c4];
        float4 yy = v;
        if (MODE == STAT_BWD) yy = y4[r * C4 + c4];
        add_row(v, yy);
      }
    }
    stat_block_reduce(sall, cgl, gl, rl, s1, s2, (int)blockIdx.y, C, part, ld);
    return;
  } else {
    for (int c = threadIdx.x; c < C; c += 256) {  // this thread is the only writer of channel c
      const float mu = mean[c];
      const float is = MODE == STAT_BWD ? invstd[c] : 0.f;
      float s1 = 0.f, s2 = 0.f;
      for (size_t r = r0; r < r1; ++r) {
        const float v = x[r * C + c];
        if (MODE == STAT_SHIFTED) {
          const float d = v - mu;
          s1 += d;
          s2 = fmaf(d, d, s2);
        } else {
          s1 += v;
          s2 = fmaf(v, (y[r * C + c] - mu) * is, s2);
        }
      }
      sm[c] += s1;
      sm[C + c] += s2;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) part[(size_t)i * ld + blockIdx.x] = sm[i];
}

// second stage of the reductions: block = channel c, wave 0 adds row S1 = part[c][0 .. n), wave 1 row S2 = part[C + c][..]
// (float4 loads, double accumulation, fixed order), then
//   STAT_SHIFTED  mean, invstd (saved for the backward pass) and the running statistics (nn.BatchNorm2d train mode:
//                 momentum 0.1, running_var takes the UNBIASED batch variance); `shift` = the first row of `pre`
//   STAT_BWD      sums2[c] = sum g = dbeta, sums2[C + c] = sum g xhat = dgamma (also written to the gradient blob)
template <int MODE>
__global__ __launch_bounds__(128) void stat_reduce_kernel(const float* __restrict__ part, int ld, int n,
                                                          const float* __restrict__ shift, float* __restrict__ o1,
                                                          float* __restrict__ o2, float* __restrict__ run_mean,
                                                          float* __restrict__ run_var, size_t M, int C) {
  __shared__ double both[2];
  const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* row = part + (size_t)(wave * C + c) * ld;
  double acc = 0.0;
  for (int i = 4 * lane; i < n; i += 256) {  // ld is a multiple of 4 and the table's padding is never read as data
    const float4 v = *reinterpret_cast<const float4*>(row + i);
    acc += (double)v.x;
    if (i + 1 < n) acc += (double)v.y;
    if (i + 2 < n) acc += (double)v.z;
    if (i + 3 < n) acc += (double)v.w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) both[wave] = acc;
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (MODE == STAT_SHIFTED) {
    const double d1 = both[0] / (double)M;
    const double m = (double)shift[c] + d1;
    double var = both[1] / (double)M - d1 * d1;  // biased batch variance
    if (var < 0.0) var = 0.0;
    o1[c] = (float)m;
    o2[c] = (float)(1.0 / sqrt(var + (double)BN_EPS));
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    run_mean[c] = (1.f - BN_MOMENTUM) * run_mean[c] + BN_MOMENTUM * (float)m;
    run_var[c] = (1.f - BN_MOMENTUM) * run_var[c] + BN_MOMENTUM * (float)unbiased;
  } else {
    const float s1 = (float)both[0], s2 = (float)both[1];
    o1[c] = s1;
    o1[C + c] = s2;
    o2[c] = s1;        // dbeta
    run_mean[c] = s2;  // dgamma (the third output pointer of this mode)
  }
}

// eval-statistics variant ("frozen" BatchNorm): mean / invstd from the running buffers
__global__ void bn_from_running_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                       float* __restrict__ mean, float* __restrict__ invstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = run_mean[c];
  invstd[c] = 1.0f / sqrtf(run_var[c] + BN_EPS);
}

// post = act(gamma * (pre - mean) * invstd + beta) (+ res).  One element per thread: 3.4 TB/s; a four-channel float4
// form measured 1.7x SLOWER (the per-channel scalars become sixteen more loads per thread)
__global__ void bn_act_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ mean,
                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, const float* __restrict__ res, float* __restrict__ post,
                                  size_t total, int C, int relu6) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = idx % C;
  float v = fmaf((pre[idx] - mean[c]) * invstd[c], gamma[c], beta[c]);
  if (relu6) v = fminf(fmaxf(v, 0.f), 6.f);
  if (res != nullptr) v += res[idx];
  post[idx] = v;
}

// g = dpost masked by the ReLU6 derivative (in place into gbuf), res_grad = dpost for residual layers
__global__ void act_bwd_kernel(const float* __restrict__ dpost, const float* __restrict__ post, float* __restrict__ g,
                               float* __restrict__ dres, size_t total, int relu6) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const float d = dpost[idx];
  if (dres != nullptr) dres[idx] = d;  // the residual branch is the FIRST writer of the block input's gradient
  float v = d;
  if (relu6) {
    const float y = post[idx];
    v = (y > 0.f && y < 6.f) ? d : 0.f;
  }
  g[idx] = v;
}

// act_bwd_kernel + colstats_kernel<STAT_BWD> in one pass (C % 4 == 0): g = dpost masked by the ReLU6 derivative is
// written AND reduced (sum g, sum g xhat) where it is produced; same thread <-> rows mapping and accumulation order as
// colstats_kernel.
template <bool RELU6, bool DRES>  // compile-time: a run-time branch around the `post` load makes every load of the row loop wait
__global__ __launch_bounds__(256) void act_bwd_stats_kernel(const float* __restrict__ dpost, const float* __restrict__ post,
                                                            const float* __restrict__ pre, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* __restrict__ g,
                                                            float* __restrict__ dres, float* __restrict__ part, int ld,
                                                            size_t M, int C, int rows_per_block, int relu6) {
  extern __shared__ float sm[];  // [2*C] (scalar fallback path)
  __shared__ __attribute__((aligned(16))) float sall[256 * 8];
  for (int i = threadIdx.x; i < 2 * C; i += 256) sm[i] = 0.f;
  __syncthreads();
  const size_t r0 = (size_t)blockIdx.x * rows_per_block;
  const size_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  const int C4 = C >> 2;
  const int cgl = C4 < STAT_GROUPS ? C4 : STAT_GROUPS;
  const int RL = 256 / cgl;
  const int gl = (int)threadIdx.x % cgl, rl = (int)threadIdx.x / cgl;
  const int c4 = (int)blockIdx.y * cgl + gl;
  const float4* d4 = reinterpret_cast<const float4*>(dpost);
  const float4* p4 = reinterpret_cast<const float4*>(post);
  const float4* y4 = reinterpret_cast<const float4*>(pre);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* r4 = reinterpret_cast<float4*>(dres);
  const bool active = c4 < C4 && rl < RL;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (active) {
    const float4 mu = *reinterpret_cast<const float4*>(mean + 4 * c4);
    const float4 is = *reinterpret_cast<const float4*>(invstd + 4 * c4);
    // rows in batches of four, all loads of a batch first (see colstats_kernel); same row order.  Native vector types for
    // the batch (arrays of HIP's float4 STRUCT were lowered load by load, each waited for on its own)
    const f32x4* d4v = reinterpret_cast<const f32x4*>(dpost);
    const f32x4* p4v = reinterpret_cast<const f32x4*>(post);
    const f32x4* y4v = reinterpret_cast<const f32x4*>(pre);
    f32x4* g4v = reinterpret_cast<f32x4*>(g);
    f32x4* r4v = reinterpret_cast<f32x4*>(dres);
    auto do_row = [&](size_t e, const f32x4 d, const f32x4 y, const f32x4 yy) {
      if (DRES) r4v[e] = d;  // the residual branch is the FIRST writer of the block input's gradient
      f32x4 v = d;
      if (RELU6) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (y[j] > 0.f && y[j] < 6.f) ? d[j] : 0.f;
      }
      g4v[e] = v;
      s1.x += v[0]; s1.y += v[1]; s1.z += v[2]; s1.w += v[3];
      s2.x = fmaf(v[0], (yy[0] - mu.x) * is.x, s2.x);
      s2.y = fmaf(v[1], (yy[1] - mu.y) * is.y, s2.y);
      s2.z = fmaf(v[2], (yy[2] - mu.z) * is.z, s2.z);
      s2.w = fmaf(v[3], (yy[3] - mu.w) * is.w, s2.w);
    };
    size_t r = r0 + rl;
    for (; r + 3 * (size_t)RL < r1; r += 4 * (size_t)RL) {
      f32x4 d[4], y[4], yy[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t e = (r + u * (size_t)RL) * C4 + c4;
        d[u] = d4v[e];
        y[u] = RELU6 ? p4v[e] : d[u];
        yy[u] = y4v[e];
      }
      __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from sinking the loads to their uses again
#pragma unroll
      for (int u = 0; u < 4; ++u) do_row((r + u * (size_t)RL) * C4 + c4, d[u], y[u], yy[u]);
    }
    for (; r < r1; r += RL) {
      const size_t e = r * C4 + c4;
      const f32x4 d = d4v[e];
      const f32x4 y = RELU6 ? p4v[e] : d;
      const f32x4 yy = y4v[e];
      do_row(e, d, y, yy);
    }
  }
  stat_block_reduce(sall, cgl, gl, rl, s1, s2, (int)blockIdx.y, C, part, ld);
}

// sums2 = (sum g, sum g * xhat) per channel = (dbeta, dgamma) and
// dpre = gamma invstd (g - dbeta/M - xhat dgamma/M)   (batch statistics)   |   gamma invstd g   (running statistics)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ pre,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ sums2,
                                    float* __restrict__ dpre, size_t total, int C, size_t M, int batch_stats) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = idx % C;
  const float is = invstd[c];
  const float xhat = (pre[idx] - mean[c]) * is;
  float v = g[idx];
  if (batch_stats) {
    v = v - sums2[c] / (float)M - xhat * (sums2[C + c] / (float)M);  // sums2 = (sum g, sum g xhat) = (dbeta, dgamma)
  }
  dpre[idx] = gamma[c] * is * v;
}
// ------------------------------------------------------------------------------------------------------------
// tail: average pool, dropout, bias / ReLU, concat
// ------------------------------------------------------------------------------------------------------------
__global__ void pool_drop_fwd_kernel(const float* __restrict__ post, const float* __restrict__ mask,
                                     float* __restrict__ pooled, int B, int P, int C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * C) return;
  const int c = idx % C, b = idx / C;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += post[((size_t)b * P + p) * C + c];
  s /= (float)P;
  pooled[idx] = mask != nullptr ? s * mask[idx] : s;
}
__global__ void pool_drop_bwd_kernel(const float* __restrict__ dpooled, const float* __restrict__ mask,
                                     float* __restrict__ dpost, int B, int P, int C) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * P * C) return;
  const int c = idx % C;
  const int b = idx / ((size_t)P * C);
  const float d = dpooled[(size_t)b * C + c] * (mask != nullptr ? mask[(size_t)b * C + c] : 1.f);
  dpost[idx] = d / (float)P;  // the only writer of the last layer's gradient
}
// x[r, 0..n) = act(x + bias)
__global__ void bias_act_kernel(float* __restrict__ x, int ld, const float* __restrict__ bias, int rows, int n, int relu) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * n) return;
  const int c = idx % n, r = idx / n;
  float v = x[(size_t)r * ld + c] + bias[c];
  if (relu) v = fmaxf(v, 0.f);
  x[(size_t)r * ld + c] = v;
}
__global__ void copy_cols_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, int rows,
                                 int n) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * n) return;
  const int c = idx % n, r = idx / n;
  dst[(size_t)r * ldd + c] = src[(size_t)r * lds_ + c];
}
// out[c] = sum_r x[r, c]: a block owns 32 columns, its 8 groups of 32 threads stride the rows (the first version was
// one thread per column walking all rows: 118 us for the flow's 512 x 192 gate gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ld, float* __restrict__ out, int rows,
                                                     int n) {
  __shared__ float part[8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  float s = 0.f;
  if (c < n) {
#pragma unroll 8
    for (int r = rg; r < rows; r += 8) s += x[(size_t)r * ld + c];  // unrolled: eight loads in flight, same sum order
  }
  part[rg][threadIdx.x & 31] = s;
  __syncthreads();
  if (rg == 0 && c < n) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += part[g][threadIdx.x];
    out[c] = t;
  }
}
// d <- d masked by relu'(y) in place AND out[c] = sum_r d[r][c] (the Linear bias gradient) in one pass
__global__ __launch_bounds__(256) void relu_bwd_colsum_kernel(float* __restrict__ d, const float* __restrict__ y, int ld,
                                                              float* __restrict__ out, int rows, int n) {
  __shared__ float part[8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  float s = 0.f;
  if (c < n) {
#pragma unroll 8
    for (int r = rg; r < rows; r += 8) {
      const size_t e = (size_t)r * ld + c;
      const float dv = d[e];  // (loaded whether or not it is kept: no branch around the load)
      const float v = y[e] > 0.f ? dv : 0.f;
      d[e] = v;
      s += v;
    }
  }
  part[rg][threadIdx.x & 31] = s;
  __syncthreads();
  if (rg == 0 && c < n) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += part[g][threadIdx.x];
    out[c] = t;
  }
}
// up to four colsum_kernel launches over column ranges of ONE matrix as one launch (the flow's four bias gradients)
struct ColsumSegs {
  int col0[4], n[4], blk0[5];  // segment i: columns col0[i] .. col0[i] + n[i], blocks blk0[i] .. blk0[i + 1]
  float* out[4];
};
__global__ __launch_bounds__(256) void colsum_segs_kernel(const float* __restrict__ x, int ld, int rows, ColsumSegs g) {
  __shared__ float part[8][32];
  int seg = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if ((int)blockIdx.x >= g.blk0[i]) seg = i;
  const int c = ((int)blockIdx.x - g.blk0[seg]) * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  const bool on = c < g.n[seg];
  float s = 0.f;
  if (on) {
    const float* xc = x + g.col0[seg] + c;
#pragma unroll 8
    for (int r = rg; r < rows; r += 8) s += xc[(size_t)r * ld];
  }
  part[rg][threadIdx.x & 31] = s;
  __syncthreads();
  if (rg == 0 && on) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += part[k][threadIdx.x];
    g.out[seg][c] = t;
  }
}
__global__ void mean_loss_kernel(const float* __restrict__ q, float* __restrict__ loss, int B) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) s += q[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int sft = 128; sft > 0; sft >>= 1) {
    if (threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = -red[0] / (float)B;  // dim/train.py:201
}

// torch.optim.Adam (defaults: betas (0.9, 0.999), eps 1e-8, amsgrad False), L2 weight decay added to the gradient;
// entries with trainable == 0 (BatchNorm running statistics) are left alone
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, const unsigned char* __restrict__ trainable, size_t n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || (trainable != nullptr && !trainable[i])) return;
  float grad = g[i];
  if (weight_decay != 0.f) grad = fmaf(weight_decay, p[i], grad);
  const float mi = m[i] + (grad - m[i]) * (1.f - beta1);  // exp_avg.lerp_(grad, 1 - beta1)
  const float vi = v[i] * beta2 + (1.f - beta2) * grad * grad;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] - (lr / bc1) * (mi / denom);
}

inline unsigned nblk(size_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// host side: parameter layout + workspace + the step
// ------------------------------------------------------------------------------------------------------------
struct TrainLayer {
  size_t w, gamma, beta, rmean, rvar;  // offsets into the packed vector
  size_t wn;
  size_t act_off;  // offset of this layer's [M, cout] activation in the per-kind activation arenas (floats per image)
  int block_in;    // layer whose post-activation is this layer's residual input, or -1
};

struct Trainer {
  EncoderPlan plan;
  std::vector<TrainLayer> tl;
  size_t cls_w, cls_b, mrg_w[3], mrg_b[3];
  size_t f_wih, f_whh, f_bih, f_bhh, f_w1, f_b1, f_w2, f_b2;
  size_t numel = 0;
  size_t act_per_image = 0;  // floats per image over all conv layers
  int C = 0, max_batch = 0, device = 0;
  // workspaces
  float *pre = nullptr, *post = nullptr, *dpost = nullptr;  // [max_batch * act_per_image]
  float *gbuf = nullptr, *dpre = nullptr;                    // [max_batch * max layer activation]
  float *stats = nullptr;                                    // per layer: mean, invstd [2 * sum cout]; sums scratch
  float *sums = nullptr;     // per layer (sum g, sum g xhat) of the backward pass [stats_floats]
  float *partial = nullptr;  // first-stage table of the per-channel reductions (stat_grid / stat_reduce_kernel)
  float* tail = nullptr;  // pooled, feat/merged, h1, h2, z and their gradients
  float* flowbuf = nullptr;
  size_t max_act = 0, stats_floats = 0, partial_floats = 0;
};

static size_t round4(size_t n) { return n; }

size_t train_numel(int in_channels) {
  const EncoderPlan plan = build_encoder_plan(in_channels);
  size_t pos = 0;
  for (const Layer& l : plan.layers) {
    const size_t per_out = l.kind == L_STEM ? (size_t)l.cin * 9 : (l.kind == L_DW ? 9 : (size_t)l.cin);
    pos += per_out * l.cout + 4 * (size_t)l.cout;
  }
  pos += (size_t)FEAT * LAST_C + FEAT;
  const int sizes[4] = {FEAT + VEC, HID, HID, HID};
  for (int i = 0; i < 3; ++i) pos += (size_t)sizes[i + 1] * sizes[i] + sizes[i + 1];
  pos += 192 * 2 + 192 * 64 + 192 + 192 + 32 * 64 + 32 + 4 * 32 + 4;
  return round4(pos);
}

// launch shape of the per-channel reduction kernels: (row chunks, 64-channel chunks), about 512 blocks in all (measured
// against 256 / 1024 / 2048 / 4096: 9.60 / 9.42 / 9.56 / 10.4 ms of kernel time per step) and at least 64 rows per block; `ld` = the row pitch of the partial table (one column per row chunk, padded to 4)
static dim3 stat_grid(size_t M, int C, int* rows_per_block, int* ld) {
  const int chunks = (C & 3) == 0 ? ((C >> 2) + STAT_GROUPS - 1) / STAT_GROUPS : 1;
  const size_t row_blocks = std::max<size_t>(1, 512 / chunks);
  *rows_per_block = (int)std::max<size_t>(64, (M + row_blocks - 1) / row_blocks);
  const unsigned gx = (unsigned)((M + *rows_per_block - 1) / *rows_per_block);
  *ld = (int)((gx + 3) & ~3u);
  return dim3(gx, (unsigned)chunks);
}

hipError_t trainer_create(Trainer** out, int in_channels, int max_batch, int device) {
  Trainer* t = new Trainer();
  t->plan = build_encoder_plan(in_channels);
  t->C = in_channels;
  t->max_batch = max_batch;
  t->device = device;
  size_t pos = 0, act = 0, chans = 0;
  const int nl = (int)t->plan.layers.size();
  t->tl.resize(nl);
  for (int i = 0; i < nl; ++i) {
    const Layer& l = t->plan.layers[i];
    TrainLayer& q = t->tl[i];
    const size_t per_out = l.kind == L_STEM ? (size_t)l.cin * 9 : (l.kind == L_DW ? 9 : (size_t)l.cin);
    q.wn = per_out * l.cout;
    q.w = pos;
    q.gamma = q.w + q.wn;
    q.beta = q.gamma + l.cout;
    q.rmean = q.beta + l.cout;
    q.rvar = q.rmean + l.cout;
    pos = q.rvar + l.cout;
    q.act_off = act;
    const size_t a = (size_t)l.h_out * l.h_out * l.cout;
    act += a;
    if (a > t->max_act) t->max_act = a;
    chans += l.cout;
    q.block_in = -1;
  }
  for (const FusedBlock& fb : t->plan.blocks) {
    const Layer& lp = t->plan.layers[fb.project];
    if (lp.residual) t->tl[fb.project].block_in = (fb.expand >= 0 ? fb.expand : fb.dw) - 1;
  }
  t->act_per_image = act;
  t->cls_w = pos;
  pos += (size_t)FEAT * LAST_C;
  t->cls_b = pos;
  pos += FEAT;
  const int sizes[4] = {FEAT + VEC, HID, HID, HID};
  for (int i = 0; i < 3; ++i) {
    t->mrg_w[i] = pos;
    pos += (size_t)sizes[i + 1] * sizes[i];
    t->mrg_b[i] = pos;
    pos += sizes[i + 1];
  }
  t->f_wih = pos;
  pos += 192 * 2;
  t->f_whh = pos;
  pos += 192 * 64;
  t->f_bih = pos;
  pos += 192;
  t->f_bhh = pos;
  pos += 192;
  t->f_w1 = pos;
  pos += 32 * 64;
  t->f_b1 = pos;
  pos += 32;
  t->f_w2 = pos;
  pos += 4 * 32;
  t->f_b2 = pos;
  pos += 4;
  t->numel = pos;
  t->stats_floats = 2 * chans;
  hipError_t e = hipSuccess;
  auto alloc = [&](float** p, size_t n) {
    if (e == hipSuccess) e = hipMalloc((void**)p, n * sizeof(float));
  };
  const size_t B = (size_t)max_batch;
  alloc(&t->pre, B * act);
  alloc(&t->post, B * act);
  alloc(&t->dpost, B * act);
  alloc(&t->gbuf, B * t->max_act);
  alloc(&t->dpre, B * t->max_act);
  alloc(&t->stats, t->stats_floats);
  alloc(&t->sums, t->stats_floats);
  for (const Layer& l : t->plan.layers) {
    int rpb, ld;
    (void)stat_grid((size_t)max_batch * l.h_out * l.h_out, l.cout, &rpb, &ld);
    t->partial_floats = std::max(t->partial_floats, 2 * (size_t)l.cout * ld);
  }
  alloc(&t->partial, t->partial_floats);
  alloc(&t->tail, B * (size_t)TRAIN_TAIL_FLOATS);
  alloc(&t->flowbuf, B * (size_t)FLOW_TRAIN_ROW_FLOATS + FW_SIZE);
  if (e != hipSuccess) {
    trainer_destroy(t);
    return e;
  }
  *out = t;
  return hipSuccess;
}

void trainer_destroy(Trainer* t) {
  if (t == nullptr) return;
  float* ptrs[] = {t->pre, t->post, t->dpost, t->gbuf, t->dpre, t->stats, t->sums, t->partial, t->tail, t->flowbuf};
  for (float* p : ptrs)
    if (p != nullptr) (void)hipFree(p);
  delete t;
}

size_t trainer_numel(const Trainer* t) { return t->numel; }
int trainer_max_batch(const Trainer* t) { return t->max_batch; }
int trainer_device(const Trainer* t) { return t->device; }

// marks the trainable entries of the packed vector (everything but the BatchNorm running statistics)
void trainer_trainable_mask(const Trainer* t, unsigned char* mask) {
  for (size_t i = 0; i < t->numel; ++i) mask[i] = 1;
  for (size_t i = 0; i < t->tl.size(); ++i) {
    const int c = t->plan.layers[i].cout;
    for (int j = 0; j < 2 * c; ++j) mask[t->tl[i].rmean + j] = 0;
  }
}

#define TRY(expr)                   \
  do {                              \
    hipError_t e_ = (expr);         \
    if (e_ != hipSuccess) return e_; \
  } while (0)

hipError_t trainer_step(Trainer* t, float* params, float* grads, const float* visual, const float* vec, const float* y,
                        const float* dropout_mask, int B, int batch_stats, float* loss, float* z_out, hipStream_t s) {
  const int nl = (int)t->plan.layers.size();
  const size_t Bz = (size_t)B;
  // activation arenas are laid out layer-major: layer i occupies [B * act_off_i, B * act_off_i + B * a_i)
  auto A = [&](float* base, int i) { return base + Bz * t->tl[i].act_off; };
  const bool forward_only = grads == nullptr;  // evaluate_step: loss and z only, no gradient buffers touched
  if (!forward_only) {
    TRY(hipMemsetAsync(grads, 0, t->numel * sizeof(float), s));
  }
  // ================================== forward ==================================
  size_t st_off = 0;
  std::vector<size_t> stat_off(nl);
  for (int i = 0; i < nl; ++i) {
    const Layer& l = t->plan.layers[i];
    const TrainLayer& q = t->tl[i];
    const size_t M = Bz * l.h_out * l.h_out;
    const size_t total = M * l.cout;
    float* pre = A(t->pre, i);
    float* post = A(t->post, i);
    const float* x = i == 0 ? visual : A(t->post, i - 1);
    if (l.kind == L_STEM) {
      if (l.cin == 2 && l.cout <= 64)
        hipLaunchKernelGGL(stem_fwd_kernel<2>, dim3(nblk(total)), dim3(256), 0, s, x, params + q.w, pre, B, l.cin, l.h_in,
                           l.h_out, l.cout);
      else if (l.cin == 4 && l.cout <= 64)
        hipLaunchKernelGGL(stem_fwd_kernel<4>, dim3(nblk(total)), dim3(256), 0, s, x, params + q.w, pre, B, l.cin, l.h_in,
                           l.h_out, l.cout);
      else
        hipLaunchKernelGGL(stem_fwd_kernel<0>, dim3(nblk(total)), dim3(256), 0, s, x, params + q.w, pre, B, l.cin, l.h_in,
                           l.h_out, l.cout);
    } else if (l.kind == L_DW) {
      hipLaunchKernelGGL(dw_fwd_kernel, dim3(nblk(total / 4)), dim3(256), 0, s, x, params + q.w, pre, B, l.cout, l.h_in, l.h_out,
                         l.stride);
    } else {
      TRY(gemm(false, true, x, l.cin, params + q.w, l.cin, pre, l.cout, (int)M, l.cout, l.cin, 0, s));
    }
    float* mean = t->stats + st_off;
    float* invstd = mean + l.cout;
    stat_off[i] = st_off;
    st_off += 2 * (size_t)l.cout;
    if (batch_stats) {
      int rows_per_block, ld;
      const dim3 sgrid = stat_grid(M, l.cout, &rows_per_block, &ld);
      // shift = the channel's value in the first row of `pre` (read in place: row 0 IS a [C] vector)
      hipLaunchKernelGGL(colstats_kernel<STAT_SHIFTED>, sgrid, dim3(256), 2 * l.cout * sizeof(float), s, pre,
                         (const float*)nullptr, pre, (const float*)nullptr, t->partial, ld, M, l.cout, rows_per_block);
      hipLaunchKernelGGL(stat_reduce_kernel<STAT_SHIFTED>, dim3(l.cout), dim3(128), 0, s, t->partial, ld, (int)sgrid.x, pre,
                         mean, invstd, params + q.rmean, params + q.rvar, M, l.cout);
    } else {
      hipLaunchKernelGGL(bn_from_running_kernel, dim3(nblk(l.cout)), dim3(256), 0, s, params + q.rmean, params + q.rvar, mean,
                         invstd, l.cout);
    }
    const float* res = q.block_in >= 0 ? A(t->post, q.block_in) : nullptr;
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(nblk(total)), dim3(256), 0, s, pre, mean, invstd, params + q.gamma,
                       params + q.beta, res, post, total, l.cout, l.relu6);
  }
  // ---- tail: pool + dropout -> classifier -> cat(vec) -> merger (3 x Linear + ReLU) -> z ----
  const Layer& ll = t->plan.layers[nl - 1];
  const int P = ll.h_out * ll.h_out;
  float* pooled = t->tail;                          // [B,1280]
  float* merged = pooled + Bz * LAST_C;             // [B,133]: feat | vec
  float* h1 = merged + Bz * (FEAT + VEC);           // [B,64]
  float* h2 = h1 + Bz * HID;
  float* zz = h2 + Bz * HID;
  float* dz = zz + Bz * HID;
  float* dh2 = dz + Bz * HID;
  float* dh1 = dh2 + Bz * HID;
  float* dmerged = dh1 + Bz * HID;                  // [B,133]
  float* dpooled = dmerged + Bz * (FEAT + VEC);     // [B,1280]
  float* qrow = dpooled + Bz * LAST_C;              // [B]
  hipLaunchKernelGGL(pool_drop_fwd_kernel, dim3(nblk(Bz * LAST_C)), dim3(256), 0, s, A(t->post, nl - 1), dropout_mask, pooled,
                     B, P, LAST_C);
  TRY(gemm(false, true, pooled, LAST_C, params + t->cls_w, LAST_C, merged, FEAT + VEC, B, FEAT, LAST_C, 0, s));
  hipLaunchKernelGGL(bias_act_kernel, dim3(nblk(Bz * FEAT)), dim3(256), 0, s, merged, FEAT + VEC, params + t->cls_b, B, FEAT, 0);
  hipLaunchKernelGGL(copy_cols_kernel, dim3(nblk(Bz * VEC)), dim3(256), 0, s, vec, VEC, merged + FEAT, FEAT + VEC, B, VEC);
  TRY(gemm(false, true, merged, FEAT + VEC, params + t->mrg_w[0], FEAT + VEC, h1, HID, B, HID, FEAT + VEC, 0, s));
  hipLaunchKernelGGL(bias_act_kernel, dim3(nblk(Bz * HID)), dim3(256), 0, s, h1, HID, params + t->mrg_b[0], B, HID, 1);
  TRY(gemm(false, true, h1, HID, params + t->mrg_w[1], HID, h2, HID, B, HID, HID, 0, s));
  hipLaunchKernelGGL(bias_act_kernel, dim3(nblk(Bz * HID)), dim3(256), 0, s, h2, HID, params + t->mrg_b[1], B, HID, 1);
  TRY(gemm(false, true, h2, HID, params + t->mrg_w[2], HID, zz, HID, B, HID, HID, 0, s));
  hipLaunchKernelGGL(bias_act_kernel, dim3(nblk(Bz * HID)), dim3(256), 0, s, zz, HID, params + t->mrg_b[2], B, HID, 1);
  if (z_out != nullptr) TRY(hipMemcpyAsync(z_out, zz, Bz * HID * sizeof(float), hipMemcpyDeviceToDevice, s));
  // ---- flow: teacher-forced inverse + adjoint (cotangent -1/B per row), per-step records for the weight gradients ----
  TRY(launch_flow_train(params + t->f_wih, params + t->f_whh, params + t->f_bih, params + t->f_bhh, params + t->f_w1,
                        params + t->f_b1, params + t->f_w2, params + t->f_b2, zz, y, B, qrow, dz, t->flowbuf, s));
  hipLaunchKernelGGL(mean_loss_kernel, dim3(1), dim3(256), 0, s, qrow, loss, B);
  if (forward_only) return hipGetLastError();
  {
    const int R = B * 4;
    const float* fb = t->flowbuf;  // [R][FLOW_TRAIN_REC]
    const int ld = FLOW_TRAIN_REC;
    // record columns: dgi 192 | dgh 192 | hprev 64 | u 2 | da1 32 | h 64 | do 4 | relu(a1) 32
    const float *dgi = fb, *dgh = fb + 192, *hprev = fb + 384, *u = fb + 448, *da1 = fb + 450, *hh = fb + 482,
                *dout = fb + 546, *ra1 = fb + 550;
    TRY(gemm(true, false, dgi, ld, u, ld, grads + t->f_wih, 2, 192, 2, R, 0, s));
    TRY(gemm(true, false, dgh, ld, hprev, ld, grads + t->f_whh, 64, 192, 64, R, 0, s));
    TRY(gemm(true, false, da1, ld, hh, ld, grads + t->f_w1, 64, 32, 64, R, 0, s));
    TRY(gemm(true, false, dout, ld, ra1, ld, grads + t->f_w2, 32, 4, 32, R, 0, s));
    ColsumSegs g;
    const float* cols[4] = {dgi, dgh, da1, dout};
    const int ns[4] = {192, 192, 32, 4};
    float* outs[4] = {grads + t->f_bih, grads + t->f_bhh, grads + t->f_b1, grads + t->f_b2};
    g.blk0[0] = 0;
    for (int i = 0; i < 4; ++i) {
      g.col0[i] = (int)(cols[i] - fb);
      g.n[i] = ns[i];
      g.out[i] = outs[i];
      g.blk0[i + 1] = g.blk0[i] + (ns[i] + 31) / 32;
    }
    hipLaunchKernelGGL(colsum_segs_kernel, dim3(g.blk0[4]), dim3(256), 0, s, fb, ld, R, g);
  }
  // ================================== backward ==================================
  // ---- merger / classifier ----
  hipLaunchKernelGGL(relu_bwd_colsum_kernel, dim3((HID + 31) / 32), dim3(256), 0, s, dz, zz, HID, grads + t->mrg_b[2], B, HID);
  TRY(gemm(true, false, dz, HID, h2, HID, grads + t->mrg_w[2], HID, HID, HID, B, 0, s));
  TRY(gemm(false, false, dz, HID, params + t->mrg_w[2], HID, dh2, HID, B, HID, HID, 0, s));
  hipLaunchKernelGGL(relu_bwd_colsum_kernel, dim3((HID + 31) / 32), dim3(256), 0, s, dh2, h2, HID, grads + t->mrg_b[1], B, HID);
  TRY(gemm(true, false, dh2, HID, h1, HID, grads + t->mrg_w[1], HID, HID, HID, B, 0, s));
  TRY(gemm(false, false, dh2, HID, params + t->mrg_w[1], HID, dh1, HID, B, HID, HID, 0, s));
  hipLaunchKernelGGL(relu_bwd_colsum_kernel, dim3((HID + 31) / 32), dim3(256), 0, s, dh1, h1, HID, grads + t->mrg_b[0], B, HID);
  TRY(gemm(true, false, dh1, HID, merged, FEAT + VEC, grads + t->mrg_w[0], FEAT + VEC, HID, FEAT + VEC, B, 0, s));
  TRY(gemm(false, false, dh1, HID, params + t->mrg_w[0], FEAT + VEC, dmerged, FEAT + VEC, B, FEAT + VEC, HID, 0, s));
  TRY(gemm(true, false, dmerged, FEAT + VEC, pooled, LAST_C, grads + t->cls_w, LAST_C, FEAT, LAST_C, B, 0, s));
  hipLaunchKernelGGL(colsum_kernel, dim3((FEAT + 31) / 32), dim3(256), 0, s, dmerged, FEAT + VEC, grads + t->cls_b, B, FEAT);
  TRY(gemm(false, false, dmerged, FEAT + VEC, params + t->cls_w, LAST_C, dpooled, LAST_C, B, LAST_C, FEAT, 0, s));
  hipLaunchKernelGGL(pool_drop_bwd_kernel, dim3(nblk(Bz * P * LAST_C)), dim3(256), 0, s, dpooled, dropout_mask,
                     A(t->dpost, nl - 1), B, P, LAST_C);
  // ---- conv stack, last layer first ----
  std::vector<char> res_writer(nl, 0);
  for (int i = 0; i < nl; ++i)
    if (t->tl[i].block_in >= 0) res_writer[t->tl[i].block_in] = 1;
  for (int i = nl - 1; i >= 0; --i) {
    const Layer& l = t->plan.layers[i];
    const TrainLayer& q = t->tl[i];
    const size_t M = Bz * l.h_out * l.h_out;
    const size_t total = M * l.cout;
    const float* mean = t->stats + stat_off[i];
    const float* invstd = mean + l.cout;
    float* dres = q.block_in >= 0 ? A(t->dpost, q.block_in) : nullptr;
    // dpost[i - 1] has at most two writers and no memset: the residual branch of the block that starts at layer i
    // (its projection layer, handled EARLIER in this loop, stores) and this layer's input gradient (adds to it, or
    // stores when there is no residual branch)
    const int acc_in = i > 0 && res_writer[i - 1] ? 1 : 0;
    float* sums_b = t->sums + stat_off[i];
    int rpb, ld;
    const dim3 sgrid = stat_grid(M, l.cout, &rpb, &ld);
    // NOTE: for residual layers `post` holds bn + res; they carry no ReLU6, so the mask is not needed there
    if ((l.cout & 3) == 0) {
#define ABS_GO(R6_, DR_)                                                                                                  \
  hipLaunchKernelGGL((act_bwd_stats_kernel<R6_, DR_>), sgrid, dim3(256), 2 * l.cout * sizeof(float), s, A(t->dpost, i), \
                     A(t->post, i), A(t->pre, i), mean, invstd, t->gbuf, dres, t->partial, ld, M, l.cout, rpb, l.relu6)
      if (l.relu6 && dres != nullptr) ABS_GO(true, true);
      else if (l.relu6) ABS_GO(true, false);
      else if (dres != nullptr) ABS_GO(false, true);
      else ABS_GO(false, false);
#undef ABS_GO
    } else {
      hipLaunchKernelGGL(act_bwd_kernel, dim3(nblk(total)), dim3(256), 0, s, A(t->dpost, i), A(t->post, i), t->gbuf, dres,
                         total, l.relu6);
      hipLaunchKernelGGL(colstats_kernel<STAT_BWD>, sgrid, dim3(256), 2 * l.cout * sizeof(float), s, t->gbuf,
                         A(t->pre, i), mean, invstd, t->partial, ld, M, l.cout, rpb);
    }
    hipLaunchKernelGGL(stat_reduce_kernel<STAT_BWD>, dim3(l.cout), dim3(128), 0, s, t->partial, ld, (int)sgrid.x,
                       (const float*)nullptr, sums_b, grads + q.beta, grads + q.gamma, (float*)nullptr, M, l.cout);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nblk(total)), dim3(256), 0, s, t->gbuf, A(t->pre, i), mean, invstd,
                       params + q.gamma, sums_b, t->dpre, total, l.cout, M, batch_stats);
    const float* x = i == 0 ? visual : A(t->post, i - 1);
    if (l.kind == L_STEM) {
      const int ppb = 512;
      if (l.cin <= 4)
        hipLaunchKernelGGL(stem_wgrad_kernel<4>, dim3(nblk(M, ppb)), dim3(256), 0, s, x, t->dpre, grads + q.w, B, l.cin, l.h_in,
                           l.h_out, l.cout, ppb);
      else
        hipLaunchKernelGGL(stem_wgrad_kernel<16>, dim3(nblk(M, ppb)), dim3(256), 0, s, x, t->dpre, grads + q.w, B, l.cin, l.h_in,
                           l.h_out, l.cout, ppb);
    } else if (l.kind == L_DW) {
      // (observation, row band) blocks: enough bands for ~2 blocks per CU, at least 4 rows each
      // ~512-1024 blocks: bands of at least 4 rows when one observation per block leaves the chip idle, several
      // observations per block when (observations x 64-channel chunks) alone is already more than that
      const int chunks = (l.cout / 4 + 15) / 16;
      const int G = std::max(1, std::min(8, (int)((long)B * chunks / 512)));
      const int groups = (B + G - 1) / G;
      const int bands = std::max(1, std::min(l.h_out / 4, (int)(1024 / ((long)groups * chunks))));
      const dim3 grid(chunks, groups * bands);
      if (l.stride == 1)
        hipLaunchKernelGGL(dw_wgrad_kernel<1>, grid, dim3(256), 0, s, x, t->dpre, grads + q.w, B, l.cout, l.h_in, l.h_out, G,
                           bands);
      else
        hipLaunchKernelGGL(dw_wgrad_kernel<2>, grid, dim3(256), 0, s, x, t->dpre, grads + q.w, B, l.cout, l.h_in, l.h_out, G,
                           bands);
      const size_t tin = Bz * l.h_in * l.h_in * l.cout;
      hipLaunchKernelGGL(dw_dgrad_kernel, dim3(nblk(tin / 4)), dim3(256), 0, s, t->dpre, params + q.w, A(t->dpost, i - 1), B, l.cout,
                         l.h_in, l.h_out, l.stride, acc_in);
    } else {
      TRY(gemm(true, false, t->dpre, l.cout, x, l.cin, grads + q.w, l.cin, l.cout, l.cin, (int)M, 1, s));  // grads are zero: add
      TRY(gemm(false, false, t->dpre, l.cout, params + q.w, l.cin, A(t->dpost, i - 1), l.cin, (int)M, l.cin, l.cout, acc_in, s));
    }
  }
  return hipGetLastError();
}

int trainer_num_layers(const Trainer* t) { return (int)t->plan.layers.size(); }

// device pointer of conv layer `i`'s saved NHWC activation of the last step (what: 0 pre-BN, 1 post-activation, 2 its
// gradient) and its element count for batch B (rip_train_peek)
float* trainer_debug_layer(Trainer* t, int i, int what, int B, size_t* numel) {
  if (i < 0 || i >= (int)t->plan.layers.size()) return nullptr;
  const Layer& l = t->plan.layers[i];
  *numel = (size_t)B * l.h_out * l.h_out * l.cout;
  float* base = what == 0 ? t->pre : (what == 1 ? t->post : t->dpost);
  return base + (size_t)B * t->tl[i].act_off;
}

hipError_t trainer_adam(float* params, const float* grads, float* m, float* v, const unsigned char* trainable, size_t n,
                        int step, float lr, float beta1, float beta2, float eps, float weight_decay, hipStream_t s) {
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
  const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(nblk(n)), dim3(256), 0, s, params, grads, m, v, trainable, n, lr, beta1, beta2, eps,
                     weight_decay, (float)bc1, (float)std::sqrt(bc2));
  return hipGetLastError();
}

}  // namespace rip
