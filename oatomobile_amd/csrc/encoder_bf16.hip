// bf16 MobileNetV2 encoder for gfx950 (BASELINE configs[2]: "bf16 encoder + fp32 flow").
//
// Activations are bf16 NHWC in HBM (half the bytes of the fp32 path), pointwise weights bf16, everything else
// (accumulation, bias, ReLU6, residual add, depthwise taps) fp32 in registers.  Pointwise convs run on
// v_mfma_f32_16x16x32_bf16: one 16-byte load per lane = 8 K-values = one MFMA k-group, so a K chunk of 32 costs one
// load per operand tile.  Same transposed orientation as the fp32 GEMM (encoder.hip): A = 16 output channels,
// B = 16 pixels, lane (n, q) ends with 4 consecutive channels of pixel n.  Both operands use the same
// "lane q <-> K values 8q..8q+7 of the chunk" assignment, so the contraction is independent of the instruction's
// internal k ordering.  features.18 is written in fp32 for the (fp32) pooling/classifier/merger tail.
#include <stdlib.h>

#include "encoder.h"
#include "flow.h"  // device_cu_count

namespace rip {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
typedef unsigned short bf16_t;

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.f), 6.f); }
__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ unsigned f2bf(float f) {  // round to nearest even
  const unsigned u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// two fp32 -> one packed bf16 pair, round to nearest even: a single v_cvt_pk_bf16_f32 (the integer sequence above is
// ~10 VALU instructions per pair, and the GEMM epilogues convert 32-64 values per lane and tile)
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
  union {
    b2_t h;
    unsigned u;
  } c;
  c.h = __builtin_convertvector(f2_t{lo, hi}, b2_t);
  return c.u;
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum16(float x) {  // sum over each row of 16 lanes, replicated in the row
  x += dpp_f<0xB1>(x);
  x += dpp_f<0x4E>(x);
  x += dpp_f<0x141>(x);
  x += dpp_f<0x140>(x);
  return x;
}

// ---- stem: fp32 NCHW input -> bf16 NHWC [K][B][Ho][Ho][32].  One block per (observation, band of output rows, model):
// the band's input rows are staged once in LDS (coalesced fp32 reads), the model's 576-float tap table too; a thread
// owns (pixel, 8 output channels) and writes 16 bytes.  Replaces per-tap scalar global loads.
constexpr int STEM_ROWS = 5;  // output rows per block -> 11 input rows of 100 floats per channel
template <int CMAX>
__global__ __launch_bounds__(256) void stem_bf16_kernel(const float* __restrict__ in, const float* __restrict__ wbase,
                                                         size_t model_stride, int k0, size_t w_off, size_t b_off,
                                                         int B, int C, int Hin, int Ho, bf16_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int IR = 2 * STEM_ROWS + 1;     // input rows of the band (incl. halo)
  float* xs = smem;                     // [C][IR][Hin + 2] with zero borders
  float* ws = xs + C * IR * (Hin + 2);  // [9][C][32]
  const int k = blockIdx.z, b = blockIdx.y, band = blockIdx.x;
  const int oy0 = band * STEM_ROWS;
  const int iy0 = oy0 * 2 - 1;
  const int tid = threadIdx.x;
  const float* w = wbase + (size_t)(k0 + k) * model_stride + w_off;
  const float* bias = wbase + (size_t)(k0 + k) * model_stride + b_off;
  const int IW = Hin + 2;
  for (int e = tid; e < C * IR * IW; e += 256) {
    const int ix = e % IW - 1, r = (e / IW) % IR, c = e / (IW * IR);
    const int iy = iy0 + r;
    xs[e] = (iy >= 0 && iy < Hin && ix >= 0 && ix < Hin) ? in[((size_t)b * C + c) * Hin * Hin + (size_t)iy * Hin + ix] : 0.f;
  }
  for (int e = tid; e < 9 * C * 32; e += 256) ws[e] = w[e];
  lds_barrier();
  const int rows = min(STEM_ROWS, Ho - oy0);
  for (int e = tid; e < rows * Ho * 4; e += 256) {
    const int oc8 = e & 3, pix = e >> 2;
    const int ox = pix % Ho, oyl = pix / Ho;
    float acc[8];
    {
      const float4 b0 = *reinterpret_cast<const float4*>(bias + oc8 * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(bias + oc8 * 8 + 4);
      acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w;
      acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
    }
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float* xr = xs + (c * IR + 2 * oyl + ky) * IW + 2 * ox;  // input col 2*ox - 1 + kx  (+1 border)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float v = xr[kx];
          const float* wp = ws + ((ky * 3 + kx) * C + c) * 32 + oc8 * 8;
          const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
          acc[0] = fmaf(v, w0.x, acc[0]);
          acc[1] = fmaf(v, w0.y, acc[1]);
          acc[2] = fmaf(v, w0.z, acc[2]);
          acc[3] = fmaf(v, w0.w, acc[3]);
          acc[4] = fmaf(v, w1.x, acc[4]);
          acc[5] = fmaf(v, w1.y, acc[5]);
          acc[6] = fmaf(v, w1.z, acc[6]);
          acc[7] = fmaf(v, w1.w, acc[7]);
        }
      }
    }
    uint4 o;
    o.x = pack2(relu6f(acc[0]), relu6f(acc[1]));
    o.y = pack2(relu6f(acc[2]), relu6f(acc[3]));
    o.z = pack2(relu6f(acc[4]), relu6f(acc[5]));
    o.w = pack2(relu6f(acc[6]), relu6f(acc[7]));
    *reinterpret_cast<uint4*>(out + (((size_t)k * B + b) * Ho * Ho + (size_t)(oy0 + oyl) * Ho + ox) * 32 + oc8 * 8) = o;
  }
}

// ---- depthwise 3x3: thread = (run of R output pixels along x, 8 channels).  Sliding window: every input column of
// the 3-row band is loaded once (16 bytes) and feeds all outputs of the run that tap it; the 9x8 fp32 tap weights are
// loaded once per thread.  ~9 load instructions per 16-byte output instead of 27 (the kernel is TA-issue bound).
template <int STRIDE, int R>
__global__ __launch_bounds__(256) void dw_bf16_kernel(const bf16_t* __restrict__ in, const float* __restrict__ wbase,
                                                       size_t model_stride, int k0, size_t w_off, size_t b_off, int B,
                                                       int C, int Hin, int Ho, bf16_t* __restrict__ out) {
  const int k = blockIdx.z;
  const float* w = wbase + (size_t)(k0 + k) * model_stride + w_off;
  const float* bias = wbase + (size_t)(k0 + k) * model_stride + b_off;
  const int C8 = C >> 3;
  const int runs = (Ho + R - 1) / R;
  const long total = (long)B * Ho * runs * C8;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c8 = (int)(idx % C8);
  long rest = idx / C8;
  const int run = (int)(rest % runs);
  rest /= runs;
  const int oy = (int)(rest % Ho), b = (int)(rest / Ho);
  const int ox0 = run * R;
  const bf16_t* ip = in + ((size_t)k * B + b) * Hin * Hin * C + c8 * 8;
  float wt[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float4 w0 = *reinterpret_cast<const float4*>(w + t * C + c8 * 8);
    const float4 w1 = *reinterpret_cast<const float4*>(w + t * C + c8 * 8 + 4);
    wt[t][0] = w0.x; wt[t][1] = w0.y; wt[t][2] = w0.z; wt[t][3] = w0.w;
    wt[t][4] = w1.x; wt[t][5] = w1.y; wt[t][6] = w1.z; wt[t][7] = w1.w;
  }
  float acc[R][8];
  {
    const float4 b0 = *reinterpret_cast<const float4*>(bias + c8 * 8);
    const float4 b1 = *reinterpret_cast<const float4*>(bias + c8 * 8 + 4);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      acc[r][0] = b0.x; acc[r][1] = b0.y; acc[r][2] = b0.z; acc[r][3] = b0.w;
      acc[r][4] = b1.x; acc[r][5] = b1.y; acc[r][6] = b1.z; acc[r][7] = b1.w;
    }
  }
  constexpr int COLS = (R - 1) * STRIDE + 3;  // input columns the run touches
  const int ix0 = ox0 * STRIDE - 1;
  // all 3 x COLS loads first, from clamped addresses (a tap off the image is multiplied as zero: acc + 0 * w == acc, the
  // sums of the skipping form): a branch per tap made every load wait at the merge — nine memory round trips in sequence
  uint4 v[3][COLS];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iyc = min(max(oy * STRIDE - 1 + ky, 0), Hin - 1);
    const bf16_t* rowp = ip + (size_t)iyc * Hin * C;
#pragma unroll
    for (int j = 0; j < COLS; ++j) v[ky][j] = *reinterpret_cast<const uint4*>(rowp + (size_t)min(max(ix0 + j, 0), Hin - 1) * C);
  }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * STRIDE - 1 + ky;
    const bool yok = iy >= 0 && iy < Hin;
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
      const int ix = ix0 + j;
      const bool ok = yok && ix >= 0 && ix < Hin;
      const unsigned vx = ok ? v[ky][j].x : 0u, vy = ok ? v[ky][j].y : 0u, vz = ok ? v[ky][j].z : 0u, vw = ok ? v[ky][j].w : 0u;
      const float f[8] = {bf2f(vx & 0xffffu), bf2f(vx >> 16), bf2f(vy & 0xffffu), bf2f(vy >> 16),
                          bf2f(vz & 0xffffu), bf2f(vz >> 16), bf2f(vw & 0xffffu), bf2f(vw >> 16)};
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int kx = j - r * STRIDE;  // compile-time after unrolling
        if (kx >= 0 && kx < 3) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[r][e] = fmaf(f[e], wt[ky * 3 + kx][e], acc[r][e]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ox = ox0 + r;
    if (ox < Ho) {
      uint4 o;
      o.x = pack2(relu6f(acc[r][0]), relu6f(acc[r][1]));
      o.y = pack2(relu6f(acc[r][2]), relu6f(acc[r][3]));
      o.z = pack2(relu6f(acc[r][4]), relu6f(acc[r][5]));
      o.w = pack2(relu6f(acc[r][6]), relu6f(acc[r][7]));
      *reinterpret_cast<uint4*>(out + (((size_t)k * B + b) * Ho * Ho + (size_t)oy * Ho + ox) * C + c8 * 8) = o;
    }
  }
}

// ---- depthwise 3x3, row-streaming variant (large launches).  A wave owns (model, observation, band of output rows,
// 64-lane group of (run of R outputs, 8 channels)) and walks down the band; everything row-dependent is wave-uniform,
// so each input row gets a buffer descriptor built on the scalar unit (num_records = 0 for rows outside the image)
// and every tap outside the row lands outside the descriptor: the hardware returns zeros for the padding and the
// loop body has no address arithmetic, no predication and no weight loads.  Loads of the next row(s) are issued
// before the current row is computed.  fp32 math runs on v_pk_fma_f32; bf16 packing on v_cvt_pk_bf16_f32.
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 u) {
  union {
    u32x4 u;
    bf16x8 v;
  } c;
  c.u = u;
  return c.v;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_srd(const bf16_t* row, int bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(row);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ f32x2 bfpair(unsigned u) {
  f32x2 r;
  r.x = __uint_as_float(u << 16);
  r.y = __uint_as_float(u & 0xffff0000u);
  return r;
}
__device__ __forceinline__ unsigned pack_relu6(f32x2 v) {
  v = __builtin_elementwise_min(__builtin_elementwise_max(v, f32x2{0.f, 0.f}), f32x2{6.f, 6.f});
  union {
    bf16x2 h;
    unsigned u;
  } c;
  c.h = __builtin_convertvector(v, bf16x2);
  return c.u;
}

constexpr int DW_OOB = 0x40000000;  // byte offset beyond any row descriptor

template <int STRIDE, int R>
__global__ __launch_bounds__(256, 2) void dw_rows_bf16_kernel(const bf16_t* __restrict__ in,
                                                               const float* __restrict__ wbase, size_t model_stride,
                                                               int k0, size_t w_off, size_t b_off, int B, int C, int Hin,
                                                               int Ho, int band_rows, int bands, int lane_groups,
                                                               bf16_t* __restrict__ out) {
  constexpr int COLS = (R - 1) * STRIDE + 3;
  constexpr int NEW = STRIDE;  // new input rows per output row
  const int k = blockIdx.z;
  const int lane = threadIdx.x & 63;
  int wi = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (wi >= B * bands * lane_groups) return;
  const int lg = wi % lane_groups;
  wi /= lane_groups;
  const int band = wi % bands, b = wi / bands;
  const int C8 = C >> 3, runs = (Ho + R - 1) / R;
  const int li = lg * 64 + lane;
  const bool active = li < runs * C8;
  const int run = active ? li / C8 : 0, c8 = active ? li - run * C8 : 0;
  const float* w = wbase + (size_t)(k0 + k) * model_stride + w_off + c8 * 8;
  const float* bias = wbase + (size_t)(k0 + k) * model_stride + b_off + c8 * 8;
  f32x2 wt[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float4 w0 = *reinterpret_cast<const float4*>(w + t * C);
    const float4 w1 = *reinterpret_cast<const float4*>(w + t * C + 4);
    wt[t][0] = f32x2{w0.x, w0.y};
    wt[t][1] = f32x2{w0.z, w0.w};
    wt[t][2] = f32x2{w1.x, w1.y};
    wt[t][3] = f32x2{w1.z, w1.w};
  }
  f32x2 bb[4];
  {
    const float4 b0 = *reinterpret_cast<const float4*>(bias);
    const float4 b1 = *reinterpret_cast<const float4*>(bias + 4);
    bb[0] = f32x2{b0.x, b0.y};
    bb[1] = f32x2{b0.z, b0.w};
    bb[2] = f32x2{b1.x, b1.y};
    bb[3] = f32x2{b1.z, b1.w};
  }
  int voff[COLS], ooff[R];
  const int ix0 = run * R * STRIDE - 1;
#pragma unroll
  for (int j = 0; j < COLS; ++j) {
    const int ix = ix0 + j;
    voff[j] = (active && ix >= 0 && ix < Hin) ? (ix * C + c8 * 8) * 2 : DW_OOB;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ox = run * R + r;
    ooff[r] = (active && ox < Ho) ? (ox * C + c8 * 8) * 2 : DW_OOB;
  }
  const int in_row_bytes = Hin * C * 2, out_row_bytes = Ho * C * 2;
  const bf16_t* img = in + ((size_t)k * B + b) * Hin * Hin * C;
  bf16_t* oimg = out + ((size_t)k * B + b) * Ho * Ho * C;
  auto load_row = [&](int iy, u32x4(&row)[COLS]) {
    const bool ok = iy >= 0 && iy < Hin;
    const __amdgpu_buffer_rsrc_t srd = row_srd(img + (size_t)(ok ? iy : 0) * Hin * C, ok ? in_row_bytes : 0);
#pragma unroll
    for (int j = 0; j < COLS; ++j) row[j] = __builtin_amdgcn_raw_buffer_load_b128(srd, voff[j], 0, 0);
  };
  const int oy0 = band * band_rows, oy1 = min(Ho, oy0 + band_rows);
  u32x4 win[3][COLS], nxt[NEW][COLS];
  load_row(oy0 * STRIDE - 1, win[0]);
  load_row(oy0 * STRIDE, win[1]);
  load_row(oy0 * STRIDE + 1, win[2]);
#pragma unroll 1
  for (int oy = oy0; oy < oy1; ++oy) {
    if (oy + 1 < oy1) {  // rows the next output row adds, in flight under this row's math
#pragma unroll
      for (int i = 0; i < NEW; ++i) load_row((oy + 1) * STRIDE + 2 - NEW + i, nxt[i]);
    }
    f32x2 acc[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[r][e] = bb[e];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int j = 0; j < COLS; ++j) {
        const u32x4 v = win[ky][j];
        const f32x2 f[4] = {bfpair(v.x), bfpair(v.y), bfpair(v.z), bfpair(v.w)};
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int kx = j - r * STRIDE;
          if (kx >= 0 && kx < 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][e] = __builtin_elementwise_fma(f[e], wt[ky * 3 + kx][e], acc[r][e]);
          }
        }
      }
    }
    const __amdgpu_buffer_rsrc_t osrd = row_srd(oimg + (size_t)oy * Ho * C, out_row_bytes);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      u32x4 o;
      o.x = pack_relu6(acc[r][0]);
      o.y = pack_relu6(acc[r][1]);
      o.z = pack_relu6(acc[r][2]);
      o.w = pack_relu6(acc[r][3]);
      __builtin_amdgcn_raw_buffer_store_b128(o, osrd, ooff[r], 0, 0);
    }
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
      if (STRIDE == 1) {
        win[0][j] = win[1][j];
        win[1][j] = win[2][j];
        win[2][j] = nxt[0][j];
      } else {
        win[0][j] = win[2][j];
        win[1][j] = nxt[0][j];
        win[2][j] = nxt[1][j];
      }
    }
  }
}

// ---- pointwise GEMM on v_mfma_f32_16x16x32_bf16 (see encoder.hip pw_kernel for the tiling / KSPLIT scheme) ----
__device__ __forceinline__ bf16x8 as_bf16x8(uint4 u) {
  union {
    uint4 u;
    bf16x8 v;
  } c;
  c.u = u;
  return c.v;
}

template <int CT, int PT, int UNROLL, int KSPLIT, bool OUT_F32>
__global__ __launch_bounds__(256) void pw_bf16_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ whbase,
                                                       const float* __restrict__ wbase, size_t model_stride, int k0,
                                                       size_t w_off, size_t b_off, const bf16_t* __restrict__ res,
                                                       void* __restrict__ out, int M, int Cin, int Cout, int flags,
                                                       size_t act_model_stride_in, size_t act_model_stride_out) {
  const int relu6 = flags & 1;  // bit 1: 4x4 average-pool epilogue (fp32 outputs only), see encoder.hip pw_kernel
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int k = blockIdx.z;
  const int ptile0 = (KSPLIT == 1 ? blockIdx.x * 4 + wave : blockIdx.x) * PT;
  const int ctile0 = blockIdx.y * CT;
  if (KSPLIT == 1 && ptile0 * 16 >= M) return;
  const bf16_t* A = whbase + (size_t)(k0 + k) * model_stride + w_off;
  const float* bias = wbase + (size_t)(k0 + k) * model_stride + b_off;
  const bf16_t* X = in + (size_t)k * act_model_stride_in;
  const bf16_t* R = res != nullptr ? res + (size_t)k * act_model_stride_out : nullptr;

  const bf16_t* arow[CT];
  bool aval[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int co = (ctile0 + ct) * 16 + n;
    aval[ct] = co < Cout;
    arow[ct] = A + (size_t)min(co, Cout - 1) * Cin + 8 * q;
  }
  const bf16_t* brow[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int p = (ptile0 + pt) * 16 + n;
    brow[pt] = X + (size_t)min(p, M - 1) * Cin + 8 * q;
  }
  f32x4 acc[CT][PT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kchunks = (Cin + 31) / 32;
  const int kper = (kchunks + KSPLIT - 1) / KSPLIT;
  const int kbeg = KSPLIT == 1 ? 0 : 32 * kper * wave;
  const int kend = KSPLIT == 1 ? Cin : min(Cin, 32 * kper * (wave + 1));
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll 1
  for (int kc0 = kbeg; kc0 < kend; kc0 += 32 * UNROLL) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int kc = kc0 + 32 * u;
      const bool kval = kc + 8 * q < kend;  // Cin is a multiple of 8: a lane's 8 values are all-valid or all-pad
      // unconditional loads (a lane past the K range reads the first group of its row), component-wise selects: a load
      // behind a branch is waited for on its own at the merge, and a ternary on the uint4 STRUCT goes through scratch
      const int koff = kval ? kc : -8 * q;
      uint4 av[CT], bv[PT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) av[ct] = *reinterpret_cast<const uint4*>(arow[ct] + koff);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) bv[pt] = *reinterpret_cast<const uint4*>(brow[pt] + koff);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const bool on = kval && aval[ct];
        av[ct] = make_uint4(on ? av[ct].x : 0u, on ? av[ct].y : 0u, on ? av[ct].z : 0u, on ? av[ct].w : 0u);
      }
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        bv[pt] = make_uint4(kval ? bv[pt].x : 0u, kval ? bv[pt].y : 0u, kval ? bv[pt].z : 0u, kval ? bv[pt].w : 0u);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
          acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(av[ct]), as_bf16x8(bv[pt]), acc[ct][pt], 0, 0, 0);
    }
  }
  if (KSPLIT > 1) {
    __shared__ float4 part[KSPLIT][CT * PT][64];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        part[wave][ct * PT + pt][lane] = make_float4(acc[ct][pt][0], acc[ct][pt][1], acc[ct][pt][2], acc[ct][pt][3]);
    lds_barrier();
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int t = ct * PT + pt;
        if ((t & (KSPLIT - 1)) == wave) {
          float4 sum = part[0][t][lane];
#pragma unroll
          for (int w2 = 1; w2 < KSPLIT; ++w2) {
            const float4 o = part[w2][t][lane];
            sum.x += o.x;
            sum.y += o.y;
            sum.z += o.z;
            sum.w += o.w;
          }
          acc[ct][pt] = f32x4{sum.x, sum.y, sum.z, sum.w};
        }
      }
  }
  // epilogue operands (bias, residual): all requested here, from clamped addresses and with no per-lane branch around
  // them (inside the store loops each load was waited for on its own: two memory round trips per tile and launch)
  float4 eb[CT];
  uint2 er[CT][PT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int co = (ctile0 + ct) * 16 + 4 * q;
    eb[ct] = *reinterpret_cast<const float4*>(bias + (co < Cout ? co : Cout - 4));
  }
  if (R != nullptr) {  // uniform
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int co = (ctile0 + ct) * 16 + 4 * q;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int p = (ptile0 + pt) * 16 + n;
        // (K-split builds: a wave finishes every KSPLIT-th tile only; the others read the first group of R, one cached line)
        const bool mine = KSPLIT == 1 || ((ct * PT + pt) & (KSPLIT - 1)) == wave;
        er[ct][pt] = *reinterpret_cast<const uint2*>(R + (mine ? (size_t)(p < M ? p : M - 1) * Cout + (co < Cout ? co : Cout - 4) : (size_t)0));
      }
    }
  }
  if (KSPLIT == 1 && !OUT_F32) {
    // bf16 epilogue through LDS: a lane's MFMA result is 4 channels (8 bytes) of one pixel; writing that straight
    // out gives 32-byte pieces per pixel and tile.  Park the wave's [PT*16 pixels][CT*16 channels] tile in LDS and
    // write it back as 16-byte chunks in NHWC order (full rows when CT covers Cout).
    constexpr int ROWB = CT * 32 + 16;  // bytes per pixel row in LDS (+16: spread the 64 lanes' b64 writes over banks)
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][PT * 16 * ROWB];
    unsigned char* st = stage[wave];
    bf16_t* O = reinterpret_cast<bf16_t*>(out) + (size_t)k * act_model_stride_out;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int co = (ctile0 + ct) * 16 + 4 * q;
      const bool cval = co < Cout;
      const float4 bb = make_float4(cval ? eb[ct].x : 0.f, cval ? eb[ct].y : 0.f, cval ? eb[ct].z : 0.f, cval ? eb[ct].w : 0.f);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int p = (ptile0 + pt) * 16 + n;
        float4 v = make_float4(acc[ct][pt][0] + bb.x, acc[ct][pt][1] + bb.y, acc[ct][pt][2] + bb.z,
                               acc[ct][pt][3] + bb.w);
        if (R != nullptr && cval && p < M) {
          const uint2 r = er[ct][pt];
          v.x += bf2f(r.x & 0xffffu);
          v.y += bf2f(r.x >> 16);
          v.z += bf2f(r.y & 0xffffu);
          v.w += bf2f(r.y >> 16);
        }
        if (relu6) {
          v.x = relu6f(v.x);
          v.y = relu6f(v.y);
          v.z = relu6f(v.z);
          v.w = relu6f(v.w);
        }
        uint2 o;
        o.x = pack2(v.x, v.y);
        o.y = pack2(v.z, v.w);
        *reinterpret_cast<uint2*>(st + (pt * 16 + n) * ROWB + ct * 32 + q * 8) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int CHUNKS_PER_ROW = CT * 2;  // 16-byte chunks (8 channels) per pixel row
    constexpr int CHUNKS = PT * 16 * CHUNKS_PER_ROW;
#pragma unroll
    for (int j = 0; j < (CHUNKS + 63) / 64; ++j) {
      const int c = lane + 64 * j;
      if (c < CHUNKS) {
        const int row = c / CHUNKS_PER_ROW, col = c - row * CHUNKS_PER_ROW;
        const int p = ptile0 * 16 + row, co = ctile0 * 16 + col * 8;
        if (p < M && co < Cout)
          *reinterpret_cast<uint4*>(O + (size_t)p * Cout + co) = *reinterpret_cast<const uint4*>(st + row * ROWB + col * 16);
      }
    }
    return;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int co = (ctile0 + ct) * 16 + 4 * q;
    if (co < Cout) {
      const float4 bb = eb[ct];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int p = (ptile0 + pt) * 16 + n;
        if (KSPLIT != 1 && ((ct * PT + pt) & (KSPLIT - 1)) != wave) continue;  // wave-uniform
        if (OUT_F32 && (flags & 2)) {
          float4 v = make_float4(acc[ct][pt][0] + bb.x, acc[ct][pt][1] + bb.y, acc[ct][pt][2] + bb.z,
                                 acc[ct][pt][3] + bb.w);
          if (relu6) v = make_float4(relu6f(v.x), relu6f(v.y), relu6f(v.z), relu6f(v.w));
          if (p >= M) v = make_float4(0.f, 0.f, 0.f, 0.f);
          v.x = row_sum16(v.x) * 0.0625f;
          v.y = row_sum16(v.y) * 0.0625f;
          v.z = row_sum16(v.z) * 0.0625f;
          v.w = row_sum16(v.w) * 0.0625f;
          float* O = reinterpret_cast<float*>(out) + (size_t)k * act_model_stride_out;
          if (n == 0 && (ptile0 + pt) * 16 < M) *reinterpret_cast<float4*>(O + (size_t)(ptile0 + pt) * Cout + co) = v;
          continue;
        }
        if (p < M) {
          float4 v = make_float4(acc[ct][pt][0] + bb.x, acc[ct][pt][1] + bb.y, acc[ct][pt][2] + bb.z,
                                 acc[ct][pt][3] + bb.w);
          if (R != nullptr) {
            const uint2 r = er[ct][pt];
            v.x += bf2f(r.x & 0xffffu);
            v.y += bf2f(r.x >> 16);
            v.z += bf2f(r.y & 0xffffu);
            v.w += bf2f(r.y >> 16);
          }
          if (relu6) {
            v.x = relu6f(v.x);
            v.y = relu6f(v.y);
            v.z = relu6f(v.z);
            v.w = relu6f(v.w);
          }
          if (OUT_F32) {
            float* O = reinterpret_cast<float*>(out) + (size_t)k * act_model_stride_out;
            *reinterpret_cast<float4*>(O + (size_t)p * Cout + co) = v;
          } else {
            bf16_t* O = reinterpret_cast<bf16_t*>(out) + (size_t)k * act_model_stride_out;
            uint2 o;
            o.x = pack2(v.x, v.y);
            o.y = pack2(v.z, v.w);
            *reinterpret_cast<uint2*>(O + (size_t)p * Cout + co) = o;
          }
        }
      }
    }
  }
}

// ---- streaming GEMM for the early, HBM-bound layers (Cin <= 32: one K chunk) ----
// The CT weight tiles (one 16-byte operand each) stay in registers; a wave walks a strided list of 32-pixel groups:
// load 2 pixel operands -> CT*2 MFMAs -> bias/ReLU6/residual -> LDS-staged 16-byte NHWC stores, with the next
// group's operands requested before the current epilogue.  Amortises the per-wave set-up the one-shot kernel pays
// per 6 KB of output (its waves live ~6 us, PMC) and keeps more bytes in flight per CU.
template <int CT>
__global__ __launch_bounds__(256) void pw_stream_bf16_kernel(const bf16_t* __restrict__ in,
                                                              const bf16_t* __restrict__ whbase,
                                                              const float* __restrict__ wbase, size_t model_stride,
                                                              int k0, size_t w_off, size_t b_off,
                                                              const bf16_t* __restrict__ res, bf16_t* __restrict__ out,
                                                              int M, int Cin, int Cout, int relu6,
                                                              size_t act_model_stride_in, size_t act_model_stride_out,
                                                              int groups_per_wave) {
  constexpr int PT = 2;
  constexpr int ROWB = CT * 32 + 16;
  __shared__ __attribute__((aligned(16))) unsigned char stage[4][PT * 16 * ROWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int k = blockIdx.z;
  const int ctile0 = blockIdx.y * CT;
  const bf16_t* A = whbase + (size_t)(k0 + k) * model_stride + w_off;
  const float* bias = wbase + (size_t)(k0 + k) * model_stride + b_off;
  const bf16_t* X = in + (size_t)k * act_model_stride_in;
  const bf16_t* R = res != nullptr ? res + (size_t)k * act_model_stride_out : nullptr;
  bf16_t* O = out + (size_t)k * act_model_stride_out;
  unsigned char* st = stage[wave];
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
  const bool kval = 8 * q < Cin;

  uint4 av[CT];
  float4 bb[CT];
  bool cval[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int co = (ctile0 + ct) * 16 + n;
    av[ct] = (kval && co < Cout) ? *reinterpret_cast<const uint4*>(A + (size_t)co * Cin + 8 * q) : zero;
    const int cb = (ctile0 + ct) * 16 + 4 * q;
    cval[ct] = cb < Cout;
    bb[ct] = cval[ct] ? *reinterpret_cast<const float4*>(bias + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int n_groups = (M + 31) / 32;
  const int g0 = (blockIdx.x * 4 + wave) * groups_per_wave;
  const int g1 = min(n_groups, g0 + groups_per_wave);
  auto load_b = [&](int g, int pt) -> uint4 {
    const int p = g * 32 + pt * 16 + n;
    return (kval && g < g1) ? *reinterpret_cast<const uint4*>(X + (size_t)min(p, M - 1) * Cin + 8 * q) : zero;
  };
  uint4 b0 = load_b(g0, 0), b1 = load_b(g0, 1);
#pragma unroll 1
  for (int g = g0; g < g1; ++g) {
    const uint4 c0 = b0, c1 = b1;
    b0 = load_b(g + 1, 0);  // next group in flight during this group's MFMAs + epilogue
    b1 = load_b(g + 1, 1);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 r0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(av[ct]), as_bf16x8(c0), z4, 0, 0, 0);
      const f32x4 r1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(av[ct]), as_bf16x8(c1), z4, 0, 0, 0);
      const int co = (ctile0 + ct) * 16 + 4 * q;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const f32x4 a4 = pt == 0 ? r0 : r1;
        const int p = g * 32 + pt * 16 + n;
        float4 v = make_float4(a4[0] + bb[ct].x, a4[1] + bb[ct].y, a4[2] + bb[ct].z, a4[3] + bb[ct].w);
        if (R != nullptr && cval[ct] && p < M) {
          const uint2 r = *reinterpret_cast<const uint2*>(R + (size_t)p * Cout + co);
          v.x += bf2f(r.x & 0xffffu);
          v.y += bf2f(r.x >> 16);
          v.z += bf2f(r.y & 0xffffu);
          v.w += bf2f(r.y >> 16);
        }
        if (relu6) v = make_float4(relu6f(v.x), relu6f(v.y), relu6f(v.z), relu6f(v.w));
        uint2 o;
        o.x = pack2(v.x, v.y);
        o.y = pack2(v.z, v.w);
        *reinterpret_cast<uint2*>(st + (pt * 16 + n) * ROWB + ct * 32 + q * 8) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int CPR = CT * 2, CHUNKS = PT * 16 * CPR;
#pragma unroll
    for (int j = 0; j < (CHUNKS + 63) / 64; ++j) {
      const int c = lane + 64 * j;
      if (c < CHUNKS) {
        const int row = c / CPR, col = c - row * CPR;
        const int p = g * 32 + row, co = ctile0 * 16 + col * 8;
        if (p < M && co < Cout)
          *reinterpret_cast<uint4*>(O + (size_t)p * Cout + co) = *reinterpret_cast<const uint4*>(st + row * ROWB + col * 16);
      }
    }
    __builtin_amdgcn_wave_barrier();  // the stage is rewritten by the next group
  }
}

template <int CT>
void launch_stream(const bf16_t* in, const bf16_t* enc_wh, const float* enc_w, size_t ms, int k0, int kc, const Layer& l,
                   const bf16_t* res, void* dst, int M, hipStream_t s) {
  const int n_groups = (M + 31) / 32, n_ct = (l.cout + 15) / 16;
  // ~8 waves per SIMD worth of blocks, each wave walking a contiguous run of groups
  const int cgroups = (n_ct + CT - 1) / CT;
  int gpw = (int)(((long)n_groups * cgroups * kc + 8191) / 8192);
  if (gpw < 1) gpw = 1;
  if (gpw > 16) gpw = 16;
  const dim3 grid((n_groups + 4 * gpw - 1) / (4 * gpw), cgroups, kc);
  note_kernel(grid, dim3(256), "pw_stream_bf16_kernel<%d>", CT);
  hipLaunchKernelGGL((pw_stream_bf16_kernel<CT>), grid, dim3(256), 0, s, in, enc_wh, enc_w, ms, k0, l.w_off, l.b_off, res,
                     reinterpret_cast<bf16_t*>(dst), M, l.cin, l.cout, l.relu6, (size_t)M * l.cin, (size_t)M * l.cout,
                     gpw);
}

// ---- LDS-tiled block GEMM for the compute-heavy layers (13x13 / 7x7 / 4x4 stages with many observations) ----
// Block tile: 128 pixels x (32*WN) channels, K steps of 32, double-buffered LDS.  Each operand row is read from
// L2 once per block instead of once per wave tile (the register-direct kernel above re-reads operands ~60x on these
// shapes and is L2-bandwidth bound).  4 waves as 2 (pixels) x 2 (channels); wave tile 64 px x 16*WN ch.
template <int WN, bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const bf16_t* __restrict__ in,
                                                         const bf16_t* __restrict__ whbase,
                                                         const float* __restrict__ wbase, size_t model_stride, int k0,
                                                         size_t w_off, size_t b_off, const bf16_t* __restrict__ res,
                                                         void* __restrict__ out, int M, int Cin, int Cout, int flags,
                                                         size_t act_model_stride_in, size_t act_model_stride_out) {
  const int relu6 = flags & 1;
  constexpr int BM = 128, BN = 32 * WN, LD = 40;  // LD: bf16 elements per LDS row (32 + 8 pad -> 80 B rows)
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * (BN + BM) * LD];
  auto As = [&](int b) -> bf16_t* { return lds + b * (BN + BM) * LD; };
  auto Bs = [&](int b) -> bf16_t* { return lds + b * (BN + BM) * LD + BN * LD; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int wp = wave >> 1, wc = wave & 1;
  const int k = blockIdx.z;
  const int p0 = blockIdx.x * BM, c0 = blockIdx.y * BN;
  const bf16_t* A = whbase + (size_t)(k0 + k) * model_stride + w_off;
  const float* bias = wbase + (size_t)(k0 + k) * model_stride + b_off;
  const bf16_t* X = in + (size_t)k * act_model_stride_in;
  const bf16_t* R = res != nullptr ? res + (size_t)k * act_model_stride_out : nullptr;

  // global -> register staging: 16-byte chunks; chunk e of a tile = (row e / 4, 8 K-values (e % 4) * 8)
  constexpr int A_CH = BN * 4 / 256, B_CH = BM * 4 / 256;  // chunks per thread
  uint4 areg[A_CH], breg[B_CH];
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int e = tid + 256 * i, row = e >> 2, kk = kt * 32 + (e & 3) * 8;
      const int co = c0 + row;
      areg[i] = (co < Cout && kk < Cin) ? *reinterpret_cast<const uint4*>(A + (size_t)co * Cin + kk) : zero;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int e = tid + 256 * i, row = e >> 2, kk = kt * 32 + (e & 3) * 8;
      const int p = p0 + row;
      breg[i] = (p < M && kk < Cin) ? *reinterpret_cast<const uint4*>(X + (size_t)p * Cin + kk) : zero;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int e = tid + 256 * i;
      *reinterpret_cast<uint4*>(As(buf) + (e >> 2) * LD + (e & 3) * 8) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int e = tid + 256 * i;
      *reinterpret_cast<uint4*>(Bs(buf) + (e >> 2) * LD + (e & 3) * 8) = breg[i];
    }
  };

  f32x4 acc[WN][4];
#pragma unroll
  for (int i = 0; i < WN; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (Cin + 31) / 32;
  load_tiles(0);
  store_tiles(0);
  lds_barrier();
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);  // in flight under this step's MFMAs
    uint4 af[WN], bf[4];
#pragma unroll
    for (int i = 0; i < WN; ++i)
      af[i] = *reinterpret_cast<const uint4*>(As(buf) + (wc * 16 * WN + 16 * i + n) * LD + 8 * q);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bf[j] = *reinterpret_cast<const uint4*>(Bs(buf) + (wp * 64 + 16 * j + n) * LD + 8 * q);
#pragma unroll
    for (int i = 0; i < WN; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(af[i]), as_bf16x8(bf[j]), acc[i][j], 0, 0, 0);
    if (kt + 1 < nk) store_tiles(buf ^ 1);  // the other buffer was last read before the previous barrier
    lds_barrier();
  }

  // ---- epilogue: bias (+ residual) (+ ReLU6); lane (n, q) holds channels 4q..4q+3 of tile i for pixel n of tile j
  if (OUT_F32) {
    float* O = reinterpret_cast<float*>(out) + (size_t)k * act_model_stride_out;
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      const int co = c0 + wc * 16 * WN + 16 * i + 4 * q;
      if (co < Cout) {
        const float4 bb = *reinterpret_cast<const float4*>(bias + co);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int p = p0 + wp * 64 + 16 * j + n;
          float4 v = make_float4(acc[i][j][0] + bb.x, acc[i][j][1] + bb.y, acc[i][j][2] + bb.z, acc[i][j][3] + bb.w);
          if (relu6) v = make_float4(relu6f(v.x), relu6f(v.y), relu6f(v.z), relu6f(v.w));
          if (flags & 2) {  // 4x4 average pool: the tile's 16 pixels are one image
            if (p >= M) v = make_float4(0.f, 0.f, 0.f, 0.f);
            v.x = row_sum16(v.x) * 0.0625f;
            v.y = row_sum16(v.y) * 0.0625f;
            v.z = row_sum16(v.z) * 0.0625f;
            v.w = row_sum16(v.w) * 0.0625f;
            const int tile = (p0 + wp * 64 + 16 * j) / 16;
            if (n == 0 && tile * 16 < M) *reinterpret_cast<float4*>(O + (size_t)tile * Cout + co) = v;
          } else if (p < M) {
            *reinterpret_cast<float4*>(O + (size_t)p * Cout + co) = v;
          }
        }
      }
    }
    return;
  }
  // bf16: stage the wave's [64 px][16*WN ch] tile in (now free) LDS, write back as 16-byte NHWC chunks
  constexpr int ROWB = WN * 32 + 16;
  unsigned char* st = reinterpret_cast<unsigned char*>(lds) + wave * 64 * ROWB;
  static_assert(4 * 64 * ROWB <= (int)sizeof(lds), "staging must fit the operand buffers");
  bf16_t* O = reinterpret_cast<bf16_t*>(out) + (size_t)k * act_model_stride_out;
#pragma unroll
  for (int i = 0; i < WN; ++i) {
    const int co = c0 + wc * 16 * WN + 16 * i + 4 * q;
    const bool cval = co < Cout;
    const float4 bb = cval ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = p0 + wp * 64 + 16 * j + n;
      float4 v = make_float4(acc[i][j][0] + bb.x, acc[i][j][1] + bb.y, acc[i][j][2] + bb.z, acc[i][j][3] + bb.w);
      if (R != nullptr && cval && p < M) {
        const uint2 r = *reinterpret_cast<const uint2*>(R + (size_t)p * Cout + co);
        v.x += bf2f(r.x & 0xffffu);
        v.y += bf2f(r.x >> 16);
        v.z += bf2f(r.y & 0xffffu);
        v.w += bf2f(r.y >> 16);
      }
      if (relu6) v = make_float4(relu6f(v.x), relu6f(v.y), relu6f(v.z), relu6f(v.w));
      uint2 o;
      o.x = pack2(v.x, v.y);
      o.y = pack2(v.z, v.w);
      *reinterpret_cast<uint2*>(st + (16 * j + n) * ROWB + i * 32 + q * 8) = o;
    }
  }
  __builtin_amdgcn_wave_barrier();
  constexpr int CPR = WN * 2;  // 16-byte chunks per pixel row
#pragma unroll
  for (int jj = 0; jj < CPR; ++jj) {  // 64 rows * CPR chunks = 64 * CPR -> CPR per lane
    const int c = lane + 64 * jj;
    const int row = c / CPR, col = c - row * CPR;
    const int p = p0 + wp * 64 + row, co = c0 + wc * 16 * WN + col * 8;
    if (p < M && co < Cout)
      *reinterpret_cast<uint4*>(O + (size_t)p * Cout + co) = *reinterpret_cast<const uint4*>(st + row * ROWB + col * 16);
  }
}

// ---- persistent variant of the block GEMM for the 7x7 / 4x4 stages with many observations ----
// Same 128 x 32*WN tile and LDS layout, but a workgroup keeps its channel slice and walks a strided list of pixel
// tiles, with the (tile, K-step) sequence flattened into one software pipeline: operand chunks are requested two
// steps ahead (two register sets), so a tile's first K-step is already in flight during the previous tile's epilogue.
// The one-shot kernel pays a full global -> LDS -> MFMA latency chain per tile, which with K = 64..160 (3-5 steps)
// is most of its time (~110 TFLOP/s on the expand layers).  Outputs go straight from the accumulators (8-byte
// stores; the 4 q-lanes of a pixel write 32 contiguous bytes).
// POOL: features.18 -- the 16 pixels of a pixel tile are one 4x4 image; the epilogue averages them and writes fp32
// [image][Cout] (the pooled feature the classifier reads) instead of bf16 activations.
// Two workgroups per CU at WN = 4 (<= 256 registers: 198 with operands requested two steps ahead), three at WN = 2.
// Round 5: the three-steps-ahead build needed 260-268 registers, i.e. ONE workgroup per CU, while the launcher sized its
// grid for two (520 workgroups on 256 slots: three rounds) — 84 us for features.18 at 512 observations x 4 models.  Capping
// the registers with three sets in flight spills, and a spill reload is a vector-memory operation: `s_waitcnt vmcnt(0)`
// in the loop, the prefetch distance gone (61 us).  Two sets, no spill, grid rounded down to the resident slots: 52 us.
// (LDS row pitch 32 + 16 instead of 32 + 8 elements: the same within noise.)
constexpr int GEMM_PERS_OCC4 = 2;
template <int WN, bool POOL>
__global__ __launch_bounds__(256, WN == 4 ? GEMM_PERS_OCC4 : 3) void gemm_pers_bf16_kernel(const bf16_t* __restrict__ in,
                                                              const bf16_t* __restrict__ whbase,
                                                              const float* __restrict__ wbase, size_t model_stride,
                                                              int k0, size_t w_off, size_t b_off,
                                                              const bf16_t* __restrict__ res, void* __restrict__ outv,
                                                              int M, int Cin, int Cout, int relu6,
                                                              size_t act_model_stride_in, size_t act_model_stride_out,
                                                              int n_ptiles, int n_slices, int walkers, int xcd_r) {
  constexpr int BM = 128, BN = 32 * WN, BK = 32, LD = BK + 8;  // (K-steps of 64 measured slower: 45.8 / 64.8 / 100.4 vs 35.0 / 52.5 / 77.8 us)
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];  // [2][BN + BM][LD]: 40 KB at WN = 4
  auto As = [&](int b) -> bf16_t* { return lds + b * (BN + BM) * LD; };
  auto Bs = [&](int b) -> bf16_t* { return lds + b * (BN + BM) * LD + BN * LD; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int wp = wave >> 1, wc = wave & 1;
  // Work of workgroup L of the 1-D grid: channel slice, model, and the pixel tiles (t_first + i * walkers) * t_mul + t_add.
  // xcd_r > 0 (the launcher: the model count divides 8): workgroup L runs on XCD L % 8 (observed placement, used for
  // speed only), and an XCD serves ONE model and every xcd_r-th pixel tile of it — each activation tile crosses the
  // fabric into one L2 instead of into several (features.18 at 512 observations x 4 models, 21 MB of activations + 3 MB of
  // weights: rocprofv3 FETCH_SIZE 157 -> 13.6 MB per launch, profiles/r5/pmc_summary_v2 / v3.csv; 54 -> 48 us).
  int k, c0, t_first, t_mul, t_add, n_loc;
  {
    const int L = blockIdx.x;
    if (xcd_r > 0) {
      const int xcd = L & 7, j = L >> 3;
      k = xcd / xcd_r;
      t_add = xcd - k * xcd_r;
      t_mul = xcd_r;
      c0 = (j % n_slices) * BN;
      t_first = j / n_slices;
      n_loc = (n_ptiles - t_add + xcd_r - 1) / xcd_r;
    } else {
      const int x = L % walkers, y = (L / walkers) % n_slices;
      k = L / (walkers * n_slices);
      c0 = y * BN;
      t_first = x;
      t_mul = 1;
      t_add = 0;
      n_loc = n_ptiles;
    }
  }
  const bf16_t* A = whbase + (size_t)(k0 + k) * model_stride + w_off;
  const float* bias = wbase + (size_t)(k0 + k) * model_stride + b_off;
  const bf16_t* X = in + (size_t)k * act_model_stride_in;
  const bf16_t* R = res != nullptr ? res + (size_t)k * act_model_stride_out : nullptr;
  bf16_t* O = reinterpret_cast<bf16_t*>(outv) + (size_t)k * act_model_stride_out;
  float* OF = reinterpret_cast<float*>(outv) + (size_t)k * act_model_stride_out;
  const int nk = (Cin + BK - 1) / BK;  // Cin % 32 == 0; a last half step loads clamped (re-read) K columns with zero weight
  const int nt = t_first < n_loc ? (n_loc - 1 - t_first) / walkers + 1 : 0;
  const int total = nt * nk;
  if (total == 0) return;

  constexpr int CPR = BK / 8;  // 16-byte chunks per row and step
  constexpr int A_CH = BN * CPR / 256, B_CH = BM * CPR / 256;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  // load stream position (tile, K-step), advanced once per load_tiles call
  int l_tile = t_first, l_kt = 0;
  // Loads are UNCONDITIONAL (rows beyond Cout / M are clamped to the last valid row: their products land in outputs
  // that are never stored; Cin is a multiple of 32 on this path): a predicated load is a branch, and behind every
  // control-flow merge the compiler's s_waitcnt falls back to vmcnt(0) — the load requested for two steps ahead was
  // waited for one step ahead, so a K-step cost one memory latency (~1900 cycles for 272 cycles of MFMAs per wave).
  auto load_tiles = [&](u32x4(&areg)[A_CH], u32x4(&breg)[B_CH]) __attribute__((always_inline)) {
    const int lt = l_tile < n_loc ? l_tile : n_loc - 1;  // the stream runs past the last step: harmless re-loads
    const int p0 = (lt * t_mul + t_add) * BM;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int e = tid + 256 * i, row = e / CPR, kk = l_kt * BK + (e % CPR) * 8;
      const int co = min(c0 + row, Cout - 1);
      // a K column beyond Cin (the second half of a last, half-filled step) is read from column kk - 32 with ZERO weight
      const u32x4 v = *reinterpret_cast<const u32x4*>(A + (size_t)co * Cin + (kk < Cin ? kk : kk - 32));
      areg[i] = kk < Cin ? v : zero;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int e = tid + 256 * i, row = e / CPR, kk = l_kt * BK + (e % CPR) * 8;
      const int p = min(p0 + row, M - 1);
      breg[i] = *reinterpret_cast<const u32x4*>(X + (size_t)p * Cin + (kk < Cin ? kk : kk - 32));
    }
    if (++l_kt == nk) {
      l_kt = 0;
      l_tile += walkers;
    }
  };
  auto store_tiles = [&](int buf, const u32x4(&areg)[A_CH], const u32x4(&breg)[B_CH]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int e = tid + 256 * i;
      *reinterpret_cast<u32x4*>(As(buf) + (e / CPR) * LD + (e % CPR) * 8) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int e = tid + 256 * i;
      *reinterpret_cast<u32x4*>(Bs(buf) + (e / CPR) * LD + (e % CPR) * 8) = breg[i];
    }
  };

  f32x4 acc[WN][4];
#pragma unroll
  for (int i = 0; i < WN; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 bb[WN];
#pragma unroll
  for (int i = 0; i < WN; ++i) {
    const int co = c0 + wc * 16 * WN + 16 * i + 4 * q;
    bb[i] = co < Cout ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  int c_tile = t_first, c_kt = 0;  // compute stream position
  auto compute = [&](int buf) __attribute__((always_inline)) {
    u32x4 af[BK / 32][WN], bf[BK / 32][4];
#pragma unroll
    for (int kh = 0; kh < BK / 32; ++kh) {
#pragma unroll
      for (int i = 0; i < WN; ++i)
        af[kh][i] = *reinterpret_cast<const u32x4*>(As(buf) + (wc * 16 * WN + 16 * i + n) * LD + 32 * kh + 8 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bf[kh][j] = *reinterpret_cast<const u32x4*>(Bs(buf) + (wp * 64 + 16 * j + n) * LD + 32 * kh + 8 * q);
    }
#pragma unroll
    for (int kh = 0; kh < BK / 32; ++kh)
#pragma unroll
      for (int i = 0; i < WN; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(af[kh][i]), as_bf16x8(bf[kh][j]), acc[i][j], 0, 0, 0);
    if (++c_kt == nk) {  // tile finished: bias (+ residual) (+ ReLU6), store, restart the accumulators
      const int p0 = (c_tile * t_mul + t_add) * BM;
#pragma unroll
      for (int i = 0; i < WN; ++i) {
        const int co = c0 + wc * 16 * WN + 16 * i + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int p = p0 + wp * 64 + 16 * j + n;
          float4 v = make_float4(acc[i][j][0] + bb[i].x, acc[i][j][1] + bb[i].y, acc[i][j][2] + bb[i].z,
                                 acc[i][j][3] + bb[i].w);
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (POOL) {
            if (relu6) v = make_float4(relu6f(v.x), relu6f(v.y), relu6f(v.z), relu6f(v.w));
            if (p >= M) v = make_float4(0.f, 0.f, 0.f, 0.f);
            v.x = row_sum16(v.x) * 0.0625f;
            v.y = row_sum16(v.y) * 0.0625f;
            v.z = row_sum16(v.z) * 0.0625f;
            v.w = row_sum16(v.w) * 0.0625f;
            const int img = (p0 + wp * 64 + 16 * j) / 16;
            if (n == 0 && img * 16 < M && co < Cout) *reinterpret_cast<float4*>(OF + (size_t)img * Cout + co) = v;
          } else if (p < M && co < Cout) {
            if (R != nullptr) {
              const uint2 r = *reinterpret_cast<const uint2*>(R + (size_t)p * Cout + co);
              v.x += bf2f(r.x & 0xffffu);
              v.y += bf2f(r.x >> 16);
              v.z += bf2f(r.y & 0xffffu);
              v.w += bf2f(r.y >> 16);
            }
            if (relu6) v = make_float4(relu6f(v.x), relu6f(v.y), relu6f(v.z), relu6f(v.w));
            uint2 o;
            o.x = pack2(v.x, v.y);
            o.y = pack2(v.z, v.w);
            *reinterpret_cast<uint2*>(O + (size_t)p * Cout + co) = o;
          }
        }
      }
      c_kt = 0;
      c_tile += walkers;
    }
  };

  // Operands are requested TWO steps ahead (two register sets = the LDS rotation) and every load is issued whether or
  // not a step s + 2 exists (clamped: see load_tiles), so the loop body has no branch between a load and its use and the
  // waits are counted (vmcnt(4..7)) instead of vmcnt(0).
  u32x4 a0[A_CH], b0[B_CH], a1[A_CH], b1[B_CH];
  load_tiles(a0, b0);
  load_tiles(a1, b1);
  int s = 0;
#define GEMM_STEP(buf_, ar_, br_)   \
  store_tiles(buf_, ar_, br_);      \
  load_tiles(ar_, br_);             \
  lds_barrier();                    \
  compute(buf_);                    \
  if (++s >= total) return;
#pragma unroll 1
  for (;;) {
    GEMM_STEP(0, a0, b0)
    GEMM_STEP(1, a1, b1)
  }
#undef GEMM_STEP
}

template <int WN, bool POOL>
void launch_gemm_pers(const bf16_t* in, const bf16_t* enc_wh, const float* enc_w, size_t ms, int k0, int kc,
                      const Layer& l, const bf16_t* res, void* dst, int M, hipStream_t s) {
  const int n_ptiles = (M + 127) / 128, n_slices = (l.cout + 32 * WN - 1) / (32 * WN);
  const int per_cu = WN == 4 ? GEMM_PERS_OCC4 : 3;  // resident workgroups per CU (registers: the kernel's launch bounds)
  // every workgroup of the grid is resident at once (rounded DOWN: 520 workgroups on 512 slots are two rounds)
  const int slots = per_cu * device_cu_count();
  // XCD-aware work list (kernel comment): the models split the eight XCDs evenly and every XCD has a walker per slice
  // (the work list's L % 8 placement holds on a whole MI355X only: device_xcd_count() is 0 elsewhere and the plain list runs)
  const int xcd_r = device_xcd_count() == 8 && (kc == 1 || kc == 2 || kc == 4 || kc == 8) && n_ptiles >= 8 && slots >= 8 * n_slices ? 8 / kc : 0;
  int walkers, wgs;
  if (xcd_r > 0) {
    walkers = slots / (8 * n_slices);
    const int most = (n_ptiles + 8 / kc - 1) / (8 / kc);  // pixel tiles of an XCD
    if (walkers > most) walkers = most;
    wgs = 8 * n_slices * walkers;
  } else {
    walkers = slots / (n_slices * kc);
    if (walkers > n_ptiles) walkers = n_ptiles;
    if (walkers < 1) walkers = 1;
    wgs = walkers * n_slices * kc;
  }
  const size_t sout = POOL ? (size_t)(M / 16) * l.cout : (size_t)M * l.cout;
  constexpr size_t lds = (size_t)2 * (32 * WN + 128) * (32 + 8) * sizeof(bf16_t);  // 40 KB at WN = 4
  static_assert(lds <= 64 * 1024, "more than 64 KB of dynamic LDS needs the per-device hipFuncSetAttribute opt-in");
  note_kernel(dim3(wgs), dim3(256), "gemm_pers_bf16_kernel<%d,%s>", WN, POOL ? "true" : "false");
  hipLaunchKernelGGL((gemm_pers_bf16_kernel<WN, POOL>), dim3(wgs), dim3(256), lds, s, in, enc_wh, enc_w, ms, k0, l.w_off,
                     l.b_off, res, dst, M, l.cin, l.cout, l.relu6, (size_t)M * l.cin, sout, n_ptiles, n_slices, walkers, xcd_r);
}

template <int WN>
void launch_gemm(const bf16_t* in, const bf16_t* enc_wh, const float* enc_w, size_t ms, int k0, int kc, const Layer& l,
                 const bf16_t* res, void* dst, int M, bool out_f32, bool pool, hipStream_t s) {
  const int flags = l.relu6 | (pool ? 2 : 0);
  const size_t sout = pool ? (size_t)(M / 16) * l.cout : (size_t)M * l.cout;
  const dim3 grid((M + 127) / 128, (l.cout + 32 * WN - 1) / (32 * WN), kc);
  note_kernel(grid, dim3(256), "gemm_bf16_kernel<%d,%s>", WN, out_f32 ? "true" : "false");
  if (out_f32)
    hipLaunchKernelGGL((gemm_bf16_kernel<WN, true>), grid, dim3(256), 0, s, in, enc_wh, enc_w, ms, k0, l.w_off, l.b_off,
                       res, dst, M, l.cin, l.cout, flags, (size_t)M * l.cin, sout);
  else
    hipLaunchKernelGGL((gemm_bf16_kernel<WN, false>), grid, dim3(256), 0, s, in, enc_wh, enc_w, ms, k0, l.w_off,
                       l.b_off, res, dst, M, l.cin, l.cout, flags, (size_t)M * l.cin, sout);
}

template <int CT, int PT, int UNROLL, int KSPLIT>
void launch_pwb(const bf16_t* in, const bf16_t* enc_wh, const float* enc_w, size_t ms, int k0, int kc, const Layer& l,
                const bf16_t* res, void* dst, int M, bool out_f32, bool pool, hipStream_t s) {
  const int flags = l.relu6 | (pool ? 2 : 0);
  const size_t sout = pool ? (size_t)(M / 16) * l.cout : (size_t)M * l.cout;
  const int n_pt = (M + 15) / 16, n_ct = (l.cout + 15) / 16;
  const int groups = (n_pt + PT - 1) / PT;
  const dim3 grid(KSPLIT == 1 ? (groups + 3) / 4 : groups, (n_ct + CT - 1) / CT, kc);
  note_kernel(grid, dim3(256), "pw_bf16_kernel<%d,%d,%d,%d,%s>", CT, PT, UNROLL, KSPLIT, out_f32 ? "true" : "false");
  if (out_f32)
    hipLaunchKernelGGL((pw_bf16_kernel<CT, PT, UNROLL, KSPLIT, true>), grid, dim3(256), 0, s, in, enc_wh, enc_w, ms, k0,
                       l.w_off, l.b_off, res, dst, M, l.cin, l.cout, flags, (size_t)M * l.cin, sout);
  else
    hipLaunchKernelGGL((pw_bf16_kernel<CT, PT, UNROLL, KSPLIT, false>), grid, dim3(256), 0, s, in, enc_wh, enc_w, ms,
                       k0, l.w_off, l.b_off, res, dst, M, l.cin, l.cout, flags, (size_t)M * l.cin, sout);
}

void dispatch_pwb(const bf16_t* in, const bf16_t* enc_wh, const float* enc_w, size_t ms, int k0, int kc,
                  const Layer& l, const bf16_t* res, void* dst, int M, bool out_f32, bool pool, hipStream_t s) {
  const long n_pt = (M + 15) / 16, n_ct = (l.cout + 15) / 16;
  auto jobs = [&](int ct, int pt) { return ((n_pt + pt - 1) / pt) * ((n_ct + ct - 1) / ct) * kc; };
  // early layers (one K chunk, lots of pixels): streaming GEMM with register-resident weights
  if (l.cin <= 32 && !out_f32 && M >= 32768) {
    if (n_ct == 1) return launch_stream<1>(in, enc_wh, enc_w, ms, k0, kc, l, res, dst, M, s);
    if (n_ct == 2) return launch_stream<2>(in, enc_wh, enc_w, ms, k0, kc, l, res, dst, M, s);
    return launch_stream<6>(in, enc_wh, enc_w, ms, k0, kc, l, res, dst, M, s);
  }
  // compute-heavy shapes (K >= 64 and enough 128-pixel tiles to fill the chip): LDS-tiled block GEMM
  if (l.cin >= 64 && M >= 1024) {
    const long blocks128 = (long)((M + 127) / 128) * ((l.cout + 127) / 128) * kc;
    // many more tiles than the chip holds at once: persistent workgroups with a cross-tile software pipeline
    if (!out_f32 && !pool && blocks128 >= 192) {
      if (l.cout > 64) return launch_gemm_pers<4, false>(in, enc_wh, enc_w, ms, k0, kc, l, res, dst, M, s);
      return launch_gemm_pers<2, false>(in, enc_wh, enc_w, ms, k0, kc, l, res, dst, M, s);
    }
    if (out_f32 && pool && blocks128 >= 192) return launch_gemm_pers<4, true>(in, enc_wh, enc_w, ms, k0, kc, l, res, dst, M, s);
    if (l.cout > 64 && blocks128 >= 192) return launch_gemm<4>(in, enc_wh, enc_w, ms, k0, kc, l, res, dst, M, out_f32, pool, s);
    return launch_gemm<2>(in, enc_wh, enc_w, ms, k0, kc, l, res, dst, M, out_f32, pool, s);
  }
  // occupancy first: these GEMMs are load-latency bound, so ask for >= 4 waves per SIMD; when one wave per tile
  // cannot deliver that and K is long enough, the block's 4 waves split K (4x the waves for the same tile).
  const long want = 1024;
  const bool ks = l.cin >= 128;
#define PWB_GO(CT_, PT_, U_, KS_) \
  return launch_pwb<CT_, PT_, U_, KS_>(in, enc_wh, enc_w, ms, k0, kc, l, res, dst, M, out_f32, pool, s)
  if (n_ct >= 5) {
    if (jobs(6, 2) >= want) PWB_GO(6, 2, 2, 1);
    if (ks && jobs(6, 2) * 4 >= want) PWB_GO(6, 2, 2, 4);
  }
  if (n_ct >= 3) {
    if (jobs(4, 2) >= want) PWB_GO(4, 2, 2, 1);
    if (ks && jobs(4, 2) * 4 >= want) PWB_GO(4, 2, 2, 4);
  }
  if (n_ct >= 2) {
    if (jobs(2, 2) >= want) PWB_GO(2, 2, 4, 1);
    if (ks && jobs(2, 2) * 4 >= want) PWB_GO(2, 2, 4, 4);
  }
  if (jobs(1, 2) >= want) PWB_GO(1, 2, 8, 1);
  if (ks && jobs(1, 2) * 4 >= want) PWB_GO(1, 2, 4, 4);
  if (ks) PWB_GO(1, 1, 4, 4);
  PWB_GO(1, 1, 8, 1);
#undef PWB_GO
}

}  // namespace

hipError_t launch_encoder_bf16(const EncoderPlan& plan, const float* enc_w, const unsigned short* enc_wh, int k0, int kc,
                               const float* visual, const float* vec, int B, float* const bufs[4], float* z,
                               float* feat, int fused_blocks, hipStream_t s, EncoderTap* tap, int variant) {
  const size_t ms = plan.blob_floats;
  // the tap: after the launch that completes layer `li`, copy its output out (as fp32) and stop
  auto tapped = [&](size_t li) -> bool {
    if (tap == nullptr || tap->layer != (int)li) return false;
    const Layer& l = plan.layers[li];
    const bool last = li + 1 == plan.layers.size();  // features.18: fp32, pooled when the map is 4x4
    const bool pooled = last && plan.final_hw == 4;
    const size_t n = (size_t)kc * B * (pooled ? 1 : (size_t)l.h_out * l.h_out) * l.cout;
    if (tap->dst != nullptr) (void)launch_tap_copy(bufs[l.dst], !last, n, tap->dst, s);
    tap->served = true;
    return true;
  };
  // inverted-residual blocks as one fused kernel each.  auto (measured, K = 4): the front kernel and the row-streaming
  // blocks (features.0-7) win at every batch size (B = 1: 304 vs 311 us, B = 4: 359 vs 401); the tile blocks
  // (features.8-17, one workgroup per 1-8 observations walking 6-15 hidden chunks in sequence) from 96 (model,
  // observation) pairs (round 5, tile vs layer-wise: B = 16: 396 vs 315 us, B = 24: 405 vs 403, B = 32: 417 vs 432, B = 64: 470 vs 602,
  // B = 128: 562 vs 736; the layer-wise kernels grow by 7.3 us per observation, the tile blocks by 1.3)
  const bool auto_sel = fused_blocks < 0;
  const bool tile_ok = !auto_sel || (long)B * kc >= 96;
  if (auto_sel) fused_blocks = 17;
  std::vector<char> in_block(plan.layers.size(), 0);
  std::vector<int> block_of(plan.layers.size(), -1);
  std::vector<char> tiled(plan.blocks.size(), 0);
  for (size_t bi = 0; bi < plan.blocks.size() && (int)bi < fused_blocks; ++bi) {
    const FusedBlock& fb = plan.blocks[bi];
    const Layer* le = fb.expand >= 0 ? &plan.layers[fb.expand] : nullptr;
    // features.2-7: the row-streaming kernel with the matrix-core depthwise (round 1's vector-unit kernel, which ran
    // features.5-7 until round 5 and everything under a variant bit, is retired: encoder_bf16_irb2.hip is faster on all six)
    const bool rows = irb2_bf16_supported(le, plan.layers[fb.dw], plan.layers[fb.project], true);
    const bool tile = !rows && tile_ok && irb_tile_bf16_supported(le, plan.layers[fb.dw], plan.layers[fb.project], (variant & ENC_VAR_F17_LAYERWISE) != 0);
    if (!rows && !tile) continue;
    tiled[bi] = tile ? 1 : 2;
    if (fb.expand >= 0) in_block[fb.expand] = 1;
    in_block[fb.dw] = 1;
    in_block[fb.project] = 2;  // the block is launched where its last layer sits
    block_of[fb.project] = (int)bi;
  }
  // stem + features.1 (block 0, t = 1) as one kernel
  bool front = false;
  if (fused_blocks >= 1 && !plan.blocks.empty() && plan.blocks[0].expand < 0 && plan.layers[0].kind == L_STEM &&
      plan.blocks[0].dw == 1 && plan.layers[1].src == plan.layers[0].dst &&
      front_bf16_supported(plan.layers[0], plan.layers[plan.blocks[0].dw], plan.layers[plan.blocks[0].project])) {
    const FusedBlock& fb = plan.blocks[0];
    front = true;
    in_block[0] = in_block[fb.dw] = in_block[fb.project] = 1;
    const bool front_old = (variant & ENC_VAR_FRONT_ROUND3) != 0;  // A/B hook
    const bool f2 = !front_old && front2_bf16_supported(plan.layers[0], plan.layers[fb.dw], plan.layers[fb.project]);
    hipError_t e = (f2 ? launch_front2_bf16 : launch_front_bf16)(plan.layers[0], plan.layers[fb.dw], plan.layers[fb.project], enc_w,
                                                                 enc_wh, ms, k0, kc, B, visual,
                                                                 reinterpret_cast<unsigned short*>(bufs[fb.dst]), s);
    if (e != hipSuccess) return e;
    if (tapped((size_t)fb.project)) return hipGetLastError();
  }
  (void)front;
  for (size_t li = 0; li < plan.layers.size(); ++li) {
    const Layer& l = plan.layers[li];
    if (in_block[li] == 1) continue;
    if (in_block[li] == 2) {
      const FusedBlock& fb = plan.blocks[block_of[li]];
      const Layer* le = fb.expand >= 0 ? &plan.layers[fb.expand] : nullptr;
      hipError_t e = (tiled[block_of[li]] == 1 ? launch_irb_tile_bf16 : launch_irb2_bf16)(
          le, plan.layers[fb.dw], plan.layers[fb.project], enc_w, enc_wh, ms, k0, kc, B,
          reinterpret_cast<const unsigned short*>(bufs[fb.src]), reinterpret_cast<unsigned short*>(bufs[fb.dst]), s);
      if (e != hipSuccess) return e;
      if (tapped(li)) return hipGetLastError();
      continue;
    }
    bf16_t* dst = reinterpret_cast<bf16_t*>(bufs[l.dst]);
    if (l.kind == L_STEM) {
      const int bands = (l.h_out + STEM_ROWS - 1) / STEM_ROWS;
      const size_t lds = ((size_t)l.cin * (2 * STEM_ROWS + 1) * (l.h_in + 2) + 9 * (size_t)l.cin * 32) * sizeof(float);
      note_kernel(dim3(bands, B, kc), dim3(256), "stem_bf16_kernel<16>");
      hipLaunchKernelGGL((stem_bf16_kernel<16>), dim3(bands, B, kc), dim3(256), lds, s, visual, enc_w, ms, k0, l.w_off,
                         l.b_off, B, l.cin, l.h_in, l.h_out, dst);
    } else if (l.kind == L_DW) {
      // runs of 4 outputs per thread once there are plenty of threads; single outputs for small launches
      const long total1 = (long)B * l.h_out * l.h_out * (l.cout / 8);
      const bf16_t* src = reinterpret_cast<const bf16_t*>(bufs[l.src]);
#define DW_GO(S_, R_)                                                                                         \
  {                                                                                                           \
    const long total = (long)B * l.h_out * ((l.h_out + R_ - 1) / R_) * (l.cout / 8);                          \
    note_kernel(dim3((unsigned)((total + 255) / 256), 1, kc), dim3(256), "dw_bf16_kernel<%d,%d>", S_, R_);         \
    hipLaunchKernelGGL((dw_bf16_kernel<S_, R_>), dim3((unsigned)((total + 255) / 256), 1, kc), dim3(256), 0, s, \
                       src, enc_w, ms, k0, l.w_off, l.b_off, B, l.cout, l.h_in, l.h_out, dst);                 \
  }
      if (total1 * kc >= 16 * 65536) {
        // row-streaming kernel: waves = observations x bands x lane groups; bands sized for >= ~8k waves
        const int R = l.stride == 1 ? 4 : 2;
        const int runs = (l.h_out + R - 1) / R;
        const int lane_groups = (runs * (l.cout / 8) + 63) / 64;
        int bands = (int)((8192 + (long)B * kc * lane_groups - 1) / ((long)B * kc * lane_groups));
        if (bands > (l.h_out + 3) / 4) bands = (l.h_out + 3) / 4;
        if (bands < 1) bands = 1;
        const int band_rows = (l.h_out + bands - 1) / bands;
        bands = (l.h_out + band_rows - 1) / band_rows;
        const long waves = (long)B * bands * lane_groups;
        const dim3 grid((unsigned)((waves + 3) / 4), 1, kc);
        note_kernel(grid, dim3(256), "dw_rows_bf16_kernel<%d,%d>", l.stride == 1 ? 1 : 2, l.stride == 1 ? 4 : 2);
        if (l.stride == 1)
          hipLaunchKernelGGL((dw_rows_bf16_kernel<1, 4>), grid, dim3(256), 0, s, src, enc_w, ms, k0, l.w_off, l.b_off, B,
                             l.cout, l.h_in, l.h_out, band_rows, bands, lane_groups, dst);
        else
          hipLaunchKernelGGL((dw_rows_bf16_kernel<2, 2>), grid, dim3(256), 0, s, src, enc_w, ms, k0, l.w_off, l.b_off, B,
                             l.cout, l.h_in, l.h_out, band_rows, bands, lane_groups, dst);
      } else if (total1 * kc >= 4 * 65536) {
        if (l.stride == 1) DW_GO(1, 4) else DW_GO(2, 4)
      } else {
        if (l.stride == 1) DW_GO(1, 1) else DW_GO(2, 1)
      }
#undef DW_GO
    } else {
      const int M = B * l.h_out * l.h_out;
      const bool last = li + 1 == plan.layers.size();  // features.18 feeds the fp32 tail
      const bf16_t* res = l.residual ? reinterpret_cast<const bf16_t*>(bufs[l.res]) : nullptr;
      const bool pool = last && plan.final_hw == 4;  // features.18: fuse the 4x4 average pool into the epilogue
      dispatch_pwb(reinterpret_cast<const bf16_t*>(bufs[l.src]), enc_wh, enc_w, ms, k0, kc, l, res,
                   reinterpret_cast<void*>(bufs[l.dst]), M, last, pool, s);
    }
    if (tapped(li)) return hipGetLastError();
  }
  if (tap != nullptr) return hipGetLastError();  // an interior layer of a fused block: not served
  return launch_tail(plan, enc_w, k0, kc, bufs[plan.final_buf], plan.final_hw == 4 ? 1 : plan.final_hw * plan.final_hw, vec,
                     B, bufs[(plan.final_buf + 1) & 3], z, feat, s);
}

}  // namespace rip
