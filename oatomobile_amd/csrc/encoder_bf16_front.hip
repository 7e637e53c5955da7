// Fused front of the bf16 encoder on gfx950: features.0 (3x3 stride-2 stem conv, C -> 32, ReLU6) and features.1
// (t = 1 inverted residual: depthwise 3x3 on 32 channels, ReLU6, linear 1x1 32 -> 16) in one kernel.
//
// torchvision v0.6.0 MobileNetV2 `features[0:2]` (reference call site oatomobile/torch/networks/perception.py:36-51),
// BN folded.  Layer by layer this front is pure traffic: per observation the stem writes 160 KB, the depthwise reads
// and writes 160 KB, the projection reads 160 KB to keep 80 KB (three launches, 0.41 ms of the 2.2 ms encoder at
// 512 observations x 4 models).  Here an observation costs its 80 KB fp32 input and its 80 KB bf16 output.
//
// A workgroup owns (model, observation, band of RB output rows):
//   1. the band's input rows (with halo) are staged in LDS, coalesced fp32 reads, zero borders;
//   2. stem: thread = (pixel, 8 output channels), fp32 FMAs in the stem kernel's order (bias; c, ky, kx), ReLU6,
//      bf16 -> LDS [RB + 2 rows][52 columns][32 channels] (zero border columns; rows off the map are zeros);
//   3. each WAVE then owns 16-pixel tiles of the band end to end, no further workgroup barrier: lane (n, q) computes
//      the depthwise of pixel n for channels 8q .. 8q+7 (9 LDS reads, packed fp32 FMAs in (ky, kx) order, ReLU6) and
//      the packed bf16 result IS the lane's B operand of the projection MFMA (K = 32 = one
//      v_mfma_f32_16x16x32_bf16, A = the 16 x 32 projection weights in registers); bias, bf16, 8-byte stores.
// Arithmetic order and rounding points are the layer-wise kernels' (stem_bf16_kernel, dw_bf16_kernel,
// pw_stream_bf16_kernel), so the outputs agree bit for bit.
#include "encoder.h"

namespace rip {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
typedef unsigned short bf16_t;

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 u) {
  union {
    u32x4 u;
    bf16x8 v;
  } c;
  c.u = u;
  return c.v;
}
__device__ __forceinline__ f32x2 bfpair(unsigned u) {
  f32x2 r;
  r.x = __uint_as_float(u << 16);
  r.y = __uint_as_float(u & 0xffff0000u);
  return r;
}
__device__ __forceinline__ unsigned pack_bf16(f32x2 v) {
  union {
    bf16x2 h;
    unsigned u;
  } c;
  c.h = __builtin_convertvector(v, bf16x2);
  return c.u;
}
__device__ __forceinline__ f32x2 relu6_2(f32x2 v) {
  return __builtin_elementwise_min(__builtin_elementwise_max(v, f32x2{0.f, 0.f}), f32x2{6.f, 6.f});
}

// development only (wrong results): 1 = no stem arithmetic, 2 = no depthwise / projection / stores, 4 = no input loads
#ifndef RIP_FRONT_ABL
#define RIP_FRONT_ABL 0
#endif

constexpr int RB = 10;        // output rows per workgroup
constexpr int SC = 32;        // stem channels
constexpr int OC = 16;        // features.1 output channels

struct FrontArgs {
  const float* in;       // [B][C][HI][HI] fp32
  bf16_t* out;           // [K][B][HS][HS][16]
  const float* wbase;    // fp32 folded blobs
  const bf16_t* whbase;  // bf16 copy (pointwise weights)
  size_t model_stride;
  int k0;
  size_t ws_off, bs_off, wd_off, bd_off, wp_off, bp_off;
  int B, C, HI, HS;
  int bands;  // row bands per observation; a workgroup loops over (observation, band) items of ONE model
};

// CC: compile-time channel count (2 = the LIDAR sensor's BEV: register-resident stem taps, one-trip input staging
// with the next item's loads in flight during step 3); 0 = any C <= 4 at run time
template <int CC>
__global__ __launch_bounds__(256, 2) void front_bf16_kernel(FrontArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int C = CC ? CC : a.C, HI = a.HI, HS = a.HS;
  const int IW = HI + 8, SW = HS + 2;  // input row: 4 zeros, HI pixels, 4 zeros (the last 4-pixel group reads to HI + 7)
  constexpr int IR = 2 * (RB + 2) + 1;  // input rows of the band incl. halo
  float* xs = reinterpret_cast<float*>(smem_raw);            // [C][IR][IW], zero borders
  float* wsm = xs + (size_t)C * IR * IW;                     // [9][C][32] stem taps
  bf16_t* ss = reinterpret_cast<bf16_t*>(wsm + 9 * C * SC);  // [RB + 2][SW][32] stem output (bf16)
  float* wdm = reinterpret_cast<float*>(ss + (size_t)(RB + 2) * SW * SC);  // [9][32] depthwise taps + [32] bias
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = blockIdx.z;
  const float* W = a.wbase + (size_t)(a.k0 + k) * a.model_stride;
  const bf16_t* Wh = a.whbase + (size_t)(a.k0 + k) * a.model_stride;

  // per-lane constants of step 3 (projection operand / bias: 8 registers; the 72 depthwise taps are re-read from an LDS
  // copy per item, like the stem's, so that neither set is live across the other's step)
  const int n = lane & 15, q = lane >> 4;
  const u32x4 apj = *reinterpret_cast<const u32x4*>(Wh + a.wp_off + (size_t)n * SC + 8 * q);  // row n = output channel
  const float4 bpj = *reinterpret_cast<const float4*>(W + a.bp_off + 4 * q);

  // once per workgroup (one model): stem taps, zero paddings
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  for (int e = tid; e < 9 * C * SC; e += 256) wsm[e] = W[a.ws_off + e];
  for (int e = tid; e < 10 * SC; e += 256) wdm[e] = e < 9 * SC ? W[a.wd_off + e] : W[a.bd_off + e - 9 * SC];
  for (int e = tid; e < (RB + 2) * 2 * (SC / 8); e += 256) {  // border columns of the stem rows
    const int c8 = e % (SC / 8), side = (e / (SC / 8)) & 1, r = e / (2 * (SC / 8));
    *reinterpret_cast<u32x4*>(ss + ((size_t)r * SW + (side ? SW - 1 : 0)) * SC + 8 * c8) = zero4;
  }
  {
    const int pads = (IW - HI) >> 2;  // float4 of padding per input row: one on the left, the rest on the right
    for (int e = tid; e < C * IR * pads; e += 256) {
      const int cr = e / pads, j = e - cr * pads;
      *reinterpret_cast<u32x4*>(xs + (size_t)cr * IW + (j == 0 ? 0 : HI + 4 * j)) = zero4;
    }
  }

  // ---- persistent loop over this model's (observation, band) items.  Staging: stem rows oy0-1 .. oy0+rows need input
  // rows 2(oy0-1)-1 .. 2(oy0+rows)+1; 16-byte loads, ALL requested before the first LDS write (a load -> store loop
  // pays the memory latency per trip); padded row = 4 zeros, HI pixels, zeros: pixel column ix sits at ix + 4, so the
  // LDS writes are 16-byte aligned.  With C == 2 the band is one trip of loads (5 per thread): the NEXT item's are
  // requested before step 3 of the current one, so the HBM latency is covered by it. ----
  constexpr int NL = 6;
  const int q4 = HI >> 2;  // float4 per input row
  const int total = C * IR * q4;
  const bool one_trip = CC == 2 && total <= 256 * NL;
  const int nitems = a.B * a.bands;
  auto load_input = [&](int item, int e0, u32x4(&v)[NL]) {
    const int ib = item / a.bands, iband = item - ib * a.bands;
    const int iiy0 = 2 * (iband * RB - 1) - 1;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int e = e0 + tid + 256 * j;
      const int cr = e / q4, x4 = e - cr * q4;
      const int c = cr / IR, r = cr - c * IR;
      const int iy = iiy0 + r;
      v[j] = (!(RIP_FRONT_ABL & 4) && e < total && iy >= 0 && iy < HI)
                 ? *reinterpret_cast<const u32x4*>(a.in + ((size_t)ib * C + c) * HI * HI + (size_t)iy * HI + 4 * x4)
                 : zero4;
    }
  };
  auto store_input = [&](int e0, const u32x4(&v)[NL]) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int e = e0 + tid + 256 * j;
      const int cr = e / q4, x4 = e - cr * q4;
      if (e < total) *reinterpret_cast<u32x4*>(xs + (size_t)cr * IW + 4 + 4 * x4) = v[j];
    }
  };
  u32x4 vnext[NL];
  if (one_trip && (int)blockIdx.x < nitems) load_input(blockIdx.x, 0, vnext);
#pragma unroll 1
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
  const int b = item / a.bands, band = item - b * a.bands;
  const int oy0 = band * RB;
  const int rows = min(RB, HS - oy0);
  if (one_trip) {
    store_input(0, vnext);
  } else {
    for (int e0 = 0; e0 < total; e0 += 256 * NL) {
      u32x4 v[NL];
      load_input(item, e0, v);
      store_input(e0, v);
    }
  }
  __syncthreads();  // (also: every wave has left step 3 of the previous item, the stem rows may be overwritten)

  // ---- 2. stem rows oy0-1 .. oy0+rows (rows off the map: zeros = the depthwise's padding).
  // thread = (row, 4 adjacent pixels, 4 output channels): one 16-byte tap read feeds 8 packed FMAs, the 9 input
  // columns of a (channel, ky) row are read once for the 4 pixels.  Per output the chain is the stem kernel's:
  // bias, then (c, ky, kx) ascending. ----
  {
    const float* bias = W + a.bs_off;
    const int npg = (HS + 3) >> 2;
    f32x2 wreg[18][2];
    int opq2 = 0;
    asm volatile("" : "+v"(opq2));  // (keeps the tap reads inside the item loop, see step 3)
    if (CC == 2) {
#pragma unroll
      for (int t = 0; t < 18; ++t) {  // t = (ky * 3 + kx) * C + c: the blob's tap order
        const float4 w = *reinterpret_cast<const float4*>(wsm + t * SC + 4 * (tid & 7) + opq2);
        wreg[t][0] = f32x2{w.x, w.y};
        wreg[t][1] = f32x2{w.z, w.w};
      }
    }
    for (int e = tid; e < (rows + 2) * npg * 8; e += 256) {
      const int c4 = e & 7, pr = e >> 3;
      const int r = pr / npg, pg = pr - r * npg;
      const int sr = oy0 - 1 + r;
      const bool rok = sr >= 0 && sr < HS;
      f32x2 acc[4][2];
      {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + 4 * c4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i][0] = f32x2{b0.x, b0.y};
          acc[i][1] = f32x2{b0.z, b0.w};
        }
      }
      if (rok && !(RIP_FRONT_ABL & 1) && CC == 2) {
        // C == 2 (the LIDAR sensor's BEV): this thread's 18 x 4 tap weights live in registers (its channel group is
        // fixed: e & 7 == tid & 7) -- half of the stem's LDS reads were tap reads, and the stem was LDS-bound
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const float* xr = xs + ((size_t)c * IR + 2 * r + ky) * IW + 8 * pg;
            const float4 x0 = *reinterpret_cast<const float4*>(xr), x1 = *reinterpret_cast<const float4*>(xr + 4);
            const float4 x2 = *reinterpret_cast<const float4*>(xr + 8);
            const float x[9] = {x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const f32x2 w01 = wreg[(ky * 3 + kx) * 2 + c][0], w23 = wreg[(ky * 3 + kx) * 2 + c][1];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const f32x2 v = {x[2 * i + kx], x[2 * i + kx]};
                acc[i][0] = __builtin_elementwise_fma(v, w01, acc[i][0]);
                acc[i][1] = __builtin_elementwise_fma(v, w23, acc[i][1]);
              }
            }
          }
        }
      } else if (rok && !(RIP_FRONT_ABL & 1)) {
        for (int c = 0; c < C; ++c) {
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            // stem row sr taps input rows 2 sr - 1 + ky = iy0 + 2 r + ky
            // pixel column ix sits at ix + 4: the group's taps 2 (4 pg) - 1 .. + 8 are padded columns 8 pg + 3 .. + 11
            const float* xr = xs + ((size_t)c * IR + 2 * r + ky) * IW + 8 * pg;
            const float4 x0 = *reinterpret_cast<const float4*>(xr), x1 = *reinterpret_cast<const float4*>(xr + 4);
            const float4 x2 = *reinterpret_cast<const float4*>(xr + 8);
            const float x[9] = {x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const float4 w = *reinterpret_cast<const float4*>(wsm + ((ky * 3 + kx) * C + c) * SC + 4 * c4);
              const f32x2 w01 = {w.x, w.y}, w23 = {w.z, w.w};
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const f32x2 v = {x[2 * i + kx], x[2 * i + kx]};
                acc[i][0] = __builtin_elementwise_fma(v, w01, acc[i][0]);
                acc[i][1] = __builtin_elementwise_fma(v, w23, acc[i][1]);
              }
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ox = 4 * pg + i;
        u32x2 o = {0u, 0u};
        if (rok) {
          o.x = pack_bf16(relu6_2(acc[i][0]));
          o.y = pack_bf16(relu6_2(acc[i][1]));
        }
        if (ox < HS) *reinterpret_cast<u32x2*>(ss + ((size_t)r * SW + ox + 1) * SC + 4 * c4) = o;
      }
    }
  }
  __syncthreads();
  if (one_trip && item + (int)gridDim.x < nitems) load_input(item + gridDim.x, 0, vnext);  // lands under step 3

  // ---- 3. depthwise + projection, one 16-pixel tile per wave at a time ----
  f32x2 wt[9][4], bd[4];
  {
    int opq = 0;
    asm volatile("" : "+v"(opq));  // the (loop-invariant) tap reads must stay inside the item loop: hoisted they are 80
                                       // registers live across both steps
    const float* wd = wdm + 8 * q + opq;
#pragma unroll
    for (int t = 0; t <= 9; ++t) {
      const float4 w0 = *reinterpret_cast<const float4*>(wd + t * SC);
      const float4 w1 = *reinterpret_cast<const float4*>(wd + t * SC + 4);
      f32x2(&dst)[4] = t < 9 ? wt[t < 9 ? t : 0] : bd;
      dst[0] = f32x2{w0.x, w0.y};
      dst[1] = f32x2{w0.z, w0.w};
      dst[2] = f32x2{w1.x, w1.y};
      dst[3] = f32x2{w1.z, w1.w};
    }
  }
  bf16_t* og = a.out + (((size_t)k * a.B + b) * HS + oy0) * HS * OC;
  const int P = rows * HS, ntiles = (P + 15) >> 4;
  // two tiles per trip: the 18 window reads of both are requested before the first value is unpacked, so one tile's
  // LDS / MFMA latency is covered by the other's arithmetic (the loop bound is a run-time value: no unrolling by
  // the compiler)
  for (int tile0 = wv; tile0 < ((RIP_FRONT_ABL & 2) ? 0 : ntiles); tile0 += 8) {
    u32x4 v[2][9];
    int pp[2];
    bool val[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int tile = tile0 + 4 * u;
      const int p = 16 * tile + n;
      val[u] = tile < ntiles && p < P;
      pp[u] = p;
      const int pc = val[u] ? p : P - 1;
      const int oyl = pc / HS, ox = pc - oyl * HS;
      // window origin: stem row oyl (= output row - 1), padded column ox (= pixel column - 1)
      const bf16_t* sp = ss + ((size_t)oyl * SW + ox) * SC + 8 * q;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) v[u][ky * 3 + kx] = *reinterpret_cast<const u32x4*>(sp + ((size_t)ky * SW + kx) * SC);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (tile0 + 4 * u >= ntiles) break;  // wave-uniform
      f32x2 s[4] = {bd[0], bd[1], bd[2], bd[3]};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        s[0] = __builtin_elementwise_fma(bfpair(v[u][t].x), wt[t][0], s[0]);
        s[1] = __builtin_elementwise_fma(bfpair(v[u][t].y), wt[t][1], s[1]);
        s[2] = __builtin_elementwise_fma(bfpair(v[u][t].z), wt[t][2], s[2]);
        s[3] = __builtin_elementwise_fma(bfpair(v[u][t].w), wt[t][3], s[3]);
      }
      u32x4 d;
      d.x = pack_bf16(relu6_2(s[0]));
      d.y = pack_bf16(relu6_2(s[1]));
      d.z = pack_bf16(relu6_2(s[2]));
      d.w = pack_bf16(relu6_2(s[3]));
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(apj), as_bf16x8(d), z4, 0, 0, 0);
      u32x2 o;
      o.x = pack_bf16(f32x2{c[0] + bpj.x, c[1] + bpj.y});
      o.y = pack_bf16(f32x2{c[2] + bpj.z, c[3] + bpj.w});
      if (val[u]) *reinterpret_cast<u32x2*>(og + (size_t)pp[u] * OC + 4 * q) = o;
    }
  }
  }  // items
}

}  // namespace

bool front_bf16_supported(const Layer& ls, const Layer& ld, const Layer& lp) {
  return ls.kind == L_STEM && ls.cout == SC && ls.stride == 2 && ls.relu6 && ls.h_out * 2 == ls.h_in && ld.kind == L_DW &&
         ld.cout == SC && ld.stride == 1 && ld.h_in == ls.h_out && ld.relu6 && lp.kind == L_PW && lp.cin == SC &&
         lp.cout == OC && !lp.relu6 && !lp.residual && ls.cin <= 4;
}

hipError_t launch_front_bf16(const Layer& ls, const Layer& ld, const Layer& lp, const float* enc_w,
                             const unsigned short* enc_wh, size_t model_stride, int k0, int kc, int B, const float* visual,
                             unsigned short* y, hipStream_t s) {
  FrontArgs a;
  a.in = visual;
  a.out = y;
  a.wbase = enc_w;
  a.whbase = enc_wh;
  a.model_stride = model_stride;
  a.k0 = k0;
  a.ws_off = ls.w_off;
  a.bs_off = ls.b_off;
  a.wd_off = ld.w_off;
  a.bd_off = ld.b_off;
  a.wp_off = lp.w_off;
  a.bp_off = lp.b_off;
  a.B = B;
  a.C = ls.cin;
  a.HI = ls.h_in;
  a.HS = ls.h_out;
  constexpr int IR = 2 * (RB + 2) + 1;
  const size_t lds = ((size_t)a.C * IR * (a.HI + 8) + 9 * a.C * SC) * sizeof(float) +
                     (size_t)(RB + 2) * (a.HS + 2) * SC * sizeof(bf16_t) + 10 * SC * sizeof(float);
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(front_bf16_kernel<2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(front_bf16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            96 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  if (lds > 96 * 1024) return hipErrorInvalidValue;
  a.bands = (a.HS + RB - 1) / RB;
  // persistent workgroups: two per CU in total (what the LDS footprint admits), each looping over the (observation,
  // band) items of one model
  int cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  int wgs = (per_cu * cus + kc - 1) / kc;
  if (B * a.bands < 2 * wgs) wgs = B * a.bands;  // small launches: one item per workgroup (no uneven 1-or-2 split)
  if (wgs < 1) wgs = 1;
  note_kernel(dim3(wgs, 1, kc), dim3(256), "front_bf16_kernel<%d>", a.C == 2 && a.HI == 100 ? 2 : 0);
  if (a.C == 2 && a.HI == 100)
    hipLaunchKernelGGL(front_bf16_kernel<2>, dim3(wgs, 1, kc), dim3(256), lds, s, a);
  else
    hipLaunchKernelGGL(front_bf16_kernel<0>, dim3(wgs, 1, kc), dim3(256), lds, s, a);
  return hipGetLastError();
}

}  // namespace rip
