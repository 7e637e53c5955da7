// Fused bf16 inverted-residual block for the large-image stages (features.1 .. features.7) on gfx950:
// expand 1x1 (MFMA) -> depthwise 3x3 -> project 1x1 (MFMA) (+ residual) in one kernel, row-streaming.
//
// torchvision v0.6.0 `InvertedResidual` (reference call site oatomobile/torch/networks/perception.py:36-51), BN
// folded.  Layer by layer these blocks are pure HBM traffic: the t-times expanded tensor is written once and read
// once (features.2 at 256 observations x 4 models: 0.49 GB each way for 0.08 GB of block input).  Here it only ever
// exists as three rows per 48-channel chunk in LDS.
//
// Decomposition: a workgroup owns (model, observation, band of output rows) and has one wave per CW hidden channels
// (CW = 48: NW = HID / 48 = 3 or 4 waves for HID = 144 / 192; CW = 32, NW = 3 for HID = 96).  Walking down the band, per output row each wave
//   1. expands the new input row(s) for ITS hidden chunk (A = 2 weight tiles held in registers, B = 16-pixel operands
//      loaded from the block input one row ahead) into its private 3-row LDS ring  -- wave-local, no barrier;
//   2. runs the 3x3 depthwise for its chunk from the ring (register window over rows, packed fp32 math) and writes the
//      bf16 result row into the shared projection operand row  ds[px][HID];
//   3. after ONE workgroup barrier, computes its share of the (pixel-tile x channel-tile) projection tiles over the full
//      hidden K from `ds` (A = projection weights held in registers for the whole band), adds bias / residual and
//      stores the output row.  `ds` is double-buffered, so the barrier per row is the only synchronisation.
// Zero padding: ring rows carry a zero pixel slot on either side; rows outside the image are replaced by zeros when
// the window is filled (wave-uniform).  Arithmetic matches the layer-wise bf16 kernels: bf16 operands, fp32
// accumulate / bias / ReLU6 / residual, bf16 rounding (RNE) where the layer-wise path rounds.
#include <stdlib.h>

#include "encoder.h"
#include "flow.h"  // device_cu_count

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

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_srd(const bf16_t* row, int bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(row);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
constexpr int OOB = 0x40000000;  // byte offset beyond any row descriptor: loads return 0, stores are dropped

struct IrbArgs {
  const bf16_t* x;      // [K][B][H_in][H_in][CIN]
  bf16_t* y;            // [K][B][H_out][H_out][COUT]
  const float* wbase;   // fp32 folded blobs (biases, depthwise taps)
  const bf16_t* whbase; // bf16 copy of the blobs (pointwise weights), same offsets
  size_t model_stride;
  int k0;
  size_t we_off, be_off, wd_off, bd_off, wp_off, bp_off;
  int B, CIN, HID, COUT, H_in, H_out, residual;
  int band_rows;
  int EW;               // pixel slots per ring row (>= H_in + 2, covers the last run's taps)
  int WP;               // H_out rounded up to 16
};


// STRIDE: depthwise stride; R: outputs per depthwise lane along x; EXPAND: false for the t = 1 block; NW: waves =
// 32-channel hidden chunks; TPW: projection tiles per wave; NPT: 16-pixel tiles per input row; WINDOW: keep the rows
// shared with the next output row in registers (stride 1 with spare registers) instead of re-reading them from LDS;
// APREG: projection weights stay in registers for the band (else re-read from L1 each row, after the depthwise);
// WLDS: depthwise taps are read from an LDS copy per use instead of living in 72 registers (register-tight variants:
// a spill reload waits on vmcnt, i.e. on every input prefetch still in flight).
template <int STRIDE, int R, bool EXPAND, int NW, int TPW, int NPT, bool WINDOW, bool APREG, bool WLDS, int CW, int KS>
__global__ __launch_bounds__(NW * 64, 2) void irb_rows_bf16_kernel(IrbArgs a) {
  constexpr int COLS = (R - 1) * STRIDE + 3;
  // CW: hidden channels per wave (32 or 48: HID = 96 / 144 / 192 split over 2 / 3 / 4 waves keeps every SIMD equally
  // loaded -- with 32-channel chunks NW = 5 or 6 waves share 4 SIMDs and the row barrier waits for the SIMD that hosts
  // two of them); KS: 32-wide K steps of the projection (ceil(HID / 32), independent of CW)
  constexpr int CG = CW / 8;        // 8-channel groups per wave = depthwise lanes per run
  constexpr int NHT = CW / 16;      // 16-channel MFMA tiles of the expansion per wave
  constexpr int ELD = CW + 8;       // bf16 elements per ring pixel slot (+8 pad: odd multiple of 16 bytes)
  constexpr int DCOLS = (NW * CW > KS * 32 ? NW * CW : KS * 32);
  constexpr int DLD = DCOLS + 8;    // bf16 elements per projection-operand pixel row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem_raw);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int k = blockIdx.z, band = blockIdx.x;
  const int CIN = a.CIN, HID = a.HID, COUT = a.COUT, H_in = a.H_in, H_out = a.H_out, EW = a.EW, WP = a.WP;
  bf16_t* es = lds + (size_t)w * 3 * EW * ELD;        // this wave's ring: [3][EW][ELD]
  bf16_t* ds = lds + (size_t)NW * 3 * EW * ELD;       // [2][WP][DLD]
  float* wl = reinterpret_cast<float*>(ds + (size_t)2 * WP * DLD);  // [9][NW*CW] depthwise taps (WLDS)
  const bf16_t* zrow = ds + (size_t)2 * WP * DLD + (WLDS ? 9 * NW * CW * 2 : 0);  // [EW][ELD] zeros: rows off the image
  const float* W = a.wbase + (size_t)(a.k0 + k) * a.model_stride;
  const bf16_t* Wh = a.whbase + (size_t)(a.k0 + k) * a.model_stride;
  // PERSISTENT over observations (round 5): the workgroup walks observations blockIdx.y, + gridDim.y, ... — the LDS
  // zeroing and the 30-odd per-wave constants (expansion / projection operands, 72 depthwise taps) are built once per
  // workgroup instead of once per observation (features.5-7 at 512 observations x 4 models: 2048 times per launch)
  const bf16_t* xin = a.x + (size_t)k * a.B * H_in * H_in * CIN;  // set per observation below (the lambdas read it by reference)
  bf16_t* yout = a.y + (size_t)k * a.B * H_out * H_out * COUT;
  const bf16_t* const xin0 = xin;
  bf16_t* const yout0 = yout;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ---- zero the LDS once: ring borders / unused channels must read as 0 (never NaN) ----
  {
    const int total16 = (NW * 3 * EW * ELD + 2 * WP * DLD + (WLDS ? 9 * NW * CW * 2 : 0) + EW * ELD) / 8;
    for (int e = threadIdx.x; e < total16; e += NW * 64) reinterpret_cast<u32x4*>(lds)[e] = zero4;
  }

  if (WLDS) {
    lds_barrier();
    for (int e = threadIdx.x; e < 9 * NW * CW; e += NW * 64) {
      const int t = e / (NW * CW), c = e - t * (NW * CW);
      wl[e] = c < HID ? W[a.wd_off + (size_t)t * HID + c] : 0.f;
    }
  }

  // ---- per-wave constants ----
  u32x4 ae[NHT];
  float4 be[NHT];
  if (EXPAND) {
#pragma unroll
    for (int ht = 0; ht < NHT; ++ht) {
      const int h_row = CW * w + 16 * ht + n;
      ae[ht] = (h_row < HID && 8 * q < CIN) ? *reinterpret_cast<const u32x4*>(Wh + a.we_off + (size_t)h_row * CIN + 8 * q)
                                            : zero4;
      const int hb = CW * w + 16 * ht + 4 * q;
      be[ht] = hb < HID ? *reinterpret_cast<const float4*>(W + a.be_off + hb) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // depthwise lane = (run, c8): CG consecutive lanes cover the chunk's CW channels of one run
  const int runs = (H_out + R - 1) / R;
  const int run_raw = lane / CG;
  const int c8 = lane - run_raw * CG;
  const int hch = CW * w + 8 * c8;  // first hidden channel of this lane
  const bool dw_active = run_raw < runs && hch < HID;
  const int run = run_raw < runs ? run_raw : 0;
  f32x2 wt[9][4], bd[4];
  {
    const float* wd = W + a.wd_off + (hch < HID ? hch : 0);
    if (!WLDS) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float4 w0 = *reinterpret_cast<const float4*>(wd + (size_t)t * HID);
        const float4 w1 = *reinterpret_cast<const float4*>(wd + (size_t)t * HID + 4);
        wt[t][0] = f32x2{w0.x, w0.y};
        wt[t][1] = f32x2{w0.z, w0.w};
        wt[t][2] = f32x2{w1.x, w1.y};
        wt[t][3] = f32x2{w1.z, w1.w};
      }
    }
    const float* bp = W + a.bd_off + (hch < HID ? hch : 0);
    const float4 b0 = *reinterpret_cast<const float4*>(bp);
    const float4 b1 = *reinterpret_cast<const float4*>(bp + 4);
    bd[0] = f32x2{b0.x, b0.y};
    bd[1] = f32x2{b0.z, b0.w};
    bd[2] = f32x2{b1.x, b1.y};
    bd[3] = f32x2{b1.z, b1.w};
  }
  // projection tiles of this wave: tile = w + NW * t -> (pixel tile pt, channel tile ct)
  const int n_ct = (COUT + 15) / 16, TT = (WP / 16) * n_ct;
  u32x4 ap[TPW][KS];
  float4 bpj[TPW];
  auto load_ap = [&]() {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = w + NW * t;
      const int ct = tile % n_ct;
      const int co = 16 * ct + n;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kk = 32 * ks + 8 * q;
        ap[t][ks] = (tile < TT && co < COUT && kk < HID)
                        ? *reinterpret_cast<const u32x4*>(Wh + a.wp_off + (size_t)co * HID + kk)
                        : zero4;
      }
      const int cb = 16 * ct + 4 * q;
      bpj[t] = (tile < TT && cb < COUT) ? *reinterpret_cast<const float4*>(W + a.bp_off + cb)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (APREG) load_ap();
  int yoff[TPW], roff[TPW];  // output / residual byte offsets inside a row (out of bounds for padding lanes)
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = w + NW * t;
    const int pt = tile / n_ct, ct = tile - pt * n_ct;
    const int px = 16 * pt + n, co = 16 * ct + 4 * q;
    const bool ok = tile < TT && px < H_out && co < COUT;
    yoff[t] = ok ? (px * COUT + co) * 2 : OOB;
    roff[t] = ok ? (px * CIN + co) * 2 : OOB;
  }

  // ---- block-input row operands, fetched one output row ahead.  Per-row buffer descriptors (0 bytes for rows off
  // the image) + per-lane offsets that are out of bounds for invalid pixels / channels: no predication in the loop.
  int xoff[NPT];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    if (EXPAND) {
      const int px = 16 * i + n;
      xoff[i] = (px < H_in && 8 * q < CIN) ? (px * CIN + 8 * q) * 2 : OOB;
    } else {  // t == 1: the block input IS the depthwise operand: lane copies (pixel, 8 channels)
      const int item = lane + 64 * i, px = item >> 2, cc = item & 3;
      xoff[i] = px < H_in ? (px * CIN + 8 * cc) * 2 : OOB;
    }
  }
  const int x_row_bytes = H_in * CIN * 2;
  auto load_x = [&](int iy, u32x4(&xr)[NPT]) {
    const bool rok = iy >= 0 && iy < H_in;
    const __amdgpu_buffer_rsrc_t srd = row_srd(xin + (size_t)(rok ? iy : 0) * H_in * CIN, rok ? x_row_bytes : 0);
#pragma unroll
    for (int i = 0; i < NPT; ++i) xr[i] = __builtin_amdgcn_raw_buffer_load_b128(srd, xoff[i], 0, 0);
  };
  // Rows off the image are expanded too (from zero operands) into a ring slot nobody reads: the window fill below
  // takes the zero row for them.  Pixels beyond the row are written as zeros (the right-hand padding slot).
  auto expand_row = [&](int iy, const u32x4(&xr)[NPT]) {
    bf16_t* ring = es + (size_t)((iy + 3) % 3) * EW * ELD;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      if (EXPAND) {
        if (16 * i >= H_in) continue;
        const int px = 16 * i + n;
        const bool pv = px < H_in;
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) {
          const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
          const f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ae[ht]), as_bf16x8(xr[i]), z4, 0, 0, 0);
          u32x2 o;
          o.x = pack_bf16(relu6_2(f32x2{c[0] + be[ht].x, c[1] + be[ht].y}));
          o.y = pack_bf16(relu6_2(f32x2{c[2] + be[ht].z, c[3] + be[ht].w}));
          o.x = pv ? o.x : 0u;
          o.y = pv ? o.y : 0u;
          *reinterpret_cast<u32x2*>(ring + (size_t)(px + 1) * ELD + 16 * ht + 4 * q) = o;
        }
      } else {
        const int item = lane + 64 * i, px = item >> 2, cc = item & 3;
        if (16 * i >= H_in) continue;  // 64 lanes = 16 pixels x 4 channel groups per step
        *reinterpret_cast<u32x4*>(ring + (size_t)(px + 1) * ELD + 8 * cc) = xr[i];  // OOB lanes loaded zeros
      }
    }
  };
  // depthwise window rows come from the ring; slot index = ix + 1, first tap of the run at ix = run*R*STRIDE - 1
  const int slot0 = run * R * STRIDE;
  auto read_row = [&](int iy, u32x4(&row)[COLS]) {
    const bool ok = iy >= 0 && iy < H_in;
    const bf16_t* ring = (ok ? es + (size_t)((iy + 3) % 3) * EW * ELD : zrow) + (size_t)slot0 * ELD + 8 * c8;
#pragma unroll
    for (int j = 0; j < COLS; ++j) row[j] = *reinterpret_cast<const u32x4*>(ring + (size_t)j * ELD);
  };

  const int oy0 = band * a.band_rows, oy1 = min(H_out, oy0 + a.band_rows);
  lds_barrier();  // LDS zeroed

  u32x4 xr[STRIDE][NPT];
  u32x4 win[3][COLS];
#pragma unroll 1
  for (int b = blockIdx.y; b < a.B; b += gridDim.y) {
  xin = xin0 + (size_t)b * H_in * H_in * CIN;
  yout = yout0 + (size_t)b * H_out * H_out * COUT;
  // prologue: rows oy0*S-1 .. oy0*S+1-S are expanded here, the remaining S rows of the first window in the loop
#pragma unroll
  for (int i = 0; i < 3 - STRIDE; ++i) {
    u32x4 x0[NPT];
    load_x(oy0 * STRIDE - 1 + i, x0);
    expand_row(oy0 * STRIDE - 1 + i, x0);
  }
#pragma unroll
  for (int i = 0; i < STRIDE; ++i) load_x(oy0 * STRIDE + 2 - STRIDE + i, xr[i]);
  if (WINDOW) {
#pragma unroll
    for (int i = 0; i < 3 - STRIDE; ++i) read_row(oy0 * STRIDE - 1 + i, win[i]);
  }

  int buf = 0;
#pragma unroll 1
  for (int oy = oy0; oy < oy1; ++oy) {
    // 1. expand the S new rows (operands fetched during the previous row), then request the next ones
    // (round 5: requested two / three output rows ahead with as many register sets — features.7, which has the registers:
    // 48.0 / 46.7 / 47.2 us; the rows do not wait for their input)
#pragma unroll
    for (int i = 0; i < STRIDE; ++i) expand_row(oy * STRIDE + 2 - STRIDE + i, xr[i]);
    // (unconditional: past the band the rows are image rows nobody uses, past the image load_x takes an empty
    // descriptor — a branch here makes the compiler wait for every load in flight at the merge)
#pragma unroll
    for (int i = 0; i < STRIDE; ++i) load_x((oy + 1) * STRIDE + 2 - STRIDE + i, xr[i]);
    // the residual operands of this row's projection tiles are requested here, a depthwise ahead of their use
    const __amdgpu_buffer_rsrc_t rsrd = row_srd(xin + (size_t)oy * H_in * CIN, a.residual ? x_row_bytes : 0);
    u32x2 rr[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) rr[t] = __builtin_amdgcn_raw_buffer_load_b64(rsrd, roff[t], 0, 0);  // zeros when there is no residual
    // 2. depthwise for this chunk
    if (WINDOW) {
#pragma unroll
      for (int i = 3 - STRIDE; i < 3; ++i) read_row(oy * STRIDE - 1 + i, win[i]);
    }
    f32x2 acc[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[r][e] = bd[e];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      if (!WINDOW) read_row(oy * STRIDE - 1 + ky, win[ky]);
      if (WLDS) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float* wp = wl + (ky * 3 + kx) * (NW * CW) + CW * w + 8 * c8;
          const float4 w0 = *reinterpret_cast<const float4*>(wp);
          const float4 w1 = *reinterpret_cast<const float4*>(wp + 4);
          wt[ky * 3 + kx][0] = f32x2{w0.x, w0.y};
          wt[ky * 3 + kx][1] = f32x2{w0.z, w0.w};
          wt[ky * 3 + kx][2] = f32x2{w1.x, w1.y};
          wt[ky * 3 + kx][3] = f32x2{w1.z, w1.w};
        }
      }
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
    bf16_t* drow = ds + (size_t)buf * WP * DLD;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int px = run * R + r;
      u32x4 o;
      o.x = pack_bf16(relu6_2(acc[r][0]));
      o.y = pack_bf16(relu6_2(acc[r][1]));
      o.z = pack_bf16(relu6_2(acc[r][2]));
      o.w = pack_bf16(relu6_2(acc[r][3]));
      *reinterpret_cast<u32x4*>(drow + (size_t)px * DLD + hch) = o;  // lanes past the last run duplicate run 0
    }
    if (WINDOW) {
#pragma unroll
      for (int j = 0; j < COLS; ++j) {
        if (STRIDE == 1) {
          win[0][j] = win[1][j];
          win[1][j] = win[2][j];
        } else {
          win[0][j] = win[2][j];
        }
      }
    }
    if (!APREG) load_ap();
    lds_barrier();  // every chunk of ds[buf] is in place
    // 3. projection tiles of this wave over the full hidden K
    const __amdgpu_buffer_rsrc_t ysrd = row_srd(yout + (size_t)oy * H_out * COUT, H_out * COUT * 2);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = w + NW * t;  // (a wave without a tile here multiplies zero weights and stores out of bounds: no branch)
      const int pt = tile / n_ct;
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      const bf16_t* brow = drow + (size_t)(16 * pt + n) * DLD + 8 * q;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 bv = *reinterpret_cast<const u32x4*>(brow + 32 * ks);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ap[t][ks]), as_bf16x8(bv), c, 0, 0, 0);
      }
      f32x2 v0 = {c[0] + bpj[t].x, c[1] + bpj[t].y}, v1 = {c[2] + bpj[t].z, c[3] + bpj[t].w};
      v0 += bfpair(rr[t].x);
      v1 += bfpair(rr[t].y);
      u32x2 o;
      o.x = pack_bf16(v0);
      o.y = pack_bf16(v1);
      __builtin_amdgcn_raw_buffer_store_b64(o, ysrd, yoff[t], 0, 0);
    }
    buf ^= 1;
  }
  lds_barrier();  // the last row's projection has read ds before the next observation's depthwise writes it
  }  // observations
}

template <int STRIDE, int R, bool EXPAND, int NW, int TPW, int NPT, bool WINDOW, bool APREG, bool WLDS, int CW, int KS>
hipError_t launch_irb(const IrbArgs& a, int kc, int bands, hipStream_t s) {
  constexpr int ELD = CW + 8;
  constexpr int DLD = (NW * CW > KS * 32 ? NW * CW : KS * 32) + 8;
  const size_t lds = ((size_t)NW * 3 * a.EW * ELD + (size_t)2 * a.WP * DLD) * sizeof(bf16_t) +
                     (WLDS ? (size_t)9 * NW * CW * sizeof(float) : 0) + (size_t)a.EW * ELD * sizeof(bf16_t);
  // two workgroups per CU stay resident (launch bounds) and walk the observations
  int wgy = (2 * device_cu_count() + bands * kc - 1) / (bands * kc);
  if (wgy > a.B) wgy = a.B;
  if (wgy < 1) wgy = 1;
  note_kernel(dim3(bands, wgy, kc), dim3(NW * 64), "irb_rows_bf16_kernel<%d,%d,%s,%d,%d,%d,%s,%s,%s,%d,%d>", STRIDE, R,
              EXPAND ? "true" : "false", NW, TPW, NPT, WINDOW ? "true" : "false", APREG ? "true" : "false",
              WLDS ? "true" : "false", CW, KS);
  hipLaunchKernelGGL((irb_rows_bf16_kernel<STRIDE, R, EXPAND, NW, TPW, NPT, WINDOW, APREG, WLDS, CW, KS>),
                     dim3(bands, wgy, kc), dim3(NW * 64), lds, s, a);
  return hipGetLastError();
}

}  // namespace

bool irb_bf16_supported(const Layer* le, const Layer& ld, const Layer& lp) {
  const int hid = ld.cout, nw = (hid + 31) / 32;
  if (ld.h_in > 64 || ld.h_in < 13) return false;  // large-image stages only (features.1 .. features.7)
  if (le == nullptr) return false;  // features.1 (t = 1, one wave per workgroup): the layer-wise pair is faster
  if (le->cin > 32) return false;
  (void)nw;
  if (hid == 96) return ld.stride == 2 && ld.h_out <= 27 && lp.cout <= 32;
  if (hid == 144) return lp.cout <= 32 && ((ld.stride == 1 && ld.h_out <= 27) || (ld.stride == 2 && ld.h_out <= 16));
  if (hid == 192) return ld.h_out <= 16 && ((ld.stride == 1 && lp.cout <= 32) || (ld.stride == 2 && ld.h_out <= 8 && lp.cout <= 64));
  return false;
}

hipError_t launch_irb_bf16(const Layer* le, const Layer& ld, const Layer& lp, const float* enc_w,
                           const unsigned short* enc_wh, size_t model_stride, int k0, int kc, int B,
                           const unsigned short* x, unsigned short* y, hipStream_t s) {
  IrbArgs a;
  a.x = x;
  a.y = y;
  a.wbase = enc_w;
  a.whbase = enc_wh;
  a.model_stride = model_stride;
  a.k0 = k0;
  a.we_off = le ? le->w_off : 0;
  a.be_off = le ? le->b_off : 0;
  a.wd_off = ld.w_off;
  a.bd_off = ld.b_off;
  a.wp_off = lp.w_off;
  a.bp_off = lp.b_off;
  a.B = B;
  a.CIN = le ? le->cin : ld.cin;
  a.HID = ld.cout;
  a.COUT = lp.cout;
  a.H_in = ld.h_in;
  a.H_out = ld.h_out;
  a.residual = lp.residual;
  // 48 hidden channels per wave (HID = 96 / 144 / 192 -> 2 / 3 / 4 waves): 6 depthwise lanes per run, so a wave
  // covers at most 10 runs: R = 3 outputs per lane on the 25-wide stages, 2 on the 13-wide, 1 on the 7-wide
  // (features.2 keeps 32-channel chunks / 3 waves / R = 2: with two 48-channel waves only 4 waves fit a CU next to
  // its 64 KB of LDS rings, 184 vs 152 us)
  const int R = le == nullptr ? 4 : (a.HID == 96 ? 2 : (ld.h_out > 16 ? 3 : (ld.h_out > 8 ? 2 : 1)));
  const int runs = (a.H_out + R - 1) / R;
  const int ew_taps = (runs * R - 1) * ld.stride + 3;
  a.EW = ew_taps > a.H_in + 2 ? ew_taps : a.H_in + 2;
  const int ew_tiles = 16 * ((a.H_in + 15) / 16) + 1;  // the expand writes whole 16-pixel tiles (zeros past the row)
  if (a.EW < ew_tiles) a.EW = ew_tiles;
  a.WP = (a.H_out + 15) & ~15;
  // bands re-expand their halo rows, so keep them >= 6 rows (a handful of observations: a latency chain, short bands)
  const int min_rows = (long)B * kc <= 16 ? 2 : 6;
  int bands = pick_row_bands((long)B * kc, a.H_out, min_rows, 2 * device_cu_count());
  a.band_rows = (a.H_out + bands - 1) / bands;
  bands = (a.H_out + a.band_rows - 1) / a.band_rows;
  if (le == nullptr) return launch_irb<1, 4, false, 1, 4, 4, false, false, true, 32, 1>(a, kc, bands, s);
  //                     STRIDE R EXPAND NW TPW NPT WINDOW APREG WLDS CW KS
  if (a.HID == 96) return launch_irb<2, 2, true, 3, 2, 4, false, false, true, 32, 3>(a, kc, bands, s);    // features.2
  if (a.HID == 144 && ld.stride == 1)
    return launch_irb<1, 3, true, 3, 2, 2, true, false, false, 48, 5>(a, kc, bands, s);                    // features.3
  if (a.HID == 144) return launch_irb<2, 2, true, 3, 1, 2, false, true, false, 48, 5>(a, kc, bands, s);   // features.4
  if (a.HID == 192 && ld.stride == 1)
    return launch_irb<1, 2, true, 4, 1, 1, true, true, false, 48, 6>(a, kc, bands, s);                     // features.5, 6
  if (a.HID == 192) return launch_irb<2, 1, true, 4, 1, 1, false, true, false, 48, 6>(a, kc, bands, s);   // features.7
  return hipErrorInvalidValue;
}

}  // namespace rip
