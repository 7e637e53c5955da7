// Fused bf16 inverted-residual block for the large-image stages (features.2 .. features.7) on gfx950, round 4:
// expand 1x1 -> depthwise 3x3 -> project 1x1 (+ residual) in one row-streaming kernel with ALL THREE convolutions on
// the matrix cores.
//
// torchvision v0.6.0 `InvertedResidual` (reference call site oatomobile/torch/networks/perception.py:36-51), BN folded.
// Round 3's kernel (encoder_bf16_irb.hip, retired in round 6) ran the depthwise on the vector unit: per output row and wave 130 bf16 -> fp32
// unpacks + 72 packed FMAs + window bookkeeping against 16 MFMAs — the block was bound by the VALU stream and its
// dependent latencies while the matrix pipe idled at 5 %.  Here the depthwise is a block-diagonal contraction:
//     out[px][c] = sum_tap  x[px + tap][c] * w[tap][c]
//   = sum over K = (tap, channel) of  A[c][(tap, c')] * B[(tap, c')][px],   A = w[tap][c] where c' == c, else 0.
// A 16-channel group and a pair of taps is one K block of 32: lane (n, q) of the B operand holds 8 consecutive channels
// (half q & 1 of the group) of tap `2 kb + (q >> 1)` at pixel n — which is ONE 16-byte read of the NHWC ring in LDS, no
// unpacking, no window registers; the A operand of lane (m, q) holds the single tap weight of channel m.  15/16 of the
// multiplies are structural zeros: the matrix pipe has the room (the useful rate, 512/16 MAC per cycle and SIMD, equals
// the fp32 vector rate) and the vector unit is left with the epilogues (ReLU6, bf16 packing).
// The depthwise taps stay fp32-grade (the bf16 oracle's definition, oracle/bf16_encoder.py): each tap is carried as
// two bf16 terms w = hi + lo (16 significant bits), K blocks [t0h t1h] .. [t6h t7h] [t8h t8l] [t0l t1l] .. [t6l t7l] —
// nine MFMAs per (16 pixels x 16 channels), the lo blocks re-use the hi blocks' B operands; the depthwise bias is the
// first MFMA's C operand (fp32).  The expansion's bias rides in the unused K columns of its single K block (C_in = 16 /
// 24: columns C_in, C_in + 1 hold bias hi / lo against B = 1.0 for valid pixels, so pixels beyond the row come out as
// ReLU6(0) = 0 with no select); C_in = 32 spends a second, nearly empty MFMA on it.
//
// Decomposition: a workgroup owns (model, observation, band of output rows), wave w owns NG 16-channel groups of the
// hidden dimension with a private 3-row ring in LDS (no barrier between expansion and depthwise); the depthwise output
// row goes to a shared, double-buffered projection operand row; after ONE barrier per output row the waves share the
// projection tiles.  Rows off the image expand from empty descriptors to zero ring rows: the row loop is branch-free.
#include <stdio.h>
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

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 u) { return __builtin_bit_cast(bf16x8, u); }
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ float relu6(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 6.f); }
__device__ __forceinline__ f32x2 bfpair(unsigned u) { return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
__device__ __forceinline__ unsigned bf16_rne(float f) {
  const unsigned u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// (hi, lo) bf16 terms of an fp32 value, hi in the low half: value ~= hi + lo to 16 significant bits
__device__ __forceinline__ unsigned split_bf16(float f) {
  const unsigned h = bf16_rne(f);
  const unsigned l = bf16_rne(f - __uint_as_float(h << 16));
  return h | (l << 16);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_srd(const bf16_t* row, int bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(row);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
#ifdef RIP_IRB2_TICKS  // development: per-phase shader cycles of the row loop, summed over the waves of a launch
__device__ unsigned long long g_irb2_ticks[8];
#define IRB2_TICK(slot_)                                            \
  do {                                                              \
    const unsigned long long now_ = __builtin_readcyclecounter();   \
    tk[slot_] += now_ - tlast;                                      \
    tlast = now_;                                                   \
  } while (0)
#else
#define IRB2_TICK(slot_) do { } while (0)
#endif
// Round 6: the bf16 encoder's depthwise taps are bf16 VALUES (rip_abi.hip: the blob these kernels read has them rounded;
// oracle/bf16_encoder.py dw_weights_bf16 = True) — "activations + weights bf16, fp32 accumulate", BASELINE configs[2].  The
// low terms of the taps are then exactly zero and their four K blocks per tile are not issued: five MFMAs per (16 pixels x
// 16 channels) instead of nine.  The oracle says what it costs: z against the fp32 encoder moves by mean 0.026 / 0.020
// instead of 0.026 / 0.015 (two models, six observations; max 0.17 / 0.19 instead of 0.15 / 0.18) — inside the storage
// format's own noise.  1 = rounds 4-5's 16-bit (hi + lo) taps, for a blob that still carries them.
#ifndef RIP_DW_TAPS_LO
#define RIP_DW_TAPS_LO 0
#endif
constexpr bool DW_TAPS_LO = RIP_DW_TAPS_LO != 0;
constexpr int OOB = 0x40000000;  // byte offset beyond any row descriptor: loads return 0, stores are dropped
constexpr unsigned ONES = 0x3F803F80u;  // two bf16 1.0

struct Irb2Args {
  const bf16_t* x;       // [K][B][H_IN][H_IN][CIN]
  bf16_t* y;             // [K][B][H_OUT][H_OUT][COUT]
  const float* wbase;    // fp32 folded blobs (biases, depthwise taps)
  const bf16_t* whbase;  // bf16 copy of the blobs (pointwise weights), same offsets
  size_t model_stride;
  int k0;
  size_t we_off, be_off, wd_off, bd_off, wp_off, bp_off;
  int B;
  int band_rows;
};

template <int S, int CIN, int HID, int COUT, int H_IN, int H_OUT, int NW, int NG>
struct Irb2Shape {
  static constexpr int NGT = HID / 16;                  // 16-channel groups of the hidden dimension
  static constexpr int NPTI = (H_IN + 15) / 16;         // 16-pixel tiles of an input row
  static constexpr int NPTO = (H_OUT + 15) / 16;        // ... of an output row
  static constexpr int KBE = CIN / 8;                   // lane blocks (q) of the expansion's K block that hold weights
  static constexpr int KE = CIN < 32 ? 1 : 2;           // MFMAs per expansion tile (the second one carries the bias)
  static constexpr int KS = (HID + 31) / 32;            // K steps of the projection
  static constexpr int ELD = 16 * NG + 8;               // bf16 elements per ring pixel slot (odd multiple of 16 bytes)
  static constexpr int DLD = 32 * KS + 8;               // bf16 elements per projection-operand pixel row
  static constexpr int EW_A = 16 * NPTI + 1, EW_B = S * (16 * NPTO - 1) + 3;
  static constexpr int EW = EW_A > EW_B ? EW_A : EW_B;  // pixel slots per ring row (slot = ix + 1; slot 0 = left border)
  static constexpr int NCT = (COUT + 15) / 16;
  static constexpr int TT = NPTO * NCT;                 // projection tiles per output row
  static constexpr int TPW = (TT + NW - 1) / NW;
  static constexpr size_t RING_EL = (size_t)NW * 3 * EW * ELD;
  static constexpr size_t DS_EL = (size_t)2 * 16 * NPTO * DLD;
  static constexpr size_t LDS_BYTES = (RING_EL + DS_EL) * 2;
  static_assert(HID % 16 == 0 && CIN % 8 == 0 && COUT % 4 == 0, "channel counts");
  static_assert(NW * NG == NGT, "every wave owns exactly NG hidden groups (a branch-free row loop)");
  static_assert(H_OUT == (H_IN + 2 - 3) / S + 1, "3x3, padding 1");
};

template <int S, int CIN, int HID, int COUT, int H_IN, int H_OUT, int NW, int NG, bool RES, int OCC, int BTD>
__global__ __launch_bounds__(NW * 64, OCC) void irb2_bf16_kernel(Irb2Args a) {
  using SH = Irb2Shape<S, CIN, HID, COUT, H_IN, H_OUT, NW, NG>;
  constexpr int NGT = SH::NGT, NPTI = SH::NPTI, NPTO = SH::NPTO, KBE = SH::KBE, KE = SH::KE, KS = SH::KS, ELD = SH::ELD,
                DLD = SH::DLD, EW = SH::EW, NCT = SH::NCT, TT = SH::TT, TPW = SH::TPW;
  constexpr bool APREG = TPW * KS <= 12;  // projection weights stay in registers for the band
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem_raw);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int k = blockIdx.z, band = blockIdx.x;
  bf16_t* es = lds + (size_t)w * 3 * EW * ELD;   // this wave's ring: [3][EW][ELD]
  bf16_t* ds = lds + SH::RING_EL;                // [2][16 * NPTO][DLD]
  const float* W = a.wbase + (size_t)(a.k0 + k) * a.model_stride;
  const bf16_t* Wh = a.whbase + (size_t)(a.k0 + k) * a.model_stride;
  // persistent over observations: the wave constants below (expansion / depthwise / projection operands, ~40 loads and
  // the LDS zeroing) are built once per workgroup, which then walks observations blockIdx.y, + gridDim.y, ...
  const bf16_t* const xin0 = a.x + (size_t)k * a.B * H_IN * H_IN * CIN;
  bf16_t* const yout0 = a.y + (size_t)k * a.B * H_OUT * H_OUT * COUT;
  const bf16_t* xin = xin0;  // this item's observation (set in the item loop; the lambdas below read it by reference)
  bf16_t* yout = yout0;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const int g0 = w * NG;  // first hidden group of this wave

  // ---- zero the LDS once: ring borders, the K padding of the projection operand rows ----
  for (int e = threadIdx.x; e < (int)(SH::LDS_BYTES / 16); e += NW * 64) reinterpret_cast<u32x4*>(lds)[e] = zero4;

  // ---- per-wave constants ----
  u32x4 ae[NG][KE];   // expansion A operands: 16 hidden channels x (C_in weights | bias hi, lo)
  u32x4 ad[NG][9];    // depthwise A operands: one tap weight of channel m per lane and K block
  f32x4 bdw[NG];      // depthwise bias of channels 4q .. 4q+3 of the group (fp32, the first MFMA's C operand)
#pragma unroll
  for (int gl = 0; gl < NG; ++gl) {
    constexpr bool gv = true;
    const int h = 16 * (g0 + gl) + n;  // the A row of this lane
    {
      // every load of this prologue is unconditional (clamped address, then a select): behind a branch each one is
      // waited for on its own — the 30-odd constants of a wave were that many memory round trips IN SEQUENCE per
      // workgroup (`s_waitcnt vmcnt(0)` at every merge), a quarter of the kernel
      const u32x4 wv = *reinterpret_cast<const u32x4*>(Wh + a.we_off + (size_t)h * CIN + 8 * (q < KBE ? q : KBE - 1));
      const unsigned bsplit = split_bf16(W[a.be_off + h]);
      u32x4 v = q < KBE ? wv : zero4;
      if (KE == 1) v.x = q == KBE ? bsplit : v.x;
      ae[gl][0] = v;
      if (KE == 2) {
        u32x4 v2 = zero4;
        v2.x = q == 0 ? bsplit : 0u;
        ae[gl][KE - 1] = v2;
      }
    }
    const bool diag = gv && (q & 1) == (n >> 3);  // this lane's 8 channels contain channel m = n
    const int pos = n & 7;
#pragma unroll
    for (int kb = 0; kb < 9; ++kb) {
      // K block kb: taps (2kb, 2kb+1) hi | (8 hi, 8 lo) | taps (2(kb-5), 2(kb-5)+1) lo; lanes q >> 1 pick the tap
      const int half = q >> 1;
      const int tap = kb < 4 ? 2 * kb + half : (kb == 4 ? 8 : 2 * (kb - 5) + half);
      const bool lo = kb > 4 || (kb == 4 && half == 1);
      const unsigned s = split_bf16(W[a.wd_off + (size_t)tap * HID + h]);
      const unsigned v16 = diag ? (lo ? (s >> 16) : (s & 0xffffu)) : 0u;
      const unsigned dw = v16 << ((pos & 1) * 16);
      u32x4 op;
      op.x = (pos >> 1) == 0 ? dw : 0u;
      op.y = (pos >> 1) == 1 ? dw : 0u;
      op.z = (pos >> 1) == 2 ? dw : 0u;
      op.w = (pos >> 1) == 3 ? dw : 0u;
      ad[gl][kb] = op;
    }
    const float4 bb = gv ? *reinterpret_cast<const float4*>(W + a.bd_off + 16 * (g0 + gl) + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    bdw[gl] = f32x4{bb.x, bb.y, bb.z, bb.w};
  }
  // the "ones" of the expansion's bias columns: valid pixels only (a pixel beyond the row expands to ReLU6(0) = 0)
  unsigned onesx[NPTI];
#pragma unroll
  for (int i = 0; i < NPTI; ++i) onesx[i] = (16 * i + n < H_IN && q == (KE == 1 ? KBE : 0)) ? ONES : 0u;

  // projection tiles of this wave: tile = w + NW * t -> (pixel tile pt, channel tile ct)
  u32x4 ap[TPW][KS];
  float4 bpj[TPW];
  auto load_ap = [&]() {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = w + NW * t;
      const int ct = tile % NCT;
      const int co = 16 * ct + n;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kk = 32 * ks + 8 * q;
        const u32x4 wv = *reinterpret_cast<const u32x4*>(Wh + a.wp_off + (size_t)(co < COUT ? co : COUT - 1) * HID + (kk < HID ? kk : HID - 8));
        ap[t][ks] = (tile < TT && co < COUT && kk < HID) ? wv : zero4;
      }
      const int cb = 16 * ct + 4 * q;
      const float4 bv4 = *reinterpret_cast<const float4*>(W + a.bp_off + (cb < COUT ? cb : COUT - 4));
      bpj[t] = (tile < TT && cb < COUT) ? bv4 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (APREG) load_ap();
  int yoff[TPW], roff[TPW];  // output / residual byte offsets inside a row (out of bounds for padding lanes)
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = w + NW * t;
    const int pt = tile / NCT, ct = tile - pt * NCT;
    const int px = 16 * pt + n, co = 16 * ct + 4 * q;
    const bool ok = tile < TT && px < H_OUT && co < COUT;
    yoff[t] = ok ? (px * COUT + co) * 2 : OOB;
    roff[t] = ok ? (px * CIN + co) * 2 : OOB;
  }

  // ---- block-input row operands, fetched one output row ahead (per-row buffer descriptors; lane offsets that are out
  // of bounds for pixels beyond the row and for the K padding: those lanes load zeros)
  int xoff[NPTI];
#pragma unroll
  for (int i = 0; i < NPTI; ++i) {
    const int px = 16 * i + n;
    xoff[i] = (px < H_IN && 8 * q < CIN) ? (px * CIN + 8 * q) * 2 : OOB;
  }
  constexpr int X_ROW_BYTES = H_IN * CIN * 2;
  auto load_x = [&](int iy, u32x4(&xr)[NPTI]) {
    const bool rok = iy >= 0 && iy < H_IN;
    const __amdgpu_buffer_rsrc_t srd = row_srd(xin + (size_t)(rok ? iy : 0) * H_IN * CIN, rok ? X_ROW_BYTES : 0);
#pragma unroll
    for (int i = 0; i < NPTI; ++i) xr[i] = __builtin_amdgcn_raw_buffer_load_b128(srd, xoff[i], 0, 0);
  };
  const int lane_ring = (n * ELD + 4 * q) * 2;  // byte offset of this lane's 4 channels of pixel n inside a ring row
  // Round 5, measured and not kept (tools/dev ticks, -DRIP_IRB2_TICKS: per row expand 0.7-1.7 k, depthwise 1.0-2.1 k,
  // barrier 0.5-0.7 k, projection 0.3-0.8 k cycles; per observation a prologue of 8-11 k cycles = 8-15 % of the kernel):
  // letting the last two rows of an observation request the first rows of the workgroup's NEXT observation (instead of
  // rows off the image) removes the prologue's own requests and changes neither the prologue's time nor the kernels'
  // (155 / 176 / 70 -> 157 / 177 / 73 us): what the prologue waits for is the memory latency of first-touch rows under
  // load, which two rows of lead do not cover either.
  // The row loop holds NO branch between a vector-memory load and its use: the s_waitcnt insertion falls back to
  // vmcnt(0) behind every control-flow merge, which waits for the operand prefetches of the NEXT rows as well (measured:
  // the "expand" phase of the stride-2 blocks and the residual add each cost a full memory latency per row that way).
  // A row off the image is therefore expanded like any other: its descriptor is empty (operands load as zeros) and the
  // bias columns' ones are masked off, so the ring row comes out as ReLU6(0) = 0.
  auto expand_row = [&](int iy, const u32x4(&xr)[NPTI]) __attribute__((always_inline)) {
    unsigned char* ring = reinterpret_cast<unsigned char*>(es + (size_t)((iy + 3) % 3) * EW * ELD);
    const unsigned rowmask = (iy >= 0 && iy < H_IN) ? 0xffffffffu : 0u;  // wave-uniform
#pragma unroll
    for (int i = 0; i < NPTI; ++i) {
      u32x4 bx = xr[i];
      u32x4 b2 = zero4;
      if (KE == 1)
        bx.x |= onesx[i] & rowmask;  // lanes of the bias block loaded zeros
      else
        b2.x = onesx[i] & rowmask;
#pragma unroll
      for (int gl = 0; gl < NG; ++gl) {
        f32x4 c = mfma_bf16(ae[gl][0], bx, f32x4{0.f, 0.f, 0.f, 0.f});
        if (KE == 2) c = mfma_bf16(ae[gl][KE - 1], b2, c);
        u32x2 o;
        o.x = pack_bf16(relu6(c[0]), relu6(c[1]));
        o.y = pack_bf16(relu6(c[2]), relu6(c[3]));
        *reinterpret_cast<u32x2*>(ring + lane_ring + ((16 * i + 1) * ELD + 16 * gl) * 2) = o;
      }
    }
  };

  const int oy0 = band * a.band_rows, oy1 = min(H_OUT, oy0 + a.band_rows);
  __syncthreads();  // LDS zeroed
  u32x4 xa[S][NPTI], xb[S][NPTI];

  // depthwise B operand of lane (n, q): pixel slot S*n (+ kx), channel half q & 1; the tap of the pair by q >> 1
  const int lane_dw = (int)(reinterpret_cast<unsigned char*>(es) - smem_raw) + (S * n * ELD + 8 * (q & 1)) * 2;
  const bool second = q >= 2;
#ifdef RIP_IRB2_TICKS
  unsigned long long tk[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
  auto row = [&](int oy, u32x4(&xr)[S][NPTI], int buf) __attribute__((always_inline)) {
    IRB2_TICK(5);
    // 0. the residual (= block input row oy, this wave's projection tiles) is requested now and added after the projection
    const __amdgpu_buffer_rsrc_t rsrd = row_srd(xin + (size_t)oy * H_IN * CIN, RES ? X_ROW_BYTES : 0);
    u32x2 rres[TPW];
    if (RES) {
#pragma unroll
      for (int t = 0; t < TPW; ++t) rres[t] = __builtin_amdgcn_raw_buffer_load_b64(rsrd, roff[t], 0, 0);
    }
    // 1. expand the S new rows (operands fetched two rows ago), then request the rows of output row oy + 2
#pragma unroll
    for (int i = 0; i < S; ++i) expand_row(oy * S + 2 - S + i, xr[i]);
#pragma unroll
    for (int i = 0; i < S; ++i) load_x((oy + 2) * S + 2 - S + i, xr[i]);  // (beyond the image: empty descriptor)
    IRB2_TICK(0);
    // 2. depthwise of this wave's groups on the matrix cores
    unsigned char* drow = reinterpret_cast<unsigned char*>(ds + (size_t)buf * 16 * NPTO * DLD);
    {
      // byte offset of tap t's ring row / column (wave-uniform); lanes q >= 2 read the second tap of a pair
      auto tap_off = [&](int t) { return (((oy * S - 1 + t / 3 + 3) % 3) * EW * ELD + (t % 3) * ELD) * 2; };
      int addr[5];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ta = tap_off(2 * j), d = tap_off(2 * j + 1) - ta;  // (an indexed T[2j + second] goes through scratch)
        addr[j] = lane_dw + ta + (second ? d : 0);
      }
      addr[4] = lane_dw + tap_off(8);
      // (pixel tile, group) items one after the other.  An item's nine MFMAs accumulate into ONE register tuple and
      // issue back to back; the operand reads of item i + 1 are issued BEFORE item i's chain, and the chain starts with
      // the K block whose operand was read last (LDS returns in order: one s_waitcnt in front of the chain covers all
      // five, nothing sits between two dependent MFMAs).  Round 5, measured and not kept: two items interleaved
      // instruction by instruction with both operand sets read up front (the nine MFMAs of an item are a dependent chain,
      // ~36 cycles each at a 16-cycle issue rate; per-row counters: the depthwise is 29-48 % of a row): features.2
      // 155 -> 159 us, features.4 70 -> 71, features.3 (the second operand set spills) 176 -> 197 — as round 4 found at
      // an earlier stage of the kernel, the chain latency is not what a row waits for.
      constexpr int NIT = NPTO * NG;
      u32x4 bt[BTD][5];  // BTD = 2: item i + 1's operands in flight during item i's chain (register-tight shapes: 1)
      auto issue_reads = [&](int item, u32x4(&dst)[5]) __attribute__((always_inline)) {
        const int imm = (16 * (item / NG) * S * ELD + 16 * (item % NG)) * 2;
#pragma unroll
        for (int j = 0; j < 5; ++j) dst[j] = *reinterpret_cast<const u32x4*>(smem_raw + addr[j] + imm);
      };
      if (BTD == 2) issue_reads(0, bt[0]);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int pt = it / NG, gl = it % NG;
        if (BTD == 2) {
          if (it + 1 < NIT) issue_reads(it + 1, bt[(it + 1) % BTD]);
        } else {
          issue_reads(it, bt[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 c = mfma_bf16(ad[gl][4], bt[it % BTD][4], bdw[gl]);  // (t8h t8l): the operand read last
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c = mfma_bf16(ad[gl][j], bt[it % BTD][j], c);
          if (DW_TAPS_LO) c = mfma_bf16(ad[gl][5 + j], bt[it % BTD][j], c);
        }
        __builtin_amdgcn_sched_barrier(0);
        u32x2 o;
        o.x = pack_bf16(relu6(c[0]), relu6(c[1]));
        o.y = pack_bf16(relu6(c[2]), relu6(c[3]));
        *reinterpret_cast<u32x2*>(drow + ((16 * pt + n) * DLD + 16 * (g0 + gl) + 4 * q) * 2) = o;
      }
    }
    if (!APREG) load_ap();
    IRB2_TICK(1);
    lds_barrier();  // every group of ds[buf] is in place (LDS-only fence: the operand prefetches stay in flight)
    IRB2_TICK(2);
    // 3. projection tiles of this wave over the full hidden K
    const __amdgpu_buffer_rsrc_t ysrd = row_srd(yout + (size_t)oy * H_OUT * COUT, H_OUT * COUT * 2);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = w + NW * t;
      if (NW * t + NW - 1 >= TT && tile >= TT) continue;  // compile-time except for the last t of some waves
      const int pt = tile / NCT;
      f32x4 c = {bpj[t].x, bpj[t].y, bpj[t].z, bpj[t].w};
      const unsigned char* brow = drow + ((16 * pt + n) * DLD + 8 * q) * 2;
      u32x4 bp[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bp[ks] = *reinterpret_cast<const u32x4*>(brow + 64 * ks);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = KS - 1; ks >= 0; --ks) c = mfma_bf16(ap[t][ks], bp[ks], c);  // back to back, last-read operand first
      __builtin_amdgcn_sched_barrier(0);
      f32x2 v0 = {c[0], c[1]}, v1 = {c[2], c[3]};
      if (RES) {
        v0 += bfpair(rres[t].x);
        v1 += bfpair(rres[t].y);
      }
      u32x2 o;
      o.x = pack_bf16(v0.x, v0.y);
      o.y = pack_bf16(v1.x, v1.y);
      __builtin_amdgcn_raw_buffer_store_b64(o, ysrd, yoff[t], 0, 0);
    }
    IRB2_TICK(3);
  };
#pragma unroll 1
  for (int b = blockIdx.y; b < a.B; b += gridDim.y) {
    xin = xin0 + (size_t)b * H_IN * H_IN * CIN;
    yout = yout0 + (size_t)b * H_OUT * H_OUT * COUT;
    // prologue: rows oy0*S-1 .. oy0*S+1-S are expanded here, the remaining S rows of the first window in the loop
#pragma unroll
    for (int i = 0; i < 3 - S; ++i) {
      u32x4 x0[NPTI];
      load_x(oy0 * S - 1 + i, x0);
      expand_row(oy0 * S - 1 + i, x0);
    }
    // block-input operands are requested TWO output rows ahead (two register sets, the row loop is unrolled by two):
    // with the depthwise off the vector unit a row takes less time than a load that misses the L2
#pragma unroll
    for (int i = 0; i < S; ++i) load_x(oy0 * S + 2 - S + i, xa[i]);
#pragma unroll
    for (int i = 0; i < S; ++i) load_x((oy0 + 1) * S + 2 - S + i, xb[i]);  // (rows past the band / image: zero-byte descriptors)
#pragma unroll 1
    for (int oy = oy0; oy < oy1; oy += 2) {
      row(oy, xa, 0);
      if (oy + 1 < oy1) row(oy + 1, xb, 1);
    }
    IRB2_TICK(5);
    lds_barrier();  // the last row's projection has read ds before the next observation's depthwise writes it
    IRB2_TICK(4);  // (development: the wait at the end of an observation)
  }
#ifdef RIP_IRB2_TICKS
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) atomicAdd(&g_irb2_ticks[i], tk[i]);
    atomicAdd(&g_irb2_ticks[6], (unsigned long long)(oy1 - oy0) * ((a.B - 1 - blockIdx.y) / gridDim.y + 1));
  }
#endif
}

template <int S, int CIN, int HID, int COUT, int H_IN, int H_OUT, int NW, int NG, bool RES, int OCC, int BTD = 2>
hipError_t launch_irb2(Irb2Args a, int B, int kc, hipStream_t s) {
  using SH = Irb2Shape<S, CIN, HID, COUT, H_IN, H_OUT, NW, NG>;
  // bands re-expand their halo rows, so keep them >= 6 rows (a handful of observations: the launch is a latency chain,
  // not a throughput problem — bands of 2-3 rows, one workgroup each, although every band re-expands its halo row)
  const int min_rows = (long)B * kc <= 16 ? 2 : 6;
  int bands = pick_row_bands((long)B * kc, H_OUT, min_rows, OCC * device_cu_count());
  a.band_rows = (H_OUT + bands - 1) / bands;
  bands = (H_OUT + a.band_rows - 1) / a.band_rows;
  auto kern = irb2_bf16_kernel<S, CIN, HID, COUT, H_IN, H_OUT, NW, NG, RES, OCC, BTD>;
  static bool attr_set[64] = {};  // > 64 KB of dynamic LDS needs the opt-in once per kernel AND device
  int dev = 0;
  if (SH::LDS_BYTES > 64 * 1024 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)SH::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  // OCC workgroups per CU stay resident and walk the observations (the wave constants are built once per workgroup)
  int wgy = (OCC * device_cu_count() + bands * kc - 1) / (bands * kc);
  if (wgy > B) wgy = B;
  if (wgy < 1) wgy = 1;
  note_kernel(dim3(bands, wgy, kc), dim3(NW * 64), "irb2_bf16_kernel<%d,%d,%d,%d,%d,%d,%d,%d,%s,%d,%d>", S, CIN, HID, COUT, H_IN,
              H_OUT, NW, NG, RES ? "true" : "false", OCC, BTD);
  hipLaunchKernelGGL(kern, dim3(bands, wgy, kc), dim3(NW * 64), SH::LDS_BYTES, s, a);
#ifdef RIP_IRB2_TICKS
  {
    unsigned long long t[8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_irb2_ticks), sizeof(t));
    const double rows = t[6] > 0 ? (double)t[6] : 1.0;  // wave-rows
    fprintf(stderr, "irb2<S=%d HID=%d NW=%d NG=%d> cycles per wave-row: expand %.0f  depthwise %.0f  barrier %.0f  project %.0f  | per observation: prologue %.0f  end barrier %.0f\n",
            S, HID, NW, NG, t[0] / rows, t[1] / rows, t[2] / rows, t[3] / rows, t[5] / rows * H_OUT, t[4] / rows * H_OUT);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_irb2_ticks), z, sizeof(z));
  }
#endif
  return hipGetLastError();
}

}  // namespace

// The shapes of torchvision's features.2 .. features.7 behind a 100 x 100 network input.  Rounds 4-5 shipped this kernel
// for features.2-4 only (profiles/r4: 287 -> 172, 230 -> 196, 111 -> 92 us at 512 observations x 4 models) and kept the
// 13x13 / 7x7-output blocks on round 1's row-streaming kernel (encoder_bf16_irb.hip): with nine-MFMA depthwise chains this
// one was 87 / 70 us there against 80 / 55.  Round 6: with bf16-valued taps the chains are five MFMAs and it wins on
// features.5-7 as well (56 / 56 / 39 us against 77 / 78 / 46): `all` is the default, RIP_OPT_ENCODER_VARIANT bit
// ENC_VAR_ROWS_F5_7 restores round 1's kernel there (tests run both).
bool irb2_bf16_supported(const Layer* le, const Layer& ld, const Layer& lp, bool all) {
  if (le == nullptr) return false;
  const int cin = le->cin, hid = ld.cout, cout = lp.cout, s = ld.stride, hi = ld.h_in, ho = ld.h_out;
  auto is = [&](int a, int b, int c, int d, int e, int f) { return cin == a && hid == b && cout == c && s == d && hi == e && ho == f; };
  if (is(16, 96, 24, 2, 50, 25) || is(24, 144, 24, 1, 25, 25) || is(24, 144, 32, 2, 25, 13)) return true;
  return all && (is(32, 192, 32, 1, 13, 13) || is(32, 192, 64, 2, 13, 7));
}

hipError_t launch_irb2_bf16(const Layer* le, const Layer& ld, const Layer& lp, const float* enc_w,
                            const unsigned short* enc_wh, size_t model_stride, int k0, int kc, int B,
                            const unsigned short* x, unsigned short* y, hipStream_t s) {
  if (le == nullptr) return hipErrorInvalidValue;
  Irb2Args a;
  a.x = x;
  a.y = y;
  a.wbase = enc_w;
  a.whbase = enc_wh;
  a.model_stride = model_stride;
  a.k0 = k0;
  a.we_off = le->w_off;
  a.be_off = le->b_off;
  a.wd_off = ld.w_off;
  a.bd_off = ld.b_off;
  a.wp_off = lp.w_off;
  a.bp_off = lp.b_off;
  a.B = B;
  a.band_rows = 0;
  const int hid = ld.cout, st = ld.stride, cout = lp.cout;
  // Waves x groups per wave: every wave owns NG groups (NW * NG = HID / 16) and the register budget is two waves per
  // SIMD (the depthwise A operands alone are 36 registers per group).  More, lighter waves (6 x 1, 9 x 1, 6 x 2 at three
  // to four waves per SIMD) measured slower: they spill, and every wave re-loads the block input and re-expands all
  // pixels for fewer channels (profiles/r4/irb2_variants.txt).  BTD = 1 where two operand sets do not fit.
  //                                   S CIN HID COUT H_IN H_OUT NW NG RES  OCC BTD
  // Round 6 (bf16-valued taps: 20 instead of 36 depthwise A operand registers per group): features.3 takes the second operand
  // set (154 -> 147 us), features.5-7 moved here from round 1's kernel; the other shapes were re-swept and stand
  // (profiles/r6/bf16_taps_v1.txt: 6 x 1 / 2 x 3 on features.2 186 / 177 us against 137; 9 x 1 on features.3 / 4 176 / 84
  // against 147 / 63; 6 x 2, 3 x 4, 12 x 1 on features.5-7 80 / 60 / 75 us against 56).
  if (hid == 96) return launch_irb2<2, 16, 96, 24, 50, 25, 3, 2, false, 2, 2>(a, B, kc, s);                     // features.2
  if (hid == 144 && st == 1) return launch_irb2<1, 24, 144, 24, 25, 25, 3, 3, true, 2, 2>(a, B, kc, s);         // features.3
  if (hid == 144) return launch_irb2<2, 24, 144, 32, 25, 13, 3, 3, false, 2, 1>(a, B, kc, s);                   // features.4
  if (hid == 192 && st == 1) return launch_irb2<1, 32, 192, 32, 13, 13, 4, 3, true, 2, 2>(a, B, kc, s);         // features.5, 6
  if (hid == 192 && cout == 64) return launch_irb2<2, 32, 192, 64, 13, 7, 4, 3, false, 2, 2>(a, B, kc, s);      // features.7
  return hipErrorInvalidValue;
}

}  // namespace rip
