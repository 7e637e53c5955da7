// fp32-GRADE fused inverted-residual block for the small-image stages of the fp32 (parity-mode) encoder on gfx950:
// expand 1x1 -> depthwise 3x3 -> project 1x1 (+ residual) in one kernel, fp32 activations in HBM, the two pointwise
// convolutions on the binary16 matrix pipe with TWO-TERM operands.
//
// torchvision v0.6.0 `InvertedResidual` (reference call site oatomobile/torch/networks/perception.py:36-51), BN
// folded.  Layer by layer (encoder.hip: pw_kernel / dw_kernel, true fp32 MFMA) the blocks features.8-17 are 30 launches,
// 3.9 ms of the 9.1 ms the fp32 encoder takes for 512 observations x 4 models: the 6x expanded tensors make two round
// trips through HBM and the fp32 matrix pipe (v_mfma_f32_16x16x4_f32) is 1/16 of the binary16 one.  Here:
//   * the expanded tensor exists only as 64-channel slices in LDS (the structure of encoder_bf16_tile.hip: a workgroup
//     owns G whole observations of one model and walks the hidden dimension in chunks of 64 channels);
//   * a pointwise product is three v_mfma_f32_16x16x32_f16 (the plan search's scheme, flow_split_dev.h):
//         W x ~= Whi xhi + Wlo xhi + Whi xlo,   fp32 accumulation, the dropped Wlo xlo term is 2^-22 relative.
//     Weights are split on the host as w 2^8 = hi + lo (`pack_split_tiles`; |w| < 240 is checked at load, rip_abi.hip — a model
//     outside keeps the layer-wise kernels): the residual of an ordinary weight is then a normal binary16 and the 2^-8
//     is one exact multiply in the epilogue.  Activations are split in the kernel, hi = f16(x), lo = f16(x - hi): the
//     expanded / depthwise tensors are ReLU6-bounded; the block input must stay below 65504 in magnitude (a
//     BatchNorm-folded MobileNetV2 is orders of magnitude inside; beyond it the result is inf / NaN, not a wrong number).
//     For |x| < 1/8 the low term is a binary16 subnormal, quantised at 2^-24: at most 2^-25 |w| per product, below the
//     fp32 accumulation error of the sum it joins.
//   * the depthwise stays on the vector unit in fp32 (fp32 taps, fp32 E tile in LDS); its ReLU6 output is split into the
//     two binary16 planes the projection reads as B operands.
// Unlike the bf16 kernel the waves are NOT specialised: with fp32 tiles LDS holds one E and one D buffer, so a step is
// expand -> barrier -> depthwise -> barrier -> project with all eight waves in every phase (two waves per SIMD fill
// each other's operand waits).
// Contract: the fp32 oracle at 1e-4 on z (tests/test_gpu_parity.py, the fp32 encoder gates); NOT bit-identical to the
// layer-wise fp32 kernels (22 significant bits per operand instead of 24, another summation order).
#include <cstdio>
#include <cstring>

#include "encoder.h"
#include "flow.h"  // device_cu_count

namespace rip {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using h16x8 = __attribute__((ext_vector_type(8))) _Float16;
using h16x2 = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
typedef unsigned short h16_t;

__device__ __forceinline__ h16x8 as_h8(u32x4 u) { return __builtin_bit_cast(h16x8, u); }
__device__ __forceinline__ f32x4 mfmah(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(a), as_h8(b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x2 relu6_2(f32x2 v) {
  return __builtin_elementwise_min(__builtin_elementwise_max(v, f32x2{0.f, 0.f}), f32x2{6.f, 6.f});
}
// two fp32 -> (hi, lo) packed binary16 pairs, lo = f16(x - hi)
__device__ __forceinline__ u32x2 split2(f32x2 x) {  // .x = hi pair, .y = lo pair
  const h16x2 h = __builtin_convertvector(x, h16x2);
  const f32x2 back = __builtin_convertvector(h, f32x2);
  const h16x2 l = __builtin_convertvector(x - back, h16x2);
  return u32x2{__builtin_bit_cast(unsigned, h), __builtin_bit_cast(unsigned, l)};
}

#ifdef RIP_SPLIT_TICKS  // development (tools/dev/split_ticks.sh): shader cycles per phase of wave 0, summed over the workgroups
__device__ unsigned long long g_split_ticks[16];
#define SPLIT_TICK(slot_)                                           \
  do {                                                              \
    const unsigned long long now_ = __builtin_readcyclecounter();   \
    tk[slot_] += now_ - tlast;                                      \
    tlast = now_;                                                   \
  } while (0)
#else
#define SPLIT_TICK(slot_) do { } while (0)
#endif

constexpr int HC = 64;        // hidden channels per chunk
constexpr int LDE = HC + 4;   // fp32 elements per E pixel row (272 B: an odd multiple of 16 bytes)
constexpr int LDD = HC + 8;   // binary16 elements per D pixel row (144 B)
constexpr float W_INV = 1.0f / 256.0f;  // the split weight planes carry w * 2^8 (SPLIT_ENC_W_SCALE in encoder.h)

struct SplitTileArgs {
  const float* x;       // [K][B][HIN][HIN][CIN] fp32
  float* y;             // [K][B][HOUT][HOUT][COUT] fp32
  const float* wbase;   // fp32 folded blobs (biases, depthwise taps)
  const h16_t* wc;      // this block's chunk records of model 0 (pack_split_tiles), models wc_stride binary16 elements apart
  size_t model_stride, wc_stride;
  int k0;
  size_t we_off, be_off, wd_off, bd_off, wp_off, bp_off;
  int B, HID, residual, G;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int HIN, int STRIDE, int CIN, int COUT, int G>
struct SplitGeom {
  static constexpr int HOUT = STRIDE == 1 ? HIN : (HIN + 1) / 2;
  static constexpr int HWI = HIN * HIN, HWO = HOUT * HOUT;
  static constexpr int PW = HIN + 2;                           // padded row: zero, HIN pixels, zero
  static constexpr int E_DUMP = G * HIN * PW;                  // row that takes the stores of lanes without a pixel
  static constexpr int E_ROWS = E_DUMP + 1;
  static constexpr int D_ROWS = ((G * HWO + 15) / 16) * 16;
  static constexpr size_t E_BYTES = (size_t)E_ROWS * LDE * sizeof(float);
  static constexpr size_t D_PLANE = (size_t)D_ROWS * LDD;      // binary16 elements per plane
  static constexpr size_t D_BYTES = 2 * D_PLANE * sizeof(h16_t);
  static constexpr size_t TP_BYTES = (size_t)2 * 3 * 1024;     // two buffers of a chunk's taps [9][64] + biases [64] (fp32): 160 lanes x 16 B, three DMA rows
  static constexpr int NFE = (HC / 16) * (CIN / 32) * 2;       // 1 KB operand fragments of a chunk's expansion weights (channel tile, K block, term)
  static constexpr int NFP = (COUT / 16) * (HC / 32) * 2;      // ... of its projection weights
  static constexpr size_t LDS_BYTES = E_BYTES + D_BYTES + TP_BYTES + (size_t)(NFE + NFP) * 1024;
};

// G: observations per workgroup at most (G * HOUT * 16 <= 512 depthwise threads: one per (observation, output column, 4 channels)).
// WCH: the eight waves split as (8 / WCH pixel partitions) x (WCH channel partitions) in both matrix phases.
//
// WEIGHTS GO THROUGH LDS.  Version 1 read a phase's A operands from global memory into registers, every wave its own:
// 8 waves x 256 workgroups asking the same few L2 lines for 64-byte pieces — ~4 k requests per L2 channel and phase, 7 k
// + 3.3 k cycles of a 13.6 k cycle step waiting for them however early they were requested (tools/dev/split_ticks.sh,
// profiles/r6/split_tile_v1.txt).  Version 2 copied a chunk's weights once per workgroup into LDS as operand fragments
// (`__builtin_amdgcn_global_load_lds`, gathered from the weight planes by the waves without depthwise work — the
// compiler puts `s_waitcnt vmcnt(0)` in front of the first LDS read that follows such a copy in program order, so a wave
// that went on to the depthwise waited right there): 1626 -> 1259 us for the ten blocks, the steps still waiting 1.2-6.8 k
// cycles for copies that had only the depthwise phase to land.  Version 3 (this one):
//   * the host packs a chunk's operands as ONE contiguous record in the order they sit in LDS (`pack_split_tiles`:
//     expansion fragments, projection fragments, 3 KB of taps / depthwise biases / expansion biases): a copy instruction
//     moves 1 KB of consecutive bytes (8 full lines instead of 16 half lines);
//   * the copies are issued from INLINE ASSEMBLY (the compiler's counter bookkeeping does not see them: no wait it did
//     not mean; its own global loads only appear outside the chunk loop) by all eight waves, each the same number of
//     instructions, and waited for with counted `s_waitcnt vmcnt(n)`:
//       expand(c) | B1 | issue WE(c+1), TP(c+1) | depthwise(c) | wait WP(c) | B2 | project(c) | wait WE(c+1), TP(c+1) | B3 | issue WP(c+1)
//     so every copy has two phases to land (the third barrier per step is what that costs).
template <int HIN, int STRIDE, int CIN, int COUT, int G, int WCH>
__global__ __launch_bounds__(512) void irb_split_tile_kernel(SplitTileArgs a) {
  using Geo = SplitGeom<HIN, STRIDE, CIN, COUT, G>;
  constexpr int HOUT = Geo::HOUT, HWI = Geo::HWI, HWO = Geo::HWO, PW = Geo::PW;
  constexpr int WP = 8 / WCH;
  constexpr int NPT_IN = (G * HWI + 15) / 16, NPT_OUT = (G * HWO + 15) / 16;  // 16-pixel tiles of a full group
  constexpr int TIN = (NPT_IN + WP - 1) / WP, TOUT = (NPT_OUT + WP - 1) / WP;
  constexpr int KSX = CIN / 32, NHT = HC / 16 / WCH, NCT = COUT / 16 / WCH, NKP = HC / 32;
  constexpr int NFE = Geo::NFE, NFP = Geo::NFP;
  static_assert(CIN % 32 == 0 && (COUT / 16) % WCH == 0 && (HC / 16) % WCH == 0, "partitions");
  static_assert(G * HOUT * 16 <= 512, "depthwise threads");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* const E = reinterpret_cast<float*>(smem_raw);                                 // [E_ROWS][LDE]
  h16_t* const Dh = reinterpret_cast<h16_t*>(smem_raw + Geo::E_BYTES);                 // [D_ROWS][LDD] hi
  h16_t* const Dl = Dh + Geo::D_PLANE;                                                 // lo
  float* const TP = reinterpret_cast<float*>(smem_raw + Geo::E_BYTES + Geo::D_BYTES);  // [2][768]: taps [9][64] + biases [64] of a chunk
  u32x4* const WE = reinterpret_cast<u32x4*>(smem_raw + Geo::E_BYTES + Geo::D_BYTES + Geo::TP_BYTES);  // [NFE][64]
  u32x4* const WPj = WE + (size_t)NFE * 64;                                             // [NFP][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int wpix = w / WCH, wch = w % WCH;
  const int ht0 = wch * NHT, ct0w = wch * NCT;
  const int k = blockIdx.z;
  const int HID = a.HID;
  const int nch = HID / HC;
  const float* W = a.wbase + (size_t)(a.k0 + k) * a.model_stride;
  const u32x4* const wcv = reinterpret_cast<const u32x4*>(a.wc + (size_t)(a.k0 + k) * a.wc_stride);  // [chunk][NFE + NFP + 3][64]
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // copies: fragment f of chunk c's record -> LDS (inline assembly, see the head of the kernel).  Every wave issues
  // NFE / 8 + 1 instructions for (WE, TP) and NFP / 8 for WP: the waits below count on it.
  constexpr int REC = NFE + NFP + 3;  // 1 KB pieces per chunk record
  static_assert(NFE % 8 == 0 && NFP % 8 == 0, "every wave issues the same number of copies");
  const unsigned lds_we = (unsigned)(size_t)(lds_ptr_t)WE, lds_wp = (unsigned)(size_t)(lds_ptr_t)WPj, lds_tp = (unsigned)(size_t)(lds_ptr_t)TP;
  auto dma1 = [&](const u32x4* src, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte), "v"(src + lane) : "memory");
  };
  auto dma_we_tp = [&](int c) {
    const u32x4* rec = wcv + (size_t)c * REC * 64;
#pragma unroll
    for (int i = 0; i < NFE / 8; ++i) dma1(rec + (size_t)(w + 8 * i) * 64, lds_we + (unsigned)(w + 8 * i) * 1024u);
    const int r = w % 3;  // (three pieces; waves 3..7 repeat one: the same bytes to the same place, and the same count for every wave)
    dma1(rec + (size_t)(NFE + NFP + r) * 64, lds_tp + (unsigned)(c & 1) * 3072u + (unsigned)r * 1024u);
  };
  auto dma_wp = [&](int c) {
    const u32x4* rec = wcv + ((size_t)c * REC + NFE) * 64;
#pragma unroll
    for (int i = 0; i < NFP / 8; ++i) dma1(rec + (size_t)(w + 8 * i) * 64, lds_wp + (unsigned)(w + 8 * i) * 1024u);
  };
  constexpr int N_WE_TP = NFE / 8 + 1;

#ifdef RIP_SPLIT_TICKS
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
  // prologue: zero E once (the padding columns are never written again)
  for (int e = tid; e < Geo::E_ROWS * LDE / 4; e += 512) reinterpret_cast<u32x4*>(E)[e] = zero4;
  SPLIT_TICK(0);

  // depthwise role of this thread: (observation, output column, 4-channel group)
  const int cg = tid & 15, dcol = (tid >> 4) % HOUT, dimg = (tid >> 4) / HOUT;
  const int dimg_c = dimg < G ? dimg : 0;  // (threads beyond G * HOUT * 16 never work; their addresses stay inside the buffers)
  const int e_off = (dimg_c * HIN * PW + dcol * STRIDE) * LDE + 4 * cg;  // first of the three padded columns
  const int d_off = (dimg_c * HWO + dcol) * LDD + 4 * cg;

  // persistent over observation groups: workgroup wgx walks the groups wgx, wgx + gridDim.x, ... of its model
  const int n_groups = (a.B + a.G - 1) / a.G;
  const int ng = (n_groups - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
  auto tile_on = [&](int t, int npt) { return !(WP * t + WP - 1 >= npt && wpix + WP * t >= npt); };  // compile-time but for the last t

  // The block input of group j + 1 is requested (raw fp32, unconditional clamped addresses: a fixed number of loads, the
  // counted wait below relies on it) where the LAST chunk of group j enters its depthwise, and split where group j + 1
  // starts: a first-touch HBM request takes 3-4 k cycles here, which used to stand in front of every group.
  // The raw values land in the registers of the two-term operands themselves (xh = first four channels, xl = the other
  // four, as bits): those are dead behind the group's last expansion, a second register set would not fit.
  constexpr int N_X = TIN * KSX * 2;
  u32x4 xh[TIN][KSX], xl[TIN][KSX];
  auto request_x = [&](int jn) {
    const int img0n = ((int)blockIdx.x + jn * (int)gridDim.x) * a.G;
    const int m_inn = min(a.G, a.B - img0n) * HWI;
    const float* xgn = a.x + ((size_t)k * a.B + img0n) * HWI * CIN;
    int n_ = n, q_ = q;  // (opaque copies: the per-lane offsets are group-invariant and would be hoisted and spilled)
    asm volatile("" : "+v"(n_), "+v"(q_));
#pragma unroll
    for (int t = 0; t < TIN; ++t) {
      const int px = 16 * (wpix + WP * t) + n_;
      const float* p = xgn + (size_t)(px < m_inn ? px : 0) * CIN + 8 * q_;
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks) {
        xh[t][ks] = *reinterpret_cast<const u32x4*>(p + 32 * ks);
        xl[t][ks] = *reinterpret_cast<const u32x4*>(p + 32 * ks + 4);
      }
    }
  };
  if (ng > 0) {
    request_x(0);
    dma_we_tp(0);  // (behind the prologue's barrier?  no LDS reader yet: E is zeroed, WE / TP are not read before the group's first barrier)
  }

#pragma unroll 1
  for (int j = 0; j < ng; ++j) {
    const int img0 = ((int)blockIdx.x + j * (int)gridDim.x) * a.G;
    const int n_img = min(a.G, a.B - img0);
    const int m_in = n_img * HWI, m_out = n_img * HWO;
    const float* xg = a.x + ((size_t)k * a.B + img0) * HWI * CIN;
    float* yg = a.y + ((size_t)k * a.B + img0) * HWO * COUT;
    const bool dw_on = dimg < G && dimg * HWI < m_in;

    // (the first chunk's expansion weights and taps were requested behind the previous group's last depthwise, or in
    // front of the group loop)
    // block input of this wave's pixel tiles as two-term B operands, resident for all chunks
    int erow[TIN];
    {
      int n_ = n;
      asm volatile("" : "+v"(n_));
#pragma unroll
      for (int t = 0; t < TIN; ++t) {
        const int px = 16 * (wpix + WP * t) + n_;
        const bool on = px < m_in;
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) {
          f32x4 v0 = __builtin_bit_cast(f32x4, xh[t][ks]), v1 = __builtin_bit_cast(f32x4, xl[t][ks]);
          if (!on) {  // (a lane without a pixel read pixel 0's values)
            v0 = f32x4{0.f, 0.f, 0.f, 0.f};
            v1 = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          const u32x2 s0 = split2(f32x2{v0[0], v0[1]}), s1 = split2(f32x2{v0[2], v0[3]});
          const u32x2 s2 = split2(f32x2{v1[0], v1[1]}), s3 = split2(f32x2{v1[2], v1[3]});
          xh[t][ks] = u32x4{s0.x, s1.x, s2.x, s3.x};
          xl[t][ks] = u32x4{s0.y, s1.y, s2.y, s3.y};
        }
        const int g = px / HWI, r = px - g * HWI, iy = r / HIN, ix = r - iy * HIN;
        erow[t] = on ? g * HIN * PW + iy * PW + ix + 1 : Geo::E_DUMP;
      }
    }
    f32x4 acc[TOUT][NCT];
#pragma unroll
    for (int t = 0; t < TOUT; ++t)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's copies have landed
    lds_barrier();                                    // ... and everybody's (also: the E zeroing of the prologue)
    dma_wp(0);  // (every wave is past the previous group's last projection)
    SPLIT_TICK(1);

#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
      // ---------------- expand chunk c -> E ----------------
      {
        float4 be[NHT];
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) be[ht] = *reinterpret_cast<const float4*>(TP + (size_t)(c & 1) * 768 + 10 * 64 + 16 * (ht0 + ht) + 4 * q);
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) {
          f32x4 v[TIN];
#pragma unroll
          for (int t = 0; t < TIN; ++t) v[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KSX; ++ks) {
            const u32x4 ah = WE[(size_t)(((ht0 + ht) * KSX + ks) * 2) * 64 + lane];
            const u32x4 al = WE[(size_t)(((ht0 + ht) * KSX + ks) * 2 + 1) * 64 + lane];
#pragma unroll
            for (int t = 0; t < TIN; ++t)
              if (tile_on(t, NPT_IN)) v[t] = mfmah(al, xh[t][ks], v[t]);  // small terms first
#pragma unroll
            for (int t = 0; t < TIN; ++t)
              if (tile_on(t, NPT_IN)) v[t] = mfmah(ah, xl[t][ks], v[t]);
#pragma unroll
            for (int t = 0; t < TIN; ++t)
              if (tile_on(t, NPT_IN)) v[t] = mfmah(ah, xh[t][ks], v[t]);
          }
#pragma unroll
          for (int t = 0; t < TIN; ++t) {
            if (!tile_on(t, NPT_IN)) continue;
            const f32x2 v0 = relu6_2(__builtin_elementwise_fma(f32x2{v[t][0], v[t][1]}, f32x2{W_INV, W_INV}, f32x2{be[ht].x, be[ht].y}));
            const f32x2 v1 = relu6_2(__builtin_elementwise_fma(f32x2{v[t][2], v[t][3]}, f32x2{W_INV, W_INV}, f32x2{be[ht].z, be[ht].w}));
            *reinterpret_cast<f32x4*>(E + (size_t)erow[t] * LDE + 16 * (ht0 + ht) + 4 * q) = f32x4{v0.x, v0.y, v1.x, v1.y};
          }
        }
      }
      SPLIT_TICK(2);
      lds_barrier();
      SPLIT_TICK(3);
      // copies that land under the depthwise and the projection: the next chunk's expansion weights (every wave is past
      // expand(c)) and its taps / biases (the other buffer)
      if (c + 1 < nch) dma_we_tp(c + 1);
      const bool x_ahead = c + 1 == nch && j + 1 < ng;
      if (x_ahead) request_x(j + 1);
      // ---------------- depthwise chunk c: E -> D (hi, lo) ----------------
      if (dw_on) {
        f32x2 wt[9][2], bd[2];
        {
          const float* wd = TP + (size_t)(c & 1) * 768 + 4 * cg;
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const float4 w0 = *reinterpret_cast<const float4*>(wd + t * 64);
            wt[t][0] = f32x2{w0.x, w0.y};
            wt[t][1] = f32x2{w0.z, w0.w};
          }
          const float4 b0 = *reinterpret_cast<const float4*>(wd + 9 * 64);
          bd[0] = f32x2{b0.x, b0.y};
          bd[1] = f32x2{b0.z, b0.w};
        }
        f32x2 sacc[HOUT][2];
#pragma unroll
        for (int iy = 0; iy < HIN; ++iy) {
          const float* r = E + e_off + iy * PW * LDE;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(r);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(r + LDE);
          const f32x4 v2 = *reinterpret_cast<const f32x4*>(r + 2 * LDE);
          const f32x2 f[3][2] = {{f32x2{v0[0], v0[1]}, f32x2{v0[2], v0[3]}},
                                 {f32x2{v1[0], v1[1]}, f32x2{v1[2], v1[3]}},
                                 {f32x2{v2[0], v2[1]}, f32x2{v2[2], v2[3]}}};
#pragma unroll
          for (int oy = 0; oy < HOUT; ++oy) {
            const int ky = iy - oy * STRIDE + 1;  // compile-time after unrolling
            if (ky < 0 || ky > 2) continue;
            if (ky == 0 || (ky == 1 && oy * STRIDE - 1 < 0)) {  // first row of this output that lies inside the map
              sacc[oy][0] = bd[0];
              sacc[oy][1] = bd[1];
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
              for (int e = 0; e < 2; ++e) sacc[oy][e] = __builtin_elementwise_fma(f[kx][e], wt[ky * 3 + kx][e], sacc[oy][e]);
            if (ky == 2 || iy == HIN - 1) {  // last row of this output inside the map: finish it
              const u32x2 s0 = split2(relu6_2(sacc[oy][0])), s1 = split2(relu6_2(sacc[oy][1]));
              *reinterpret_cast<u32x2*>(Dh + d_off + oy * HOUT * LDD) = u32x2{s0.x, s1.x};
              *reinterpret_cast<u32x2*>(Dl + d_off + oy * HOUT * LDD) = u32x2{s0.y, s1.y};
            }
          }
        }
      }
      SPLIT_TICK(4);
      // this chunk's projection weights (requested behind the previous projection) have landed; the copies issued above may
      // still be in flight (loads complete in order)
      if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_WE_TP) : "memory");
      else if (x_ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_X) : "memory");  // (the next group's block input stays in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      SPLIT_TICK(5);
      if (x_ahead) dma_we_tp(0);  // the NEXT group's first chunk (its model's, the same): WE and both tap buffers are free behind this barrier
      // ---------------- project chunk c: D -> acc ----------------
      {
        u32x4 bh[TOUT][NKP], bl[TOUT][NKP];  // (a ragged group's missing pixels read rows of D nobody wrote: never stored)
#pragma unroll
        for (int t = 0; t < TOUT; ++t) {
          if (!tile_on(t, NPT_OUT)) continue;
#pragma unroll
          for (int ks = 0; ks < NKP; ++ks) {
            const size_t o = (size_t)(16 * (wpix + WP * t) + n) * LDD + 32 * ks + 8 * q;
            bh[t][ks] = *reinterpret_cast<const u32x4*>(Dh + o);
            bl[t][ks] = *reinterpret_cast<const u32x4*>(Dl + o);
          }
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int ks = 0; ks < NKP; ++ks) {
            const u32x4 ah = WPj[(size_t)(((ct0w + ct) * NKP + ks) * 2) * 64 + lane];
            const u32x4 al = WPj[(size_t)(((ct0w + ct) * NKP + ks) * 2 + 1) * 64 + lane];
#pragma unroll
            for (int t = 0; t < TOUT; ++t)
              if (tile_on(t, NPT_OUT)) acc[t][ct] = mfmah(al, bh[t][ks], acc[t][ct]);
#pragma unroll
            for (int t = 0; t < TOUT; ++t)
              if (tile_on(t, NPT_OUT)) acc[t][ct] = mfmah(ah, bl[t][ks], acc[t][ct]);
#pragma unroll
            for (int t = 0; t < TOUT; ++t)
              if (tile_on(t, NPT_OUT)) acc[t][ct] = mfmah(ah, bh[t][ks], acc[t][ct]);
          }
      }
      SPLIT_TICK(6);
      if (c + 1 < nch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next chunk's expansion weights and taps have landed
        lds_barrier();                                    // ... everybody's, and everybody is past this projection:
        dma_wp(c + 1);                                    // its weights may be overwritten
      }
      SPLIT_TICK(7);
    }

    // ---------------- epilogue: 2^-8, bias (+ residual = block input), fp32 out ----------------
    {
      int n_ = n, q_ = q;
      asm volatile("" : "+v"(n_), "+v"(q_));
      // (all operand requests first: one memory latency for the epilogue, not one per pixel tile)
      float4 bp[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) bp[ct] = *reinterpret_cast<const float4*>(W + a.bp_off + 16 * (ct0w + ct) + 4 * q_);
      if (a.residual) {
#pragma unroll
        for (int t = 0; t < TOUT; ++t) {
          const int p = 16 * (wpix + WP * t) + n_;
          if (!tile_on(t, NPT_OUT) || p >= m_out) continue;
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(xg + (size_t)p * CIN + 16 * (ct0w + ct) + 4 * q_);
            acc[t][ct] = __builtin_elementwise_fma(r, f32x4{256.f, 256.f, 256.f, 256.f}, acc[t][ct]);  // (the accumulators carry 2^8; exact)
          }
        }
      }
#pragma unroll
      for (int t = 0; t < TOUT; ++t) {
        const int p = 16 * (wpix + WP * t) + n_;
        if (!tile_on(t, NPT_OUT) || p >= m_out) continue;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const f32x4 v = {fmaf(acc[t][ct][0], W_INV, bp[ct].x), fmaf(acc[t][ct][1], W_INV, bp[ct].y),
                           fmaf(acc[t][ct][2], W_INV, bp[ct].z), fmaf(acc[t][ct][3], W_INV, bp[ct].w)};
          *reinterpret_cast<f32x4*>(yg + (size_t)p * COUT + 16 * (ct0w + ct) + 4 * q_) = v;
        }
      }
    }
    SPLIT_TICK(1);
  }
#ifdef RIP_SPLIT_TICKS
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&g_split_ticks[i], tk[i]);
    atomicAdd(&g_split_ticks[8], 1ull);
    atomicAdd(&g_split_ticks[9], (unsigned long long)ng);
  }
#endif
}

// development sweeps (tools/dev/split_sweep.sh rebuilds with -D...)
#ifndef RIP_ST_G64
#define RIP_ST_G64 3
#endif
#ifndef RIP_ST_G96
#define RIP_ST_G96 3
#endif
#ifndef RIP_ST_G160
#define RIP_ST_G160 4
#endif
#ifndef RIP_ST_G320
#define RIP_ST_G320 2
#endif
#ifndef RIP_ST_W17
#define RIP_ST_W17 4
#endif
constexpr int ST_G64 = RIP_ST_G64, ST_G96 = RIP_ST_G96, ST_G160 = RIP_ST_G160, ST_G320 = RIP_ST_G320, ST_W17 = RIP_ST_W17;

template <int HIN, int STRIDE, int CIN, int COUT, int GMAX, int WCH>
hipError_t launch_split_tile(SplitTileArgs a, int kc, hipStream_t s) {
  using Geo = SplitGeom<HIN, STRIDE, CIN, COUT, GMAX>;
  static_assert(Geo::LDS_BYTES <= 160 * 1024, "LDS budget");
  if (a.HID != 6 * CIN) return hipErrorInvalidValue;
  // observations per workgroup: one workgroup per CU when the launch is large enough, never more than GMAX
  int G = (int)(((long)a.B * kc + 255) / 256);
  if (G > GMAX) G = GMAX;
  if (G < 1) G = 1;
  a.G = G;
  static bool attr_set[64] = {};  // per device: > 64 KB of dynamic LDS needs the opt-in
  auto kern = irb_split_tile_kernel<HIN, STRIDE, CIN, COUT, GMAX, WCH>;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)Geo::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  const int n_groups = (a.B + G - 1) / G;
  int gx = device_cu_count() / kc;
  if (gx < 1) gx = 1;
  if (gx > n_groups) gx = n_groups;
  note_kernel(dim3(gx, 1, kc), dim3(512), "irb_split_tile_kernel<%d,%d,%d,%d,%d,%d> G=%d", HIN, STRIDE, CIN, COUT, GMAX, WCH, G);
  hipLaunchKernelGGL(kern, dim3(gx, 1, kc), dim3(512), Geo::LDS_BYTES, s, a);
#ifdef RIP_SPLIT_TICKS
  {
    unsigned long long t[16];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_split_ticks), sizeof(t));
    const double n = t[8] > 0 ? (double)t[8] : 1.0, st = (double)(a.HID / HC) * (double)t[9] / n;
    fprintf(stderr, "split tile<%d,%d,%d,%d,G%d,%d> cycles per workgroup (%.1f groups): prologue %.0f | per group: setup + epilogue %.0f | per step (%d): "
            "expand %.0f barrier %.0f depthwise %.0f wait + barrier %.0f project %.0f wait + barrier %.0f\n",
            HIN, STRIDE, CIN, COUT, G, WCH, t[9] / n, t[0] / n, t[1] / (double)t[9], a.HID / HC, t[2] / n / st, t[3] / n / st,
            t[4] / n / st, t[5] / n / st, t[6] / n / st, t[7] / n / st);
    unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_split_ticks), z, sizeof(z));
  }
#endif
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// SMALL LAUNCHES of the tile-block layers (under SPLIT_TILE_MIN_PAIRS pairs: a single observation is 4 pairs): the tile
// kernel above walks an observation's chunks serially (25 us per block), the layer-wise kernels are three dependent
// latency-bound launches (~5 us each).  This kernel is the first TWO of them in one: a workgroup = (observation, 64-channel
// chunk of the hidden dimension) expands its chunk (split-f16 MFMA, the chunk's record of `pack_split_tiles`: expansion
// fragments + taps / biases, copied into LDS while the block input is fetched and split), runs the depthwise on it in LDS
// and writes the depthwise OUTPUT (fp32, ReLU6) where the layer-wise `dw_kernel` would: the projection stays the layer-wise
// K-split `pw_kernel`.  No cross-workgroup reduction, no atomics; one memory latency + two short phases.
// ------------------------------------------------------------------------------------------------------------------
struct ExpDwArgs {
  const float* x;      // [K][B][HIN][HIN][CIN] fp32
  float* d;            // [K][B][HOUT][HOUT][HID] fp32: the depthwise layer's output
  const h16_t* wc;     // the block's chunk records of model 0
  size_t wc_stride;
  int k0, B, HID, cout;  // (cout: the record's projection fragments are skipped)
};

template <int HIN, int STRIDE, int CIN>
__global__ __launch_bounds__(512) void irb_split_expdw_kernel(ExpDwArgs a) {
  constexpr int HOUT = STRIDE == 1 ? HIN : (HIN + 1) / 2, HWI = HIN * HIN, PW = HIN + 2;
  constexpr int E_DUMP = HIN * PW, E_ROWS = E_DUMP + 1;
  constexpr int NPT = (HWI + 15) / 16, KSX = CIN / 32, NFE = (HC / 16) * KSX * 2;
  constexpr int WPX = NPT > 2 ? 4 : 2, WCH = 8 / WPX, NHT = (HC / 16) / WCH;  // 4 pixel tiles x 2 channel halves (7x7), 2 x 4 -> 1 tile x 1..2 channel tiles (4x4: one pixel tile)
  static_assert(NPT <= WPX && (HC / 16) % WCH == 0 && NFE % 8 == 0 && HOUT * HOUT * 16 <= 1024, "shapes");
  __shared__ __attribute__((aligned(16))) float E[E_ROWS * LDE];
  __shared__ __attribute__((aligned(16))) u32x4 WE[NFE * 64];
  __shared__ __attribute__((aligned(16))) float TP[768];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int k = blockIdx.z, nch = a.HID / HC;
  const int b = blockIdx.x / nch, c = blockIdx.x - b * nch;
  const int nfp = (a.cout / 16) * (HC / 32) * 2, rec = NFE + nfp + 3;
  const u32x4* const recp = reinterpret_cast<const u32x4*>(a.wc + (size_t)(a.k0 + k) * a.wc_stride) + (size_t)c * rec * 64;
  // the chunk's expansion fragments and taps / biases -> LDS (every wave the same number of copies)
  {
    const unsigned lds_we = (unsigned)(size_t)(lds_ptr_t)WE, lds_tp = (unsigned)(size_t)(lds_ptr_t)TP;
#pragma unroll
    for (int i = 0; i < NFE / 8; ++i) {
      const int f = w + 8 * i;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_we + (unsigned)f * 1024u), "v"(recp + (size_t)f * 64 + lane) : "memory");
    }
    const int r = w % 3;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_tp + (unsigned)r * 1024u), "v"(recp + (size_t)(NFE + nfp + r) * 64 + lane) : "memory");
  }
  for (int e = tid; e < E_ROWS * LDE / 4; e += 512) reinterpret_cast<u32x4*>(E)[e] = u32x4{0u, 0u, 0u, 0u};
  // block input of this wave's pixel tile as two-term B operands
  const int wpx = w % WPX, wch = w / WPX;
  const int px = 16 * wpx + n;
  const bool on = wpx < NPT && px < HWI;
  u32x4 xh[KSX], xl[KSX];
  {
    const float* xp = a.x + (((size_t)k * a.B + b) * HWI + (on ? px : 0)) * CIN + 8 * q;
#pragma unroll
    for (int ks = 0; ks < KSX; ++ks) {
      f32x4 v0 = *reinterpret_cast<const f32x4*>(xp + 32 * ks), v1 = *reinterpret_cast<const f32x4*>(xp + 32 * ks + 4);
      if (!on) {
        v0 = f32x4{0.f, 0.f, 0.f, 0.f};
        v1 = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const u32x2 s0 = split2(f32x2{v0[0], v0[1]}), s1 = split2(f32x2{v0[2], v0[3]});
      const u32x2 s2 = split2(f32x2{v1[0], v1[1]}), s3 = split2(f32x2{v1[2], v1[3]});
      xh[ks] = u32x4{s0.x, s1.x, s2.x, s3.x};
      xl[ks] = u32x4{s0.y, s1.y, s2.y, s3.y};
    }
  }
  const int iy = px / HIN, ix = px - iy * HIN;
  const int erow = on ? iy * PW + ix + 1 : E_DUMP;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  // ---------------- expand -> E ----------------
  if (wpx < NPT) {
#pragma unroll
    for (int hi = 0; hi < NHT; ++hi) {
      const int ht = wch * NHT + hi;
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f}, vc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks) {
        const u32x4 ah = WE[(size_t)((ht * KSX + ks) * 2) * 64 + lane], al = WE[(size_t)((ht * KSX + ks) * 2 + 1) * 64 + lane];
        va = mfmah(al, xh[ks], va);
        vb = mfmah(ah, xl[ks], vb);
        vc = mfmah(ah, xh[ks], vc);
      }
      const f32x4 v = vc + (va + vb);
      const float4 be = *reinterpret_cast<const float4*>(TP + 10 * 64 + 16 * ht + 4 * q);
      const f32x2 v0 = relu6_2(__builtin_elementwise_fma(f32x2{v[0], v[1]}, f32x2{W_INV, W_INV}, f32x2{be.x, be.y}));
      const f32x2 v1 = relu6_2(__builtin_elementwise_fma(f32x2{v[2], v[3]}, f32x2{W_INV, W_INV}, f32x2{be.z, be.w}));
      *reinterpret_cast<f32x4*>(E + (size_t)erow * LDE + 16 * ht + 4 * q) = f32x4{v0.x, v0.y, v1.x, v1.y};
    }
  }
  lds_barrier();
  // ---------------- depthwise -> global (thread = (output pixel, 4 channels); HOUT^2 * 16 items) ----------------
  float* dp = a.d + ((size_t)k * a.B + b) * HOUT * HOUT * a.HID + (size_t)c * HC;
  for (int it = tid; it < HOUT * HOUT * 16; it += 512) {
    const int cg = it & 15, op = it >> 4, oy = op / HOUT, ox = op - oy * HOUT;
    const float4 b0 = *reinterpret_cast<const float4*>(TP + 9 * 64 + 4 * cg);
    f32x2 s0 = {b0.x, b0.y}, s1 = {b0.z, b0.w};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int r = oy * STRIDE - 1 + ky;
      if (r < 0 || r >= HIN) continue;  // (rows off the map: zero contributions; columns off the map are E's zero columns)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const f32x4 e = *reinterpret_cast<const f32x4*>(E + (size_t)(r * PW + ox * STRIDE + kx) * LDE + 4 * cg);
        const float4 wv = *reinterpret_cast<const float4*>(TP + (ky * 3 + kx) * 64 + 4 * cg);
        s0 = __builtin_elementwise_fma(f32x2{e[0], e[1]}, f32x2{wv.x, wv.y}, s0);
        s1 = __builtin_elementwise_fma(f32x2{e[2], e[3]}, f32x2{wv.z, wv.w}, s1);
      }
    }
    s0 = relu6_2(s0);
    s1 = relu6_2(s1);
    *reinterpret_cast<f32x4*>(dp + (size_t)op * a.HID + 4 * cg) = f32x4{s0.x, s0.y, s1.x, s1.y};
  }
}

// ------------------------------------------------------------------------------------------------------------------
// features.18 (1x1, 320 -> 1280, ReLU6) + the 4x4 average pool of the fp32 encoder in the same number format: the last
// true-fp32 GEMM of the fp32 mode (`pw_kernel` with the pooled epilogue, 285 us at 512 observations x 4 models).
// A wave owns ONE observation (its 16 pixels = one MFMA pixel tile; block input resident as two-term B operands, 80
// registers); the 1280 output channels stream through LDS in chunks of 32 (`pack_split_tiles`: 40 KB of operand
// fragments + the biases per chunk, one contiguous record), three buffers deep: the copy of chunk c + 2 is issued (inline
// assembly, all eight waves, six instructions each) where chunk c starts.  Epilogue per chunk: 2^-8, bias, ReLU6, mean
// over the tile's 16 pixels (a butterfly over the 16 lanes of a row), one float4 per (observation, 4 channels).
// ------------------------------------------------------------------------------------------------------------------
constexpr int HD_CIN = 320, HD_COUT = 1280, HD_CH = 32, HD_KS = HD_CIN / 32, HD_NF = (HD_CH / 16) * HD_KS * 2, HD_REC = HD_NF + 1;
constexpr int HD_NCH = HD_COUT / HD_CH, HD_G = 8;
constexpr size_t HD_LDS = (size_t)3 * HD_REC * 1024;

struct HeadArgs {
  const float* x;    // [K][B][16][320] fp32
  float* y;          // [K][B][1280] fp32 (pooled)
  const h16_t* wc;   // the head's chunk records of model 0, models wc_stride binary16 elements apart
  size_t wc_stride;
  int k0, B;
};

__global__ __launch_bounds__(512) void head_split_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* const WB = reinterpret_cast<u32x4*>(smem_raw);  // [3][HD_REC][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int k = blockIdx.z;
  const u32x4* const wcv = reinterpret_cast<const u32x4*>(a.wc + (size_t)(a.k0 + k) * a.wc_stride);  // [chunk][HD_REC][64]
  const unsigned lds_wb = (unsigned)(size_t)(lds_ptr_t)WB;
  static_assert(HD_NF % 8 == 0, "every wave issues the same number of copies");
  constexpr int N_DMA = HD_NF / 8 + 1;
  auto dma_chunk = [&](int c) {
    const u32x4* rec = wcv + (size_t)c * HD_REC * 64;
    const unsigned dst = lds_wb + (unsigned)(c % 3) * (unsigned)(HD_REC * 1024);
#pragma unroll
    for (int i = 0; i < HD_NF / 8; ++i) {
      const int f = w + 8 * i;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst + (unsigned)f * 1024u), "v"(rec + (size_t)f * 64 + lane) : "memory");
    }
    // (the biases' piece by every wave: the same bytes to the same place, and the same count for every wave)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst + (unsigned)HD_NF * 1024u), "v"(rec + (size_t)HD_NF * 64 + lane) : "memory");
  };
  const int n_groups = (a.B + HD_G - 1) / HD_G;
#pragma unroll 1
  for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int img = g * HD_G + w;
    const bool on = img < a.B;
    dma_chunk(0);
    dma_chunk(1);
    u32x4 xh[HD_KS], xl[HD_KS];
    {
      const float* xp = a.x + (((size_t)k * a.B + (on ? img : 0)) * 16 + n) * HD_CIN + 8 * q;
#pragma unroll
      for (int ks = 0; ks < HD_KS; ++ks) {
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        if (on) {
          v0 = *reinterpret_cast<const f32x4*>(xp + 32 * ks);
          v1 = *reinterpret_cast<const f32x4*>(xp + 32 * ks + 4);
        }
        const u32x2 s0 = split2(f32x2{v0[0], v0[1]}), s1 = split2(f32x2{v0[2], v0[3]});
        const u32x2 s2 = split2(f32x2{v1[0], v1[1]}), s3 = split2(f32x2{v1[2], v1[3]});
        xh[ks] = u32x4{s0.x, s1.x, s2.x, s3.x};
        xl[ks] = u32x4{s0.y, s1.y, s2.y, s3.y};
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    float* yp = a.y + ((size_t)k * a.B + (on ? img : 0)) * HD_COUT + 4 * q;
#pragma unroll 1
    for (int c = 0; c < HD_NCH; ++c) {
      if (c + 2 < HD_NCH) dma_chunk(c + 2);  // (its buffer was chunk c - 1's: every wave is behind the barrier that ended it)
      const u32x4* wb = WB + (size_t)(c % 3) * HD_REC * 64;
      f32x4 acc[2][3];
#pragma unroll
      for (int ht = 0; ht < 2; ++ht)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[ht][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HD_KS; ++ks)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
          const u32x4 ah = wb[(size_t)((ht * HD_KS + ks) * 2) * 64 + lane], al = wb[(size_t)((ht * HD_KS + ks) * 2 + 1) * 64 + lane];
          acc[ht][0] = mfmah(al, xh[ks], acc[ht][0]);  // one accumulator per term: six independent chains
          acc[ht][1] = mfmah(ah, xl[ks], acc[ht][1]);
          acc[ht][2] = mfmah(ah, xh[ks], acc[ht][2]);
        }
#pragma unroll
      for (int ht = 0; ht < 2; ++ht) {
        const float4 bb = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(wb + (size_t)HD_NF * 64) + 16 * ht + 4 * q);
        const f32x4 s4 = acc[ht][2] + (acc[ht][0] + acc[ht][1]);  // (small terms first)
        f32x2 v0 = relu6_2(__builtin_elementwise_fma(f32x2{s4[0], s4[1]}, f32x2{W_INV, W_INV}, f32x2{bb.x, bb.y}));
        f32x2 v1 = relu6_2(__builtin_elementwise_fma(f32x2{s4[2], s4[3]}, f32x2{W_INV, W_INV}, f32x2{bb.z, bb.w}));
        float r[4] = {v0.x, v0.y, v1.x, v1.y};
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // sum over the 16 pixels = the 16 lanes n of this q
          r[e] += __shfl_xor(r[e], 1);
          r[e] += __shfl_xor(r[e], 2);
          r[e] += __shfl_xor(r[e], 4);
          r[e] += __shfl_xor(r[e], 8);
        }
        if (on && n == 0)
          *reinterpret_cast<f32x4*>(yp + HD_CH * c + 16 * ht) = f32x4{r[0] * 0.0625f, r[1] * 0.0625f, r[2] * 0.0625f, r[3] * 0.0625f};
      }
      // chunk c + 1 (requested where chunk c - 1 started) has landed; the six copies issued above may still be in flight
      // (loads complete in order; this wave's stores only make the wait longer)
      if (c + 2 < HD_NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_DMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
    }
  }
}

}  // namespace

hipError_t launch_irb_split_expdw(const Layer* le, const Layer& ld, const Layer& lp, const unsigned short* wc, size_t wc_stride,
                                  int k0, int kc, int B, const float* x, float* d, hipStream_t s) {
  if (!irb_split_tile_supported(le, ld, lp)) return hipErrorInvalidValue;
  ExpDwArgs a;
  a.x = x;
  a.d = d;
  a.wc = wc;
  a.wc_stride = wc_stride;
  a.k0 = k0;
  a.B = B;
  a.HID = ld.cout;
  a.cout = lp.cout;
  const dim3 grid((unsigned)(B * (ld.cout / HC)), 1, kc);
  const int cin = le->cin;
  if (ld.h_in == 7 && ld.stride == 1 && cin == 64) {
    note_kernel(grid, dim3(512), "irb_split_expdw_kernel<7,1,64>");
    hipLaunchKernelGGL((irb_split_expdw_kernel<7, 1, 64>), grid, dim3(512), 0, s, a);
  } else if (ld.h_in == 7 && ld.stride == 1 && cin == 96) {
    note_kernel(grid, dim3(512), "irb_split_expdw_kernel<7,1,96>");
    hipLaunchKernelGGL((irb_split_expdw_kernel<7, 1, 96>), grid, dim3(512), 0, s, a);
  } else if (ld.h_in == 7 && ld.stride == 2 && cin == 96) {
    note_kernel(grid, dim3(512), "irb_split_expdw_kernel<7,2,96>");
    hipLaunchKernelGGL((irb_split_expdw_kernel<7, 2, 96>), grid, dim3(512), 0, s, a);
  } else if (ld.h_in == 4 && ld.stride == 1 && cin == 160) {
    note_kernel(grid, dim3(512), "irb_split_expdw_kernel<4,1,160>");
    hipLaunchKernelGGL((irb_split_expdw_kernel<4, 1, 160>), grid, dim3(512), 0, s, a);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

bool head_split_supported(const Layer& l, int final_hw) {
  return l.kind == L_PW && l.cin == HD_CIN && l.cout == HD_COUT && l.h_in == 4 && final_hw == 4 && l.relu6 && !l.residual;
}

hipError_t launch_head_split(const Layer& l, const unsigned short* wc, size_t wc_stride, int k0, int kc, int B, const float* x,
                             float* y, hipStream_t s) {
  (void)l;
  HeadArgs a;
  a.x = x;
  a.y = y;
  a.wc = wc;
  a.wc_stride = wc_stride;
  a.k0 = k0;
  a.B = B;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HD_LDS);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  const int n_groups = (B + HD_G - 1) / HD_G;
  int gx = device_cu_count() / kc;
  if (gx < 1) gx = 1;
  if (gx > n_groups) gx = n_groups;
  note_kernel(dim3(gx, 1, kc), dim3(512), "head_split_kernel");
  hipLaunchKernelGGL(head_split_kernel, dim3(gx, 1, kc), dim3(512), HD_LDS, s, a);
  return hipGetLastError();
}

bool irb_split_tile_supported(const Layer* le, const Layer& ld, const Layer& lp) {
  if (le == nullptr) return false;
  const int cin = le->cin, hid = ld.cout, cout = lp.cout;
  if (hid != 6 * cin || hid % HC != 0) return false;
  if (ld.h_in == 7 && ld.stride == 1) return (cin == 64 && (cout == 64 || cout == 96)) || (cin == 96 && cout == 96);
  if (ld.h_in == 7 && ld.stride == 2) return cin == 96 && cout == 160;
  if (ld.h_in == 4 && ld.stride == 1) return cin == 160 && (cout == 160 || cout == 320);
  return false;
}

SplitTileLayout split_tile_layout(const EncoderPlan& plan) {
  SplitTileLayout L;
  L.off.assign(plan.blocks.size(), (size_t)-1);
  size_t off = 0;
  for (size_t bi = 0; bi < plan.blocks.size(); ++bi) {
    const FusedBlock& fb = plan.blocks[bi];
    const Layer* le = fb.expand >= 0 ? &plan.layers[fb.expand] : nullptr;
    if (!irb_split_tile_supported(le, plan.layers[fb.dw], plan.layers[fb.project])) continue;
    const int cin = le->cin, hid = plan.layers[fb.dw].cout, cout = plan.layers[fb.project].cout;
    const size_t rec = (size_t)((HC / 16) * (cin / 32) * 2 + (cout / 16) * (HC / 32) * 2 + 3) * 512;  // binary16 elements per chunk record
    L.off[bi] = off;
    off += rec * (hid / HC);
  }
  const Layer& last = plan.layers.back();
  if (head_split_supported(last, plan.final_hw)) {
    L.head_off = off;
    off += (size_t)HD_NCH * HD_REC * 512;
  }
  L.total = off;
  return L;
}

// One model's folded fp32 blob -> the chunk records of every tile block.  Record of chunk c (1 KB pieces, in LDS order):
//   expansion fragments (ht, ks, term): lane (n, q), element j = We[64 c + 16 ht + n][32 ks + 8 q + j];
//   projection fragments (ct, ks, term): Wp[16 ct + n][64 c + 32 ks + 8 q + j];
//   (weights as w 2^8 split into hi = f16(.), lo = f16(. - hi): term 0 / 1)
//   3 KB of fp32: depthwise taps [9][64], depthwise biases [64], expansion biases [64] of the chunk's channels (+ padding).
void pack_split_tiles(const EncoderPlan& plan, const SplitTileLayout& L, const float* enc, unsigned short* out) {
  auto put = [&](size_t idx, float wv, int term) {
    const float v = wv * SPLIT_ENC_W_SCALE;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const _Float16 r = term ? lo : hi;
    std::memcpy(&out[idx], &r, 2);
  };
  for (size_t bi = 0; bi < plan.blocks.size(); ++bi) {
    if (L.off[bi] == (size_t)-1) continue;
    const FusedBlock& fb = plan.blocks[bi];
    const Layer &le = plan.layers[fb.expand], &ld = plan.layers[fb.dw], &lp = plan.layers[fb.project];
    const int cin = le.cin, hid = ld.cout, cout = lp.cout, ksx = cin / 32, nctp = cout / 16;
    const int nfe = (HC / 16) * ksx * 2, nfp = nctp * (HC / 32) * 2;
    const size_t rec = (size_t)(nfe + nfp + 3) * 512;
    for (int c = 0; c < hid / HC; ++c) {
      size_t o = L.off[bi] + (size_t)c * rec;
      for (int ht = 0; ht < HC / 16; ++ht)
        for (int ks = 0; ks < ksx; ++ks)
          for (int term = 0; term < 2; ++term)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 8; ++j)
                put(o++, enc[le.w_off + (size_t)(c * HC + 16 * ht + (lane & 15)) * cin + 32 * ks + 8 * (lane >> 4) + j], term);
      for (int ct = 0; ct < nctp; ++ct)
        for (int ks = 0; ks < HC / 32; ++ks)
          for (int term = 0; term < 2; ++term)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 8; ++j)
                put(o++, enc[lp.w_off + (size_t)(16 * ct + (lane & 15)) * hid + c * HC + 32 * ks + 8 * (lane >> 4) + j], term);
      float tp[768];
      for (int i = 0; i < 768; ++i) tp[i] = 0.f;
      for (int t = 0; t < 9; ++t)
        for (int i = 0; i < HC; ++i) tp[t * HC + i] = enc[ld.w_off + (size_t)t * hid + c * HC + i];
      for (int i = 0; i < HC; ++i) {
        tp[9 * HC + i] = enc[ld.b_off + c * HC + i];
        tp[10 * HC + i] = enc[le.b_off + c * HC + i];
      }
      std::memcpy(&out[o], tp, sizeof(tp));
    }
  }
  if (L.head_off != (size_t)-1) {  // features.18: chunk c = output channels 32 c ..: fragments (ht, ks, term), then 1 KB with the 32 biases
    const Layer& l = plan.layers.back();
    for (int c = 0; c < HD_NCH; ++c) {
      size_t o = L.head_off + (size_t)c * HD_REC * 512;
      for (int ht = 0; ht < HD_CH / 16; ++ht)
        for (int ks = 0; ks < HD_KS; ++ks)
          for (int term = 0; term < 2; ++term)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 8; ++j)
                put(o++, enc[l.w_off + (size_t)(c * HD_CH + 16 * ht + (lane & 15)) * HD_CIN + 32 * ks + 8 * (lane >> 4) + j], term);
      float bb[256];
      for (int i = 0; i < 256; ++i) bb[i] = i < HD_CH ? enc[l.b_off + c * HD_CH + i] : 0.f;
      std::memcpy(&out[o], bb, sizeof(bb));
    }
  }
}

hipError_t launch_irb_split_tile(const Layer* le, const Layer& ld, const Layer& lp, const float* enc_w, const unsigned short* wc,
                                 size_t wc_stride, size_t model_stride, int k0, int kc, int B, const float* x, float* y, hipStream_t s) {
  SplitTileArgs a;
  a.x = x;
  a.y = y;
  a.wbase = enc_w;
  a.wc = wc;
  a.wc_stride = wc_stride;
  a.model_stride = model_stride;
  a.k0 = k0;
  a.we_off = le->w_off;
  a.be_off = le->b_off;
  a.wd_off = ld.w_off;
  a.bd_off = ld.b_off;
  a.wp_off = lp.w_off;
  a.bp_off = lp.b_off;
  a.B = B;
  a.HID = ld.cout;
  a.residual = lp.residual;
  a.G = 1;
  const int cin = le->cin, cout = lp.cout;
  //                                                HIN S CIN COUT GMAX WCH
  if (ld.h_in == 7 && ld.stride == 1) {
    if (cin == 64 && cout == 64) return launch_split_tile<7, 1, 64, 64, ST_G64, 2>(a, kc, s);    // features.8-10
    if (cin == 64 && cout == 96) return launch_split_tile<7, 1, 64, 96, ST_G64, 2>(a, kc, s);    // features.11
    if (cin == 96 && cout == 96) return launch_split_tile<7, 1, 96, 96, ST_G96, 2>(a, kc, s);    // features.12, 13
  }
  if (ld.h_in == 7 && ld.stride == 2 && cin == 96 && cout == 160) return launch_split_tile<7, 2, 96, 160, ST_G96, 2>(a, kc, s);  // 14
  if (ld.h_in == 4 && ld.stride == 1 && cin == 160) {
    if (cout == 160) return launch_split_tile<4, 1, 160, 160, ST_G160, 2>(a, kc, s);            // features.15, 16
    if (cout == 320) return launch_split_tile<4, 1, 160, 320, ST_G320, ST_W17>(a, kc, s);       // features.17
  }
  return hipErrorInvalidValue;
}

}  // namespace rip
