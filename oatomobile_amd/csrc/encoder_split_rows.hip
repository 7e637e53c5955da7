// fp32-GRADE fused inverted-residual block for the large-image stages of the fp32 (parity-mode) encoder on gfx950
// (features.2 .. features.7: 50x50 -> 7x7 maps): expand 1x1 -> depthwise 3x3 -> project 1x1 (+ residual) in one kernel,
// fp32 activations in HBM, both pointwise convolutions on the binary16 matrix pipe with TWO-TERM operands
// (encoder_split_tile.hip has the number format: weights as w 2^8 = hi + lo, activations hi = f16(x), lo = f16(x - hi),
// three v_mfma_f32_16x16x32_f16 per product, fp32 accumulation).
//
// torchvision v0.6.0 `InvertedResidual` (reference call site oatomobile/torch/networks/perception.py:36-51), BN folded.
// Rounds 1-5 ran features.1-3 as `irb_kernel` (encoder_fused.hip: true fp32 MFMA, one workgroup per row tile, its phases
// waiting on weight loads: 0.9-1.0 ms per block at 512 observations x 4 models) and features.4-7 layer by layer (0.2-0.56
// ms per block).  Here a workgroup STREAMS an observation's rows:
//   * a block's pointwise weights are small (<= 49 KB as two binary16 terms): they live in LDS for the kernel's lifetime,
//     packed on the host as the MFMA operand fragments they are used as (`pack_split_rows`: lane (n, q) of fragment
//     (channel tile, K block, term) = 8 consecutive K values of row 16 tile + n; zero rows / columns where a dimension is
//     padded to the 16 x 32 tile) and copied in once;
//   * the expanded tensor exists as a ring of R = (RB - 1) S + 3 rows in LDS (fp32, zero columns either side), the
//     depthwise output as RB = 2 rows of two binary16 planes;
//   * a step expands RB S new input rows (their pixels flattened into 16-pixel tiles: 200 pixels of a 50x50 map are 13
//     tiles, not 4 x 4), runs the depthwise for the RB output rows whose window is now complete and projects them:
//         split x(s) | request x(s + 1) | expand -> E ring | barrier | depthwise -> D | barrier | project -> y
//     The block input is fetched one step ahead (the last step of an observation requests the first rows of the
//     workgroup's next one); nothing else comes from global memory inside the loop but the residual.
//   * the depthwise stays on the vector unit in fp32: thread = (4 channels, pixel slot), its 36 taps in registers for
//     the whole kernel.
// Contract: the fp32 oracle at 1e-4 on z (tests/test_gpu_parity.py); not bit-identical to the layer-wise fp32 kernels.
#include <cmath>
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

__device__ __forceinline__ f32x4 mfmah(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x2 relu6_2(f32x2 v) {
  return __builtin_elementwise_min(__builtin_elementwise_max(v, f32x2{0.f, 0.f}), f32x2{6.f, 6.f});
}
__device__ __forceinline__ u32x2 split2(f32x2 x) {  // .x = hi pair, .y = lo pair
  const h16x2 h = __builtin_convertvector(x, h16x2);
  const f32x2 back = __builtin_convertvector(h, f32x2);
  const h16x2 l = __builtin_convertvector(x - back, h16x2);
  return u32x2{__builtin_bit_cast(unsigned, h), __builtin_bit_cast(unsigned, l)};
}

constexpr float W_INV = 1.0f / SPLIT_ENC_W_SCALE;
constexpr int RB = 2;  // output rows per step

#ifdef RIP_ROWS_TICKS  // development (tools/dev/rows_ticks.sh): shader cycles per phase of wave 0, summed over the workgroups
__device__ unsigned long long g_rows_ticks[16];
#define ROWS_TICK(slot_)                                            \
  do {                                                              \
    const unsigned long long now_ = __builtin_readcyclecounter();   \
    tk[slot_] += now_ - tlast;                                      \
    tlast = now_;                                                   \
  } while (0)
#else
#define ROWS_TICK(slot_) do { } while (0)
#endif

template <int HIN, int S, int CIN, int HID, int COUT>
struct RowsGeom {
  static constexpr int W = HIN, HOUT = S == 1 ? HIN : (HIN + 1) / 2, PW = W + 2;
  static constexpr int NRI = RB * S;             // new input rows per step
  static constexpr int R = (RB - 1) * S + 3;     // ring rows
  static constexpr int NS = (HOUT - 1) / 2 + 1;  // steps per observation (step s finishes output rows ob(s), ob(s) + 1; ob = 2 s - 1 (S = 1), 2 s (S = 2))
  static constexpr int LDE = HID + 4;            // fp32 elements per E pixel (an odd multiple of 16 bytes for the three widths)
  static constexpr int HIDP = (HID + 31) / 32 * 32;
  static constexpr int LDD = HIDP + 8;           // binary16 elements per D pixel
  static constexpr int NCTE = HID / 16, NCTP = (COUT + 15) / 16, NKP = HIDP / 32;
  static constexpr int NPT = (NRI * W + 15) / 16;      // input pixel tiles per step
  static constexpr int NPO = (RB * HOUT + 15) / 16;    // output pixel tiles per step
  static constexpr int DROWS = NPO * 16;
  static constexpr int E_DUMP = R * PW;                // pixel row that takes the stores of lanes without a pixel
  static constexpr size_t E_BYTES = (size_t)(E_DUMP + 1) * LDE * sizeof(float);
  static constexpr size_t D_PLANE = (size_t)DROWS * LDD;
  static constexpr size_t D_BYTES = 2 * D_PLANE * sizeof(h16_t);
  static constexpr int NFE = NCTE * 2, NFP = NCTP * NKP * 2;  // 1 KB operand fragments
  static constexpr size_t P_FLOATS = (size_t)HID + NCTP * 16;  // expansion biases, projection biases (padded)
  static constexpr size_t LDS_BYTES = E_BYTES + D_BYTES + (size_t)(NFE + NFP) * 1024 + P_FLOATS * sizeof(float);
  static_assert(HID % 16 == 0 && CIN % 8 == 0 && CIN <= 32 && COUT % 4 == 0, "shapes");
  static_assert(NPO * NCTP <= 8, "one projection tile per wave");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

struct RowsArgs {
  const float* x;      // [K][B][HIN][HIN][CIN] fp32
  float* y;            // [K][B][HOUT][HOUT][COUT] fp32
  const float* wbase;  // fp32 folded blobs (biases, depthwise taps)
  const h16_t* wfrag;  // this block's operand fragments of model 0 (pack_split_rows), models wr_stride apart
  size_t model_stride, wr_stride;
  int k0;
  size_t be_off, wd_off, bd_off, bp_off;
  int B, residual;
  int NB;  // row bands per observation (small launches: a workgroup walks (observation, band) items; a band = a run of steps)
};

template <int HIN, int S, int CIN, int HID, int COUT>
__global__ __launch_bounds__(512) void irb_split_rows_kernel(RowsArgs a) {
  using Geo = RowsGeom<HIN, S, CIN, HID, COUT>;
  constexpr int W = Geo::W, HOUT = Geo::HOUT, PW = Geo::PW, NRI = Geo::NRI, R = Geo::R, NS = Geo::NS, LDE = Geo::LDE, LDD = Geo::LDD;
  constexpr int NCTE = Geo::NCTE, NCTP = Geo::NCTP, NKP = Geo::NKP, NPT = Geo::NPT, NPO = Geo::NPO;
  // expansion: the eight waves as WPX pixel partitions x WCH channel partitions
  constexpr int WPX = NPT > 4 ? 8 : (NPT > 2 ? 4 : 2), WCH = 8 / WPX;
  constexpr int TIN = (NPT + WPX - 1) / WPX, NCTW = (NCTE + WCH - 1) / WCH;
  // depthwise: thread = (4-channel group, pixel slot)
  constexpr int NCG = HID / 4, PSL = 512 / NCG, NOP = RB * HOUT, JMAX = (NOP + PSL - 1) / PSL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* const E = reinterpret_cast<float*>(smem_raw);                                   // [R][PW][LDE] + dump pixel
  h16_t* const Dh = reinterpret_cast<h16_t*>(smem_raw + Geo::E_BYTES);                   // [DROWS][LDD] hi
  h16_t* const Dl = Dh + Geo::D_PLANE;                                                   // lo
  u32x4* const WE = reinterpret_cast<u32x4*>(smem_raw + Geo::E_BYTES + Geo::D_BYTES);    // [NFE][64]
  u32x4* const WP = WE + (size_t)Geo::NFE * 64;                                           // [NFP][64]
  float* const PB = reinterpret_cast<float*>(WP + (size_t)Geo::NFP * 64);                 // be [HID], bp [NCTP * 16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int k = blockIdx.z;
  const float* Wf = a.wbase + (size_t)(a.k0 + k) * a.model_stride;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#ifdef RIP_ROWS_TICKS
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif

  // ---- prologue: zero E and D (padding columns / padded K columns are never written again), weights and biases into LDS ----
  for (int e = tid; e < (int)(Geo::E_BYTES / 16); e += 512) reinterpret_cast<u32x4*>(E)[e] = zero4;
  for (int e = tid; e < (int)(Geo::D_BYTES / 16); e += 512) reinterpret_cast<u32x4*>(Dh)[e] = zero4;
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.wfrag + (size_t)(a.k0 + k) * a.wr_stride);
    for (int e = tid; e < (Geo::NFE + Geo::NFP) * 64; e += 512) WE[e] = src[e];
    for (int e = tid; e < HID; e += 512) PB[e] = Wf[a.be_off + e];
    for (int e = tid; e < NCTP * 16; e += 512) PB[HID + e] = e < COUT ? Wf[a.bp_off + e] : 0.f;
  }
  // depthwise role: taps and bias of this thread's four channels, for the whole kernel
  const int cg = tid % NCG, pslot = tid / NCG;
  const bool dw_thread = pslot < PSL;
  f32x2 wt[9][2], bd[2];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float4 w0 = *reinterpret_cast<const float4*>(Wf + a.wd_off + (size_t)t * HID + 4 * cg);
    wt[t][0] = f32x2{w0.x, w0.y};
    wt[t][1] = f32x2{w0.z, w0.w};
  }
  {
    const float4 b0 = *reinterpret_cast<const float4*>(Wf + a.bd_off + 4 * cg);
    bd[0] = f32x2{b0.x, b0.y};
    bd[1] = f32x2{b0.z, b0.w};
  }
  // expansion role
  const int wpx = w % WPX, wch = w / WPX;
  const int ct_lo = wch * NCTW, ct_hi = ct_lo + NCTW < NCTE ? ct_lo + NCTW : NCTE;
  int t_irow[TIN], t_ix[TIN];
  bool t_on[TIN];
#pragma unroll
  for (int t = 0; t < TIN; ++t) {
    const int p = 16 * (wpx + WPX * t) + n;
    t_on[t] = p < NRI * W;
    t_irow[t] = p / W;
    t_ix[t] = p - t_irow[t] * W;
  }
  // projection role: wave w owns (output pixel tile, channel tile) pair w
  const bool pj_wave = w < NPO * NCTP;
  const int pj_pt = w % NPO, pj_ct = w / NPO;
  const int pj_j = 16 * pj_pt + n, pj_orow = pj_j / HOUT, pj_ox = pj_j - pj_orow * HOUT;
  const int pj_ch = 16 * pj_ct + 4 * q;
  lds_barrier();
  ROWS_TICK(0);

  // block input rows of (observation b, step s) -> raw fp32 registers (zeros off the map / beyond CIN)
  f32x4 xr[TIN][2];
  auto request_x = [&](int b, int s) {
    const float* xb = a.x + ((size_t)k * a.B + b) * HIN * W * CIN;
#pragma unroll
    for (int t = 0; t < TIN; ++t) {
      const int row = s * NRI + t_irow[t];
      xr[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      xr[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t_on[t] && row < HIN && 8 * q < CIN) {
        const float* p = xb + ((size_t)row * W + t_ix[t]) * CIN + 8 * q;
        xr[t][0] = *reinterpret_cast<const f32x4*>(p);
        xr[t][1] = *reinterpret_cast<const f32x4*>(p + 4);
      }
    }
  };

  // The split of the next step's input sits BEHIND the depthwise barrier and IN FRONT of the projection's stores: the
  // compiler's wait for the request (issued at the top of the step) cannot tell how many of the predicated stores of
  // the projection were issued after it, so at the top of the next step it was `vmcnt(0)` — every projecting wave waiting
  // for its own stores to be acknowledged (1-2 k cycles per step).
  u32x4 xh[TIN], xl[TIN];
  auto split_x = [&]() {
#pragma unroll
    for (int t = 0; t < TIN; ++t) {
      const u32x2 s0 = split2(f32x2{xr[t][0][0], xr[t][0][1]}), s1 = split2(f32x2{xr[t][0][2], xr[t][0][3]});
      const u32x2 s2 = split2(f32x2{xr[t][1][0], xr[t][1][1]}), s3 = split2(f32x2{xr[t][1][2], xr[t][1][3]});
      xh[t] = u32x4{s0.x, s1.x, s2.x, s3.x};
      xl[t] = u32x4{s0.y, s1.y, s2.y, s3.y};
    }
  };
  // Work items = (observation, band of steps).  One band per observation when the launch fills the device; a small launch
  // cuts an observation into NB bands (launcher: `pick_bands`) so that its rows are streamed by NB workgroups at once.  A
  // band that does not start at the top runs the step in front of it as a WARM-UP: expansion only, which leaves the 3 - S
  // rows its first step shares with that step in the ring.
  const int NB = a.NB, SPB = (NS + NB - 1) / NB, n_items = a.B * NB;
  auto item_of = [&](int item, int& b_, int& s0_, int& s1_, int& sa_) {
    b_ = item / NB;
    s0_ = (item - b_ * NB) * SPB;
    s1_ = s0_ + SPB < NS ? s0_ + SPB : NS;
    sa_ = s0_ > 0 ? s0_ - 1 : 0;
  };
  int item = blockIdx.x;
  {
    int b_, s0_, s1_, sa_;
    item_of(item < n_items ? item : 0, b_, s0_, s1_, sa_);
    if (item < n_items) request_x(b_, sa_);
  }
  split_x();
#pragma unroll 1
  for (; item < n_items; item += gridDim.x) {
    int b, s0, s1, sa;
    item_of(item, b, s0, s1, sa);
    // the ring slot of input row -1 is a zero row (slot R - 1; the last depthwise of the previous item is behind its barrier)
    if (s0 == 0)
      for (int e = tid; e < PW * LDE / 4; e += 512) reinterpret_cast<u32x4*>(E + (size_t)(R - 1) * PW * LDE)[e] = zero4;
    float* yb = a.y + ((size_t)k * a.B + b) * HOUT * HOUT * COUT;
    const float* xb = a.x + ((size_t)k * a.B + b) * HIN * W * CIN;
    ROWS_TICK(1);
#pragma unroll 1
    for (int s = sa; s < s1; ++s) {
      const bool warm = s < s0;
      // ---------------- expand the input rows s NRI .. s NRI + NRI - 1 -> E ring ----------------
      // (xh / xl hold this step's rows; the next step's — or the next item's first — are requested now)
      if (s + 1 < s1) request_x(b, s + 1);
      else if (item + (int)gridDim.x < n_items) {
        int b_, s0_, s1_, sa_;
        item_of(item + (int)gridDim.x, b_, s0_, s1_, sa_);
        request_x(b_, sa_);
      }
      const int ob = S == 1 ? 2 * s - 1 : 2 * s;  // output rows of this step: ob, ob + 1
      f32x4 res = {0.f, 0.f, 0.f, 0.f};              // the projection's residual, requested a step's length ahead
      const bool pj_on = !warm && pj_wave && pj_j < NOP && ob + pj_orow >= 0 && ob + pj_orow < HOUT && pj_ch < COUT;
      if (a.residual && pj_on) res = *reinterpret_cast<const f32x4*>(xb + ((size_t)(ob + pj_orow) * W + pj_ox) * CIN + pj_ch);  // (CIN == COUT, S == 1)
      int eoff[TIN];
      bool in_map[TIN];
#pragma unroll
      for (int t = 0; t < TIN; ++t) {
        const int row = s * NRI + t_irow[t];
        in_map[t] = row < HIN;
        eoff[t] = t_on[t] ? ((row % R) * PW + t_ix[t] + 1) * LDE : Geo::E_DUMP * LDE;
      }
      // Channel tiles in batches of CB: all operand reads of a batch, then its MFMAs term by term (3 CB TIN independent
      // chains instead of CB serial ones of three dependent MFMAs each — left as one loop the compiler emits read,
      // MFMA, MFMA, MFMA, epilogue, store per tile and nothing overlaps: ~300 cycles per tile), then the epilogues.
      // (A partition's missing last tile recomputes the previous one: the same values stored twice, no branch.)
      constexpr int CB = NCTW > 5 ? (NCTW + 1) / 2 : NCTW;
#pragma unroll
      for (int c0 = 0; c0 < NCTW; c0 += CB) {
        u32x4 ah[CB], al[CB];
        float4 be[CB];
        int ctv[CB];
#pragma unroll
        for (int ci = 0; ci < CB; ++ci) {
          const int ct = ct_lo + c0 + ci < ct_hi ? ct_lo + c0 + ci : ct_hi - 1;
          ctv[ci] = ct;
          ah[ci] = WE[(size_t)(ct * 2) * 64 + lane];
          al[ci] = WE[(size_t)(ct * 2 + 1) * 64 + lane];
          be[ci] = *reinterpret_cast<const float4*>(PB + 16 * ct + 4 * q);
        }
        f32x4 v[CB][TIN];
#pragma unroll
        for (int ci = 0; ci < CB; ++ci)
#pragma unroll
          for (int t = 0; t < TIN; ++t) v[ci][t] = mfmah(al[ci], xh[t], f32x4{0.f, 0.f, 0.f, 0.f});  // small terms first
#pragma unroll
        for (int ci = 0; ci < CB; ++ci)
#pragma unroll
          for (int t = 0; t < TIN; ++t) v[ci][t] = mfmah(ah[ci], xl[t], v[ci][t]);
#pragma unroll
        for (int ci = 0; ci < CB; ++ci)
#pragma unroll
          for (int t = 0; t < TIN; ++t) v[ci][t] = mfmah(ah[ci], xh[t], v[ci][t]);
#pragma unroll
        for (int ci = 0; ci < CB; ++ci)
#pragma unroll
          for (int t = 0; t < TIN; ++t) {
            f32x2 v0 = relu6_2(__builtin_elementwise_fma(f32x2{v[ci][t][0], v[ci][t][1]}, f32x2{W_INV, W_INV}, f32x2{be[ci].x, be[ci].y}));
            f32x2 v1 = relu6_2(__builtin_elementwise_fma(f32x2{v[ci][t][2], v[ci][t][3]}, f32x2{W_INV, W_INV}, f32x2{be[ci].z, be[ci].w}));
            if (!in_map[t]) {  // a row below the map is a zero row of the EXPANDED tensor (the depthwise pads its input)
              v0 = f32x2{0.f, 0.f};
              v1 = f32x2{0.f, 0.f};
            }
            *reinterpret_cast<f32x4*>(E + eoff[t] + 16 * ctv[ci] + 4 * q) = f32x4{v0.x, v0.y, v1.x, v1.y};
          }
      }
      ROWS_TICK(2);
      if (warm) {  // (uniform) the rows are in the ring; the first real step's barrier orders them in front of its depthwise
        split_x();
        continue;
      }
      lds_barrier();
      ROWS_TICK(3);
      // ---------------- depthwise: output rows ob, ob + 1 -> D (hi, lo) ----------------
      int slotv[R];  // ring slot of input row ob S - 1 + i
#pragma unroll
      for (int i = 0; i < R; ++i) slotv[i] = (ob * S - 1 + i + R) % R;
      if (dw_thread) {
#pragma unroll
        for (int jj = 0; jj < JMAX; ++jj) {
          const int j = pslot + jj * PSL;
          if (j >= NOP) break;
          const int orow = j >= HOUT ? 1 : 0, ox = j - orow * HOUT;
          f32x2 s0 = bd[0], s1 = bd[1];
          f32x4 e[9];  // all nine reads first: one LDS latency per pixel, not one per tap row
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int slot = orow ? slotv[S + ky] : slotv[ky];
            const float* r = E + (size_t)(slot * PW + ox * S) * LDE + 4 * cg;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) e[ky * 3 + kx] = *reinterpret_cast<const f32x4*>(r + kx * LDE);
          }
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            s0 = __builtin_elementwise_fma(f32x2{e[t][0], e[t][1]}, wt[t][0], s0);
            s1 = __builtin_elementwise_fma(f32x2{e[t][2], e[t][3]}, wt[t][1], s1);
          }
          const u32x2 p0 = split2(relu6_2(s0)), p1 = split2(relu6_2(s1));
          *reinterpret_cast<u32x2*>(Dh + (size_t)j * LDD + 4 * cg) = u32x2{p0.x, p1.x};
          *reinterpret_cast<u32x2*>(Dl + (size_t)j * LDD + 4 * cg) = u32x2{p0.y, p1.y};
        }
      }
      ROWS_TICK(4);
      lds_barrier();
      ROWS_TICK(5);
      split_x();  // the next step's rows (requested at the top of this step)
      // ---------------- project the two rows -> y ----------------
      if (pj_wave) {
        const int o = ob + pj_orow;
        const bool on = pj_on;
        // one accumulator per term: three independent chains of NKP MFMAs instead of one of 3 NKP dependent ones
        f32x4 acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f}, acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKP; ++ks) {
          const size_t od = (size_t)pj_j * LDD + 32 * ks + 8 * q;
          const u32x4 bh = *reinterpret_cast<const u32x4*>(Dh + od), bl = *reinterpret_cast<const u32x4*>(Dl + od);
          const u32x4 ah = WP[(size_t)((pj_ct * NKP + ks) * 2) * 64 + lane], al = WP[(size_t)((pj_ct * NKP + ks) * 2 + 1) * 64 + lane];
          acc_a = mfmah(al, bh, acc_a);
          acc_b = mfmah(ah, bl, acc_b);
          acc = mfmah(ah, bh, acc);
        }
        acc += acc_a + acc_b;  // (small terms first)
        if (on) {
          const float4 bp = *reinterpret_cast<const float4*>(PB + HID + pj_ch);
          const f32x4 v = {fmaf(acc[0], W_INV, bp.x) + res[0], fmaf(acc[1], W_INV, bp.y) + res[1], fmaf(acc[2], W_INV, bp.z) + res[2],
                           fmaf(acc[3], W_INV, bp.w) + res[3]};
          *reinterpret_cast<f32x4*>(yb + ((size_t)o * HOUT + pj_ox) * COUT + pj_ch) = v;
        }
      }
      ROWS_TICK(6);
      // (no barrier: the next expansion writes ring rows the depthwise above is done with; the next depthwise writes D
      // behind the barrier that follows that expansion, which every wave reaches after this projection)
    }
  }
#ifdef RIP_ROWS_TICKS
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&g_rows_ticks[i], tk[i]);
    atomicAdd(&g_rows_ticks[8], 1ull);
  }
#endif
}

// Row bands per observation: `slots` resident workgroups walk B * bands items; an item costs its steps plus one warm-up step
// when the observation is cut.  The count that minimises rounds x (steps + warm-up); ties go to fewer bands.
int pick_bands(int B, int steps, int slots) {
  int best = 1;
  long best_cost = -1;
  for (int nb = 1; nb <= steps; ++nb) {
    const int spb = (steps + nb - 1) / nb, nbe = (steps + spb - 1) / spb;
    if (nbe != nb) continue;  // (the same cut as a smaller count)
    const long rounds = ((long)B * nbe + slots - 1) / slots;
    const long cost = rounds * (spb + (nbe > 1 ? 1 : 0));
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = nbe;
    }
  }
  return best;
}

template <int HIN, int S, int CIN, int HID, int COUT>
hipError_t launch_rows(RowsArgs a, int kc, hipStream_t s) {
  using Geo = RowsGeom<HIN, S, CIN, HID, COUT>;
  static bool attr_set[64] = {};  // per device: > 64 KB of dynamic LDS needs the opt-in
  auto kern = irb_split_rows_kernel<HIN, S, CIN, HID, COUT>;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Geo::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  // one workgroup per CU is resident (LDS); each walks the (observation, band) items wgx, wgx + gx, ... of its model
  int slots = device_cu_count() / kc;
  if (slots < 1) slots = 1;
  a.NB = pick_bands(a.B, Geo::NS, slots);
  int gx = slots;
  if (gx > a.B * a.NB) gx = a.B * a.NB;
  note_kernel(dim3(gx, 1, kc), dim3(512), "irb_split_rows_kernel<%d,%d,%d,%d,%d> NB=%d", HIN, S, CIN, HID, COUT, a.NB);
  hipLaunchKernelGGL(kern, dim3(gx, 1, kc), dim3(512), Geo::LDS_BYTES, s, a);
#ifdef RIP_ROWS_TICKS
  {
    unsigned long long t[16];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_rows_ticks), sizeof(t));
    const double n = t[8] > 0 ? (double)t[8] : 1.0, imgs = (double)a.B * kc / n, st = imgs * Geo::NS;
    fprintf(stderr, "split rows<%d,%d,%d,%d,%d> cycles per workgroup (%.1f observations x %d steps): prologue %.0f | per observation: zero row %.0f | per step: "
            "expand %.0f barrier %.0f depthwise %.0f barrier %.0f project %.0f\n",
            HIN, S, CIN, HID, COUT, imgs, Geo::NS, t[0] / n, t[1] / n / imgs, t[2] / n / st, t[3] / n / st, t[4] / n / st, t[5] / n / st, t[6] / n / st);
    unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rows_ticks), z, sizeof(z));
  }
#endif
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// The front of the fp32 encoder in the same structure: stem (3x3, stride 2, C -> 32, fp32 on the vector unit) ->
// features.1 (t = 1 block: depthwise 3x3 on 32 channels -> projection 32 -> 16, split-f16 MFMA) in one kernel.
// Rounds 1-5: `stem_kernel` (0.44 ms at 512 observations x 4 models: 0.65 GB written) + `irb_kernel<2,32,false>` (0.90 ms).
// The "expansion" of a step is the stem itself: two rows of its output are computed from five rows of the BEV (staged
// in LDS one step ahead, zero columns either side) straight into the ring the depthwise reads.
// ------------------------------------------------------------------------------------------------------------------
struct FrontArgs {
  const float* visual;  // [B][C][100][100]
  float* y;             // [K][B][50][50][16]
  const float* wbase;
  const h16_t* wfrag;   // features.1's projection fragments of model 0, models wr_stride apart
  size_t model_stride, wr_stride;
  int k0;
  size_t ws_off, bs_off, wd_off, bd_off, bp_off;
  int B;
  int NB;  // row bands per observation (see irb_split_rows_kernel)
};

template <int C>
struct FrontGeom {
  static constexpr int HIN = 100, W = 50, HOUT = 50, PW = W + 2, HID = 32, COUT = 16;
  static constexpr int R = 4, NS = HOUT / 2 + 1;  // step s: stem rows 2 s, 2 s + 1; depthwise / projection of rows 2 s - 1, 2 s
  static constexpr int LDE = HID + 4, LDD = HID + 8;
  static constexpr int NOP = RB * HOUT, NPO = (NOP + 15) / 16, DROWS = NPO * 16;
  static constexpr int RAWW = 4 + HIN + 4, RAWR = 5;  // BEV window: rows 4 s - 1 .. 4 s + 3, column x at index x + 4
  static constexpr size_t E_BYTES = (size_t)R * PW * LDE * sizeof(float);
  static constexpr size_t D_PLANE = (size_t)DROWS * LDD;
  static constexpr size_t D_BYTES = 2 * D_PLANE * sizeof(h16_t);
  static constexpr size_t RAW_BYTES = (size_t)C * RAWR * RAWW * sizeof(float);
  static constexpr size_t LDS_BYTES = E_BYTES + D_BYTES + RAW_BYTES + 2 * 1024 + 16 * sizeof(float);
  static constexpr int NRAW4 = C * RAWR * (HIN / 4);  // float4 pieces of a window
  static_assert(NRAW4 <= 512 && NPO <= 8, "one request per thread, one projection tile per wave");
};

template <int C>
__global__ __launch_bounds__(512) void front_split_kernel(FrontArgs a) {
  using Geo = FrontGeom<C>;
  constexpr int HIN = Geo::HIN, W = Geo::W, HOUT = Geo::HOUT, PW = Geo::PW, HID = Geo::HID, COUT = Geo::COUT, R = Geo::R, NS = Geo::NS;
  constexpr int LDE = Geo::LDE, LDD = Geo::LDD, NOP = Geo::NOP, NPO = Geo::NPO, RAWW = Geo::RAWW, RAWR = Geo::RAWR;
  constexpr int NCG = HID / 4, PSL = 512 / NCG, JMAX = (NOP + PSL - 1) / PSL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* const E = reinterpret_cast<float*>(smem_raw);                                  // [R][PW][LDE]
  h16_t* const Dh = reinterpret_cast<h16_t*>(smem_raw + Geo::E_BYTES);                  // [DROWS][LDD] hi
  h16_t* const Dl = Dh + Geo::D_PLANE;
  float* const RAW = reinterpret_cast<float*>(smem_raw + Geo::E_BYTES + Geo::D_BYTES);  // [C][RAWR][RAWW]
  u32x4* const WP = reinterpret_cast<u32x4*>(smem_raw + Geo::E_BYTES + Geo::D_BYTES + Geo::RAW_BYTES);  // [2][64]
  float* const PB = reinterpret_cast<float*>(WP + 128);                                  // bp [16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int k = blockIdx.z;
  const float* Wf = a.wbase + (size_t)(a.k0 + k) * a.model_stride;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#ifdef RIP_ROWS_TICKS
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
  for (int e = tid; e < (int)(Geo::E_BYTES / 16); e += 512) reinterpret_cast<u32x4*>(E)[e] = zero4;
  for (int e = tid; e < (int)(Geo::D_BYTES / 16); e += 512) reinterpret_cast<u32x4*>(Dh)[e] = zero4;
  for (int e = tid; e < (int)(Geo::RAW_BYTES / 16); e += 512) reinterpret_cast<u32x4*>(RAW)[e] = zero4;
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.wfrag + (size_t)(a.k0 + k) * a.wr_stride);
    if (tid < 128) WP[tid] = src[tid];
    if (tid < 16) PB[tid] = Wf[a.bp_off + tid];
  }
  // this thread's four channels (of the stem's 32 outputs = the depthwise's 32 channels): stem taps, depthwise taps, biases
  const int cg = tid % NCG, pslot = tid / NCG;
  f32x2 ws[9 * C][2], bs[2], wt[9][2], bd[2];
#pragma unroll
  for (int t = 0; t < 9 * C; ++t) {
    const float4 w0 = *reinterpret_cast<const float4*>(Wf + a.ws_off + (size_t)t * 32 + 4 * cg);  // [(ky 3 + kx) C + c][32]
    ws[t][0] = f32x2{w0.x, w0.y};
    ws[t][1] = f32x2{w0.z, w0.w};
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float4 w0 = *reinterpret_cast<const float4*>(Wf + a.wd_off + (size_t)t * HID + 4 * cg);
    wt[t][0] = f32x2{w0.x, w0.y};
    wt[t][1] = f32x2{w0.z, w0.w};
  }
  {
    const float4 b0 = *reinterpret_cast<const float4*>(Wf + a.bs_off + 4 * cg), b1 = *reinterpret_cast<const float4*>(Wf + a.bd_off + 4 * cg);
    bs[0] = f32x2{b0.x, b0.y};
    bs[1] = f32x2{b0.z, b0.w};
    bd[0] = f32x2{b1.x, b1.y};
    bd[1] = f32x2{b1.z, b1.w};
  }
  // BEV window of step s: thread tid < NRAW4 owns the float4 (channel c, window row r, columns 4 i ..)
  const int rw_c = tid / (RAWR * (HIN / 4)), rw_r = (tid / (HIN / 4)) % RAWR, rw_i = tid % (HIN / 4);
  f32x4 rawv = {0.f, 0.f, 0.f, 0.f};
  auto request_raw = [&](int b, int s) {
    const int row = 4 * s - 1 + rw_r;
    rawv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < Geo::NRAW4 && row >= 0 && row < HIN)
      rawv = *reinterpret_cast<const f32x4*>(a.visual + (((size_t)b * C + rw_c) * HIN + row) * HIN + 4 * rw_i);
  };
  auto publish_raw = [&]() {
    if (tid < Geo::NRAW4) *reinterpret_cast<f32x4*>(RAW + ((size_t)rw_c * RAWR + rw_r) * RAWW + 4 + 4 * rw_i) = rawv;
  };
  // projection role: wave w owns output pixel tile w
  const bool pj_wave = w < NPO;
  const int pj_j = 16 * w + n, pj_orow = pj_j / HOUT, pj_ox = pj_j - pj_orow * HOUT;
  lds_barrier();
  ROWS_TICK(0);

  // (observation, band) items and warm-up steps as in irb_split_rows_kernel; the BEV windows follow the sequence of
  // EXECUTED steps across items: `rq` = the step whose window is in flight (two steps ahead of the one that runs)
  const int NB = a.NB, SPB = (NS + NB - 1) / NB, n_items = a.B * NB;
  auto item_of = [&](int item_, int& b_, int& s0_, int& s1_, int& sa_) {
    b_ = item_ / NB;
    s0_ = (item_ - b_ * NB) * SPB;
    s1_ = s0_ + SPB < NS ? s0_ + SPB : NS;
    sa_ = s0_ > 0 ? s0_ - 1 : 0;
  };
  int rq_item = blockIdx.x, rq_s = 0;
  auto rq_request = [&]() {  // request the window of step (rq_item, rq_s), if there is one
    if (rq_item < n_items) request_raw(rq_item / NB, rq_s);
  };
  auto rq_advance = [&]() {
    if (rq_item >= n_items) return;
    int b_, s0_, s1_, sa_;
    item_of(rq_item, b_, s0_, s1_, sa_);
    if (rq_s + 1 < s1_) {
      ++rq_s;
    } else {
      rq_item += (int)gridDim.x;
      if (rq_item < n_items) {
        item_of(rq_item, b_, s0_, s1_, sa_);
        rq_s = sa_;
      }
    }
  };
  {
    int b_, s0_, s1_, sa_;
    item_of(rq_item < n_items ? rq_item : 0, b_, s0_, s1_, sa_);
    rq_s = sa_;
    rq_request();
    publish_raw();
    rq_advance();
    rq_request();
  }
  lds_barrier();
#pragma unroll 1
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    int b, s0b, s1b, sab;
    item_of(item, b, s0b, s1b, sab);
    if (s0b == 0)
      for (int e = tid; e < PW * LDE / 4; e += 512) reinterpret_cast<u32x4*>(E + (size_t)(R - 1) * PW * LDE)[e] = zero4;  // stem row -1
    float* yb = a.y + ((size_t)k * a.B + b) * HOUT * HOUT * COUT;
    ROWS_TICK(1);
#pragma unroll 1
    for (int s = sab; s < s1b; ++s) {
      const bool warm = s < s0b;
      // ---------------- stem rows 2 s, 2 s + 1 -> E ring (rows >= 50: zero rows) ----------------
#pragma unroll
      for (int jj = 0; jj < JMAX; ++jj) {
        const int j = pslot + jj * PSL;
        if (j >= NOP) break;
        const int r = j >= W ? 1 : 0, ox = j - r * W, row = 2 * s + r;
        f32x2 s0 = bs[0], s1 = bs[1];
        float rv[9 * C];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const float* rp = RAW + ((size_t)c * RAWR + 2 * r + ky) * RAWW + 2 * ox + 3;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) rv[(ky * 3 + kx) * C + c] = rp[kx];
          }
#pragma unroll
        for (int t = 0; t < 9 * C; ++t) {  // (tap order (ky, kx) outer, channel inner: the layer-wise kernel's is channel outer — fp32 rounding order only)
          s0 = __builtin_elementwise_fma(f32x2{rv[t], rv[t]}, ws[t][0], s0);
          s1 = __builtin_elementwise_fma(f32x2{rv[t], rv[t]}, ws[t][1], s1);
        }
        s0 = relu6_2(s0);
        s1 = relu6_2(s1);
        if (row >= HOUT) {
          s0 = f32x2{0.f, 0.f};
          s1 = f32x2{0.f, 0.f};
        }
        *reinterpret_cast<f32x4*>(E + (size_t)((row % R) * PW + ox + 1) * LDE + 4 * cg) = f32x4{s0.x, s0.y, s1.x, s1.y};
      }
      ROWS_TICK(2);
      lds_barrier();
      ROWS_TICK(3);
      // the next executed step's BEV window (requested a step ago) into LDS, the one after it requested
      publish_raw();
      rq_advance();
      rq_request();
      if (warm) {  // (uniform) a band's warm-up: its stem rows are in the ring, no output rows yet
        lds_barrier();
        continue;
      }
      // ---------------- depthwise: output rows 2 s - 1, 2 s -> D (hi, lo) ----------------
      const int ob = 2 * s - 1;
      int slotv[R];
#pragma unroll
      for (int i = 0; i < R; ++i) slotv[i] = (ob - 1 + i + R) % R;
#pragma unroll
      for (int jj = 0; jj < JMAX; ++jj) {
        const int j = pslot + jj * PSL;
        if (j >= NOP) break;
        const int orow = j >= HOUT ? 1 : 0, ox = j - orow * HOUT;
        f32x2 s0 = bd[0], s1 = bd[1];
        f32x4 e[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int slot = orow ? slotv[1 + ky] : slotv[ky];
          const float* rp = E + (size_t)(slot * PW + ox) * LDE + 4 * cg;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) e[ky * 3 + kx] = *reinterpret_cast<const f32x4*>(rp + kx * LDE);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          s0 = __builtin_elementwise_fma(f32x2{e[t][0], e[t][1]}, wt[t][0], s0);
          s1 = __builtin_elementwise_fma(f32x2{e[t][2], e[t][3]}, wt[t][1], s1);
        }
        const u32x2 p0 = split2(relu6_2(s0)), p1 = split2(relu6_2(s1));
        *reinterpret_cast<u32x2*>(Dh + (size_t)j * LDD + 4 * cg) = u32x2{p0.x, p1.x};
        *reinterpret_cast<u32x2*>(Dl + (size_t)j * LDD + 4 * cg) = u32x2{p0.y, p1.y};
      }
      ROWS_TICK(4);
      lds_barrier();
      ROWS_TICK(5);
      // ---------------- project the two rows -> y ----------------
      if (pj_wave) {
        const int o = ob + pj_orow;
        const size_t od = (size_t)pj_j * LDD + 8 * q;
        const u32x4 bh = *reinterpret_cast<const u32x4*>(Dh + od), bl = *reinterpret_cast<const u32x4*>(Dl + od);
        const u32x4 ah = WP[lane], al = WP[64 + lane];
        f32x4 acc = mfmah(al, bh, f32x4{0.f, 0.f, 0.f, 0.f});
        acc = mfmah(ah, bl, acc);
        acc = mfmah(ah, bh, acc);
        if (pj_j < NOP && o >= 0 && o < HOUT) {
          const float4 bp = *reinterpret_cast<const float4*>(PB + 4 * q);
          *reinterpret_cast<f32x4*>(yb + ((size_t)o * HOUT + pj_ox) * COUT + 4 * q) =
              f32x4{fmaf(acc[0], W_INV, bp.x), fmaf(acc[1], W_INV, bp.y), fmaf(acc[2], W_INV, bp.z), fmaf(acc[3], W_INV, bp.w)};
        }
      }
      ROWS_TICK(6);
    }
  }
#ifdef RIP_ROWS_TICKS
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&g_rows_ticks[i], tk[i]);
    atomicAdd(&g_rows_ticks[8], 1ull);
  }
#endif
}

struct RowsShape {
  int hin, stride, cin, hid, cout;
};
constexpr RowsShape ROWS_SHAPES[] = {{50, 2, 16, 96, 24}, {25, 1, 24, 144, 24}, {25, 2, 24, 144, 32},
                                     {13, 1, 32, 192, 32}, {13, 2, 32, 192, 64}};

int rows_shape_index(const Layer* le, const Layer& ld, const Layer& lp) {
  if (le == nullptr) return -1;
  for (int i = 0; i < (int)(sizeof(ROWS_SHAPES) / sizeof(ROWS_SHAPES[0])); ++i) {
    const RowsShape& r = ROWS_SHAPES[i];
    if (ld.h_in == r.hin && ld.stride == r.stride && le->cin == r.cin && ld.cout == r.hid && lp.cout == r.cout) return i;
  }
  return -1;
}

// operand fragments of one block: halves per model
size_t rows_frag_halves(const Layer* le, const Layer& ld, const Layer& lp) {
  const int nfe = le != nullptr ? ld.cout / 16 * 2 : 0, nfp = ((lp.cout + 15) / 16) * ((ld.cout + 31) / 32) * 2;
  return (size_t)(nfe + nfp) * 512;
}

template <int C>
hipError_t launch_front(FrontArgs a, int kc, hipStream_t s) {
  using Geo = FrontGeom<C>;
  static bool attr_set[64] = {};
  auto kern = front_split_kernel<C>;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Geo::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  int slots = device_cu_count() / kc;
  if (slots < 1) slots = 1;
  a.NB = pick_bands(a.B, Geo::NS, slots);
  int gx = slots;
  if (gx > a.B * a.NB) gx = a.B * a.NB;
  note_kernel(dim3(gx, 1, kc), dim3(512), "front_split_kernel<%d> NB=%d", C, a.NB);
  hipLaunchKernelGGL(kern, dim3(gx, 1, kc), dim3(512), Geo::LDS_BYTES, s, a);
#ifdef RIP_ROWS_TICKS
  {
    unsigned long long t[16];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_rows_ticks), sizeof(t));
    const double n = t[8] > 0 ? (double)t[8] : 1.0, imgs = (double)a.B * kc / n, st = imgs * Geo::NS;
    fprintf(stderr, "split front<%d> cycles per workgroup (%.1f observations x %d steps): prologue %.0f | per observation: zero row %.0f | per step: "
            "stem %.0f barrier %.0f depthwise %.0f barrier %.0f project %.0f\n",
            C, imgs, Geo::NS, t[0] / n, t[1] / n / imgs, t[2] / n / st, t[3] / n / st, t[4] / n / st, t[5] / n / st, t[6] / n / st);
    unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rows_ticks), z, sizeof(z));
  }
#endif
  return hipGetLastError();
}

}  // namespace

bool irb_split_rows_supported(const Layer* le, const Layer& ld, const Layer& lp) { return rows_shape_index(le, ld, lp) >= 0; }

bool front_split_supported(const Layer& ls, const Layer& ld, const Layer& lp) {
  return ls.kind == L_STEM && ls.cin == 2 && ls.cout == 32 && ls.h_in == 100 && ls.stride == 2 && ld.kind == L_DW && ld.cout == 32 &&
         ld.h_in == 50 && ld.stride == 1 && lp.cin == 32 && lp.cout == 16 && !lp.residual;
}

SplitRowsLayout split_rows_layout(const EncoderPlan& plan) {
  SplitRowsLayout L;
  L.off.assign(plan.blocks.size(), (size_t)-1);
  size_t off = 0;
  for (size_t bi = 0; bi < plan.blocks.size(); ++bi) {
    const FusedBlock& fb = plan.blocks[bi];
    const Layer* le = fb.expand >= 0 ? &plan.layers[fb.expand] : nullptr;
    const bool front = bi == 0 && fb.expand < 0 && fb.dw == 1 && front_split_supported(plan.layers[0], plan.layers[fb.dw], plan.layers[fb.project]);
    if (!front && !irb_split_rows_supported(le, plan.layers[fb.dw], plan.layers[fb.project])) continue;
    L.off[bi] = off;
    off += rows_frag_halves(le, plan.layers[fb.dw], plan.layers[fb.project]);
  }
  L.total = off;
  return L;
}

// One model's folded fp32 blob -> the operand fragments of every supported block (binary16 bit patterns):
// expansion fragment (ct, term): lane (n, q), element j = We[16 ct + n][8 q + j] (0 beyond CIN);
// projection fragment (ct, ks, term): Wp[16 ct + n][32 ks + 8 q + j] (0 beyond COUT / HID); values are w 2^8 split into
// hi = f16(.), lo = f16(. - hi).
void pack_split_rows(const EncoderPlan& plan, const SplitRowsLayout& L, const float* enc, unsigned short* out) {
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
    const Layer &ld = plan.layers[fb.dw], &lp = plan.layers[fb.project];
    const Layer& le = plan.layers[fb.expand >= 0 ? fb.expand : fb.dw];  // (the front's block has no expansion: no expansion fragments)
    const int cin = le.cin, hid = ld.cout, cout = lp.cout;
    const int ncte = fb.expand >= 0 ? hid / 16 : 0, nctp = (cout + 15) / 16, nkp = (hid + 31) / 32;
    size_t o = L.off[bi];
    for (int ct = 0; ct < ncte; ++ct)
      for (int term = 0; term < 2; ++term)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int row = 16 * ct + (lane & 15), kk = 8 * (lane >> 4) + j;
            put(o++, kk < cin ? enc[le.w_off + (size_t)row * cin + kk] : 0.f, term);
          }
    for (int ct = 0; ct < nctp; ++ct)
      for (int ks = 0; ks < nkp; ++ks)
        for (int term = 0; term < 2; ++term)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int row = 16 * ct + (lane & 15), kk = 32 * ks + 8 * (lane >> 4) + j;
              put(o++, (row < cout && kk < hid) ? enc[lp.w_off + (size_t)row * hid + kk] : 0.f, term);
            }
  }
}

hipError_t launch_front_split(const Layer& ls, const Layer& ld, const Layer& lp, const float* enc_w, const unsigned short* wfrag,
                              size_t wr_stride, size_t model_stride, int k0, int kc, int B, const float* visual, float* y,
                              hipStream_t s) {
  if (!front_split_supported(ls, ld, lp)) return hipErrorInvalidValue;
  FrontArgs a;
  a.visual = visual;
  a.y = y;
  a.wbase = enc_w;
  a.wfrag = wfrag;
  a.model_stride = model_stride;
  a.wr_stride = wr_stride;
  a.k0 = k0;
  a.ws_off = ls.w_off;
  a.bs_off = ls.b_off;
  a.wd_off = ld.w_off;
  a.bd_off = ld.b_off;
  a.bp_off = lp.b_off;
  a.B = B;
  a.NB = 1;
  return launch_front<2>(a, kc, s);
}

hipError_t launch_irb_split_rows(const Layer* le, const Layer& ld, const Layer& lp, const float* enc_w, const unsigned short* wfrag,
                                 size_t wr_stride, size_t model_stride, int k0, int kc, int B, const float* x, float* y,
                                 hipStream_t s) {
  RowsArgs a;
  a.x = x;
  a.y = y;
  a.wbase = enc_w;
  a.wfrag = wfrag;
  a.model_stride = model_stride;
  a.wr_stride = wr_stride;
  a.k0 = k0;
  a.be_off = le->b_off;
  a.wd_off = ld.w_off;
  a.bd_off = ld.b_off;
  a.bp_off = lp.b_off;
  a.B = B;
  a.residual = lp.residual;
  a.NB = 1;
  switch (rows_shape_index(le, ld, lp)) {
    case 0: return launch_rows<50, 2, 16, 96, 24>(a, kc, s);    // features.2
    case 1: return launch_rows<25, 1, 24, 144, 24>(a, kc, s);   // features.3
    case 2: return launch_rows<25, 2, 24, 144, 32>(a, kc, s);   // features.4
    case 3: return launch_rows<13, 1, 32, 192, 32>(a, kc, s);   // features.5, 6
    case 4: return launch_rows<13, 2, 32, 192, 64>(a, kc, s);   // features.7
  }
  return hipErrorInvalidValue;
}

}  // namespace rip
