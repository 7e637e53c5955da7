// Fused bf16 inverted-residual block for the small-image stages (features.8 .. features.17: 7x7 and 4x4 maps) on
// gfx950: expand 1x1 (MFMA) -> depthwise 3x3 -> project 1x1 (MFMA) (+ residual) in one kernel.
//
// torchvision v0.6.0 `InvertedResidual` (reference call site oatomobile/torch/networks/perception.py:36-51), BN
// folded.  Layer by layer these ten blocks are 30 launches that write and re-read the 6x expanded tensor three times
// (features.8 at 512 observations x 4 models: 77 MB each way per layer for 13 MB of block input); they were half of
// the bf16 encoder's time.  Here the expanded tensor only ever exists as 64-channel slices in LDS.
//
// Decomposition (the row-streaming kernels of encoder_bf16_irb2.hip do not fit: a 7x7 map has no rows to stream):
// a workgroup owns G whole observations of one model (G*H*H pixel rows) and walks the hidden dimension in chunks of
// 64 channels.  Its 8 waves are SPECIALISED and pipelined over the chunks, one barrier per step:
//   waves 0-3 (matrix waves), step s:  expand chunk s    -> E[s & 1]      project chunk s-2  <- D[s & 1]
//   waves 4-7 (vector waves), step s:  depthwise chunk s-1: E[(s-1) & 1] -> D[(s-1) & 1]
// so each SIMD hosts one matrix wave and one vector wave: the depthwise (the VALU-bound part: 9 fp32 FMAs per output,
// bf16 unpacking, ReLU6, packing) runs under the other wave's MFMAs instead of in a phase of its own, and neither role
// carries the other's registers (the matrix waves hold the block input as B operands, the projection accumulators and
// the pointwise weights; the vector waves two sets of 72 fp32 taps, so the next chunk's taps load a whole step ahead).
//   expand: wave w owns the 16-pixel tiles w, w+4, ...; B = block input (K = CIN), resident in registers for all
//     chunks; A = the chunk's 64 x CIN weight rows (requested one step ahead).  bias + ReLU6 + bf16 -> E.
//   E is stored with a zero column either side of every image row, so the depthwise has no border predicates; rows
//     above / below the map are skipped at compile time.
//   depthwise: thread = (observation, output column, 8-channel group) walks DOWN its column: each input row is read
//     (3 pixels) and unpacked once and scattered into the <= 3 output rows it belongs to.  bias + ReLU6 + bf16 -> D.
//   project: wave w owns output tiles w, w+4, ...: one B read from D feeds all COUT/16 channel tiles; accumulators
//     persist in registers across the chunks.
// Epilogue: bias (+ residual from the block input) -> bf16.  Arithmetic order is the layer-wise kernels': MFMA chain
// over ascending K from zero, then bias; depthwise = bias, then taps in (ky, kx) order; RNE rounding at the same points.
#include <cstdio>

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

// development only (tools/dev/tile_abl.sh rebuilds with -DRIP_TILE_ABL=<bits>; wrong results): 1 = no depthwise,
// 2 = no matrix work, 4 = no tap loads, 8 = no LDS zeroing, 16 = no epilogue stores (the compiler then drops the matrix
// work as well), 32 = pointwise weights of chunk 0 only, 64 = epilogue stores predicated off at run time
#ifndef RIP_TILE_ABL
#define RIP_TILE_ABL 0
#endif

#ifdef RIP_TILE_TICKS  // development (tools/dev/tile_ticks.sh): shader cycles per phase of matrix wave 0 / vector wave 4, summed over the workgroups
__device__ unsigned long long g_tile_ticks[16];
#define TILE_TICK(slot_)                                            \
  do {                                                              \
    const unsigned long long now_ = __builtin_readcyclecounter();   \
    tk[slot_] += now_ - tlast;                                      \
    tlast = now_;                                                   \
  } while (0)
#else
#define TILE_TICK(slot_) do { } while (0)
#endif

constexpr int HC = 64;        // hidden channels per chunk
constexpr int LD = HC + 8;    // bf16 elements per LDS pixel row (odd multiple of 16 bytes)

struct TileArgs {
  const bf16_t* x;       // [K][B][HIN][HIN][CIN]
  bf16_t* y;             // [K][B][HOUT][HOUT][COUT]
  const float* wbase;    // fp32 folded blobs (biases, depthwise taps)
  const bf16_t* whbase;  // bf16 copy of the blobs (pointwise weights), same offsets
  size_t model_stride;
  int k0;
  size_t we_off, be_off, wd_off, bd_off, wp_off, bp_off;
  int B, HID, residual, G;
};

template <int HIN, int STRIDE, int G>
struct TileGeom {
  static constexpr int HOUT = STRIDE == 1 ? HIN : (HIN + 1) / 2;
  static constexpr int HWI = HIN * HIN, HWO = HOUT * HOUT;
  static constexpr int PW = HIN + 2;                                  // padded row: zero, HIN pixels, zero
  static constexpr int E_DUMP = G * HIN * PW;                         // row that takes the stores of lanes without a pixel
  static constexpr int E_ROWS = E_DUMP + 1;                            // per E buffer
  static constexpr int D_ROWS = ((G * HWO + 15) / 16) * 16;           // per D buffer
  static constexpr int TIN = (G * HWI + 63) / 64, TOUT = (G * HWO + 63) / 64;  // 16-pixel tiles per matrix wave
  static constexpr size_t ED_BYTES = (size_t)2 * (E_ROWS + D_ROWS) * LD * sizeof(bf16_t);
  // + the block's depthwise taps [9][HID] and biases [HID] (fp32): every vector thread of every workgroup used to fetch
  // its 18 + 2 float4 per chunk from global memory — 73 KB per step through the CU's texture path, 10 us of a 45 us kernel
  static constexpr size_t lds_bytes(int hid) { return ED_BYTES + (size_t)10 * hid * sizeof(float); }
};

// G: observations per workgroup at most (G * HOUT * 8 <= 256 depthwise threads); a.G <= G is what the host chose.
// AEF / APF: the expansion / projection weights of the next step are requested a step ahead (register budget
// permitting: KSX * 16 / NCT * 8 more live registers); otherwise they are requested where the phase starts and the
// wave's stall is covered by the vector wave on the same SIMD.
// WCH: the four matrix waves split as (4 / WCH pixel partitions) x (WCH channel partitions).  WCH = 2 for the 4x4
// blocks: a wave then streams HALF of a chunk's weights for twice as many pixel tiles (the weight loads are what
// bounds those blocks: K = 160 / 960 against 128 pixels per workgroup).
template <int HIN, int STRIDE, int CIN, int COUT, int G, bool AEF, bool APF, int WCH = 1>
__global__ __launch_bounds__(512) void irb_tile_bf16_kernel(TileArgs a) {
  using Geo = TileGeom<HIN, STRIDE, G>;
  constexpr int HOUT = Geo::HOUT, HWI = Geo::HWI, HWO = Geo::HWO, PW = Geo::PW;
  constexpr int WP = 4 / WCH;  // pixel partitions of the matrix waves
  constexpr int TIN = (Geo::TIN * 4 + WP - 1) / WP, TOUT = (Geo::TOUT * 4 + WP - 1) / WP;
  constexpr int KSX = CIN / 32, NCT = COUT / 16 / WCH, NHT = HC / 16 / WCH, NKP = HC / 32;
  static_assert((COUT / 16) % WCH == 0 && (HC / 16) % WCH == 0, "channel partitions");
  constexpr int NPT_IN = (G * HWI + 15) / 16, NPT_OUT = (G * HWO + 15) / 16;  // 16-pixel tiles of a full group
  constexpr int CTG = NCT > 10 ? 10 : NCT;  // channel tiles per projection pass (weights of one pass are live at a time)
  static_assert(NCT % CTG == 0 && (CTG == NCT || !APF), "projection passes");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* const Ebuf = reinterpret_cast<bf16_t*>(smem_raw);              // [2][E_ROWS][LD]
  bf16_t* const Dbuf = Ebuf + (size_t)2 * Geo::E_ROWS * LD;              // [2][D_ROWS][LD]
  float* const Tl = reinterpret_cast<float*>(smem_raw + Geo::ED_BYTES);   // [9][HID] taps, [HID] biases
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = blockIdx.z;
  const int HID = a.HID;
  const int nch = HID / HC;
  const float* W = a.wbase + (size_t)(a.k0 + k) * a.model_stride;
  const bf16_t* Wh = a.whbase + (size_t)(a.k0 + k) * a.model_stride;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // PERSISTENT over observation groups (round 5): one workgroup per CU is resident (LDS), so a large launch used to be
  // two rounds of workgroups; now a workgroup walks the groups blockIdx.x, + gridDim.x, ... of its model: the LDS
  // prologue (taps, zeroing: 4-8 k cycles of a 50-140 k cycle group, tools/dev/tile_ticks.sh) is paid once, and a
  // group's output stores drain under the next group's operand loads: encoder 1.445 -> 1.42 ms.  (Also built: the
  // (group, chunk) sequence flattened into ONE pipeline — the next group's expansions under the previous group's last
  // two projections.  It needs the next group's block input live during the epilogue; with the per-lane offsets behind
  // opaque copies it compiles without scratch for the 7x7 blocks and measures the same as this loop (+-2 us per block),
  // while features.17 keeps 52 bytes of scratch with reloads inside the step loop: 65 -> 103 us.  Not shipped.)
  const int n_groups = (a.B + a.G - 1) / a.G;
  const int ng = (n_groups - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;  // (the launcher: gridDim.x <= n_groups)
  int img0 = 0, m_in = 0, m_out = 0;  // the current group (set by enter_group)
  const bf16_t* xg = nullptr;
  bf16_t* yg = nullptr;
  auto enter_group = [&](int j) {
    img0 = ((int)blockIdx.x + j * (int)gridDim.x) * a.G;
    const int n_img = min(a.G, a.B - img0);
    m_in = n_img * HWI;
    m_out = n_img * HWO;
    xg = a.x + ((size_t)k * a.B + img0) * HWI * CIN;
    yg = a.y + ((size_t)k * a.B + img0) * HWO * COUT;
  };

#ifdef RIP_TILE_TICKS
  unsigned long long tk[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
  // zero both E buffers once: the padding columns are never written again
  if (!(RIP_TILE_ABL & 8))
    for (int e = tid; e < 2 * Geo::E_ROWS * LD / 8; e += 512) reinterpret_cast<u32x4*>(Ebuf)[e] = zero4;
  {
    const int HIDs = a.HID;
    const float* Wk = a.wbase + (size_t)(a.k0 + blockIdx.z) * a.model_stride;
    for (int e = tid; e < 9 * HIDs / 4; e += 512)
      reinterpret_cast<float4*>(Tl)[e] = *reinterpret_cast<const float4*>(Wk + a.wd_off + 4 * e);
    for (int e = tid; e < HIDs / 4; e += 512)
      reinterpret_cast<float4*>(Tl + 9 * HIDs)[e] = *reinterpret_cast<const float4*>(Wk + a.bd_off + 4 * e);
  }
  lds_barrier();
  TILE_TICK(0);  // prologue: LDS zeroing, taps

  if (w < 4) {
    // ================= matrix waves: expand + project =================
    const int n = lane & 15, q = lane >> 4;
    const int wpix = w / WCH, wch = w % WCH;  // this wave's pixel partition / channel partition
    const int ht0 = wch * NHT, ct0w = wch * NCT;
    u32x4 xb[TIN][KSX];   // block input of the current group, B operands
    int erow[TIN];        // padded E row of this lane's pixel per tile (the dump row beyond the group's pixels)
    f32x4 acc[TOUT][NCT];

    u32x4 ae[NHT][KSX], ap[CTG][NKP];
    float4 be[NHT];
    // (the row pointers of a weight request are chunk-invariant per lane: left alone they are hoisted above the group
    // loop — one 64-bit pointer per channel tile, 16 * HID elements apart, beyond any immediate offset — and spill;
    // behind an opaque copy of the lane coordinates they cost a handful of integer instructions per request)
    auto load_ae = [&](int c) {  // expand weights / bias of chunk c
      int n_ = n, q_ = q;
      asm volatile("" : "+v"(n_), "+v"(q_));
#pragma unroll
      for (int ht = 0; ht < NHT; ++ht) {
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks)
          ae[ht][ks] = *reinterpret_cast<const u32x4*>(Wh + a.we_off + (size_t)(c * HC + 16 * (ht0 + ht) + n_) * CIN + 32 * ks + 8 * q_);
        be[ht] = *reinterpret_cast<const float4*>(W + a.be_off + c * HC + 16 * (ht0 + ht) + 4 * q_);
      }
    };
    auto load_ap = [&](int c, int ct0) {  // projection weights of chunk c, channel tiles ct0 .. ct0 + CTG - 1
      int n_ = n, q_ = q;
      asm volatile("" : "+v"(n_), "+v"(q_));
      const bf16_t* wp = Wh + a.wp_off + (size_t)(16 * (ct0w + ct0) + n_) * HID + c * HC + 8 * q_;
#pragma unroll
      for (int ct = 0; ct < CTG; ++ct)
#pragma unroll
        for (int ks = 0; ks < NKP; ++ks) ap[ct][ks] = *reinterpret_cast<const u32x4*>(wp + (size_t)16 * ct * HID + 32 * ks);
    };
    // Both phases are software-pipelined by hand across pixel tiles: the MFMAs of tile t + 1 are issued before the
    // epilogue (expansion) / behind the operand reads (projection) of tile t.  Left to itself the compiler put every
    // chain into ONE accumulator tuple: MFMA pair, wait for the result, ten VALU instructions, next pair — the matrix
    // waves (one per SIMD, nothing else to issue) ran at a quarter of their MFMA time.
    auto tile_on = [&](int t, int npt) { return !(WP * t + WP - 1 >= npt && wpix + WP * t >= npt); };  // compile-time but for the last t
    auto expand = [&](int c) {  // chunk c -> E[c & 1]
      bf16_t* E = Ebuf + (size_t)(c & 1) * Geo::E_ROWS * LD;
      if (!AEF && !((RIP_TILE_ABL & 32) && c > 0)) load_ae(c);
      // no run-time branch per tile (a ragged last group's missing pixels are computed on zero operands and stored to
      // the dump row; only tiles no FULL group has are skipped, which is known at compile time except for the last t)
      f32x4 v[2][NHT];
      auto mm = [&](int t, f32x4(&o)[NHT]) {
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) o[ht] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
          for (int ht = 0; ht < NHT; ++ht)
            o[ht] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ae[ht][ks]), as_bf16x8(xb[t][ks]), o[ht], 0, 0, 0);
      };
      auto epi = [&](int t, const f32x4(&o)[NHT]) {
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht) {
          u32x2 w2;
          w2.x = pack_bf16(relu6_2(f32x2{o[ht][0] + be[ht].x, o[ht][1] + be[ht].y}));
          w2.y = pack_bf16(relu6_2(f32x2{o[ht][2] + be[ht].z, o[ht][3] + be[ht].w}));
          *reinterpret_cast<u32x2*>(E + (size_t)erow[t] * LD + 16 * (ht0 + ht) + 4 * q) = w2;
        }
      };
      if (tile_on(0, NPT_IN)) mm(0, v[0]);
#pragma unroll
      for (int t = 0; t < TIN; ++t) {
        if (t + 1 < TIN && tile_on(t + 1, NPT_IN)) mm(t + 1, v[(t + 1) & 1]);
        if (tile_on(t, NPT_IN)) epi(t, v[t & 1]);
      }
      if (AEF && c + 1 < nch && !(RIP_TILE_ABL & 32)) load_ae(c + 1);  // lands during the projection and the barrier
    };
    auto project = [&](int c) {  // chunk c <- D[c & 1]
      const bf16_t* D = Dbuf + (size_t)(c & 1) * Geo::D_ROWS * LD;
#pragma unroll
      for (int cg = 0; cg < NCT; cg += CTG) {
        if (!APF && !((RIP_TILE_ABL & 32) && c > 0)) load_ap(c, cg);
        u32x4 bv[2][NKP];  // (a ragged group's missing pixels read rows of D nobody wrote: never stored)
        auto rd = [&](int t, u32x4(&o)[NKP]) {
#pragma unroll
          for (int ks = 0; ks < NKP; ++ks)
            o[ks] = *reinterpret_cast<const u32x4*>(D + (size_t)(16 * (wpix + WP * t) + n) * LD + 32 * ks + 8 * q);
        };
        if (tile_on(0, NPT_OUT)) rd(0, bv[0]);
#pragma unroll
        for (int t = 0; t < TOUT; ++t) {
          if (t + 1 < TOUT && tile_on(t + 1, NPT_OUT)) rd(t + 1, bv[(t + 1) & 1]);
          if (!tile_on(t, NPT_OUT)) continue;
#pragma unroll
          for (int ks = 0; ks < NKP; ++ks)
#pragma unroll
            for (int ct = 0; ct < CTG; ++ct)
              acc[t][cg + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ap[ct][ks]), as_bf16x8(bv[t & 1][ks]), acc[t][cg + ct], 0, 0, 0);
        }
      }
      if (APF && c + 1 < nch && !(RIP_TILE_ABL & 32)) load_ap(c + 1, 0);  // projected in the next step
    };
    const bool mx_on = !(RIP_TILE_ABL & 2);
#pragma unroll 1
    for (int j = 0; j < ng; ++j) {  // this workgroup's observation groups
    enter_group(j);
    // (lane coordinates re-derived behind an opaque copy per group and again for the epilogue: otherwise the per-lane
    // offsets of the group's loads and stores are group-invariant, get hoisted above this loop and spill — 150-190 bytes
    // of scratch per lane, reloaded inside the step loop)
    int ng_ = n, qg_ = q;
    asm volatile("" : "+v"(ng_), "+v"(qg_));
#pragma unroll
    for (int t = 0; t < TIN; ++t) {
      const int px = 16 * (wpix + WP * t) + ng_;
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks)
        xb[t][ks] = px < m_in ? *reinterpret_cast<const u32x4*>(xg + (size_t)px * CIN + 32 * ks + 8 * qg_) : zero4;
      const int g = px / HWI, r = px - g * HWI, iy = r / HIN, ix = r - iy * HIN;
      erow[t] = px < m_in ? g * HIN * PW + iy * PW + ix + 1 : Geo::E_DUMP;
    }
#pragma unroll
    for (int t = 0; t < TOUT; ++t)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (AEF) load_ae(0);
    if (APF) load_ap(0, 0);
    TILE_TICK(1);  // operand setup (block input, first weights requested)
#pragma unroll 1
    for (int s = 0; s < nch; ++s) {  // steps with an expansion
      if (mx_on) expand(s);
      TILE_TICK(2);
      if (s >= 2 && mx_on) project(s - 2);
      TILE_TICK(3);
      lds_barrier();
      TILE_TICK(4);
    }
    // drain: the last two projections; the epilogue's operands (bias, residual = block input) are requested first
    float4 bpj[NCT];
    u32x2 rres[TOUT][NCT];
    int ne_ = n, qe_ = q;
    asm volatile("" : "+v"(ne_), "+v"(qe_));
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) bpj[ct] = *reinterpret_cast<const float4*>(W + a.bp_off + 16 * (ct0w + ct) + 4 * qe_);
#pragma unroll
    for (int t = 0; t < TOUT; ++t) {
      const int p = 16 * (wpix + WP * t) + ne_;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
        rres[t][ct] = (a.residual && p < m_out) ? *reinterpret_cast<const u32x2*>(xg + (size_t)p * CIN + 16 * (ct0w + ct) + 4 * qe_)
                                                : u32x2{0u, 0u};
    }
    TILE_TICK(5);  // epilogue operands requested
    if (nch >= 2 && mx_on) project(nch - 2);
    TILE_TICK(6);
    lds_barrier();
    TILE_TICK(7);
    if (mx_on) project(nch - 1);
    TILE_TICK(6);
    lds_barrier();
    TILE_TICK(7);
    // Epilogue.  Every workgroup of a round reaches this point together and the store burst is 10-14 % of a workgroup's
    // time (tools/dev/tile_ticks.sh: 5-12 k cycles; 0.5-1.4 k with the stores predicated off, RIP_TILE_ABL bit 64) — close
    // to what the memory system takes for 256 x 40 KB at once (tools/micro/store_burst.hip: 4.0 TB/s sustained in this
    // lane layout — 64 separate 8-byte pieces per instruction —, 6.0 TB/s with 64 contiguous bytes per pixel).  Measured
    // and not kept: (a) exchanging two channel tiles between the lane pairs (q, q ^ 1) and transposing the lane grid with
    // ds_bpermute (16-byte stores, four consecutive lanes = 64 contiguous bytes per pixel): within +-2 us on every block;
    // (b) the block-input / residual loads unconditional (clamped address + select): 20 us slower over the ten blocks.
    // What would hide the burst is other work on the CU while it drains — a second resident workgroup (LDS: one fits) or
    // a persistent workgroup that starts the next observation group's expansion under it.
#pragma unroll
    for (int t = 0; t < TOUT; ++t) {
      const int p = 16 * (wpix + WP * t) + ne_;
      if (p >= m_out || (RIP_TILE_ABL & 16)) continue;
      if ((RIP_TILE_ABL & 64) && a.k0 >= 0) continue;  // development: the epilogue's arithmetic stays, its stores never run
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        f32x2 v0 = {acc[t][ct][0] + bpj[ct].x, acc[t][ct][1] + bpj[ct].y};
        f32x2 v1 = {acc[t][ct][2] + bpj[ct].z, acc[t][ct][3] + bpj[ct].w};
        v0 += bfpair(rres[t][ct].x);  // zeros when the block has no residual (as the layer-wise kernels: + 0 is exact)
        v1 += bfpair(rres[t][ct].y);
        u32x2 o;
        o.x = pack_bf16(v0);
        o.y = pack_bf16(v1);
        *reinterpret_cast<u32x2*>(yg + (size_t)p * COUT + 16 * (ct0w + ct) + 4 * qe_) = o;
      }
    }
    TILE_TICK(8);  // epilogue
    }  // groups
#ifdef RIP_TILE_TICKS
    if (tid == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) atomicAdd(&g_tile_ticks[i], tk[i]);
      atomicAdd(&g_tile_ticks[9], 1ull);
    }
#endif
  } else {
    // ================= vector waves: depthwise 3x3 =================
    const int vt = tid - 256;
    const int c8 = vt & 7, dcol = (vt >> 3) % HOUT, dimg = (vt >> 3) / HOUT;
    const int dimg_c = dimg < G ? dimg : 0;  // (threads beyond G * HOUT * 8 never work; their addresses stay inside the buffers)
    // first of the three padded columns this thread reads (input column dcol*S - 1 -> padded index dcol*S)
    const int e_off = (dimg_c * HIN * PW + dcol * STRIDE) * LD + 8 * c8;
    const int d_off = (dimg_c * HWO + dcol) * LD + 8 * c8;
    f32x2 wt[2][9][4], bd[2][4];
    auto load_taps = [&](int c, f32x2(&wt_)[9][4], f32x2(&bd_)[4]) {
      const float* wd = Tl + c * HC + 8 * c8;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float4 w0 = *reinterpret_cast<const float4*>(wd + (size_t)t * HID);
        const float4 w1 = *reinterpret_cast<const float4*>(wd + (size_t)t * HID + 4);
        wt_[t][0] = f32x2{w0.x, w0.y};
        wt_[t][1] = f32x2{w0.z, w0.w};
        wt_[t][2] = f32x2{w1.x, w1.y};
        wt_[t][3] = f32x2{w1.z, w1.w};
      }
      const float4 b0 = *reinterpret_cast<const float4*>(Tl + 9 * HID + c * HC + 8 * c8);
      const float4 b1 = *reinterpret_cast<const float4*>(Tl + 9 * HID + c * HC + 8 * c8 + 4);
      bd_[0] = f32x2{b0.x, b0.y};
      bd_[1] = f32x2{b0.z, b0.w};
      bd_[2] = f32x2{b1.x, b1.y};
      bd_[3] = f32x2{b1.z, b1.w};
    };
    auto depthwise = [&](const bf16_t* E, bf16_t* D, const f32x2(&wt_)[9][4], const f32x2(&bd_)[4]) {
      f32x2 sacc[HOUT][4];
#pragma unroll
      for (int iy = 0; iy < HIN; ++iy) {
        const bf16_t* r = E + e_off + iy * PW * LD;
        const u32x4 v0 = *reinterpret_cast<const u32x4*>(r);
        const u32x4 v1 = *reinterpret_cast<const u32x4*>(r + LD);
        const u32x4 v2 = *reinterpret_cast<const u32x4*>(r + 2 * LD);
        const f32x2 f[3][4] = {{bfpair(v0.x), bfpair(v0.y), bfpair(v0.z), bfpair(v0.w)},
                               {bfpair(v1.x), bfpair(v1.y), bfpair(v1.z), bfpair(v1.w)},
                               {bfpair(v2.x), bfpair(v2.y), bfpair(v2.z), bfpair(v2.w)}};
#pragma unroll
        for (int oy = 0; oy < HOUT; ++oy) {
          const int ky = iy - oy * STRIDE + 1;  // compile-time after unrolling
          if (ky < 0 || ky > 2) continue;
          if (ky == 0 || (ky == 1 && oy * STRIDE - 1 < 0)) {  // first row of this output that lies inside the map
#pragma unroll
            for (int e = 0; e < 4; ++e) sacc[oy][e] = bd_[e];
          }
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int e = 0; e < 4; ++e) sacc[oy][e] = __builtin_elementwise_fma(f[kx][e], wt_[ky * 3 + kx][e], sacc[oy][e]);
          if (ky == 2 || iy == HIN - 1) {  // last row of this output inside the map: finish it
            u32x4 o;
            o.x = pack_bf16(relu6_2(sacc[oy][0]));
            o.y = pack_bf16(relu6_2(sacc[oy][1]));
            o.z = pack_bf16(relu6_2(sacc[oy][2]));
            o.w = pack_bf16(relu6_2(sacc[oy][3]));
            *reinterpret_cast<u32x4*>(D + d_off + oy * HOUT * LD) = o;
          }
        }
      }
    };
#pragma unroll 1
    for (int j = 0; j < ng; ++j) {  // this workgroup's observation groups (the same barrier sequence as the matrix waves)
    enter_group(j);
    const bool dw_on = dimg * HWI < m_in;  // this thread's observation exists in the group (the last one may be ragged)
    load_taps(0, wt[0], bd[0]);
    TILE_TICK(1);
#pragma unroll 1
    for (int s = 0; s < nch + 2; s += 2) {
      // even step s: depthwise of chunk s-1 (odd buffers), taps of chunk s arrive in set 0
      if (s >= 1 && s <= nch && dw_on && !(RIP_TILE_ABL & 1))
        depthwise(Ebuf + (size_t)Geo::E_ROWS * LD, Dbuf + (size_t)Geo::D_ROWS * LD, wt[1], bd[1]);
      if (s + 1 < nch && !(RIP_TILE_ABL & 4)) load_taps(s + 1, wt[1], bd[1]);
      TILE_TICK(2);
      lds_barrier();
      TILE_TICK(3);
      if (s + 1 >= nch + 2) break;
      // odd step s+1: depthwise of chunk s (even buffers)
      if (s + 1 <= nch && dw_on && !(RIP_TILE_ABL & 1)) depthwise(Ebuf, Dbuf, wt[0], bd[0]);
      if (s + 2 < nch && !(RIP_TILE_ABL & 4)) load_taps(s + 2, wt[0], bd[0]);
      TILE_TICK(2);
      lds_barrier();
      TILE_TICK(3);
    }
    }  // groups
#ifdef RIP_TILE_TICKS
    if (tid == 256) {
      atomicAdd(&g_tile_ticks[10], tk[2]);
      atomicAdd(&g_tile_ticks[11], tk[3]);
    }
#endif
  }
}

template <int HIN, int STRIDE, int CIN, int COUT, int GMAX, bool AEF, bool APF, int WCH = 1>
hipError_t launch_tile(TileArgs a, int kc, hipStream_t s) {
  using Geo = TileGeom<HIN, STRIDE, GMAX>;
  static_assert(GMAX * Geo::HOUT * 8 <= 256, "one depthwise thread per (observation, column, 8 channels)");
  // observations per workgroup: one workgroup per CU when the launch is large enough, never more than GMAX
  int G = (int)(((long)a.B * kc + 255) / 256);
  if (G > GMAX) G = GMAX;
  if (G < 1) G = 1;
  a.G = G;
  static bool attr_set[64] = {};  // per device: > 64 KB of dynamic LDS needs the opt-in
  auto kern = irb_tile_bf16_kernel<HIN, STRIDE, CIN, COUT, GMAX, AEF, APF, WCH>;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)Geo::lds_bytes(6 * CIN));
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  if (a.HID != 6 * CIN) return hipErrorInvalidValue;
  static_assert(Geo::lds_bytes(6 * CIN) <= 160 * 1024, "LDS budget");
  // one workgroup per CU is resident (LDS); each walks the observation groups wgx, wgx + gx, ... of its model
  const int n_groups = (a.B + G - 1) / G;
  int gx = device_cu_count() / kc;
  if (gx < 1) gx = 1;
  if (gx > n_groups) gx = n_groups;
  note_kernel(dim3(gx, 1, kc), dim3(512), "irb_tile_bf16_kernel<%d,%d,%d,%d,%d,%s,%s,%d> G=%d", HIN, STRIDE, CIN, COUT,
              GMAX, AEF ? "true" : "false", APF ? "true" : "false", WCH, G);
  hipLaunchKernelGGL(kern, dim3(gx, 1, kc), dim3(512), Geo::lds_bytes(a.HID), s, a);
#ifdef RIP_TILE_TICKS
  {
    unsigned long long t[16];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_tile_ticks), sizeof(t));
    const double n = t[9] > 0 ? (double)t[9] : 1.0, st = a.HID / HC;
    fprintf(stderr, "tile<%d,%d,%d,%d,G%d,%d> cycles per workgroup: prologue %.0f setup %.0f | per step (%d): expand %.0f project %.0f barrier %.0f | "
            "drain: requests %.0f projections 2 x %.0f barriers 2 x %.0f epilogue %.0f | vector wave per step: work %.0f barrier %.0f\n",
            HIN, STRIDE, CIN, COUT, G, WCH, t[0] / n, t[1] / n, (int)st, t[2] / n / st, t[3] / n / st, t[4] / n / st, t[5] / n, t[6] / n / 2,
            t[7] / n / 2, t[8] / n, t[10] / n / (st + 2), t[11] / n / (st + 2));
    unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_ticks), z, sizeof(z));
  }
#endif
  return hipGetLastError();
}

}  // namespace

bool irb_tile_bf16_supported(const Layer* le, const Layer& ld, const Layer& lp, bool f17_layerwise) {
  if (le == nullptr) return false;
  const int cin = le->cin, hid = ld.cout, cout = lp.cout;
  if (hid != 6 * cin || hid % HC != 0) return false;
  if (ld.h_in == 7 && ld.stride == 1) return (cin == 64 && (cout == 64 || cout == 96)) || (cin == 96 && cout == 96);
  if (ld.h_in == 7 && ld.stride == 2) return cin == 96 && cout == 160;
  // features.17 (160 -> 960 -> 320 at 4x4): round 4 kept it layer-wise (with 320 output channels per wave a wave's
  // accumulators leave room for one 16-pixel tile only: 424 us vs ~150 us); round 5 splits the channels over the waves
  if (ld.h_in == 4 && ld.stride == 1) return cin == 160 && (cout == 160 || (!f17_layerwise && cout == 320));
  return false;
}

hipError_t launch_irb_tile_bf16(const Layer* le, const Layer& ld, const Layer& lp, const float* enc_w,
                                const unsigned short* enc_wh, size_t model_stride, int k0, int kc, int B,
                                const unsigned short* x, unsigned short* y, hipStream_t s) {
  TileArgs a;
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
  a.HID = ld.cout;
  a.residual = lp.residual;
  a.G = 1;
  const int cin = le->cin, cout = lp.cout;
  //                                          HIN S CIN COUT GMAX AEF APF
  if (ld.h_in == 7 && ld.stride == 1) {
    if (cin == 64 && cout == 64) return launch_tile<7, 1, 64, 64, 4, true, true, 2>(a, kc, s);    // features.8-10
    if (cin == 64 && cout == 96) return launch_tile<7, 1, 64, 96, 4, true, true, 2>(a, kc, s);    // features.11
    // (features.12 / 13 and 15 / 16 with fewer observations per workgroup and the weight prefetch on — the recipe that paid
    // for features.17 — measured 6-90 us SLOWER: profiles/r5/tile17_variants.txt)
    if (cin == 96 && cout == 96) return launch_tile<7, 1, 96, 96, 4, false, false, 2>(a, kc, s);    // features.12, 13
  }
  if (ld.h_in == 7 && ld.stride == 2 && cin == 96 && cout == 160) return launch_tile<7, 2, 96, 160, 4, true, true, 2>(a, kc, s);  // 14
  if (ld.h_in == 4 && ld.stride == 1 && cin == 160) {
    if (cout == 160) return launch_tile<4, 1, 160, 160, 8, false, false, 2>(a, kc, s);              // features.15, 16
    // features.17 (round 5): the 320 output channels split FOUR ways over the matrix waves (WCH = 4: every wave owns 5
    // channel tiles of all pixel tiles — 20 accumulators, 245 registers, no scratch) and 4 observations per workgroup;
    // 139 -> 68 us against the three layer-wise launches (expand GEMM 57, depthwise 27, projection GEMM 57).  G = 2 / 3 and
    // the builds without weight prefetch measured 1499-1567 us of encoder time against 1475 (profiles/r5/tile17_variants.txt)
    if (cout == 320) return launch_tile<4, 1, 160, 320, 4, true, true, 4>(a, kc, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace rip
