// Device functions of the split-f16 plan search (flow_split.hip; also compiled into tools/micro/split_f16.hip).
//
// The GRU / head contractions of flow_phase.hip, moved from v_mfma_f32_16x16x4_f32 (32 cycles, K = 4) to
// v_mfma_f32_16x16x32_f16 (16 cycles, K = 32) without giving up fp32-grade results: both operands are carried as two
// binary16 terms, x ~= hi + lo with hi = f16(x), lo ~ x - hi, and a product is three MFMAs, fp32 accumulation,
//     W x ~= Whi xhi + Whi xlo + Wlo xhi        (the dropped Wlo xlo term is 2^-22 relative)
// (binary16 products are exact in fp32).  22 significant bits per operand instead of 24: the teacher-forced 1e-4 tests
// are the gate (tests/test_gpu_parity.py).
// Round 5, FORWARD side (hidden states, |h| <= max(1, |z|), forward weight rows): all three products of a tile go into
// ONE accumulator.  The weight rows keep lo' = f16((w - hi) 2^11) (weights are small: an unscaled residual would sit in
// the subnormal range of binary16, quantised at 2^-24 absolute — measured: 3x the fp32 kernel's error at |z| ~ 1e3 and
// O(1) gradient errors where the search amplifies rounding 1e5-fold), and the 2^-11 moves to the other operand: the
// (Wlo' x hi) product uses hs = hi * 2^-11, an exact binary16 scaling (v_pk_mul_f16, four instructions per K block).
// The state's own low term is lo = f16(x - hi), unscaled: for |x| < 1/8 it is subnormal, quantised at 2^-24, i.e. at most
// 2^-25 |W| per product — below the fp32 accumulation error of the sum it joins (the f16 matrix pipe takes subnormal
// inputs as they are; tools/micro/split_f16.hip checks).  Gone: the second accumulator of every tile, its
// `a + l * 2^-11` combine and one multiply per split pair — a wave's time here is the SUM of its matrix-pipe cycles,
// 4 cycles per vector instruction and 16 per transcendental (MFMAs and vector work of ONE wave do not overlap,
// DESIGN_HISTORY §4.1), so instructions are what there is to remove.  (Packing the r / z rows pre-multiplied by -log2(e)
// saves two more multiplies per unit pair and measured no faster: 2.654 vs 2.640 ms; not kept.)  The ADJOINT side: see
// TW_SHIFT below.
// The weights are split on the host (flow_split_pack.h); activations / gradients are split here, per candidate:
//   * hidden states (|h| <= max(1, |z|)) and the head's hidden layer are split as they are;
//   * adjoint quantities (unbounded either way) are first scaled by a per-candidate power of two that puts the
//     candidate's largest entry at 2^13, and the contraction's result is scaled back — exact, and neither overflow
//     (binary16 max 65504) nor the subnormal range is ever reached by values that matter.
// A 64-unit vector in the "H layout" (lane (c, q) holds units 16u + 4q + r in H[4u + r]) needs NO data movement to
// become a B operand: K block kb of a lane = its H[8kb .. 8kb+7]; the unit permutation lives in the operand rows.
#pragma once
#include <hip/hip_runtime.h>

#include "flow.h"
#include "flow_math.h"

namespace rip {
namespace split {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using h16x8 = __attribute__((ext_vector_type(8))) _Float16;
using h16x2 = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int T = 4;
constexpr int CB = 16;  // candidates per wave

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfmah(h16x8 a, h16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 zero4() {
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return z;
}
__device__ __forceinline__ h16x8 as_h8(uint4 u) {
  u32x4 v = {u.x, u.y, u.z, u.w};
  return __builtin_bit_cast(h16x8, v);
}
__device__ __forceinline__ float4 as_f4(uint4 u) {
  return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}

// B operands of a 64-unit vector (or of any 16 values per lane): two K blocks, two terms each
struct BSplit {
  h16x8 hi[2], lo[2];
  h16x8 hs[2];  // hi * 2^-11 (forward states only): the B operand of the (Wlo' x hi) product, see the head of this file
};

// ADJOINT side: one accumulator as well, by scaling the WEIGHTS instead of their residuals.  The transposed rows and the
// W_ih^T table are packed as w * 2^8 (flow_split_pack.h: exact; |w| < 255): hi = f16(256 w), lo = f16(256 w - hi) is then
// a normal binary16 for every weight that matters, no 2^11 anywhere, and the 2^-8 is folded into the power of two the
// contraction's result is scaled back by anyway (`pow2_scale`'s inverse: free).  Gradient operands: hi = f16(x s),
// lo = f16(x s - hi) with the candidate's largest entry at 2^13: entries more than 2^16 below it keep only their hi bits
// (11), i.e. 2^-27 of the largest — nothing that matters.
constexpr float LO_INV = 1.0f / 2048.0f;      // forward: hs = hi * 2^-11
constexpr int TW_SHIFT = 8;                   // transposed rows are packed as w * 2^8

// 8 fp32 -> (hi, lo) packed halves, lo = f16(x - hi).  SCALED (adjoint quantities): `s` = a power-of-two pre-scale.
// Per pair: [v_pk_mul (s)], v_cvt_pk_f16_f32, 2 x v_cvt_f32_f16, v_pk_add, v_cvt_pk_f16_f32.
template <bool SCALED>
__device__ __forceinline__ void split8(const float* v, float s, h16x8& hi, h16x8& lo) {
  u32x4 uh, ul;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    f32x2 x = {v[2 * p], v[2 * p + 1]};
    if (SCALED) x = x * f32x2{s, s};
    const h16x2 h = __builtin_convertvector(x, h16x2);
    // (round 5: the residual as v_fma_mix_f32(h.half, -1, x) from inline assembly — four instead of five instructions
    // per pair, 241 fewer in the kernel — measured 2.491 vs 2.494 ms: not kept)
    const f32x2 back = __builtin_convertvector(h, f32x2);
    const h16x2 l = __builtin_convertvector(x - back, h16x2);
    uh[p] = __builtin_bit_cast(unsigned, h);
    ul[p] = __builtin_bit_cast(unsigned, l);
  }
  hi = __builtin_bit_cast(h16x8, uh);
  lo = __builtin_bit_cast(h16x8, ul);
}
__device__ __forceinline__ void split16(const float (&v)[16], BSplit& b) {
  split8<false>(&v[0], 1.f, b.hi[0], b.lo[0]);
  split8<false>(&v[8], 1.f, b.hi[1], b.lo[1]);
  const _Float16 k = (_Float16)LO_INV;  // exact; v_pk_mul_f16 on the packed halves, four instructions per K block
  const h16x8 k8 = {k, k, k, k, k, k, k, k};
  b.hs[0] = b.hi[0] * k8;
  b.hs[1] = b.hi[1] * k8;
}

// max |.| over the 4 q-lanes of a candidate (lanes c, c+16, c+32, c+48): two swaps on the VALU, no LDS
__device__ __forceinline__ float qmax(float m) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}
// power of two that moves `amax` (>= 0) to [2^13, 2^14), and its inverse (times 2^-TW_SHIFT); amax == 0 -> 1.  The exponent is clamped at
// -100: a candidate whose gate gradients have all but vanished (saturated gates: r (1 - r) ~ 1e-35 at |z| ~ 1e3) would
// otherwise ask for 2^130 = inf, and inf * 0 = NaN in every zero entry (found by test_split_kernel_operand_ranges;
// below 2^-100 the scaled values are simply smaller than 2^13, which loses nothing that matters).
__device__ __forceinline__ void pow2_scale(float amax, float& s, float& inv) {
  int e = amax > 0.f ? __builtin_amdgcn_frexp_expf(amax) : 14;  // amax = f * 2^e, f in [0.5, 1)
  e = e < -100 ? -100 : e;
  s = __builtin_ldexpf(1.0f, 14 - e);
  inv = __builtin_ldexpf(1.0f, e - 14 - TW_SHIFT);  // (the transposed weight rows carry 2^TW_SHIFT)
}

#ifdef RIP_ISA_MARKS  // development: comment lines in the ISA listing (tools/dev/isa_regions.py counts instructions between them)
#define RIP_MARK(name_) asm volatile("; RIPMARK " name_)
#else
#define RIP_MARK(name_)
#endif
#ifndef RIP_PRIO
#define RIP_PRIO 1
#endif

#define SPLIT_PRIO_BURST() do { if (RIP_PRIO) __builtin_amdgcn_s_setprio(1); } while (0)
#define SPLIT_PRIO_VALU() do { if (RIP_PRIO) __builtin_amdgcn_s_setprio(0); } while (0)
#ifndef RIP_ABL
#define RIP_ABL 0  // development only: 1 = no tape loads, 3 = no tape stores (wrong results)
#endif

// Adjoint tape of one heavy step, per lane: rows 0..15 = (r, z, -, gh_n) of the 4 unit tiles (slot 2 of a tile stays
// unused: n is recomputed), rows 16..19 = hprev (not written at t = 1: that is the prefix H1), then one dword with the
// ReLU mask of a1 (8 bits).
constexpr int TAPE_ROWS = 20;
constexpr int TAPE_STEP_F4 = TAPE_ROWS * 64 + 16;
constexpr int TAPE_SLOT_F4 = 3 * TAPE_STEP_F4;

__device__ __forceinline__ void tape_st(float4* p, float a, float b, float c, float d) {
  if (RIP_ABL != 3) *p = make_float4(a, b, c, d);
}
__device__ __forceinline__ float4 tape_ld(const float4* p) {
  if (RIP_ABL == 1) return make_float4(0.3f, 0.4f, 0.5f, 0.6f);
  return *p;
}
__device__ __forceinline__ float4* trow(float4* base, int r, unsigned loff) {
  return reinterpret_cast<float4*>(reinterpret_cast<char*>(base + r * 64) + loff);
}
__device__ __forceinline__ const float4* trow(const float4* base, int r, unsigned loff) {
  return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base + r * 64) + loff);
}

struct StepTape {
  float hp[16], r[16], z[16], n[16], gh[16];
  unsigned mask;
};
enum { SAVE_NONE = 0, SAVE_TAPE = 1, SAVE_TAPE_NOHP = 2, SAVE_REGS = 3 };
enum { MODE_FWD = 0, MODE_INV = 1 };

// gate math of one unit tile (flow_phase.hip:gru_gates)
__device__ __forceinline__ void gru_gates(const f32x4& ar, const f32x4& az, const f32x4& agn, const f32x4& ahn,
                                          const float* Hold, float* Hn, float (&rr)[4], float (&zz)[4], float (&nn)[4]) {
  const f32x2 one = {1.0f, 1.0f}, two = {2.0f, 2.0f};
  constexpr float L2E = 1.4426950408889634f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x2 pr = f32x2{ar[2 * h], ar[2 * h + 1]} * f32x2{-L2E, -L2E};
    const f32x2 pz = f32x2{az[2 * h], az[2 * h + 1]} * f32x2{-L2E, -L2E};
    const f32x2 er = {__builtin_amdgcn_exp2f(pr.x), __builtin_amdgcn_exp2f(pr.y)};
    const f32x2 ez = {__builtin_amdgcn_exp2f(pz.x), __builtin_amdgcn_exp2f(pz.y)};
    const f32x2 dr = er + one, dz = ez + one;
    const f32x2 r2 = {rcpf_(dr.x), rcpf_(dr.y)};
    const f32x2 z2 = {rcpf_(dz.x), rcpf_(dz.y)};
    const f32x2 pre = __builtin_elementwise_fma(r2, f32x2{ahn[2 * h], ahn[2 * h + 1]}, f32x2{agn[2 * h], agn[2 * h + 1]});
    const f32x2 pn = pre * f32x2{2.0f * L2E, 2.0f * L2E};
    const f32x2 en = {__builtin_amdgcn_exp2f(pn.x), __builtin_amdgcn_exp2f(pn.y)};
    const f32x2 dn = en + one;
    const f32x2 in2 = {rcpf_(dn.x), rcpf_(dn.y)};
    const f32x2 n2 = one - two * in2;
    const f32x2 hold = {Hold[2 * h], Hold[2 * h + 1]};
    const f32x2 hn = __builtin_elementwise_fma(z2, hold - n2, n2);  // (1-z)*n + z*h
    rr[2 * h] = r2.x, rr[2 * h + 1] = r2.y;
    zz[2 * h] = z2.x, zz[2 * h + 1] = z2.y;
    nn[2 * h] = n2.x, nn[2 * h + 1] = n2.y;
    Hn[2 * h] = hn.x, Hn[2 * h + 1] = hn.y;
  }
}

// B operand of the input / bias k-steps (round 6): the K block flow_split_pack.h's MHF_KS rows contract against.
// y / 4 as three binary16 terms (hi, mid, lo: 33 significant bits, i.e. the fp32 value exactly; the / 4 keeps waypoints up
// to 2.6e5 m inside binary16, the rows hold W x 4), and their exact 2^-11 / 2^-22 multiples for the weights' second /
// third terms:  slots 0..5 = y0 (hi, mid, lo, hi 2^-11, mid 2^-11, hi 2^-22), 6..11 = y1 likewise, 12..14 = (1, 2^-11,
// 2^-22) for the bias terms, 15 = 0; lane block q = 0 holds slots 0..7, q = 1 slots 8..15, q = 2, 3 zeros.
__device__ __forceinline__ h16x8 ybuild(float yp0, float yp1, int q) {
  const f32x2 ys = f32x2{yp0, yp1} * f32x2{0.25f, 0.25f};
  const h16x2 h = __builtin_convertvector(ys, h16x2);
  const f32x2 r1 = ys - __builtin_convertvector(h, f32x2);
  const h16x2 m = __builtin_convertvector(r1, h16x2);
  const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
  const h16x2 l = __builtin_convertvector(r2, h16x2);
  const _Float16 k1 = (_Float16)LO_INV, k2 = (_Float16)(LO_INV * LO_INV);
  const h16x2 hs = h * h16x2{k1, k1}, ms = m * h16x2{k1, k1}, hss = h * h16x2{k2, k2};
  const h16x8 b0 = {h[0], m[0], l[0], hs[0], ms[0], hss[0], h[1], m[1]};
  const h16x8 b1 = {l[1], hs[1], ms[1], hss[1], (_Float16)1.0f, k1, k2, (_Float16)0.0f};
  const u32x4 u0 = __builtin_bit_cast(u32x4, b0), u1 = __builtin_bit_cast(u32x4, b1);
  u32x4 u;
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = q == 0 ? u0[i] : (q == 1 ? u1[i] : 0u);
  return __builtin_bit_cast(h16x8, u);
}

// One GRU + head step for 16 candidates.  `wl` = this lane's column of the forward operand rows in LDS (MHF_* in
// flow.h); H = the hidden state (H layout), `hs` = its split B operands — both are replaced by the new state's.
// Round 6: no fp32 MFMA is left in the step (a wave's time is the SUM of its matrix-pipe and vector-issue cycles, and the
// 27 K = 4 fp32 MFMAs of rounds 3-5 — input / bias k-steps, b1, W2, b2 — were 864 of its 2208 matrix cycles for 1 % of
// its flops).  Per unit tile: 3 k-step MFMAs (one f16 K block each: ybuild) + 2 K blocks x 3 gates x 3 f16 MFMAs, b_hn as
// the accumulator image of gh_n's chain; head: b1 as accumulator images, 12 MFMAs, then W2 as ONE K block over the 32
// head units (3 MFMAs on the split ReLU output, b2 as the accumulator image).  99 f16 MFMAs = 1584 matrix-pipe cycles
// (rounds 3-5: 2208; flow_phase.hip: 251 x 32 = 8032).
template <int SAVE, bool PIPE = false>
__device__ __forceinline__ void fwd_step(const uint4* wl, float (&H)[16], BSplit& hs, float yp0, float yp1, int q,
                                         unsigned lane, float4* __restrict__ tape, StepTape* tr, float (&o)[4]) {
  (void)PIPE;  // (round 3's tile pipelining: MFMAs and vector work of one wave do not overlap, removed in round 5)
  RIP_MARK("fwd_begin");
  unsigned loff = lane * 16u;
  asm volatile("" : "+v"(loff));  // flow_phase.hip: keeps the tape addressing scalar base + one lane offset
  const h16x8 by = ybuild(yp0, yp1, q);
  // Operand rows of group gk = up * 2 + kb (one K block of one unit tile): hi row of gate g at
  // ((g * 4 + up) * 2 + kb) * 2, lo row behind it.  A group is nine MFMAs into the tile's three accumulators, ordered
  // (hi hi) r z n, (hi lo) r z n, (lo hi) r z n so that an accumulator is touched every THIRD instruction (back-to-back
  // MFMAs on one accumulator wait for each other).  The hi rows of group gk + 1 are requested at the top of group gk
  // (also across tiles: they land under the gate math), the lo rows of gk at its top (six MFMAs ahead of their use).
  auto row_of = [](int gk, int g, int term) {
    const int up = gk >> 1, kb = gk & 1;
    return (((g * 4 + up) * 2 + kb) * 2 + term) * 64;
  };
  uint4 RH[2][3], RL[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) RH[0][g] = wl[row_of(0, g, 0)];
  float Hn[16];
  struct TileAcc {
    f32x4 a[3], agn;  // pre_r, pre_z, gh_n; gi_n
  };
  auto issue = [&](int up, TileAcc& t) __attribute__((always_inline)) {
    // the tile's k-step rows (r, z, gi_n) and the b_hn image: requested first, used first
    const uint4 kr = wl[(MHF_KS + 0 + up) * 64], kz = wl[(MHF_KS + 4 + up) * 64], kn = wl[(MHF_KS + 8 + up) * 64];
    const float4 bh = as_f4(wl[(MHF_GHB + up) * 64]);
    SPLIT_PRIO_BURST();
    t.a[0] = mfmah(as_h8(kr), by, zero4());
    t.a[1] = mfmah(as_h8(kz), by, zero4());
    t.agn = mfmah(as_h8(kn), by, zero4());
    t.a[2] = f32x4{bh.x, bh.y, bh.z, bh.w};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int gk = up * 2 + kb;
      const h16x8 bhh = hs.hi[kb], bl = hs.lo[kb], bs = hs.hs[kb];
#pragma unroll
      for (int g = 0; g < 3; ++g) RL[g] = wl[row_of(gk, g, 1)];
      if (gk + 1 < 8) {
#pragma unroll
        for (int g = 0; g < 3; ++g) RH[(gk + 1) & 1][g] = wl[row_of(gk + 1, g, 0)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 3; ++g) t.a[g] = mfmah(as_h8(RH[gk & 1][g]), bhh, t.a[g]);
#pragma unroll
      for (int g = 0; g < 3; ++g) t.a[g] = mfmah(as_h8(RH[gk & 1][g]), bl, t.a[g]);
#pragma unroll
      for (int g = 0; g < 3; ++g) t.a[g] = mfmah(as_h8(RL[g]), bs, t.a[g]);
    }
    SPLIT_PRIO_VALU();
  };
  auto finish = [&](int up, const TileAcc& t) __attribute__((always_inline)) {
    const f32x4 ahn = t.a[2];
    float rr[4], zz[4], nn[4];
    gru_gates(t.a[0], t.a[1], t.agn, ahn, &H[up * 4], &Hn[up * 4], rr, zz, nn);
    // pin the tile's gate math HERE: it has no side effect, and left alone it sinks below the MFMAs of ALL later tiles
    // (to its first use), which keeps four tiles of accumulators alive
    asm volatile("" : "+v"(Hn[up * 4]), "+v"(Hn[up * 4 + 1]), "+v"(Hn[up * 4 + 2]), "+v"(Hn[up * 4 + 3]));
    if (SAVE == SAVE_TAPE || SAVE == SAVE_TAPE_NOHP) {
      tape_st(trow(tape, up * 4 + 0, loff), rr[0], rr[1], rr[2], rr[3]);
      tape_st(trow(tape, up * 4 + 1, loff), zz[0], zz[1], zz[2], zz[3]);
      tape_st(trow(tape, up * 4 + 3, loff), ahn[0], ahn[1], ahn[2], ahn[3]);
      if (SAVE == SAVE_TAPE) tape_st(trow(tape, 16 + up, loff), H[up * 4], H[up * 4 + 1], H[up * 4 + 2], H[up * 4 + 3]);
    }
    if (SAVE == SAVE_REGS) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        tr->hp[up * 4 + r] = H[up * 4 + r];
        tr->r[up * 4 + r] = rr[r];
        tr->z[up * 4 + r] = zz[r];
        tr->n[up * 4 + r] = nn[r];
        tr->gh[up * 4 + r] = ahn[r];
      }
    }
  };
  // tile by tile (two waves per SIMD: the partner wave's MFMAs cover this wave's gate math; one wave: nothing does)
#pragma unroll
  for (int up = 0; up < 4; ++up) {
    TileAcc t;
    issue(up, t);
    finish(up, t);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) H[i] = Hn[i];
  RIP_MARK("fwd_tiles_done");
  split16(H, hs);  // the head's B operands == the next step's
  // ---- head: rows 52..59 = W1 ((tile mt, kb) x (hi, lo)); MHF_B1 / MHF_B2 accumulator images, MHF_W2 (hi, lo') ----
  const float4 b1a = as_f4(wl[(MHF_B1 + 0) * 64]), b1b = as_f4(wl[(MHF_B1 + 1) * 64]);
  const uint4 w2h = wl[(MHF_W2 + 0) * 64], w2l = wl[(MHF_W2 + 1) * 64];
  const float4 b2v = as_f4(wl[MHF_B2 * 64]);
  SPLIT_PRIO_BURST();
  f32x4 a0 = {b1a.x, b1a.y, b1a.z, b1a.w}, a1 = {b1b.x, b1b.y, b1b.z, b1b.w};
  {
    // rows 52 + (mt * 2 + kb) * 2 + term; 12 MFMAs, the two tiles' accumulators alternate
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const uint4 w0h = wl[(52 + kb * 2) * 64], w0l = wl[(53 + kb * 2) * 64];
      const uint4 w1h = wl[(56 + kb * 2) * 64], w1l = wl[(57 + kb * 2) * 64];
      a0 = mfmah(as_h8(w0h), hs.hi[kb], a0);
      a1 = mfmah(as_h8(w1h), hs.hi[kb], a1);
      a0 = mfmah(as_h8(w0h), hs.lo[kb], a0);
      a1 = mfmah(as_h8(w1h), hs.lo[kb], a1);
      a0 = mfmah(as_h8(w0l), hs.hs[kb], a0);
      a1 = mfmah(as_h8(w1l), hs.hs[kb], a1);
    }
  }
  SPLIT_PRIO_VALU();
  if (SAVE != SAVE_NONE) {
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      m |= a0[r] > 0.f ? (1u << r) : 0u;
      m |= a1[r] > 0.f ? (16u << r) : 0u;
    }
    if (SAVE == SAVE_REGS) {
      tr->mask = m;
    } else if (RIP_ABL != 3) {
      *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(tape + TAPE_ROWS * 64) + (loff >> 2)) = m;
    }
  }
  // W2 relu(a1): the lane's 8 head units (tile 0: 4q + r, tile 1: 16 + 4q + r) ARE K slots 8q .. 8q + 7 of the W2 K block
  // (the permutation lives in the rows); relu(a1) / 4 as two binary16 terms (a1 up to 2.6e5: |h| < 2^14 times a W1 row sum
  // of 16), the rows hold W2 x 4
  {
    float av[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      av[r] = fmaxf(a0[r], 0.f);
      av[4 + r] = fmaxf(a1[r], 0.f);
    }
    h16x8 ah, al;
    split8<true>(av, 0.25f, ah, al);
    const _Float16 k = (_Float16)LO_INV;
    const h16x8 k8 = {k, k, k, k, k, k, k, k};
    const h16x8 as = ah * k8;
    SPLIT_PRIO_BURST();
    f32x4 oa = mfmah(as_h8(w2h), ah, f32x4{b2v.x, b2v.y, b2v.z, b2v.w});
    oa = mfmah(as_h8(w2h), al, oa);
    oa = mfmah(as_h8(w2l), as, oa);
    SPLIT_PRIO_VALU();
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = oa[r];
  }
  RIP_MARK("fwd_end");
}


struct Prefix16 {
  float H1[16];
  float dloc0, dloc1, s0, s1, lad;
};
constexpr int PRE_FLOATS = 72;  // per (model, observation): H1[64], dloc0, dloc1, s0, s1, lad, pad

__device__ __forceinline__ Prefix16 load_prefix(const float* __restrict__ p, int q) {
  Prefix16 pre;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float4 v = *reinterpret_cast<const float4*>(p + 16 * u + 4 * q);
    pre.H1[u * 4 + 0] = v.x, pre.H1[u * 4 + 1] = v.y, pre.H1[u * 4 + 2] = v.z, pre.H1[u * 4 + 3] = v.w;
  }
  pre.dloc0 = p[64], pre.dloc1 = p[65], pre.s0 = p[66], pre.s1 = p[67], pre.lad = p[68];
  return pre;
}

struct PassOut {
  float lad, sq;
};

// forward (x -> y, in place in `io`) or inverse (reads y from `io`) pass of the current model for this wave's 16
// candidates (flow_phase.hip:pass_forward with the split-f16 step).  MODE_FWD tapes all three heavy steps, MODE_INV
// tapes steps 1, 2 and hands step 3 over in registers (`last[2]`); REGTAPE: all three in registers.
template <int MODE, bool REGTAPE = false, bool PIPE = false>
__device__ __forceinline__ PassOut pass_forward(const uint4* wl, const Prefix16& pre, float (*io)[8], float (*st)[6][CB],
                                                float4* __restrict__ tape, StepTape* last, int c, int q, unsigned lane) {
  PassOut po;
  po.lad = pre.lad;
  po.sq = 0.f;
  float yp0, yp1;
  {
    float x0, x1;
    if (MODE == MODE_FWD) {
      x0 = io[c][0];
      x1 = io[c][1];
      yp0 = pre.dloc0 + pre.s0 * x0;
      yp1 = pre.dloc1 + pre.s1 * x1;
      po.sq = fmaf(x0, x0, x1 * x1);
      __builtin_amdgcn_wave_barrier();
      if (q == 0) {
        io[c][0] = yp0;
        io[c][1] = yp1;
      }
    } else {
      yp0 = io[c][0];
      yp1 = io[c][1];
      x0 = (yp0 - pre.dloc0) * rcpf_(pre.s0);
      x1 = (yp1 - pre.dloc1) * rcpf_(pre.s1);
      po.sq = fmaf(x0, x0, x1 * x1);
    }
    if (q == 0) {
      st[0][0][c] = x0;
      st[0][1][c] = x1;
      st[0][2][c] = pre.s0;
      st[0][3][c] = pre.s1;
    }
  }
  float H[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) H[i] = pre.H1[i];
  BSplit hs;
  split16(H, hs);
  auto coupling = [&](int t, const float (&o)[4]) __attribute__((always_inline)) {
    const float s0 = softplusf_(o[2]) + 1e-3f;  // sequence.py:133
    const float s1 = softplusf_(o[3]) + 1e-3f;
    float x0, x1, y0, y1;
    if (MODE == MODE_FWD) {
      x0 = io[c][2 * t];
      x1 = io[c][2 * t + 1];
      y0 = (yp0 + o[0]) + s0 * x0;  // sequence.py:136
      y1 = (yp1 + o[1]) + s1 * x1;
      po.sq = fmaf(x0, x0, fmaf(x1, x1, po.sq));
      __builtin_amdgcn_wave_barrier();
      if (q == 0) {
        io[c][2 * t] = y0;
        io[c][2 * t + 1] = y1;
      }
    } else {
      y0 = io[c][2 * t];
      y1 = io[c][2 * t + 1];
      x0 = (y0 - (yp0 + o[0])) * rcpf_(s0);  // sequence.py:196
      x1 = (y1 - (yp1 + o[1])) * rcpf_(s1);
      po.sq = fmaf(x0, x0, fmaf(x1, x1, po.sq));
    }
    po.lad += __logf(s0 * s1);
    if (q == 0) {
      st[t][0][c] = x0;
      st[t][1][c] = x1;
      st[t][2][c] = s0;
      st[t][3][c] = s1;
      st[t][4][c] = softplus_gradf_(o[2]);
      st[t][5][c] = softplus_gradf_(o[3]);
    }
    yp0 = y0;
    yp1 = y1;
  };
  // an opaque zero per step: the (loop-invariant) operand reads must not be merged across steps
  {
    float o[4];
    int zero = 0;
    asm volatile("" : "+v"(zero));
    if (REGTAPE)
      fwd_step<SAVE_REGS, PIPE>(wl + zero, H, hs, yp0, yp1, q, lane, nullptr, &last[0], o);
    else
      fwd_step<SAVE_TAPE_NOHP, PIPE>(wl + zero, H, hs, yp0, yp1, q, lane, tape, nullptr, o);
    coupling(1, o);
  }
  {
    float o[4];
    int zero = 0;
    asm volatile("" : "+v"(zero));
    if (REGTAPE)
      fwd_step<SAVE_REGS, PIPE>(wl + zero, H, hs, yp0, yp1, q, lane, nullptr, &last[1], o);
    else
      fwd_step<SAVE_TAPE, PIPE>(wl + zero, H, hs, yp0, yp1, q, lane, tape + TAPE_STEP_F4, nullptr, o);
    coupling(2, o);
  }
  {
    float o[4];
    int zero = 0;
    asm volatile("" : "+v"(zero));
    if (MODE == MODE_FWD)
      fwd_step<SAVE_TAPE, PIPE>(wl + zero, H, hs, yp0, yp1, q, lane, tape + 2 * TAPE_STEP_F4, nullptr, o);
    else
      fwd_step<SAVE_REGS, PIPE>(wl + zero, H, hs, yp0, yp1, q, lane, nullptr, &last[2], o);
    coupling(3, o);
  }
  return po;
}

// Gate gradients of one adjoint step as B operands, split at ONE per-candidate power-of-two scale:
// K blocks 0, 1 = d pre_r; 2, 3 = d pre_z (both feed W_hh^T of the next-earlier step and W_ih^T of this one);
// gn = d gh_n (W_hh^T), and d pre_n (W_ih^T only) is split locally.  `inv` = 1 / scale.
struct GSplit {
  h16x8 rz_hi[4], rz_lo[4];
  h16x8 gn_hi[2], gn_lo[2];
  float inv;
};

__device__ __forceinline__ float amax8(const float* v, float m) {
  m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])));
  m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
  m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[4]), __builtin_fabsf(v[5])));
  m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[6]), __builtin_fabsf(v[7])));
  return m;
}

// One step t of the adjoint (flow_phase.hip:adj_step with split-f16 contractions).
//   dh_t = W1^T da1_t [12 f16 MFMAs, own scale] + W_hh^T (dpr, dpz, dgh_n)_{t+1} [72 f16 MFMAs, `gs` from step t+1]
//   du_t = W_ih^T (dpr, dpz, dpn)_t [18 f16 MFMAs, the scale of this step's gate gradients]
// + 2 (W2^T) + 4 (gi_n) fp32 MFMAs: 1824 matrix-pipe cycles (flow_phase.hip: 278 x 32 = 8896).  One accumulator per tile
// (TW_SHIFT above).
// tw: this lane's column of the transposed rows; wtab: this lane's entry (q * 2 + (c & 1)) of the W_ih^T table, one
// group of 8 entries per (kb, term); wl: the forward rows (gi_n's k-step).  LASTSTEP (t = 1): nothing consumes the
// gate gradients as W_hh^T operands any more.
template <int MODE, int TS, bool FROM_REGS, bool TAPE_EARLY = true>
__device__ __forceinline__ void adj_step(const uint4* tw_in, const uint4* wtab_in, const uint4* wl_in, const float (*io)[8],
                                         const float (*gin)[8], const float (*st)[6][CB], const float4* __restrict__ tp,
                                         const StepTape* tr, const float* hp1, int c, int q, float w0, float (&dhz)[16],
                                         GSplit& gs, float& carry0, float& carry1, float (&res)[8]) {
  constexpr bool FIRST = TS == T - 1;
  constexpr bool LASTSTEP = TS == 1;
  RIP_MARK("adj_begin");
  int zero = 0;
  asm volatile("" : "+v"(zero));  // keeps the operand reads of this step from being merged with another step's
  const uint4* tw = tw_in + zero;
  const uint4* wtab = wtab_in + zero;
  // ---- the step's tape ----
  // TAPE_EARLY: all rows are requested BEFORE the contraction (their latency hides under its MFMAs; 64 more live
  // registers during it: the one-wave-per-SIMD builds).  Otherwise only the ReLU mask comes first and the rows are
  // requested after the contraction (two waves per SIMD: 256 registers each, the partner wave covers the latency).
  StepTape tl;
  const StepTape* tv = tr;
  unsigned tape_loff = 0;
  if (!FROM_REGS) {
    const unsigned lane = (unsigned)(q * 16 + c);
    tape_loff = lane * 16u;
    asm volatile("" : "+v"(tape_loff));
    tl.mask = RIP_ABL == 1 ? 0x5au
                           : *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(tp + TAPE_ROWS * 64) + (tape_loff >> 2));
    tv = &tl;
  }
  auto load_tape_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int up = 0; up < 4; ++up) {
      const float4 rr = tape_ld(trow(tp, up * 4 + 0, tape_loff)), zz = tape_ld(trow(tp, up * 4 + 1, tape_loff));
      const float4 gh = tape_ld(trow(tp, up * 4 + 3, tape_loff));
      float4 hp;
      if (TS == 1)
        hp = *reinterpret_cast<const float4*>(hp1 + 16 * up + 4 * q);  // prefix H1 (global, L2)
      else
        hp = tape_ld(trow(tp, 16 + up, tape_loff));
      tl.r[up * 4 + 0] = rr.x, tl.r[up * 4 + 1] = rr.y, tl.r[up * 4 + 2] = rr.z, tl.r[up * 4 + 3] = rr.w;
      tl.z[up * 4 + 0] = zz.x, tl.z[up * 4 + 1] = zz.y, tl.z[up * 4 + 2] = zz.z, tl.z[up * 4 + 3] = zz.w;
      tl.gh[up * 4 + 0] = gh.x, tl.gh[up * 4 + 1] = gh.y, tl.gh[up * 4 + 2] = gh.z, tl.gh[up * 4 + 3] = gh.w;
      tl.hp[up * 4 + 0] = hp.x, tl.hp[up * 4 + 1] = hp.y, tl.hp[up * 4 + 2] = hp.z, tl.hp[up * 4 + 3] = hp.w;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  if (!FROM_REGS && TAPE_EARLY) load_tape_rows();
  const float x0 = st[TS][0][c], x1 = st[TS][1][c], s0 = st[TS][2][c], s1 = st[TS][3][c];
  const float sg0 = st[TS][4][c], sg1 = st[TS][5][c];
  float dd0, dd1, dos0, dos1, c0, c1;
  if (MODE == MODE_INV) {
    const float i0 = rcpf_(s0), i1 = rcpf_(s1);
    const float xs0 = x0 * i0, xs1 = x1 * i1;
    res[2 * TS] = carry0 - xs0;
    res[2 * TS + 1] = carry1 - xs1;
    c0 = xs0;
    c1 = xs1;
    dd0 = xs0;
    dd1 = xs1;
    dos0 = (x0 * x0 - 1.0f) * i0 * sg0;
    dos1 = (x1 * x1 - 1.0f) * i1 * sg1;
  } else {
    const float D0 = gin[c][2 * TS] + carry0;
    const float D1 = gin[c][2 * TS + 1] + carry1;
    res[2 * TS] = fmaf(D0, s0, w0 * x0);
    res[2 * TS + 1] = fmaf(D1, s1, w0 * x1);
    c0 = D0;
    c1 = D1;
    dd0 = D0;
    dd1 = D1;
    dos0 = (D0 * x0 + w0 * rcpf_(s0)) * sg0;
    dos1 = (D1 * x1 + w0 * rcpf_(s1)) * sg1;
  }
  // ---- head adjoint: da1 = relu'(a1) * W2^T do (fp32 MFMA: 4 inputs per candidate) ----
  const float4 w2t = as_f4(tw[0]);
  const float bdo = q == 0 ? dd0 : (q == 1 ? dd1 : (q == 2 ? dos0 : dos1));
  const f32x4 da0 = mfma4(w2t.x, bdo, zero4());
  const f32x4 da1 = mfma4(w2t.y, bdo, zero4());
  const unsigned mask = tv->mask;
  float da1r[8];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    da1r[r] = (mask >> r) & 1u ? da0[r] : 0.f;
    da1r[4 + r] = (mask >> (4 + r)) & 1u ? da1[r] : 0.f;
  }
  // ---- dh_t, part 1: W1^T da1 at da1's own per-candidate scale ----
  f32x4 dh[4];
  {
    float sa, ia;
    pow2_scale(qmax(amax8(da1r, 0.f)), sa, ia);
    h16x8 ah, al;
    split8<true>(da1r, sa, ah, al);
    SPLIT_PRIO_BURST();
    // rows 1 + ut * 2 (hi), 2 + ut * 2 (lo); term by term, so that an accumulator is touched every fourth MFMA
    f32x4 a[4];
#pragma unroll
    for (int ut = 0; ut < 4; ++ut) a[ut] = mfmah(as_h8(tw[(1 + ut * 2) * 64]), ah, zero4());
#pragma unroll
    for (int ut = 0; ut < 4; ++ut) a[ut] = mfmah(as_h8(tw[(1 + ut * 2) * 64]), al, a[ut]);
#pragma unroll
    for (int ut = 0; ut < 4; ++ut) a[ut] = mfmah(as_h8(tw[(2 + ut * 2) * 64]), ah, a[ut]);
    SPLIT_PRIO_VALU();
#pragma unroll
    for (int ut = 0; ut < 4; ++ut) dh[ut] = a[ut] * ia;
  }
  // ---- part 2: W_hh^T (dpr, dpz, dgh_n)_{t+1}: 6 K blocks x 4 unit tiles, rows 9 + (kb * 4 + ut) * 2 + term ----
  if (!FIRST) {
    // dh'_{t+1} z_{t+1} joins here, before the big contraction, so that its 16 registers are free during it
#pragma unroll
    for (int i = 0; i < 16; ++i) dh[i >> 2][i & 3] += dhz[i];
    // group (half, kb): unit tiles ut = 2 half + u (u = 0, 1: 8 accumulator registers), hi row 9 + (kb * 4 + ut) * 2, lo
    // row behind it; six MFMAs (hi hi) u0 u1, (hi lo) u0 u1, (lo hi) u0 u1: the two accumulators alternate.  The hi rows of the next group are requested at the top of a group, its own lo rows as well.
    auto row_of = [](int gi, int u, int term) {
      const int half = gi / 6, kb = gi % 6;
      return (9 + (kb * 4 + 2 * half + u) * 2 + term) * 64;
    };
    const float ig = gs.inv;
    uint4 RH[2][2], RL[2];
    SPLIT_PRIO_BURST();
    RH[0][0] = tw[row_of(0, 0, 0)];
    RH[0][1] = tw[row_of(0, 1, 0)];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 a[2] = {zero4(), zero4()};
#pragma unroll
      for (int kb = 0; kb < 6; ++kb) {
        const int gi = half * 6 + kb;
        const h16x8 bh = kb < 4 ? gs.rz_hi[kb < 4 ? kb : 0] : gs.gn_hi[kb < 4 ? 0 : kb - 4];
        const h16x8 bl = kb < 4 ? gs.rz_lo[kb < 4 ? kb : 0] : gs.gn_lo[kb < 4 ? 0 : kb - 4];
        RL[0] = tw[row_of(gi, 0, 1)];
        RL[1] = tw[row_of(gi, 1, 1)];
        if (gi + 1 < 12) {
          RH[(gi + 1) & 1][0] = tw[row_of(gi + 1, 0, 0)];
          RH[(gi + 1) & 1][1] = tw[row_of(gi + 1, 1, 0)];
        }
        __builtin_amdgcn_sched_barrier(0);
        a[0] = mfmah(as_h8(RH[gi & 1][0]), bh, a[0]);
        a[1] = mfmah(as_h8(RH[gi & 1][1]), bh, a[1]);
        a[0] = mfmah(as_h8(RH[gi & 1][0]), bl, a[0]);
        a[1] = mfmah(as_h8(RH[gi & 1][1]), bl, a[1]);
        a[0] = mfmah(as_h8(RL[0]), bh, a[0]);
        a[1] = mfmah(as_h8(RL[1]), bh, a[1]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) dh[2 * half + u] = dh[2 * half + u] + a[u] * ig;
    }
    SPLIT_PRIO_VALU();
  }
  if (!FROM_REGS && !TAPE_EARLY) {
    __builtin_amdgcn_sched_barrier(0);
    load_tape_rows();
  }
  // ---- n of this step: tanh(gi_n + r gh_n), gi_n = the (W_in[.][0], W_in[.][1], b_in, 0) k-step on y_{t-1} ----
  float nrec[16];
  if (!FROM_REGS) {
    const float4 wxg = as_f4((wl_in + zero)[50 * 64]);
    const float yp0 = io[c][2 * (TS - 1)], yp1 = io[c][2 * (TS - 1) + 1];
    const float bin = q == 0 ? yp0 : (q == 1 ? yp1 : (q == 2 ? 1.f : 0.f));
    f32x4 agn_t[4];
    agn_t[0] = mfma4(wxg.x, bin, zero4());
    agn_t[1] = mfma4(wxg.y, bin, zero4());
    agn_t[2] = mfma4(wxg.z, bin, zero4());
    agn_t[3] = mfma4(wxg.w, bin, zero4());
    constexpr float L2E = 1.4426950408889634f;
    const f32x2 one = {1.0f, 1.0f}, two = {2.0f, 2.0f};
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const f32x2 r2 = {tl.r[i], tl.r[i + 1]}, ghn = {tl.gh[i], tl.gh[i + 1]};
      const f32x2 gin2 = {agn_t[i >> 2][i & 3], agn_t[i >> 2][(i & 3) + 1]};
      const f32x2 pre = __builtin_elementwise_fma(r2, ghn, gin2);
      const f32x2 pn = pre * f32x2{2.0f * L2E, 2.0f * L2E};
      const f32x2 en = {__builtin_amdgcn_exp2f(pn.x), __builtin_amdgcn_exp2f(pn.y)};
      const f32x2 dn = en + one;
      const f32x2 in2 = {rcpf_(dn.x), rcpf_(dn.y)};
      const f32x2 n2 = one - two * in2;
      nrec[i] = n2.x;
      nrec[i + 1] = n2.y;
    }
  }
  // ---- GRUCell adjoint, lane-local in the H layout ----
  float dpn[16], dgh[48];  // dgh: d pre_r (0-15), d pre_z (16-31), d gh_n (32-47)
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    const f32x2 hp2 = {tv->hp[i], tv->hp[i + 1]}, rr2 = {tv->r[i], tv->r[i + 1]}, zz2 = {tv->z[i], tv->z[i + 1]};
    const f32x2 nn2 = {FROM_REGS ? tv->n[i] : nrec[i], FROM_REGS ? tv->n[i + 1] : nrec[i + 1]}, gh2 = {tv->gh[i], tv->gh[i + 1]};
    const f32x2 one = {1.0f, 1.0f};
    const f32x2 d = {dh[i >> 2][i & 3], dh[i >> 2][(i & 3) + 1]};  // (dh'_{t+1} z_{t+1} already added)
    const f32x2 dn = d * (one - zz2);
    const f32x2 dzg = d * (hp2 - nn2);
    const f32x2 dhzn = d * zz2;
    const f32x2 dp = dn * (one - nn2 * nn2);
    const f32x2 dr = dp * gh2;
    const f32x2 dgn = dp * rr2;
    const f32x2 dpr = dr * rr2 * (one - rr2);
    const f32x2 dpz = dzg * zz2 * (one - zz2);
    dhz[i] = dhzn.x, dhz[i + 1] = dhzn.y;
    dpn[i] = dp.x, dpn[i + 1] = dp.y;
    dgh[32 + i] = dgn.x, dgh[33 + i] = dgn.y;
    dgh[i] = dpr.x, dgh[1 + i] = dpr.y;
    dgh[16 + i] = dpz.x, dgh[17 + i] = dpz.y;
  }
  // ---- this step's gate gradients as B operands: one per-candidate scale ----
  float m = amax8(&dgh[0], 0.f);
  m = amax8(&dgh[8], m);
  m = amax8(&dgh[16], m);
  m = amax8(&dgh[24], m);
  if (!LASTSTEP) {
    m = amax8(&dgh[32], m);
    m = amax8(&dgh[40], m);
  }
  m = amax8(&dpn[0], m);
  m = amax8(&dpn[8], m);
  float sg, ig;
  pow2_scale(qmax(m), sg, ig);
  gs.inv = ig;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) split8<true>(&dgh[8 * kb], sg, gs.rz_hi[kb], gs.rz_lo[kb]);
  if (!LASTSTEP) {
    split8<true>(&dgh[32], sg, gs.gn_hi[0], gs.gn_lo[0]);
    split8<true>(&dgh[40], sg, gs.gn_hi[1], gs.gn_lo[1]);
  }
  h16x8 pn_hi[2], pn_lo[2];
  split8<true>(&dpn[0], sg, pn_hi[0], pn_lo[0]);
  split8<true>(&dpn[8], sg, pn_hi[1], pn_lo[1]);
  // ---- du = W_ih^T (dpr, dpz, dpn): 6 K blocks, one 16-row tile whose rows m hold input dim m & 1 ----
  f32x4 ua = zero4(), ul = zero4(), ub = zero4();
  SPLIT_PRIO_BURST();
#pragma unroll
  for (int kb = 0; kb < 6; ++kb) {
    const h16x8 wh = as_h8(wtab[(kb * 2 + 0) * 8]), wo = as_h8(wtab[(kb * 2 + 1) * 8]);
    const h16x8 bh = kb < 4 ? gs.rz_hi[kb < 4 ? kb : 0] : pn_hi[kb < 4 ? 0 : kb - 4];
    const h16x8 bl = kb < 4 ? gs.rz_lo[kb < 4 ? kb : 0] : pn_lo[kb < 4 ? 0 : kb - 4];
    ua = mfmah(wh, bh, ua);
    ul = mfmah(wh, bl, ul);
    ub = mfmah(wo, bh, ub);
  }
  SPLIT_PRIO_VALU();
  carry0 = c0 + (ua[0] + (ul[0] + ub[0])) * ig;  // (three accumulators: one would make the 18 MFMAs a dependent chain)
  carry1 = c1 + (ua[1] + (ul[1] + ub[1])) * ig;
  RIP_MARK("adj_end");
}

// adjoint pass of the current model (flow_phase.hip:pass_backward)
template <int MODE, bool REGTAPE = false, bool TAPE_EARLY = true>
__device__ __forceinline__ void pass_backward(const uint4* tw, const uint4* wtab, const uint4* wl, const float (*io)[8],
                                              const float (*gin)[8], const float (*st)[6][CB], const float4* __restrict__ tape,
                                              const StepTape* last, const float* hp1, int c, int q, float (&res)[8], float w0) {
  float dhz[16];
  GSplit gs;
  float carry0 = 0.f, carry1 = 0.f;
  if (MODE == MODE_INV)
    adj_step<MODE, 3, true>(tw, wtab, wl, io, gin, st, nullptr, &last[2], hp1, c, q, w0, dhz, gs, carry0, carry1, res);
  else
    adj_step<MODE, 3, false, TAPE_EARLY>(tw, wtab, wl, io, gin, st, tape + 2 * TAPE_STEP_F4, nullptr, hp1, c, q, w0, dhz, gs, carry0, carry1, res);
  if (REGTAPE) {
    adj_step<MODE, 2, true>(tw, wtab, wl, io, gin, st, nullptr, &last[1], hp1, c, q, w0, dhz, gs, carry0, carry1, res);
    adj_step<MODE, 1, true>(tw, wtab, wl, io, gin, st, nullptr, &last[0], hp1, c, q, w0, dhz, gs, carry0, carry1, res);
  } else {
    adj_step<MODE, 2, false, TAPE_EARLY>(tw, wtab, wl, io, gin, st, tape + TAPE_STEP_F4, nullptr, hp1, c, q, w0, dhz, gs, carry0, carry1, res);
    adj_step<MODE, 1, false, TAPE_EARLY>(tw, wtab, wl, io, gin, st, tape, nullptr, hp1, c, q, w0, dhz, gs, carry0, carry1, res);
  }
  const float x0 = st[0][0][c], x1 = st[0][1][c], s0 = st[0][2][c], s1 = st[0][3][c];
  if (MODE == MODE_INV) {
    res[0] = carry0 - x0 * rcpf_(s0);
    res[1] = carry1 - x1 * rcpf_(s1);
  } else {
    res[0] = fmaf(gin[c][0] + carry0, s0, w0 * x0);
    res[1] = fmaf(gin[c][1] + carry1, s1, w0 * x1);
  }
}

}  // namespace split
}  // namespace rip
