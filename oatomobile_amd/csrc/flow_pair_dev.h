// Device functions of the PAIRED split-f16 plan search (flow_pair.hip), round 6.
//
// flow_split_dev.h runs a 16-candidate block on ONE wave at one wave per SIMD: the wave's time is the SUM of its
// matrix-pipe cycles (2208 per GRU + head step) and its vector-issue cycles (~2240) — MFMAs and vector work of one
// wave do not overlap on this chip, two waves on a SIMD do.  The two-waves-per-SIMD build of that kernel
// has 256 registers per wave, which is less than its register tape (240): its tape goes to global memory and it loses.
// Here a block is run by a PAIR of waves that split the 64 hidden units: wave `hw` owns unit tiles 2 hw, 2 hw + 1
// (H layout: its lanes hold 8 of the 16 values a lane of the one-wave kernel holds), i.e. half of the gate math, half
// of the MFMAs, half of the tape (120 registers) — eight waves per workgroup, two per SIMD, the tape still in registers.
// What crosses between the two waves of a pair, through LDS (2.5 KB per wave, `PairXchg`):
//   forward step   (a) after the gates: the new state's B operand of the own K block (hi, lo: 32 B per lane) — a K block
//                      of the 64-deep contraction IS the two unit tiles of one wave;
//                  (b) after the head: the partial head output W2 relu(a1_own tile) (4 floats) and the tile's ReLU mask;
//   adjoint step   the W_hh^T contraction is split over K (each wave contracts its OWN gate gradients against the rows
//                      of all four output tiles): the partial dh of the PEER's tiles (32 B per lane) and the partial
//                      du = W_ih^T (...) (2 floats) go across, once per step; each wave scales by its own power of two.
// Everything per-candidate (couplings, log-dets, goal terms, Adam) is computed by BOTH waves from bit-identical inputs
// (sums of the two partials are written as own + peer: commutative), so the two waves take the same branches and may
// write the same values to the pair's shared scalars without further synchronisation.
// The arithmetic per product is flow_split_dev.h's (two-term binary16 operands, fp32 accumulate); what differs is the
// order of some fp32 sums (own K block first; partial sums over K in the adjoint).  Gates: the same tests.
#pragma once
#include "flow_split_dev.h"

namespace rip {
namespace split {

// ---- the exchange between the two waves of a pair ----
// One slot per wave (2 rows of 64 x 16 B + 64 x 8 B), a flag (payloads published) and an ack (peer payloads consumed).
// LDS executes one wave's instructions in order, so "payload stores, then the flag store" and "payload loads, then the
// ack store" need no fence; every access is volatile so that the compiler keeps them in program order as well.
// The pointers carry the LDS address space explicitly: through generic pointers every access of the exchange compiled
// to a FLAT instruction with system scope and `s_waitcnt vmcnt(0) lgkmcnt(0)` — each poll of the peer's word waited for
// the wave's outstanding global tape stores as well (first build: 3.58 ms per launch against the one-wave shape's 2.51).
#define RIP_LDS __attribute__((address_space(3)))
// wave priority around matrix bursts.  RIP_PAIR_PRIO: 0 = never touched, 1 = raised for MFMA bursts (flow_split_dev.h's scheme),
// 2 = raised for the vector phases instead, 3 = static: the younger half of the workgroup (waves 4..7) runs at priority 1
#ifndef RIP_PAIR_PRIO
#define RIP_PAIR_PRIO 1
#endif
#define PAIR_PRIO_BURST() do { if (RIP_PAIR_PRIO == 1) __builtin_amdgcn_s_setprio(1); else if (RIP_PAIR_PRIO == 2) __builtin_amdgcn_s_setprio(0); } while (0)
#define PAIR_PRIO_VALU() do { if (RIP_PAIR_PRIO == 1) __builtin_amdgcn_s_setprio(0); else if (RIP_PAIR_PRIO == 2) __builtin_amdgcn_s_setprio(1); } while (0)

struct PairXchg {
  volatile RIP_LDS u32x4* my_rows;          // + lane; row r at [r * 64]  (native vectors: HIP's uint4 is a struct without volatile members)
  const volatile RIP_LDS u32x4* peer_rows;
  volatile RIP_LDS f32x2* my_extra;         // + lane
  const volatile RIP_LDS f32x2* peer_extra;
  volatile RIP_LDS unsigned* my_ctl;        // [0] flag, [1] ack
  const volatile RIP_LDS unsigned* peer_ctl;
  unsigned seq;                     // payloads published so far (both waves of a pair publish in lockstep)
#ifdef RIP_PROFILE_TICKS
  long long spin_ack = 0, spin_data = 0;  // development: cycles spent waiting for the peer
#endif
};

#ifndef RIP_PAIR_SLEEP
#define RIP_PAIR_SLEEP 1  // s_sleep argument between two polls of the peer's word (0: poll back to back)
#endif
__device__ __forceinline__ void xch_spin(const volatile RIP_LDS unsigned* p, unsigned need) {
  while ((unsigned)__builtin_amdgcn_readfirstlane((int)*p) < need) {
    if (RIP_PAIR_SLEEP > 0) __builtin_amdgcn_s_sleep(RIP_PAIR_SLEEP);
  }
}
#ifdef RIP_PROFILE_TICKS
#define XCH_TIMED(acc_, stmt_) do { const long long t0_ = clock64(); stmt_; acc_ += clock64() - t0_; } while (0)
#else
#define XCH_TIMED(acc_, stmt_) do { stmt_; } while (0)
#endif
// before overwriting my slot: the peer has consumed my previous payload
__device__ __forceinline__ void xch_begin(PairXchg& x) { XCH_TIMED(x.spin_ack, xch_spin(x.peer_ctl + 1, x.seq)); }
__device__ __forceinline__ void xch_publish(PairXchg& x) {
  x.seq += 1;
  x.my_ctl[0] = x.seq;
}
// the peer's payload number `seq` (the one that matches my last published one) is in its slot
__device__ __forceinline__ void xch_wait(PairXchg& x) { XCH_TIMED(x.spin_data, xch_spin(x.peer_ctl, x.seq)); }
__device__ __forceinline__ void xch_done(PairXchg& x) { x.my_ctl[1] = x.seq; }

__device__ __forceinline__ u32x4 h8_bits(h16x8 v) { return __builtin_bit_cast(u32x4, v); }
__device__ __forceinline__ u32x4 f4_bits(float a, float b, float c, float d) {
  const u32x4 u = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
  return u;
}

// B operands of the 64-unit state as one wave of a pair sees them: its own K block and the peer's
struct HSplit {
  h16x8 own_hi, own_lo, own_hs, peer_hi, peer_lo, peer_hs;
};
__device__ __forceinline__ h16x8 scale_hs(h16x8 hi) {
  const _Float16 k = (_Float16)LO_INV;
  const h16x8 k8 = {k, k, k, k, k, k, k, k};
  return hi * k8;
}

// adjoint tape of one step, own units only
struct HalfTape {
  float hp[8], r[8], z[8], n[8], gh[8];
  unsigned mask;  // ReLU mask of BOTH a1 tiles (8 bits)
};

__device__ __forceinline__ float pick4(const float4& v, int i) {  // i is wave-uniform
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// ---- forward / inverse pass, this wave's half (flow_split_dev.h:fwd_step / pass_forward), software-pipelined over the
// two exchanges of a step.  The 64-deep contraction of a unit tile is the sum of two K blocks, the wave's own (its own
// units of h) and the peer's; the k-steps that multiply y_{t-1} go LAST into the same accumulators.  So the tile MFMAs
// of step t + 1 are issued inside step t, where the wave would otherwise wait for its peer:
//     k-steps(y_{t-1}) -> gates(t) -> split own h -> publish (a)
//     -> own-K-block tile MFMAs of step t + 1, own-K-block head MFMAs        [the peer's K block is in flight]
//     -> wait (a) -> peer-K-block head MFMAs, W2 -> publish (b)
//     -> peer-K-block tile MFMAs of step t + 1                               [the peer's partial head output is in flight]
//     -> wait (b) -> coupling(t)
// (a) = the new state's own K block as B operands (hi, lo: 32 B per lane), (b) = the partial head output W2 relu(a1_own)
// (4 floats) and the own a1 tile's ReLU mask.  `H` = the own 8 units of the state, `hs` = the B operands of the WHOLE state.
struct TileAcc2 {
  f32x4 a[2][3];  // own tiles u = 0, 1: pre_r, pre_z, gh_n (without their y / bias k-steps)
};

// the 2 x 9 tile MFMAs of one K block: rows (gate g, own tile u, term) at `wk` + (g * 16 + 4 u + term) * 64
template <bool INIT>
__device__ __forceinline__ void pair_tiles_kb(const uint4* wk, h16x8 bhi, h16x8 blo, h16x8 bhs, TileAcc2& t) {
  auto row = [](int g, int u, int term) { return (g * 16 + 4 * u + term) * 64; };
  uint4 RH[2][3], RL[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) RH[0][g] = wk[row(g, 0, 0)];
  PAIR_PRIO_BURST();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
#pragma unroll
    for (int g = 0; g < 3; ++g) RL[g] = wk[row(g, u, 1)];
    if (u == 0) {
#pragma unroll
      for (int g = 0; g < 3; ++g) RH[1][g] = wk[row(g, 1, 0)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 3; ++g) t.a[u][g] = mfmah(as_h8(RH[u][g]), bhi, INIT ? zero4() : t.a[u][g]);
#pragma unroll
    for (int g = 0; g < 3; ++g) t.a[u][g] = mfmah(as_h8(RH[u][g]), blo, t.a[u][g]);
#pragma unroll
    for (int g = 0; g < 3; ++g) t.a[u][g] = mfmah(as_h8(RL[g]), bhs, t.a[u][g]);
  }
  PAIR_PRIO_VALU();
}

template <int MODE>
__device__ __forceinline__ PassOut pass_forward_pair(const uint4* wl, int hw, const Prefix16& pre, const float (*xs)[8],
                                                     float (*ys)[8], float (*st)[6][CB], float4* __restrict__ tape,
                                                     HalfTape* last, int c, int q, unsigned lane, PairXchg& x) {
  PassOut po;
  po.lad = pre.lad;
  po.sq = 0.f;
  float yp0, yp1;
  {
    float x0, x1;
    if (MODE == MODE_FWD) {
      x0 = xs[c][0];
      x1 = xs[c][1];
      yp0 = pre.dloc0 + pre.s0 * x0;
      yp1 = pre.dloc1 + pre.s1 * x1;
      if (q == 0) {
        ys[c][0] = yp0;
        ys[c][1] = yp1;
      }
    } else {
      yp0 = ys[c][0];
      yp1 = ys[c][1];
      x0 = (yp0 - pre.dloc0) * rcpf_(pre.s0);
      x1 = (yp1 - pre.dloc1) * rcpf_(pre.s1);
    }
    po.sq = fmaf(x0, x0, x1 * x1);
    if (q == 0) {
      st[0][0][c] = x0;
      st[0][1][c] = x1;
      st[0][2][c] = pre.s0;
      st[0][3][c] = pre.s1;
    }
  }
  // the prefix state h_1 is candidate independent and known to both waves: no exchange for the first step's operands
  HSplit hs;
  float H[8];
  {
    float h_own[8], h_peer[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      h_own[i] = hw ? pre.H1[8 + i] : pre.H1[i];
      h_peer[i] = hw ? pre.H1[i] : pre.H1[8 + i];
      H[i] = h_own[i];
    }
    split8<false>(h_own, 1.f, hs.own_hi, hs.own_lo);
    split8<false>(h_peer, 1.f, hs.peer_hi, hs.peer_lo);
    hs.own_hs = scale_hs(hs.own_hi);
    hs.peer_hs = scale_hs(hs.peer_hi);
  }
  unsigned loff = lane * 16u;
  asm volatile("" : "+v"(loff));
  // operand rows (flow.h MHF_*): W_hh row of (gate g, tile up, K block kb, term) = g * 16 + up * 4 + kb * 2 + term with
  // up = 2 hw + u; own K block kb = hw, the peer's 1 - hw.  Head rows of the own a1 tile mt = hw: 52 + (mt * 2 + kb) * 2 + term.
  const uint4* wown = wl + (10 * hw) * 64;
  const uint4* wpeer = wl + (6 * hw + 2) * 64;
  const uint4* h_own = wl + (52 + 6 * hw) * 64;
  const uint4* h_peer = wl + (54 + 2 * hw) * 64;
  const float bone = q == 2 ? 1.f : 0.f;
  TileAcc2 acc;
  pair_tiles_kb<true>(wown, hs.own_hi, hs.own_lo, hs.own_hs, acc);
  pair_tiles_kb<false>(wpeer, hs.peer_hi, hs.peer_lo, hs.peer_hs, acc);
#pragma unroll
  for (int t = 1; t < T; ++t) {
    RIP_MARK("fwd_begin");
    int zero = 0;
    asm volatile("" : "+v"(zero));  // the (loop-invariant) operand reads must not be merged across steps
    const uint4* wz = wl + zero;
    float4* tp = MODE == MODE_FWD ? tape + (t - 1) * TAPE_STEP_F4 : nullptr;
    HalfTape* tr = MODE == MODE_INV ? &last[t - 1] : nullptr;
    // ---- the y / bias k-steps on top of the tile accumulators, then the gates of the own two tiles ----
    const float bin = q == 0 ? yp0 : (q == 1 ? yp1 : (q == 2 ? 1.f : 0.f));
    const float4 wxr = as_f4(wz[48 * 64]), wxz = as_f4(wz[49 * 64]), wxg = as_f4(wz[50 * 64]), wxh = as_f4(wz[51 * 64]);
    float Hn[8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int up = 2 * hw + u;
      PAIR_PRIO_BURST();
      const f32x4 ar = mfma4(pick4(wxr, up), bin, acc.a[u][0]);
      const f32x4 az = mfma4(pick4(wxz, up), bin, acc.a[u][1]);
      const f32x4 agn = mfma4(pick4(wxg, up), bin, zero4());
      const f32x4 ahn = mfma4(pick4(wxh, up), bin, acc.a[u][2]);
      PAIR_PRIO_VALU();
      float rr[4], zz[4], nn[4];
      gru_gates(ar, az, agn, ahn, &H[u * 4], &Hn[u * 4], rr, zz, nn);
      asm volatile("" : "+v"(Hn[u * 4]), "+v"(Hn[u * 4 + 1]), "+v"(Hn[u * 4 + 2]), "+v"(Hn[u * 4 + 3]));
      if (MODE == MODE_FWD) {
        float4* tpu = tp + (8 * hw + 4 * u) * 64;
        tape_st(trow(tpu, 0, loff), rr[0], rr[1], rr[2], rr[3]);
        tape_st(trow(tpu, 1, loff), zz[0], zz[1], zz[2], zz[3]);
        tape_st(trow(tpu, 3, loff), ahn[0], ahn[1], ahn[2], ahn[3]);
        if (t > 1) tape_st(trow(tp + (16 + 2 * hw + u) * 64, 0, loff), H[u * 4], H[u * 4 + 1], H[u * 4 + 2], H[u * 4 + 3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          tr->hp[u * 4 + r] = H[u * 4 + r];
          tr->r[u * 4 + r] = rr[r];
          tr->z[u * 4 + r] = zz[r];
          tr->n[u * 4 + r] = nn[r];
          tr->gh[u * 4 + r] = ahn[r];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) H[i] = Hn[i];
    RIP_MARK("fwd_tiles_done");
    // ---- exchange (a): the new state's own K block ----
    split8<false>(H, 1.f, hs.own_hi, hs.own_lo);
    xch_begin(x);
    x.my_rows[0] = h8_bits(hs.own_hi);
    x.my_rows[64] = h8_bits(hs.own_lo);
    xch_publish(x);
    hs.own_hs = scale_hs(hs.own_hi);
    if (t + 1 < T) pair_tiles_kb<true>(wown + zero, hs.own_hi, hs.own_lo, hs.own_hs, acc);  // step t + 1, own K block
    const float4 t60 = as_f4(wz[60 * 64]), t61 = as_f4(wz[61 * 64]), t62 = as_f4(wz[62 * 64]);
    const uint4 woh = (h_own + zero)[0], wol = (h_own + zero)[64];
    const uint4 wph = (h_peer + zero)[0], wpl = (h_peer + zero)[64];
    PAIR_PRIO_BURST();
    f32x4 a0 = mfma4(hw ? t60.y : t60.x, bone, zero4());
    a0 = mfmah(as_h8(woh), hs.own_hi, a0);
    f32x4 a1 = mfmah(as_h8(woh), hs.own_lo, zero4());
    a0 = mfmah(as_h8(wol), hs.own_hs, a0);
    PAIR_PRIO_VALU();
    xch_wait(x);
    {
      const u32x4 ph = x.peer_rows[0], pl = x.peer_rows[64];
      hs.peer_hi = __builtin_bit_cast(h16x8, ph);
      hs.peer_lo = __builtin_bit_cast(h16x8, pl);
    }
    xch_done(x);
    hs.peer_hs = scale_hs(hs.peer_hi);
    PAIR_PRIO_BURST();
    a1 = mfmah(as_h8(wph), hs.peer_hi, a1);
    a0 = mfmah(as_h8(wph), hs.peer_lo, a0);
    a1 = mfmah(as_h8(wpl), hs.peer_hs, a1);
    PAIR_PRIO_VALU();
    const f32x4 av = a0 + a1;
    unsigned m4 = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) m4 |= av[r] > 0.f ? (1u << r) : 0u;
    // W2 k-steps of the own a1 tile: tile 0 = (t60.z, t60.w, t61.x, t61.y), tile 1 = (t61.z, t61.w, t62.x, t62.y); b2 = t62.z (wave 0)
    PAIR_PRIO_BURST();
    f32x4 oa = mfma4(hw ? t61.z : t60.z, fmaxf(av[0], 0.f), zero4());
    oa = mfma4(hw ? t61.w : t60.w, fmaxf(av[1], 0.f), oa);
    oa = mfma4(hw ? t62.x : t61.x, fmaxf(av[2], 0.f), oa);
    oa = mfma4(hw ? t62.y : t61.y, fmaxf(av[3], 0.f), oa);
    oa = mfma4(hw ? 0.f : t62.z, bone, oa);
    PAIR_PRIO_VALU();
    // ---- exchange (b): the partial head output and the tile's ReLU mask ----
    xch_begin(x);
    x.my_rows[0] = f4_bits(oa[0], oa[1], oa[2], oa[3]);
    *x.my_extra = f32x2{__uint_as_float(m4), 0.f};
    xch_publish(x);
    if (t + 1 < T) pair_tiles_kb<false>(wpeer + zero, hs.peer_hi, hs.peer_lo, hs.peer_hs, acc);  // step t + 1, the peer's K block
    xch_wait(x);
    const u32x4 pob = x.peer_rows[0];
    const f32x2 pe = *x.peer_extra;
    xch_done(x);
    const unsigned m4p = __float_as_uint(pe.x);
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = oa[r] + __uint_as_float(pob[r]);
    const unsigned mask8 = hw ? (m4p | (m4 << 4)) : (m4 | (m4p << 4));
    if (MODE == MODE_INV) tr->mask = mask8;
    if (MODE == MODE_FWD && RIP_ABL != 3)
      *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(tp + TAPE_ROWS * 64) + (loff >> 2)) = mask8;  // (both waves: the same word)
    RIP_MARK("fwd_end");
    // ---- coupling (sequence.py:133-136 / :193-196), on both waves ----
    {
      const float s0 = softplusf_(o[2]) + 1e-3f;
      const float s1 = softplusf_(o[3]) + 1e-3f;
      float x0, x1, y0, y1;
      if (MODE == MODE_FWD) {
        x0 = xs[c][2 * t];
        x1 = xs[c][2 * t + 1];
        y0 = (yp0 + o[0]) + s0 * x0;
        y1 = (yp1 + o[1]) + s1 * x1;
        if (q == 0) {
          ys[c][2 * t] = y0;
          ys[c][2 * t + 1] = y1;
        }
      } else {
        y0 = ys[c][2 * t];
        y1 = ys[c][2 * t + 1];
        x0 = (y0 - (yp0 + o[0])) * rcpf_(s0);
        x1 = (y1 - (yp1 + o[1])) * rcpf_(s1);
      }
      po.sq = fmaf(x0, x0, fmaf(x1, x1, po.sq));
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
    }
  }
  return po;
}

// what an adjoint step hands to the next-earlier one
struct AdjCarry {
  float dhz[8];      // dh'_{t+1} z_{t+1}, own units
  float whh[8];      // W_hh^T (own gate gradients of step t+1) for the own tiles, already scaled back
  float c0, c1;      // the coupling's part of the carry
  float du0, du1;    // W_ih^T (own gate gradients): the own partial of du, scaled back
};

// One step t of the adjoint, this wave's half (flow_split_dev.h:adj_step).  The W_hh^T and W_ih^T contractions are split
// over K: each wave contracts the gate gradients of its OWN units (its own power-of-two scale) against the rows of all
// four / the one output tile(s); the partial results for the peer's tiles cross through the exchange at the END of the
// step and are picked up at the top of the next one (FIRST = TS == 3 has nothing to pick up).
// Per wave: 2 + (2) fp32 and 6 + 9 + 36 f16 MFMAs.
template <int MODE, int TS, bool FROM_REGS>
__device__ __forceinline__ void adj_step_pair(const uint4* tw_in, const uint4* wtab_in, const uint4* wl_in, int hw,
                                              const float (*ys)[8], const float (*gin)[8], const float (*st)[6][CB],
                                              const float4* __restrict__ tp, const HalfTape* tr, const float* hp1, int c, int q,
                                              float w0, AdjCarry& cy, float (&res)[8], PairXchg& x) {
  constexpr bool FIRST = TS == T - 1;
  constexpr bool LASTSTEP = TS == 1;
  RIP_MARK("adj_begin");
  int zero = 0;
  asm volatile("" : "+v"(zero));
  const uint4* tw = tw_in + zero;
  const uint4* wtab = wtab_in + zero + hw * 16;
  // ---- the step's tape (own units) ----
  HalfTape tl;
  const HalfTape* tv = tr;
  if (!FROM_REGS) {
    const unsigned lane = (unsigned)(q * 16 + c);
    unsigned tape_loff = lane * 16u;
    asm volatile("" : "+v"(tape_loff));
    tl.mask = RIP_ABL == 1 ? 0x5au
                           : *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(tp + TAPE_ROWS * 64) + (tape_loff >> 2));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float4* tpu = tp + (8 * hw + 4 * u) * 64;
      const float4 rr = tape_ld(trow(tpu, 0, tape_loff)), zz = tape_ld(trow(tpu, 1, tape_loff));
      const float4 gh = tape_ld(trow(tpu, 3, tape_loff));
      float4 hp;
      if (TS == 1)
        hp = *reinterpret_cast<const float4*>(hp1 + 16 * (2 * hw + u) + 4 * q);  // prefix H1 (global, L2)
      else
        hp = tape_ld(trow(tp + (16 + 2 * hw + u) * 64, 0, tape_loff));
      tl.r[u * 4 + 0] = rr.x, tl.r[u * 4 + 1] = rr.y, tl.r[u * 4 + 2] = rr.z, tl.r[u * 4 + 3] = rr.w;
      tl.z[u * 4 + 0] = zz.x, tl.z[u * 4 + 1] = zz.y, tl.z[u * 4 + 2] = zz.z, tl.z[u * 4 + 3] = zz.w;
      tl.gh[u * 4 + 0] = gh.x, tl.gh[u * 4 + 1] = gh.y, tl.gh[u * 4 + 2] = gh.z, tl.gh[u * 4 + 3] = gh.w;
      tl.hp[u * 4 + 0] = hp.x, tl.hp[u * 4 + 1] = hp.y, tl.hp[u * 4 + 2] = hp.z, tl.hp[u * 4 + 3] = hp.w;
    }
    tv = &tl;
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- what the peer contributed at the end of step t + 1: its part of dh (my tiles) and of du ----
  float peer_dh[8];
  float carry0 = 0.f, carry1 = 0.f;
  if (!FIRST) {
    xch_wait(x);
    const u32x4 p0 = x.peer_rows[0], p1 = x.peer_rows[64];
    const f32x2 pe = *x.peer_extra;
    xch_done(x);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      peer_dh[i] = __uint_as_float(p0[i]);
      peer_dh[4 + i] = __uint_as_float(p1[i]);
    }
    carry0 = cy.c0 + (cy.du0 + pe.x);  // (own + peer: the same float on both waves)
    carry1 = cy.c1 + (cy.du1 + pe.y);
  }
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
  cy.c0 = c0;
  cy.c1 = c1;
  // ---- head adjoint: da1 = relu'(a1) * W2^T do, BOTH a1 tiles on both waves (W1^T da1 contracts over all 32) ----
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
  // ---- dh_t (own tiles), part 1: W1^T da1; rows 1 + ut * 2 + term, ut = 2 hw + u ----
  float dh[8];
  {
    float sa, ia;
    pow2_scale(qmax(amax8(da1r, 0.f)), sa, ia);
    h16x8 ah, al;
    split8<true>(da1r, sa, ah, al);
    const uint4* t1 = tw + (1 + 4 * hw) * 64;
    const uint4 r0h = t1[0], r0l = t1[64], r1h = t1[128], r1l = t1[192];
    PAIR_PRIO_BURST();
    f32x4 a[2];
    a[0] = mfmah(as_h8(r0h), ah, zero4());
    a[1] = mfmah(as_h8(r1h), ah, zero4());
    a[0] = mfmah(as_h8(r0h), al, a[0]);
    a[1] = mfmah(as_h8(r1h), al, a[1]);
    a[0] = mfmah(as_h8(r0l), ah, a[0]);
    a[1] = mfmah(as_h8(r1l), ah, a[1]);
    PAIR_PRIO_VALU();
#pragma unroll
    for (int i = 0; i < 8; ++i) dh[i] = a[i >> 2][i & 3] * ia;
  }
  // ---- part 2: dh'_{t+1} z_{t+1} + W_hh^T (gate gradients of step t + 1): own K blocks (kept) + peer K blocks (received) ----
  if (!FIRST) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dh[i] += cy.dhz[i] + (cy.whh[i] + peer_dh[i]);
  }
  // ---- n of this step (own tiles): tanh(gi_n + r gh_n) ----
  float nrec[8];
  if (!FROM_REGS) {
    const float4 wxg = as_f4((wl_in + zero)[50 * 64]);
    const float yp0 = ys[c][2 * (TS - 1)], yp1 = ys[c][2 * (TS - 1) + 1];
    const float bin = q == 0 ? yp0 : (q == 1 ? yp1 : (q == 2 ? 1.f : 0.f));
    f32x4 agn_t[2];
    agn_t[0] = mfma4(pick4(wxg, 2 * hw), bin, zero4());
    agn_t[1] = mfma4(pick4(wxg, 2 * hw + 1), bin, zero4());
    constexpr float L2E = 1.4426950408889634f;
    const f32x2 one = {1.0f, 1.0f}, two = {2.0f, 2.0f};
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
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
  // ---- GRUCell adjoint of the own units ----
  float dpn[8], dgr[8], dgz[8], dgn[8];
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const f32x2 hp2 = {tv->hp[i], tv->hp[i + 1]}, rr2 = {tv->r[i], tv->r[i + 1]}, zz2 = {tv->z[i], tv->z[i + 1]};
    const f32x2 nn2 = {FROM_REGS ? tv->n[i] : nrec[i], FROM_REGS ? tv->n[i + 1] : nrec[i + 1]}, gh2 = {tv->gh[i], tv->gh[i + 1]};
    const f32x2 one = {1.0f, 1.0f};
    const f32x2 d = {dh[i], dh[i + 1]};
    const f32x2 dn = d * (one - zz2);
    const f32x2 dzg = d * (hp2 - nn2);
    const f32x2 dhzn = d * zz2;
    const f32x2 dp = dn * (one - nn2 * nn2);
    const f32x2 dr = dp * gh2;
    const f32x2 dgn2 = dp * rr2;
    const f32x2 dpr = dr * rr2 * (one - rr2);
    const f32x2 dpz = dzg * zz2 * (one - zz2);
    cy.dhz[i] = dhzn.x, cy.dhz[i + 1] = dhzn.y;
    dpn[i] = dp.x, dpn[i + 1] = dp.y;
    dgn[i] = dgn2.x, dgn[i + 1] = dgn2.y;
    dgr[i] = dpr.x, dgr[i + 1] = dpr.y;
    dgz[i] = dpz.x, dgz[i + 1] = dpz.y;
  }
  // ---- the own gate gradients as B operands at the wave's own per-candidate scale ----
  float m = amax8(dgr, 0.f);
  m = amax8(dgz, m);
  if (!LASTSTEP) m = amax8(dgn, m);
  m = amax8(dpn, m);
  float sg, ig;
  pow2_scale(qmax(m), sg, ig);
  h16x8 r_hi, r_lo, z_hi, z_lo, n_hi, n_lo, p_hi, p_lo;
  split8<true>(dgr, sg, r_hi, r_lo);
  split8<true>(dgz, sg, z_hi, z_lo);
  split8<true>(dpn, sg, p_hi, p_lo);
  // ---- du (own partial) = W_ih^T over the own K blocks hw, 2 + hw, 4 + hw: table entry ((kb * 2 + term) * 8) ----
  {
    f32x4 ua = zero4(), ul = zero4(), ub = zero4();
    PAIR_PRIO_BURST();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const h16x8 wh = as_h8(wtab[(j * 4 + 0) * 8]), wo = as_h8(wtab[(j * 4 + 1) * 8]);
      const h16x8 bh = j == 0 ? r_hi : (j == 1 ? z_hi : p_hi);
      const h16x8 bl = j == 0 ? r_lo : (j == 1 ? z_lo : p_lo);
      ua = mfmah(wh, bh, ua);
      ul = mfmah(wh, bl, ul);
      ub = mfmah(wo, bh, ub);
    }
    PAIR_PRIO_VALU();
    cy.du0 = (ua[0] + (ul[0] + ub[0])) * ig;
    cy.du1 = (ua[1] + (ul[1] + ub[1])) * ig;
  }
  // ---- W_hh^T (own gate gradients): rows 9 + (kb * 4 + ut) * 2 + term, kb = hw + 2 j; the PEER's tiles first (they cross) ----
  if (!LASTSTEP) {
    split8<true>(dgn, sg, n_hi, n_lo);
    auto contract = [&](const uint4* base, float (&out)[8]) __attribute__((always_inline)) {
      f32x4 a[2] = {zero4(), zero4()};
      uint4 RH[2], RL[2];
      PAIR_PRIO_BURST();
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const h16x8 bh = j == 0 ? r_hi : (j == 1 ? z_hi : n_hi);
        const h16x8 bl = j == 0 ? r_lo : (j == 1 ? z_lo : n_lo);
        RH[0] = base[(16 * j + 0) * 64];
        RL[0] = base[(16 * j + 1) * 64];
        RH[1] = base[(16 * j + 2) * 64];
        RL[1] = base[(16 * j + 3) * 64];
        a[0] = mfmah(as_h8(RH[0]), bh, a[0]);
        a[1] = mfmah(as_h8(RH[1]), bh, a[1]);
        a[0] = mfmah(as_h8(RH[0]), bl, a[0]);
        a[1] = mfmah(as_h8(RH[1]), bl, a[1]);
        a[0] = mfmah(as_h8(RL[0]), bh, a[0]);
        a[1] = mfmah(as_h8(RL[1]), bh, a[1]);
      }
      PAIR_PRIO_VALU();
#pragma unroll
      for (int i = 0; i < 8; ++i) out[i] = a[i >> 2][i & 3] * ig;
    };
    float part[8];
    contract(tw + (13 + 4 * hw) * 64, part);  // peer tiles ut = 2 (1 - hw) + u
    xch_begin(x);
    x.my_rows[0] = f4_bits(part[0], part[1], part[2], part[3]);
    x.my_rows[64] = f4_bits(part[4], part[5], part[6], part[7]);
    *x.my_extra = f32x2{cy.du0, cy.du1};
    xch_publish(x);
    contract(tw + (9 + 12 * hw) * 64, cy.whh);  // own tiles ut = 2 hw + u
  } else {
    xch_begin(x);
    *x.my_extra = f32x2{cy.du0, cy.du1};
    xch_publish(x);
  }
  RIP_MARK("adj_end");
}

// adjoint pass of the current model, this wave's half (flow_split_dev.h:pass_backward)
template <int MODE>
__device__ __forceinline__ void pass_backward_pair(const uint4* tw, const uint4* wtab, const uint4* wl, int hw,
                                                   const float (*ys)[8], const float (*gin)[8], const float (*st)[6][CB],
                                                   const float4* __restrict__ tape, const HalfTape* last, const float* hp1, int c,
                                                   int q, float (&res)[8], float w0, PairXchg& x) {
  AdjCarry cy;
  cy.c0 = cy.c1 = cy.du0 = cy.du1 = 0.f;
  if (MODE == MODE_INV) {
    adj_step_pair<MODE, 3, true>(tw, wtab, wl, hw, ys, gin, st, nullptr, &last[2], hp1, c, q, w0, cy, res, x);
    adj_step_pair<MODE, 2, true>(tw, wtab, wl, hw, ys, gin, st, nullptr, &last[1], hp1, c, q, w0, cy, res, x);
    adj_step_pair<MODE, 1, true>(tw, wtab, wl, hw, ys, gin, st, nullptr, &last[0], hp1, c, q, w0, cy, res, x);
  } else {
    adj_step_pair<MODE, 3, false>(tw, wtab, wl, hw, ys, gin, st, tape + 2 * TAPE_STEP_F4, nullptr, hp1, c, q, w0, cy, res, x);
    adj_step_pair<MODE, 2, false>(tw, wtab, wl, hw, ys, gin, st, tape + TAPE_STEP_F4, nullptr, hp1, c, q, w0, cy, res, x);
    adj_step_pair<MODE, 1, false>(tw, wtab, wl, hw, ys, gin, st, tape, nullptr, hp1, c, q, w0, cy, res, x);
  }
  xch_wait(x);
  const f32x2 pe = *x.peer_extra;
  xch_done(x);
  const float carry0 = cy.c0 + (cy.du0 + pe.x), carry1 = cy.c1 + (cy.du1 + pe.y);
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
