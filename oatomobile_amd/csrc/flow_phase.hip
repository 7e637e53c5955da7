// Phase-sequential MFMA plan search for gfx950: the throughput kernel of RIPAgent.__call__ (rip/agent.py:78-137).
//
// Same arithmetic as round 1's wave-per-model kernel (removed in round 5; DESIGN_HISTORY §4.1b: 16 candidates per wave on v_mfma_f32_16x16x4_f32, hidden
// state in the "H layout", transposed products, the same operand blobs), another decomposition:
//   * ONE WAVE owns a block of 16 candidates for the whole search and runs ALL K models on it, one after the other:
//       F_0 -> for k = 1..K-1: inverse_k, adjoint_k -> ensemble aggregation -> adjoint-F_0 + Adam.
//     Posteriors, dq_k/dy and the Adam state never leave the wave (no cross-wave counters, no K <= 4 limit).
//   * a workgroup is 8 such waves (two per SIMD) that advance through the model phases together, so the operands of the
//     CURRENT model are shared: they live in LDS — a 63 KB buffer with the forward operands (F-buf) and a 58.5 KB buffer
//     with the transposed ones (T-buf), filled by direct global->LDS DMA (global_load_lds_dwordx4, one 1 KB lane-major
//     row per wave instruction) — and every MFMA A operand is a ds_read_b128 away.  The L2 operand stream of the
//     wave-per-model kernel (235 KB per block, model and Adam step; the reason a second wave per SIMD made that kernel
//     slower) becomes 121.5 KB per WORKGROUP, model and step: 15x less.
//   * nothing is register resident across phases, so two waves share each SIMD (256 registers each: the 8-wave build
//     fills them and spills 84 VGPRs to scratch, the 4- / 2-wave builds have the whole file and do not) and one wave's
//     gate math, tape traffic and LDS waits run under the other wave's MFMAs.
//   * per-candidate exchange buffers shrink to 1 KB per wave (x -> y in place, dLoss/dy), the gate gradients of the
//     adjoint are registers (the contraction is fully unrolled: its A operands have static LDS offsets).
//   * the adjoint tape (what a step's adjoint needs from its forward: r, z, gh_n, hprev + the ReLU mask; n is
//     recomputed from r, gh_n and a 4-MFMA k-step on the step's input) goes to global memory, 16 KB per wave and taped
//     step; the last inverse step of a pass is handed to its adjoint in registers.
//   * launches with fewer than 8 blocks per CU use 4- or 2-wave workgroups (template WPB) so that every CU has one.
// LDS: 121.5 KB operands + 8 x 4 KB = 153.5 KB per workgroup, one workgroup per CU.
#include "flow.h"
#include "flow_math.h"

namespace rip {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int T = 4;
constexpr int CB = 16;                        // candidates per wave
constexpr int WPB_MAX = 8;                    // waves (= candidate blocks) per workgroup: 8 (two per SIMD), 4 or 2
// Adjoint tape of one heavy step, per lane: rows 0..15 = (r, z, n, gh_n) of the 4 unit tiles, rows 16..19 = hprev (not
// written at t = 1: that is the prefix H1), then ONE dword with the ReLU mask of a1 (8 bits).  A pass has 3 heavy steps.
constexpr int TAPE_ROWS = 20;
constexpr int TAPE_STEP_F4 = TAPE_ROWS * 64 + 16;  // + 64 mask dwords
constexpr int TAPE_SLOT_F4 = 3 * TAPE_STEP_F4;
constexpr int F_ROWS = 63;                    // forward operand rows (lane-major float4, MWF_* in flow.h)
constexpr int T_ROWS = 57;                    // transposed rows: 0 = W2^T, 1..8 = W1^T, 9..56 = W_hh^T
constexpr int PRE_FLOATS = 72;                // per (model, observation): H1[64], dloc0, dloc1, s0, s1, lad, pad

enum { MODE_FWD = 0, MODE_INV = 1 };

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 zero4() {
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return z;
}
// Two waves share a SIMD (waves w and w + 4 of the workgroup).  Arbitration is by age: the older wave runs at its solo
// speed, the younger one fills its gaps and finishes alone (tools/search_ticks.py: F_0 pass 16 k vs 21 k cycles per step).
// A wave raises its priority for its MFMA bursts and drops it for the VALU sections between them (gate math, tape
// traffic), so that the partner's burst wins the issue port while this wave does VALU work: 4.74 -> 4.64 ms.  Measured
// and rejected: balancing the two waves' progress tile by tile through LDS counters (both then reach the barrier
// together, but evenly interleaved bursts cost more than the hand-over: 5.62 ms).
#ifndef RIP_PRIO
#define RIP_PRIO 1
#endif
#define PRIO_BURST() do { if (RIP_PRIO) __builtin_amdgcn_s_setprio(1); } while (0)
#define PRIO_VALU() do { if (RIP_PRIO) __builtin_amdgcn_s_setprio(0); } while (0)

// the adjoint recomputes n = tanh(gi_n + r * gh_n) from the taped r and gh_n (one 4-MFMA k-step for gi_n and 16 tanh per
// step) instead of reading it back: 16 instead of 20 rows per taped step (RIP_TAPE_N = 1 restores the taped n)
#ifndef RIP_TAPE_N
#define RIP_TAPE_N 0
#endif
#ifndef RIP_REGTAPE
#define RIP_REGTAPE 1  // 4- / 2-wave workgroups keep the inverse passes' whole tape in registers
#endif
#ifndef RIP_ADJ_PF
#define RIP_ADJ_PF 3  // adjoint contraction: operand rows in flight
#endif
#ifndef RIP_PREFETCH
#define RIP_PREFETCH 1  // operand rows requested one MFMA group ahead (development switch)
#endif
#ifndef RIP_ABL
#define RIP_ABL 0  // development only: 1 = no tape loads, 2 = no operand reloads, 3 = no tape stores (wrong results)
#endif
__device__ __forceinline__ void tape_st(float4* p, float a, float b, float c, float d) {
  if (RIP_ABL != 3) *p = make_float4(a, b, c, d);
}
__device__ __forceinline__ float4 tape_ld(const float4* p) {
  if (RIP_ABL == 1) return make_float4(0.3f, 0.4f, 0.5f, 0.6f);
  return *p;
}

template <int WPB>
struct PShared {
  float4 fbuf[F_ROWS * 64];      // forward operands of the current model
  float4 tbuf[T_ROWS * 64];      // transposed operands of the current model
  float4 wihc[12 * 8];           // its W_ih^T operands: one float4 per (tile g, q, input dim)
  float io[WPB][CB][8];          // per wave: x in, y out (in place)
  float gy[WPB][CB][8];          // per wave: dLoss/dy handed to the F_0 adjoint
  float stape[WPB][2][T][6][CB]; // per wave: per-candidate scalars of the F_0 pass [0] and of the current inverse [1]
};

// gate math of one unit tile on unit pairs (same formulas as sigmoidf_ / tanhf_ in flow_math.h; v_pk_* where possible)
__device__ __forceinline__ void gru_gates(const f32x4& ar, const f32x4& az, const f32x4& agn, const f32x4& ahn,
                                          const float* Hold, float* Hn, float (&rr)[4], float (&zz)[4], float (&nn)[4]) {
  using f2 = __attribute__((ext_vector_type(2))) float;
  const f2 one = {1.0f, 1.0f}, two = {2.0f, 2.0f};
  constexpr float L2E = 1.4426950408889634f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f2 pr = f2{ar[2 * h], ar[2 * h + 1]} * f2{-L2E, -L2E};
    const f2 pz = f2{az[2 * h], az[2 * h + 1]} * f2{-L2E, -L2E};
    const f2 er = {__builtin_amdgcn_exp2f(pr.x), __builtin_amdgcn_exp2f(pr.y)};
    const f2 ez = {__builtin_amdgcn_exp2f(pz.x), __builtin_amdgcn_exp2f(pz.y)};
    const f2 dr = er + one, dz = ez + one;
    const f2 r2 = {rcpf_(dr.x), rcpf_(dr.y)};
    const f2 z2 = {rcpf_(dz.x), rcpf_(dz.y)};
    const f2 pre = __builtin_elementwise_fma(r2, f2{ahn[2 * h], ahn[2 * h + 1]}, f2{agn[2 * h], agn[2 * h + 1]});
    const f2 pn = pre * f2{2.0f * L2E, 2.0f * L2E};
    const f2 en = {__builtin_amdgcn_exp2f(pn.x), __builtin_amdgcn_exp2f(pn.y)};
    const f2 dn = en + one;
    const f2 in2 = {rcpf_(dn.x), rcpf_(dn.y)};
    const f2 n2 = one - two * in2;
    const f2 hold = {Hold[2 * h], Hold[2 * h + 1]};
    const f2 hn = __builtin_elementwise_fma(z2, hold - n2, n2);  // (1-z)*n + z*h
    rr[2 * h] = r2.x;
    rr[2 * h + 1] = r2.y;
    zz[2 * h] = z2.x;
    zz[2 * h + 1] = z2.y;
    nn[2 * h] = n2.x;
    nn[2 * h + 1] = n2.y;
    Hn[2 * h] = hn.x;
    Hn[2 * h + 1] = hn.y;
  }
}

// row `r` of a lane-major tape whose base is wave uniform: scalar base + immediate + ONE zero-extended 32-bit per-lane
// byte offset (`loff` = 16 * lane), so no 64-bit address pair per row is ever materialised in vector registers
__device__ __forceinline__ float4* trow(float4* base, int r, unsigned loff) {
  return reinterpret_cast<float4*>(reinterpret_cast<char*>(base + r * 64) + loff);
}
__device__ __forceinline__ const float4* trow(const float4* base, int r, unsigned loff) {
  return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base + r * 64) + loff);
}

// what the adjoint needs from one heavy step, per lane (the "tape" of that step)
struct StepTape {
  float hp[16], r[16], z[16], n[16], gh[16];
  unsigned mask;  // bit j: a1[j] > 0
};

enum { SAVE_NONE = 0, SAVE_TAPE = 1, SAVE_TAPE_NOHP = 2, SAVE_REGS = 3 };

// One GRU + head step for 16 candidates with the 251 A operands read, tile by tile, from `wl` = this lane's column of
// the 63 lane-major float4 rows of the F-buf (value i of the MW order, fold_and_pack, sits in row i / 4, component
// i % 4).  Same MFMA order per accumulator as fwd_step there: bitwise the same results.
// SAVE_TAPE: the step's tape goes to `tape` (global); SAVE_TAPE_NOHP: without hprev (step 1: hprev is the prefix);
// SAVE_REGS: it stays in `tr` (the last inverse step is consumed by its adjoint right away).
// `tape` is the WAVE-UNIFORM base of the step's tape (row r of lane l at tape[r * 64 + l]): scalar base + one per-lane
// offset register + immediates, instead of one 64-bit address pair per row.
template <int SAVE>
__device__ __forceinline__ void fwd_step_lds(const float4* wl, float (&H)[16], float yp0, float yp1, int q, unsigned lane,
                                             float4* __restrict__ tape, StepTape* tr, float (&o)[4]) {
  const float bin = q == 0 ? yp0 : (q == 1 ? yp1 : (q == 2 ? 1.f : 0.f));
  float Hn[16];
  // the per-lane tape offset is made opaque HERE: otherwise base + offset is hoisted out of the search loops as one
  // 64-bit vector address per group of rows (60 registers, spilled) instead of staying scalar base + this register
  unsigned loff = lane * 16u;
  asm volatile("" : "+v"(loff));
  // rows 48..51: (W_ih[.][0], W_ih[.][1], bias, 0) k-step of gate a in {r, z, gi_n, gh_n}; component = unit tile
  const float4 wxr = wl[48 * 64], wxz = wl[49 * 64], wxg = wl[50 * 64], wxh = wl[51 * 64];
  const float wxra[4] = {wxr.x, wxr.y, wxr.z, wxr.w}, wxza[4] = {wxz.x, wxz.y, wxz.z, wxz.w};
  const float wxga[4] = {wxg.x, wxg.y, wxg.z, wxg.w}, wxha[4] = {wxh.x, wxh.y, wxh.z, wxh.w};
  // operand rows one group (12 MFMAs = 384 cycles) ahead of their use: the LDS latency of a group's three reads is
  // covered by the previous group's MFMAs instead of by its last two (what the scheduler does on its own)
  float4 cr = wl[((0 * 4 + 0) * 4 + 0) * 64], cz = wl[((1 * 4 + 0) * 4 + 0) * 64], ch = wl[((2 * 4 + 0) * 4 + 0) * 64];
#pragma unroll
  for (int up = 0; up < 4; ++up) {
    f32x4 ar = zero4(), az = zero4(), agn = zero4(), ahn = zero4();
    PRIO_BURST();
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // 4 k-steps per operand row
      const float4 wr = cr, wz = cz, wh = ch;
      if (RIP_PREFETCH) {
        const int nj = (j + 1) & 3, nup = up + (j == 3 ? 1 : 0);
        if (nup < 4) {
          cr = wl[((0 * 4 + nup) * 4 + nj) * 64];
          cz = wl[((1 * 4 + nup) * 4 + nj) * 64];
          ch = wl[((2 * 4 + nup) * 4 + nj) * 64];
        } else {  // the head's first rows
          cr = wl[52 * 64];
          cz = wl[56 * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      ar = mfma(wr.x, H[4 * j + 0], ar);
      az = mfma(wz.x, H[4 * j + 0], az);
      ahn = mfma(wh.x, H[4 * j + 0], ahn);
      ar = mfma(wr.y, H[4 * j + 1], ar);
      az = mfma(wz.y, H[4 * j + 1], az);
      ahn = mfma(wh.y, H[4 * j + 1], ahn);
      ar = mfma(wr.z, H[4 * j + 2], ar);
      az = mfma(wz.z, H[4 * j + 2], az);
      ahn = mfma(wh.z, H[4 * j + 2], ahn);
      ar = mfma(wr.w, H[4 * j + 3], ar);
      az = mfma(wz.w, H[4 * j + 3], az);
      ahn = mfma(wh.w, H[4 * j + 3], ahn);
      if (!RIP_PREFETCH && !(up == 3 && j == 3)) {
        const int nj = (j + 1) & 3, nup = up + (j == 3 ? 1 : 0);
        cr = wl[((0 * 4 + nup) * 4 + nj) * 64];
        cz = wl[((1 * 4 + nup) * 4 + nj) * 64];
        ch = wl[((2 * 4 + nup) * 4 + nj) * 64];
      }
    }
    ar = mfma(wxra[up], bin, ar);
    az = mfma(wxza[up], bin, az);
    agn = mfma(wxga[up], bin, agn);
    ahn = mfma(wxha[up], bin, ahn);
    PRIO_VALU();
    float rr[4], zz[4], nn[4];
    gru_gates(ar, az, agn, ahn, &H[up * 4], &Hn[up * 4], rr, zz, nn);
    if (SAVE == SAVE_TAPE || SAVE == SAVE_TAPE_NOHP) {
      tape_st(trow(tape, up * 4 + 0, loff), rr[0], rr[1], rr[2], rr[3]);
      tape_st(trow(tape, up * 4 + 1, loff), zz[0], zz[1], zz[2], zz[3]);
      if (RIP_TAPE_N) tape_st(trow(tape, up * 4 + 2, loff), nn[0], nn[1], nn[2], nn[3]);
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
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) H[i] = Hn[i];
  // ---- head: rows 52..59 = W1 (two 16-row tiles x 16 k-steps), row 60 = (b1 tile 0, b1 tile 1, W2 k-steps 0, 1),
  // row 61 = W2 k-steps 2..5, row 62 = (W2 k-steps 6, 7, b2, -) ----
  const float bone = q == 2 ? 1.f : 0.f;
  f32x4 a0 = zero4(), a1 = zero4();
  PRIO_BURST();
  float4 ca = RIP_PREFETCH ? cr : wl[52 * 64], cb = RIP_PREFETCH ? cz : wl[56 * 64];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 wa = ca, wb = cb;
    if (j < 3) {
      ca = wl[(52 + j + 1) * 64];
      cb = wl[(56 + j + 1) * 64];
      if (RIP_PREFETCH) __builtin_amdgcn_sched_barrier(0);
    }
    a0 = mfma(wa.x, H[4 * j + 0], a0);
    a1 = mfma(wb.x, H[4 * j + 0], a1);
    a0 = mfma(wa.y, H[4 * j + 1], a0);
    a1 = mfma(wb.y, H[4 * j + 1], a1);
    a0 = mfma(wa.z, H[4 * j + 2], a0);
    a1 = mfma(wb.z, H[4 * j + 2], a1);
    a0 = mfma(wa.w, H[4 * j + 3], a0);
    a1 = mfma(wb.w, H[4 * j + 3], a1);
  }
  const float4 t60 = wl[60 * 64], t61 = wl[61 * 64], t62 = wl[62 * 64];
  a0 = mfma(t60.x, bone, a0);
  a1 = mfma(t60.y, bone, a1);
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
  f32x4 oa = zero4(), ob = zero4();
  oa = mfma(t60.z, fmaxf(a0[0], 0.f), oa);
  ob = mfma(t61.z, fmaxf(a1[0], 0.f), ob);
  oa = mfma(t60.w, fmaxf(a0[1], 0.f), oa);
  ob = mfma(t61.w, fmaxf(a1[1], 0.f), ob);
  oa = mfma(t61.x, fmaxf(a0[2], 0.f), oa);
  ob = mfma(t62.x, fmaxf(a1[2], 0.f), ob);
  oa = mfma(t61.y, fmaxf(a0[3], 0.f), oa);
  ob = mfma(t62.y, fmaxf(a1[3], 0.f), ob);
  oa = mfma(t62.z, bone, oa);
  PRIO_VALU();
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = oa[r] + ob[r];
}

struct Prefix16 {
  float H1[16];
  float dloc0, dloc1, s0, s1, lad;
};

__device__ __forceinline__ Prefix16 load_prefix(const float* __restrict__ p, int q) {
  Prefix16 pre;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float4 v = *reinterpret_cast<const float4*>(p + 16 * u + 4 * q);
    pre.H1[u * 4 + 0] = v.x;
    pre.H1[u * 4 + 1] = v.y;
    pre.H1[u * 4 + 2] = v.z;
    pre.H1[u * 4 + 3] = v.w;
  }
  pre.dloc0 = p[64];
  pre.dloc1 = p[65];
  pre.s0 = p[66];
  pre.s1 = p[67];
  pre.lad = p[68];
  return pre;
}

struct PassOut {
  float lad, sq;
};

// forward (x -> y, in place in `io`) or inverse (reads y from `io`) pass of the current model for this wave's 16
// candidates.  Steps 1..3 are "heavy"; step 0 is the candidate-independent prefix.  MODE_FWD tapes all three heavy steps
// (its adjoint runs K-1 model phases later); MODE_INV tapes steps 1, 2 and hands step 3 over in registers (`last`).
// REGTAPE (inverse passes of the 4- / 2-wave workgroups: one wave per SIMD owns the whole register file): ALL three
// steps are handed to the adjoint in registers (`last[0..2]`, spilled to AGPRs by the compiler), no tape traffic.
template <int MODE, bool REGTAPE = false>
__device__ __forceinline__ PassOut pass_forward(const float4* wl, const Prefix16& pre, float (*io)[8],
                                                float (*st)[6][CB], float4* __restrict__ tape, StepTape* last, int c,
                                                int q, unsigned lane) {
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
  // An opaque zero per step: the (loop-invariant) operand reads must not be merged across steps, which would make
  // all 251 operands register resident.
  {
    float o[4];
    int zero = 0;
    asm volatile("" : "+v"(zero));
    if (REGTAPE)
      fwd_step_lds<SAVE_REGS>(wl + zero, H, yp0, yp1, q, lane, nullptr, &last[0], o);
    else
      fwd_step_lds<SAVE_TAPE_NOHP>(wl + zero, H, yp0, yp1, q, lane, tape, nullptr, o);
    coupling(1, o);
  }
  {
    float o[4];
    int zero = 0;
    asm volatile("" : "+v"(zero));
    if (REGTAPE)
      fwd_step_lds<SAVE_REGS>(wl + zero, H, yp0, yp1, q, lane, nullptr, &last[1], o);
    else
      fwd_step_lds<SAVE_TAPE>(wl + zero, H, yp0, yp1, q, lane, tape + TAPE_STEP_F4, nullptr, o);
    coupling(2, o);
  }
  {
    float o[4];
    int zero = 0;
    asm volatile("" : "+v"(zero));
    if (MODE == MODE_FWD)
      fwd_step_lds<SAVE_TAPE>(wl + zero, H, yp0, yp1, q, lane, tape + 2 * TAPE_STEP_F4, nullptr, o);
    else
      fwd_step_lds<SAVE_REGS>(wl + zero, H, yp0, yp1, q, lane, nullptr, &last[2], o);
    coupling(3, o);
  }
  return po;
}

// One step t of the adjoint (see pass_backward).  FROM_REGS: the step's tape is `tr` (registers); else it is read from
// `tp` (global) — all rows requested BEFORE the contraction, so their latency hides under its 34 / 226 MFMAs.
// HP_PREFIX (t = 1): hprev is the prefix H1 (`hp1`).  FIRST (t = T-1): nothing flows in from a later step.
template <int MODE, int TS, bool FROM_REGS>
__device__ __forceinline__ void adj_step(const float4* tw_in, const float4* wq4_in, const float4* wl_in,
                                         const float (*io)[8], const float (*gin)[8],
                                         const float (*st)[6][CB], const float4* __restrict__ tp, const StepTape* tr,
                                         const float* hp1, int c, int q, float w0, float (&dhz)[16], float (&dgh)[48],
                                         float& carry0, float& carry1, float (&res)[8]) {
  constexpr bool FIRST = TS == T - 1;
  int zero = 0;
  asm volatile("" : "+v"(zero));  // keeps the operand reads of this step from being merged with another step's
  const float4* tw = tw_in + zero;
  const float4* wq4 = wq4_in + zero;
  // ---- the step's tape ----
  StepTape tl;
  const StepTape* tv = tr;
  if (!FROM_REGS) {
    const unsigned lane = (unsigned)(q * 16 + c);  // unsigned: scalar base + zero-extended 32-bit lane offset
    unsigned loff = lane * 16u;
    asm volatile("" : "+v"(loff));  // see fwd_step_lds
    tl.mask = RIP_ABL == 1 ? 0x5au
                           : *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(tp + TAPE_ROWS * 64) + (loff >> 2));
#pragma unroll
    for (int up = 0; up < 4; ++up) {
      const float4 rr = tape_ld(trow(tp, up * 4 + 0, loff)), zz = tape_ld(trow(tp, up * 4 + 1, loff));
      const float4 gh = tape_ld(trow(tp, up * 4 + 3, loff));
      float4 nn = make_float4(0.f, 0.f, 0.f, 0.f);
      if (RIP_TAPE_N) nn = tape_ld(trow(tp, up * 4 + 2, loff));
      float4 hp;
      if (TS == 1)
        hp = *reinterpret_cast<const float4*>(hp1 + 16 * up + 4 * q);  // prefix H1 (global, L2): units 16 up + 4 q + r
      else
        hp = tape_ld(trow(tp, 16 + up, loff));
      tl.r[up * 4 + 0] = rr.x, tl.r[up * 4 + 1] = rr.y, tl.r[up * 4 + 2] = rr.z, tl.r[up * 4 + 3] = rr.w;
      tl.z[up * 4 + 0] = zz.x, tl.z[up * 4 + 1] = zz.y, tl.z[up * 4 + 2] = zz.z, tl.z[up * 4 + 3] = zz.w;
      tl.n[up * 4 + 0] = nn.x, tl.n[up * 4 + 1] = nn.y, tl.n[up * 4 + 2] = nn.z, tl.n[up * 4 + 3] = nn.w;
      tl.gh[up * 4 + 0] = gh.x, tl.gh[up * 4 + 1] = gh.y, tl.gh[up * 4 + 2] = gh.z, tl.gh[up * 4 + 3] = gh.w;
      tl.hp[up * 4 + 0] = hp.x, tl.hp[up * 4 + 1] = hp.y, tl.hp[up * 4 + 2] = hp.z, tl.hp[up * 4 + 3] = hp.w;
    }
    tv = &tl;
    __builtin_amdgcn_sched_barrier(0);  // the requests stay ahead of the contraction
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
    // w0 != 0: this candidate's loss also holds -w0 * q_0 with q_0 = -0.5|x|^2 - logabsdet_F(x) evaluated on the
    // forward pass itself (inverse_0(F_0(x)) == x): d/dx_t = w0 x_t, d/ds_t = w0 / s_t.
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
  // ---- head adjoint: da1 = relu'(a1) * W2^T do ----
  const float4 w2t = tw[0];
  const float bdo = q == 0 ? dd0 : (q == 1 ? dd1 : (q == 2 ? dos0 : dos1));
  const f32x4 da0 = mfma(w2t.x, bdo, zero4());
  const f32x4 da1 = mfma(w2t.y, bdo, zero4());
  const unsigned mask = tv->mask;
  float da1r[8];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    da1r[r] = (mask >> r) & 1u ? da0[r] : 0.f;
    da1r[4 + r] = (mask >> (4 + r)) & 1u ? da1[r] : 0.f;
  }
  // ---- dh_t = W1^T da1_t + W_hh^T dgh_{t+1} (+ dh'_{t+1} z_{t+1} below) ----
  f32x4 acc0 = zero4(), acc1 = zero4(), acc2 = zero4(), acc3 = zero4();
  PRIO_BURST();
  {
    // rows 1..8 (W1^T, B = da1) then 9..56 (W_hh^T, B = dgh), each feeding 4 MFMAs; requested PF rows (PF * 128
    // cycles of MFMAs) ahead of their use
    constexpr int NROW = FIRST ? 8 : 56;
    constexpr int PF = RIP_ADJ_PF;
    float4 ring[PF];
#pragma unroll
    for (int e = 0; e < PF; ++e) ring[e] = tw[(1 + e) * 64];
#pragma unroll
    for (int e = 0; e < NROW; ++e) {
      const float4 w = ring[e % PF];
      if (RIP_PREFETCH) {
        if (e + PF < NROW) ring[e % PF] = tw[(1 + e + PF) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
      const float bop = e < 8 ? da1r[e < 8 ? e : 0] : dgh[e < 8 ? 0 : e - 8];
      acc0 = mfma(w.x, bop, acc0);
      acc1 = mfma(w.y, bop, acc1);
      acc2 = mfma(w.z, bop, acc2);
      acc3 = mfma(w.w, bop, acc3);
      if (!RIP_PREFETCH && e + PF < NROW) ring[e % PF] = tw[(1 + e + PF) * 64];
    }
  }
  PRIO_VALU();
  // ---- GRUCell adjoint, lane-local in the H layout (unit pairs: v_pk_mul_f32 / v_pk_fma_f32) ----
  using f2 = __attribute__((ext_vector_type(2))) float;
  // gi_n of this step = the (W_in[.][0], W_in[.][1], b_in, 0) k-step on the step's input y_{t-1} (still in `io`),
  // issued after the contraction (16 accumulator registers held across it were spilled); then the same expressions as
  // gru_gates
  f32x4 agn_t[4];
  if (!FROM_REGS && !RIP_TAPE_N) {
    const float4 wxg = (wl_in + zero)[50 * 64];
    const float yp0 = io[c][2 * (TS - 1)], yp1 = io[c][2 * (TS - 1) + 1];
    const float bin = q == 0 ? yp0 : (q == 1 ? yp1 : (q == 2 ? 1.f : 0.f));
    agn_t[0] = mfma(wxg.x, bin, zero4());
    agn_t[1] = mfma(wxg.y, bin, zero4());
    agn_t[2] = mfma(wxg.z, bin, zero4());
    agn_t[3] = mfma(wxg.w, bin, zero4());
  }
  float nrec[16];
  if (!FROM_REGS && !RIP_TAPE_N) {
    constexpr float L2E = 1.4426950408889634f;
    const f2 one = {1.0f, 1.0f}, two = {2.0f, 2.0f};
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const f2 r2 = {tl.r[i], tl.r[i + 1]}, ghn = {tl.gh[i], tl.gh[i + 1]};
      const f2 gin2 = {agn_t[i >> 2][i & 3], agn_t[i >> 2][(i & 3) + 1]};
      const f2 pre = __builtin_elementwise_fma(r2, ghn, gin2);
      const f2 pn = pre * f2{2.0f * L2E, 2.0f * L2E};
      const f2 en = {__builtin_amdgcn_exp2f(pn.x), __builtin_amdgcn_exp2f(pn.y)};
      const f2 dn = en + one;
      const f2 in2 = {rcpf_(dn.x), rcpf_(dn.y)};
      const f2 n2 = one - two * in2;
      nrec[i] = n2.x;
      nrec[i + 1] = n2.y;
    }
  }
  const f32x4 accs[4] = {acc0, acc1, acc2, acc3};
  float dpn[16];
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    const f2 hp2 = {tv->hp[i], tv->hp[i + 1]}, rr2 = {tv->r[i], tv->r[i + 1]}, zz2 = {tv->z[i], tv->z[i + 1]};
    const bool n_taped = FROM_REGS || RIP_TAPE_N;
    const f2 nn2 = {n_taped ? tv->n[i] : nrec[i], n_taped ? tv->n[i + 1] : nrec[i + 1]}, gh2 = {tv->gh[i], tv->gh[i + 1]};
    const f2 one = {1.0f, 1.0f};
    f2 dh = {accs[i >> 2][i & 3], accs[i >> 2][(i & 3) + 1]};
    if (!FIRST) dh = dh + f2{dhz[i], dhz[i + 1]};
    const f2 dn = dh * (one - zz2);
    const f2 dzg = dh * (hp2 - nn2);
    const f2 dhzn = dh * zz2;
    const f2 dp = dn * (one - nn2 * nn2);
    const f2 dr = dp * gh2;
    const f2 dgn = dp * rr2;
    const f2 dpr = dr * rr2 * (one - rr2);
    const f2 dpz = dzg * zz2 * (one - zz2);
    dhz[i] = dhzn.x;
    dhz[i + 1] = dhzn.y;
    dpn[i] = dp.x;        // d pre_n
    dpn[i + 1] = dp.y;
    dgh[32 + i] = dgn.x;  // d gh_n
    dgh[33 + i] = dgn.y;
    dgh[i] = dpr.x;       // d pre_r
    dgh[1 + i] = dpr.y;
    dgh[16 + i] = dpz.x;  // d pre_z
    dgh[17 + i] = dpz.y;
  }
  // ---- du = W_ih^T (dpr, dpz, dpn): rows m <-> input dim m & 1 ----
  f32x4 dua = zero4(), dub = zero4();
  PRIO_BURST();
#pragma unroll
  for (int g = 0; g < 12; ++g) {
    const float4 wq = wq4[g * 8];
    const float b0 = g < 8 ? dgh[4 * g + 0] : dpn[4 * (g - 8) + 0];
    const float b1 = g < 8 ? dgh[4 * g + 1] : dpn[4 * (g - 8) + 1];
    const float b2 = g < 8 ? dgh[4 * g + 2] : dpn[4 * (g - 8) + 2];
    const float b3 = g < 8 ? dgh[4 * g + 3] : dpn[4 * (g - 8) + 3];
    dua = mfma(wq.x, b0, dua);
    dub = mfma(wq.y, b1, dub);
    dua = mfma(wq.z, b2, dua);
    dub = mfma(wq.w, b3, dub);
  }
  PRIO_VALU();
  carry0 = c0 + (dua[0] + dub[0]);
  carry1 = c1 + (dua[1] + dub[1]);
}

// adjoint pass of the current model.  MODE_INV: writes dq/dy (q = -0.5|x|^2 - logabsdet) to res[8], step 3's tape is
// `last` (registers); MODE_FWD: takes dL/dy from gin[c][*] and writes dL/dx to res[8], all steps from `tape`.
// tw: this lane's column of the T-buf rows; wq4: this lane's entry of the W_ih^T table; hp1: the prefix record of
// (model, observation) in global memory (its H1 is step 1's hprev).
// All gate gradients are registers: the 8 + 48 entries of the dh contraction are unrolled, their A operands are
// ds_read_b128 at static offsets.
template <int MODE, bool REGTAPE = false>
__device__ __forceinline__ void pass_backward(const float4* tw, const float4* wq4, const float4* wl, const float (*io)[8],
                                              const float (*gin)[8],
                                              const float (*st)[6][CB], const float4* __restrict__ tape,
                                              const StepTape* last, const float* hp1, int c, int q, float (&res)[8],
                                              float w0) {
  float dhz[16];  // dh'_{t+1} z_{t+1}, carried to the next (earlier) step
  float dgh[48];  // d pre_r (0-15), d pre_z (16-31), d gh_n (32-47) of step t+1: B operands of the W_hh^T contraction
  float carry0 = 0.f, carry1 = 0.f;
  if (MODE == MODE_INV)
    adj_step<MODE, 3, true>(tw, wq4, wl, io, gin, st, nullptr, &last[2], hp1, c, q, w0, dhz, dgh, carry0, carry1, res);
  else
    adj_step<MODE, 3, false>(tw, wq4, wl, io, gin, st, tape + 2 * TAPE_STEP_F4, nullptr, hp1, c, q, w0, dhz, dgh, carry0, carry1, res);
  if (REGTAPE) {
    adj_step<MODE, 2, true>(tw, wq4, wl, io, gin, st, nullptr, &last[1], hp1, c, q, w0, dhz, dgh, carry0, carry1, res);
    adj_step<MODE, 1, true>(tw, wq4, wl, io, gin, st, nullptr, &last[0], hp1, c, q, w0, dhz, dgh, carry0, carry1, res);
  } else {
    adj_step<MODE, 2, false>(tw, wq4, wl, io, gin, st, tape + TAPE_STEP_F4, nullptr, hp1, c, q, w0, dhz, dgh, carry0, carry1, res);
    adj_step<MODE, 1, false>(tw, wq4, wl, io, gin, st, tape, nullptr, hp1, c, q, w0, dhz, dgh, carry0, carry1, res);
  }
  // ---- t = 0: coupling only ----
  const float x0 = st[0][0][c], x1 = st[0][1][c], s0 = st[0][2][c], s1 = st[0][3][c];
  if (MODE == MODE_INV) {
    res[0] = carry0 - x0 * rcpf_(s0);
    res[1] = carry1 - x1 * rcpf_(s1);
  } else {
    res[0] = fmaf(gin[c][0] + carry0, s0, w0 * x0);
    res[1] = fmaf(gin[c][1] + carry1, s1, w0 * x1);
  }
}

// ---- operand staging: direct global -> LDS DMA, one 1 KB lane-major row per wave instruction ----
template <int WPB>
__device__ __forceinline__ void dma_rows(const float4* __restrict__ src, float4* dst, int rows, int wave, int lane) {
  for (int r = wave; r < rows; r += WPB)
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + r * 64 + lane), (lds_ptr_t)(dst + r * 64), 16, 0, 0);
}
template <int WPB>
__device__ __forceinline__ void load_fbuf(PShared<WPB>& sh, const float* __restrict__ mwk, int wave, int lane) {
  dma_rows<WPB>(reinterpret_cast<const float4*>(mwk), sh.fbuf, F_ROWS, wave, lane);
}
template <int WPB>
__device__ __forceinline__ void load_tbuf(PShared<WPB>& sh, const float* __restrict__ mwk, int wave, int lane, int tid) {
  static_assert(WPB * 64 >= 96, "the W_ih^T table is copied by 96 threads");
  const float4* src = reinterpret_cast<const float4*>(mwk + MWF_FLOATS);
  dma_rows<WPB>(src, sh.tbuf, T_ROWS, wave, lane);
  // W_ih^T table: entry (g, q, d) = the lane-major row 57 + g at lane 16 q + d (its 16 rows only differ in m & 1)
  if (tid < 96) sh.wihc[tid] = src[(T_ROWS + (tid >> 3)) * 64 + ((tid & 7) >> 1) * 16 + (tid & 1)];
}

// prefix of every (model, observation): step 0 from h_0 = z_k, y_0 = 0 is candidate independent.  One wave each, with
// the operands read straight from global memory (L2) — a 2 x 251-MFMA kernel in front of the search.
__global__ __launch_bounds__(64) void phase_prefix_kernel(SearchArgs a, const float* __restrict__ mw_all,
                                                          float* __restrict__ pre_out) {
  if (a.run_if_flag != nullptr && __builtin_nontemporal_load(a.run_if_flag) == 0u) return;  // fallback launch, not needed
  const int lane = threadIdx.x, q = lane >> 4;
  const int b = blockIdx.x, k = blockIdx.y;
  const float4* wl = reinterpret_cast<const float4*>(mw_all + (size_t)(a.k0 + k) * MW_SIZE) + lane;
  float H[16];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) H[u * 4 + r] = a.z[((size_t)k * a.B + b) * 64 + 16 * u + 4 * q + r];
  float o[4];
  fwd_step_lds<SAVE_NONE>(wl, H, 0.f, 0.f, q, (unsigned)lane, nullptr, nullptr, o);
  float* p = pre_out + ((size_t)k * a.B + b) * PRE_FLOATS;
  if ((lane & 15) == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      *reinterpret_cast<float4*>(p + 16 * u + 4 * q) = make_float4(H[u * 4], H[u * 4 + 1], H[u * 4 + 2], H[u * 4 + 3]);
  }
  if (lane == 0) {
    const float s0 = softplusf_(o[2]) + 1e-3f, s1 = softplusf_(o[3]) + 1e-3f;
    p[64] = o[0];
    p[65] = o[1];
    p[66] = s0;
    p[67] = s1;
    p[68] = __logf(s0 * s1);
  }
}

#ifdef RIP_PROFILE_TICKS  // development (tools/search_ticks.py): where a wave's cycles go; one workgroup prints at the end
#define TK_DECL() long long tk_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk0_ = 0
#define TK_START() tk0_ = clock64()
#define TK_STOP(i_) tk_[i_] += clock64() - tk0_
#else
#define TK_DECL()
#define TK_START()
#define TK_STOP(i_)
#endif

// WPB: waves per workgroup.  8 = two per SIMD (full launches); 4 / 2 for launches that would otherwise leave CUs idle
// (one workgroup per CU holds the operand buffers: 128 observations x 8 blocks are 128 eight-wave workgroups on 256
// CUs, but 256 four-wave ones).
template <bool TRACE, int WPB>
__global__ __launch_bounds__(WPB * 64) void search_phase_kernel(SearchArgs a, const float* __restrict__ mw_all,
                                                                const float* __restrict__ pre_all,
                                                                float4* __restrict__ tape_all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  PShared<WPB>& sh = *reinterpret_cast<PShared<WPB>*>(smem_raw);
  // launched behind the split-f16 kernel as its operand-range fallback: nothing to do unless that launch raised the word
  if (a.run_if_flag != nullptr && __builtin_nontemporal_load(a.run_if_flag) == 0u) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int c = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K;
  const int blocks_per_obs = a.N / CB;
  const int items = a.B * blocks_per_obs;
  const int item = blockIdx.x * WPB + wave;
  const bool active = item < items;               // a tail workgroup may carry idle waves (they still serve the DMA)
  const int it = active ? item : items - 1;
  const int b = it / blocks_per_obs;
  const int n0 = (it - b * blocks_per_obs) * CB;
  const size_t row = (size_t)b * a.N + n0 + c;
  const float* goal = a.goal != nullptr ? a.goal + (size_t)b * a.G * 2 : nullptr;
  const float* mw0 = mw_all + (size_t)a.k0 * MW_SIZE;

  float (*io)[8] = sh.io[wave];
  float (*gy)[8] = sh.gy[wave];
  float (*stF)[6][CB] = sh.stape[wave][0];
  float (*stI)[6][CB] = sh.stape[wave][1];
  const float4* wl = sh.fbuf + lane;
  const float4* tw = sh.tbuf + lane;
  const float4* wq4 = sh.wihc + q * 2 + (c & 1);
  // wave-uniform tape bases (scalar registers): lanes add their own 16-byte column at each access
  float4* tapeF = tape_all + ((size_t)item * 2 + (RIP_ABL == 4 ? 1 : 0)) * TAPE_SLOT_F4;  // ABL 4: aliased tapes
  float4* tapeI = tape_all + ((size_t)item * 2 + 1) * TAPE_SLOT_F4;

  // Adam state: lane (c, q) owns latent coordinates 2q, 2q+1 of candidate c
  float xv0 = a.x0[row * 8 + 2 * q], xv1 = a.x0[row * 8 + 2 * q + 1];
  float am0 = 0.f, am1 = 0.f, av0 = 0.f, av1 = 0.f;
  float xb0 = xv0, xb1 = xv1, lbest = 1000.0f;
  double b1p = 1.0, b2p = 1.0;
  const bool mean_mode = a.algorithm == ALGO_MA;
  const float inv_k = 1.0f / (float)K;

  load_fbuf(sh, mw0, wave, lane);
  load_tbuf(sh, K > 1 ? mw0 + MW_SIZE : mw0, wave, lane, tid);

  const int S = a.num_steps;
  TK_DECL();
#pragma unroll 1
  for (int step = 0; step <= S; ++step) {
    const bool final_pass = step == S;
    // ================= F_0: x -> y (F-buf = model 0) =================
    io[c][2 * q] = final_pass ? xb0 : xv0;
    io[c][2 * q + 1] = final_pass ? xb1 : xv1;
    TK_START();
    __syncthreads();  // F-buf (and, at step 0, T-buf) landed; io visible within the wave
    TK_STOP(0);
    float q_sel, gl = 0.f, gg0 = 0.f, gg1 = 0.f, w0;
    int ksel = 0;
    float gsel[8];
    {
      const Prefix16 pre = load_prefix(pre_all + ((size_t)0 * a.B + b) * PRE_FLOATS, q);
      TK_START();
      const PassOut po = pass_forward<MODE_FWD>(wl, pre, io, stF, tapeF, nullptr, c, q, (unsigned)lane);
      TK_STOP(1);
      __builtin_amdgcn_wave_barrier();
      if (final_pass) break;
      if (goal != nullptr) gl = goal_ll(goal, a.G, a.epsilon, io[c][6], io[c][7], &gg0, &gg1);
      q_sel = (-0.5f * po.sq - 4.0f * LOG_2PI) - po.lad;  // model 0's posterior through the self-inverse shortcut
      if (TRACE && a.trace_post != nullptr && q == 0 && active)
        a.trace_post[(((size_t)step * K + 0) * a.B + b) * a.N + n0 + c] = q_sel + gl;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) gsel[i] = 0.f;
    float q_sum = q_sel;
    // ================= models 1..K-1: inverse, adjoint, streaming aggregation =================
#pragma unroll 1
    for (int k = 1; k < K; ++k) {
      TK_START();
      __syncthreads();  // every wave is done with the F-buf (F_0 or inverse_{k-1}) and the T-buf (adjoint_{k-1})
      TK_STOP(2);
      TK_START();
      const float* mwk = mw_all + (size_t)(a.k0 + k) * MW_SIZE;
      if (RIP_ABL != 2) {
        load_fbuf(sh, mwk, wave, lane);
        if (k > 1) load_tbuf(sh, mwk, wave, lane, tid);  // (model 1's T-buf was requested under F_0)
      }
      __syncthreads();  // operands of model k landed
      TK_STOP(3);
      TK_START();
      const Prefix16 pre = load_prefix(pre_all + ((size_t)k * a.B + b) * PRE_FLOATS, q);
      constexpr bool REGTAPE = RIP_REGTAPE && WPB <= 4;
      StepTape last[3];
      const PassOut po = pass_forward<MODE_INV, REGTAPE>(wl, pre, io, stI, tapeI, last, c, q, (unsigned)lane);
      TK_STOP(4);
      const float qk = (-0.5f * po.sq - 4.0f * LOG_2PI) - po.lad;  // rip/agent.py:111-112
      if (TRACE && a.trace_post != nullptr && q == 0 && active)
        a.trace_post[(((size_t)step * K + k) * a.B + b) * a.N + n0 + c] = qk + gl;
      q_sum += qk;
      // rip/agent.py:121-127 as coded: WCM = min_k(-q) = the largest posterior, BCM = the smallest (first on ties)
      const bool take = a.algorithm == ALGO_WCM ? (qk > q_sel) : (qk < q_sel);
      if (mean_mode || __any(take)) {
        __builtin_amdgcn_wave_barrier();
        float res[8];
        TK_START();
        pass_backward<MODE_INV, REGTAPE>(tw, wq4, wl, io, nullptr, stI, tapeI, last, pre_all + ((size_t)k * a.B + b) * PRE_FLOATS, c, q,
                                res, 0.f);
        if (a.stats != nullptr && lane == 0 && active) atomicAdd(a.stats, 1ull);  // executed inverse-pass adjoints (bench.py)
        TK_STOP(5);
        if (mean_mode) {
#pragma unroll
          for (int i = 0; i < 8; ++i) gsel[i] += inv_k * res[i];
        } else if (take) {
#pragma unroll
          for (int i = 0; i < 8; ++i) gsel[i] = res[i];
        }
      }
      if (!mean_mode && take) {
        q_sel = qk;
        ksel = k;
      }
    }
    const float loss = -((mean_mode ? q_sum * inv_k : q_sel) + gl);
    w0 = (mean_mode ? inv_k : (ksel == 0 ? 1.0f : 0.0f)) * a.grad_scale;
    // ================= adjoint of F_0 + Adam (T-buf = model 0) =================
    if (K > 1) {
      TK_START();
      __syncthreads();  // every wave is done with model K-1's buffers
      TK_STOP(6);
      if (RIP_ABL != 2) {
        load_tbuf(sh, mw0, wave, lane, tid);
        load_fbuf(sh, mw0, wave, lane);  // next step's F_0
      }
    }
    // dLoss/dy = -(sum_k w_k dq_k/dy + d gl/dy_T): lane (c, q) fills coordinates 2q, 2q+1
    {
      float ga = q == 0 ? gsel[0] : q == 1 ? gsel[2] : q == 2 ? gsel[4] : gsel[6];
      float gb = q == 0 ? gsel[1] : q == 1 ? gsel[3] : q == 2 ? gsel[5] : gsel[7];
      if (q == 3) {
        ga += gg0;
        gb += gg1;
      }
      gy[c][2 * q] = -ga * a.grad_scale;
      gy[c][2 * q + 1] = -gb * a.grad_scale;
    }
    TK_START();
    if (K > 1) __syncthreads();  // model 0's T-buf (and next step's F-buf) landed
    TK_STOP(7);
    __builtin_amdgcn_wave_barrier();
    float res[8];
    TK_START();
    pass_backward<MODE_FWD>(tw, wq4, wl, io, gy, stF, tapeF, nullptr, pre_all + ((size_t)0 * a.B + b) * PRE_FLOATS, c, q, res, w0);
    TK_STOP(8);
    const float g0 = q == 0 ? res[0] : q == 1 ? res[2] : q == 2 ? res[4] : res[6];
    const float g1 = q == 0 ? res[1] : q == 1 ? res[3] : q == 2 ? res[5] : res[7];
    // ---- Adam (torch.optim.Adam defaults) + bookkeeping ----
    b1p *= 0.9;
    b2p *= 0.999;
    const float step_size = (float)((double)a.lr / (1.0 - b1p));
    const float bc2s = (float)sqrt(1.0 - b2p);
    am0 = am0 + (g0 - am0) * 0.1f;
    am1 = am1 + (g1 - am1) * 0.1f;
    av0 = av0 * 0.999f + 0.001f * g0 * g0;
    av1 = av1 * 0.999f + 0.001f * g1 * g1;
    xv0 = xv0 - step_size * (am0 / (sqrtf(av0) / bc2s + 1e-8f));
    xv1 = xv1 - step_size * (am1 / (sqrtf(av1) / bc2s + 1e-8f));
    if (loss < lbest) {  // post-step x vs pre-step loss (rip/agent.py:131-135)
      xb0 = xv0;
      xb1 = xv1;
      lbest = loss;
    }
    if (TRACE && active) {
      const size_t srow = (size_t)step * a.B * a.N + row;
      if (a.trace_grad != nullptr) {
        a.trace_grad[srow * 8 + 2 * q] = g0;
        a.trace_grad[srow * 8 + 2 * q + 1] = g1;
      }
      if (a.trace_x != nullptr) {
        a.trace_x[srow * 8 + 2 * q] = xv0;
        a.trace_x[srow * 8 + 2 * q + 1] = xv1;
      }
      if (a.trace_loss != nullptr && q == 0) a.trace_loss[srow] = loss;
    }
    if (K > 1 && S > 0) {
      // model 1's transposed operands for the next step, requested once every wave has left the T-buf
      __syncthreads();
      if (step + 1 < S && RIP_ABL != 2) load_tbuf(sh, mw0 + MW_SIZE, wave, lane, tid);
    }
  }
#ifdef RIP_PROFILE_TICKS
  if (blockIdx.x == 7 && lane == 0)
    printf("ticks wave %d: barrier-top %lld | F %lld | barrier-done %lld barrier-dma %lld | inv %lld adj %lld | "
           "barrier-last %lld barrier-dma0 %lld | adjF %lld\n", wave, tk_[0], tk_[1], tk_[2], tk_[3], tk_[4], tk_[5], tk_[6],
           tk_[7], tk_[8]);
#endif
  // plan = F_0(x_best) is in io (rip/agent.py:137)
  if (active) {
    if (a.plans != nullptr) {
      a.plans[row * 8 + 2 * q] = io[c][2 * q];
      a.plans[row * 8 + 2 * q + 1] = io[c][2 * q + 1];
    }
    if (a.loss_best != nullptr && q == 0) a.loss_best[row] = lbest;
  }
}

static bool wants_trace(const SearchArgs& a) {
  return a.trace_post != nullptr || a.trace_x != nullptr || a.trace_loss != nullptr || a.trace_grad != nullptr;
}

}  // namespace

bool search_phase_supported(const SearchArgs& a) { return a.K >= 1 && a.K <= MAX_MODELS && a.N % CB == 0; }

// scratch of one launch: the prefix table [K][B][PRE_FLOATS] followed by two tape slots per 16-candidate block
size_t search_phase_scratch_bytes(int B, int N, int K) {
  if (N < CB) return 0;
  const size_t pre = ((size_t)K * B * PRE_FLOATS * sizeof(float) + 255) / 256 * 256;
  const size_t items = ((size_t)B * (N / CB) + WPB_MAX - 1) / WPB_MAX * WPB_MAX;
  return pre + items * 2 * TAPE_SLOT_F4 * sizeof(float4);
}

namespace {
template <int WPB>
hipError_t launch_phase_wpb(const SearchArgs& a, const float* mw_all, const float* pre, float4* tape, int items, hipStream_t s) {
  hipError_t e = allow_lds(reinterpret_cast<const void*>(search_phase_kernel<false, WPB>));
  if (e != hipSuccess) return e;
  e = allow_lds(reinterpret_cast<const void*>(search_phase_kernel<true, WPB>));
  if (e != hipSuccess) return e;
  const dim3 grid((items + WPB - 1) / WPB);
  if (wants_trace(a))
    hipLaunchKernelGGL((search_phase_kernel<true, WPB>), grid, dim3(WPB * 64), sizeof(PShared<WPB>), s, a, mw_all, pre, tape);
  else
    hipLaunchKernelGGL((search_phase_kernel<false, WPB>), grid, dim3(WPB * 64), sizeof(PShared<WPB>), s, a, mw_all, pre, tape);
  return hipGetLastError();
}
}  // namespace

// waves per workgroup: as many as keep ~256 workgroups (one per CU: the operand buffers fill its LDS) in the launch.
// Cost model from the measurements: an 8-wave workgroup (two waves per SIMD) takes twice as long as a 4-wave one (2.3 vs
// 1.15 ms for 10 Adam steps at K = 4), a 2-wave one about as long as a 4-wave one (1.1 ms); a launch is
// ceil(workgroups / CUs) rounds of that.  Ties go to the larger workgroup (fewer operand DMA streams).
static int phase_pick_wpb(int items) {
#ifdef RIP_FORCE_WPB  // development: pin the workgroup shape (profiling the register-tape builds at full launches)
  return RIP_FORCE_WPB;
#endif
  const int cus = device_cu_count();
  auto rounds = [&](int wpb) { return (double)((items + wpb * cus - 1) / (wpb * cus)); };
  const double c8 = 2.0 * rounds(8), c4 = rounds(4), c2 = 0.96 * rounds(2);
  if (c8 <= c4 && c8 <= c2) return 8;
  return c4 <= c2 ? 4 : 2;
}

// MFMA instructions (v_mfma_f32_16x16x4_f32) of a launch per 16-candidate block, same slots as search_split_info (the
// f16 slots are 0): fwd_step_lds = 251, an adjoint step = 274 (82 at t = T-1) + 4 when its tape is read back (gi_n).
void search_phase_info(int B, int N, int K, int out[9]) {
  (void)K;
  const int wpb = phase_pick_wpb(B * (N / CB));
  const bool regtape = RIP_REGTAPE && wpb <= 4;
  out[0] = wpb;
  out[1] = 0, out[2] = 3 * 251;
  out[3] = 0, out[4] = 2 * 274 + 82 + (regtape ? 0 : 2 * 4);
  out[5] = 0, out[6] = 2 * 274 + 82 + 3 * 4;
  out[7] = 0, out[8] = 251;
}

hipError_t launch_search_phase(const SearchArgs& a, const float* mw_all, void* scratch, hipStream_t s) {
  float* pre = reinterpret_cast<float*>(scratch);
  const size_t pre_bytes = ((size_t)a.K * a.B * PRE_FLOATS * sizeof(float) + 255) / 256 * 256;
  float4* tape = reinterpret_cast<float4*>(reinterpret_cast<char*>(scratch) + pre_bytes);
  hipLaunchKernelGGL(phase_prefix_kernel, dim3(a.B, a.K), dim3(64), 0, s, a, mw_all, pre);
  const int items = a.B * (a.N / CB);
  switch (phase_pick_wpb(items)) {
    case 8: return launch_phase_wpb<8>(a, mw_all, pre, tape, items, s);
    case 4: return launch_phase_wpb<4>(a, mw_all, pre, tape, items, s);
    default: return launch_phase_wpb<2>(a, mw_all, pre, tape, items, s);
  }
}

}  // namespace rip
