// MFMA-batched RIP plan search for gfx950: 16 candidate plans per wavefront on
// v_mfma_f32_16x16x4_f32 (fp32 in / fp32 accumulate, bitwise an fmaf chain).
//
// Same algorithm and phase structure as search_kernel (flow.hip; rip/agent.py:78-137), different
// mapping — the throughput kernel for large candidate counts:
//   * workgroup = 16 candidates of one observation x K models, wave k = model k.
//   * lane (c = lane & 15, q = lane >> 4) holds, for candidate c, hidden units i = 16u + 4q + r in 16
//     registers H[u][r].  Every product is computed transposed, OUT^T[j][cand] = sum_i W[j][i] H[cand][i]:
//     A = a 16-row weight tile, B = H.  The k-step (u, r) contracts units {16u+4q+r : q}, and the result tile
//     u' comes back as lane (c, q) reg r' <-> unit 16u'+4q+r' — exactly the H layout.  The output of a step is
//     therefore the B operand of the next one with no data movement; the unit permutation is absorbed into
//     the host-side weight layout (fold_and_pack_mfma).
//   * biases and the GRU's 2-wide input are one extra k-step with B = (y0, y1, 1, 0) over q.
//   * forward operands (251 values/lane) are register resident (MFMA reads AGPRs directly); the transposed
//     operands of the adjoint (274 values/lane) are streamed from L2 as lane-major float4.
//   * the per-step tape (88 values/lane/step) goes to a global scratch (L2 resident), per-candidate scalars to LDS.
#include "flow.h"
#include "flow_math.h"

namespace rip {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int T = 4;
constexpr int CB = 16;                        // candidates per workgroup
constexpr int TAPE_F4 = 22;                   // float4 per lane per heavy step: 4 x (hprev,r,z,n,ghn) + 2 x a1
constexpr int TAPE_STEP_F4 = TAPE_F4 * 64;    // float4 per heavy step
constexpr int TAPE_SLOT_F4 = 3 * TAPE_STEP_F4;

enum { MODE_FWD = 0, MODE_INV = 1 };

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 zero4() {
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return z;
}

// Tape accessors.  Non-temporal hints were measured and rejected: `nt` stores +14 % on the kernel (the adjoint then
// reads the tape from HBM instead of the memory-side cache), `nt` loads +-0.
__device__ __forceinline__ void tape_st(float4* p, float a, float b, float c, float d) {
  *p = make_float4(a, b, c, d);
}
__device__ __forceinline__ float4 tape_ld(const float4* p) {
  return *p;
}

// register-resident forward operands of one model (see MWF_* in flow.h)
struct MW {
  float wf[192];  // [(g*4+u')*16 + (u*4+r)]  W_hh[g*64+16u'+m][16u+4q+r]
  float wx[16];   // [a*4+u'], a in {r, z, gi_n, gh_n}: (W_ih[.][0], W_ih[.][1], bias, 0) over q
  float w1f[32];  // [mt*16 + (u*4+r)]        W1[16mt+m][16u+4q+r]
  float w1x[2];   // q==2 ? b1[16mt+m] : 0
  float w2f[8];   // [mt*4+r']                W2[m&3][16mt+4q+r']
  float w2x;      // q==2 ? b2[m&3] : 0
};

__device__ __forceinline__ void load_mw(MW& W, const float* __restrict__ blob, int lane) {
  const float4* p = reinterpret_cast<const float4*>(blob) + lane;
  float tmp[252];
#pragma unroll
  for (int i = 0; i < 63; ++i) {
    const float4 v = p[i * 64];
    tmp[4 * i + 0] = v.x;
    tmp[4 * i + 1] = v.y;
    tmp[4 * i + 2] = v.z;
    tmp[4 * i + 3] = v.w;
  }
#pragma unroll
  for (int i = 0; i < 192; ++i) W.wf[i] = tmp[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) W.wx[i] = tmp[192 + i];
#pragma unroll
  for (int i = 0; i < 32; ++i) W.w1f[i] = tmp[208 + i];
  W.w1x[0] = tmp[240];
  W.w1x[1] = tmp[241];
#pragma unroll
  for (int i = 0; i < 8; ++i) W.w2f[i] = tmp[242 + i];
  W.w2x = tmp[250];
}

// One GRU + head step for 16 candidates.  H (in/out): hidden state; (yp0, yp1): GRU input of this lane's
// candidate; o: head output (dloc0, dloc1, pre-softplus scale0, scale1), replicated over q.
template <bool SAVE>
__device__ __forceinline__ void fwd_step(const MW& W, float (&H)[16], float yp0, float yp1, int q,
                                         float4* __restrict__ tape, float (&o)[4]) {
  const float bin = q == 0 ? yp0 : (q == 1 ? yp1 : (q == 2 ? 1.f : 0.f));
  float Hn[16];
#pragma unroll
  for (int up = 0; up < 4; ++up) {
    f32x4 ar = zero4(), az = zero4(), agn = zero4(), ahn = zero4();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float b = H[s];
      ar = mfma(W.wf[(0 * 4 + up) * 16 + s], b, ar);
      az = mfma(W.wf[(1 * 4 + up) * 16 + s], b, az);
      ahn = mfma(W.wf[(2 * 4 + up) * 16 + s], b, ahn);
    }
    ar = mfma(W.wx[0 * 4 + up], bin, ar);
    az = mfma(W.wx[1 * 4 + up], bin, az);
    agn = mfma(W.wx[2 * 4 + up], bin, agn);
    ahn = mfma(W.wx[3 * 4 + up], bin, ahn);
    float rr[4], zz[4], nn[4];
    {
      // gates on unit pairs: the non-transcendental half of the math runs as v_pk_* (same formulas as sigmoidf_ /
      // tanhf_ in flow_math.h; VALU instructions of a wave do not overlap its own MFMAs, so fewer is faster)
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
        const f2 hold = {H[up * 4 + 2 * h], H[up * 4 + 2 * h + 1]};
        const f2 hn = __builtin_elementwise_fma(z2, hold - n2, n2);  // (1-z)*n + z*h
        rr[2 * h] = r2.x;
        rr[2 * h + 1] = r2.y;
        zz[2 * h] = z2.x;
        zz[2 * h + 1] = z2.y;
        nn[2 * h] = n2.x;
        nn[2 * h + 1] = n2.y;
        Hn[up * 4 + 2 * h] = hn.x;
        Hn[up * 4 + 2 * h + 1] = hn.y;
      }
    }
    if (SAVE) {
      tape_st(tape + (up * 5 + 0) * 64, H[up * 4], H[up * 4 + 1], H[up * 4 + 2], H[up * 4 + 3]);
      tape_st(tape + (up * 5 + 1) * 64, rr[0], rr[1], rr[2], rr[3]);
      tape_st(tape + (up * 5 + 2) * 64, zz[0], zz[1], zz[2], zz[3]);
      tape_st(tape + (up * 5 + 3) * 64, nn[0], nn[1], nn[2], nn[3]);
      tape_st(tape + (up * 5 + 4) * 64, ahn[0], ahn[1], ahn[2], ahn[3]);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) H[i] = Hn[i];
  // ---- head: a1 = W1 h + b1 (32 x cand), o = W2 relu(a1) + b2 (4 x cand) ----
  const float bone = q == 2 ? 1.f : 0.f;
  f32x4 a0 = zero4(), a1 = zero4();
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    a0 = mfma(W.w1f[s], H[s], a0);
    a1 = mfma(W.w1f[16 + s], H[s], a1);
  }
  a0 = mfma(W.w1x[0], bone, a0);
  a1 = mfma(W.w1x[1], bone, a1);
  if (SAVE) {
    tape_st(tape + 20 * 64, a0[0], a0[1], a0[2], a0[3]);
    tape_st(tape + 21 * 64, a1[0], a1[1], a1[2], a1[3]);
  }
  f32x4 oa = zero4(), ob = zero4();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    oa = mfma(W.w2f[r], fmaxf(a0[r], 0.f), oa);
    ob = mfma(W.w2f[4 + r], fmaxf(a1[r], 0.f), ob);
  }
  oa = mfma(W.w2x, bone, oa);
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = oa[r] + ob[r];
}

struct MPrefix {
  float H1[16];
  float dloc0, dloc1, s0, s1, lad;
};

struct MShared {
  float xbuf[CB][8];
  float ybuf[CB][8];
  float gsum[CB][8];
  float gl[3][CB];
  float q[4][CB];
  float gk[4][CB][8];
  float goal[2 * MAX_GOALS];
  float stape[4 + 1][T][6][CB];  // per-candidate scalars of every pass: x0,x1,s0,s1,sg0,sg1
  float dg[4][88 * 64];                   // per wave: gate gradients + carried dh of the adjoint (pass_backward)
  float4 wiht[4][12 * 64];                // per wave: W_ih^T operands of its model (48 contraction steps)
};

struct PassOut {
  float lad, sq;
};

// forward / inverse pass for this wave's 16 candidates (steps 1..3 are "heavy"; step 0 is the shared prefix)
__device__ __forceinline__ PassOut pass_forward(int mode, const MW& W, const MPrefix& pre, const float (*in)[8],
                                                float (*out)[8], float (*st)[6][CB], float4* __restrict__ tape,
                                                int c, int q) {
  PassOut po;
  po.lad = pre.lad;
  po.sq = 0.f;
  float yp0, yp1;
  {
    float x0, x1;
    if (mode == MODE_FWD) {
      x0 = in[c][0];
      x1 = in[c][1];
      yp0 = pre.dloc0 + pre.s0 * x0;
      yp1 = pre.dloc1 + pre.s1 * x1;
      po.sq = fmaf(x0, x0, x1 * x1);
      if (q == 0) {
        out[c][0] = yp0;
        out[c][1] = yp1;
      }
    } else {
      yp0 = in[c][0];
      yp1 = in[c][1];
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
#pragma unroll 1
  for (int t = 1; t < T; ++t) {
    float o[4];
    fwd_step<true>(W, H, yp0, yp1, q, tape + (t - 1) * TAPE_STEP_F4, o);
    const float s0 = softplusf_(o[2]) + 1e-3f;  // sequence.py:133
    const float s1 = softplusf_(o[3]) + 1e-3f;
    float x0, x1, y0, y1;
    if (mode == MODE_FWD) {
      x0 = in[c][2 * t];
      x1 = in[c][2 * t + 1];
      y0 = (yp0 + o[0]) + s0 * x0;  // sequence.py:136
      y1 = (yp1 + o[1]) + s1 * x1;
      po.sq = fmaf(x0, x0, fmaf(x1, x1, po.sq));
      if (q == 0) {
        out[c][2 * t] = y0;
        out[c][2 * t + 1] = y1;
      }
    } else {
      y0 = in[c][2 * t];
      y1 = in[c][2 * t + 1];
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
  }
  return po;
}

// adjoint pass.  MODE_INV: writes dq/dy (q = -0.5|x|^2 - logabsdet) to res[8]; MODE_FWD: takes dL/dy from
// gin[c][*] and writes dL/dx to res[8].  bw: this model's streamed transposed operands (lane base, stride 64);
// dgl: this wave's LDS scratch [64 slots][64 lanes]: slots 0-15 d pre_r, 16-31 d pre_z, 32-47 d gh_n, 48-63 d pre_n
// (kept in LDS so the 48-step contractions can be rolled loops with dynamic slot indices); slots 64-79 carry
// dh'_{t+1} z_{t+1}, slots 80-87 hold da1_t.  wiht4: LDS copy of the W_ih^T operands (lane base, stride 64).
__device__ __forceinline__ void pass_backward(int mode, const float4* __restrict__ bw,
                                              const float4* __restrict__ wiht4, const float (*gin)[8],
                                              const float (*st)[6][CB], const float4* __restrict__ tape,
                                              float* __restrict__ dgl, int c, int q, float (&res)[8],
                                              const float4 (&rw)[16], float w0 = 0.f, long long* ptk = nullptr) {
#ifdef RIP_PROFILE_TICKS
  long long pt0_ = 0;
#define PSTART() pt0_ = clock64()
#define PSTOP(i_) \
  if (ptk) ptk[i_] += clock64() - pt0_
#else
#define PSTART()
#define PSTOP(i_)
#endif
#pragma unroll
  for (int i = 0; i < 16; ++i) dgl[(64 + i) * 64] = 0.f;
  float carry0 = 0.f, carry1 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) res[i] = 0.f;
#pragma unroll 1
  for (int t = T - 1; t >= 1; --t) {
    PSTART();
    const float x0 = st[t][0][c], x1 = st[t][1][c], s0 = st[t][2][c], s1 = st[t][3][c];
    const float sg0 = st[t][4][c], sg1 = st[t][5][c];
    float dd0, dd1, dos0, dos1, c0, c1, r0, r1;
    if (mode == MODE_INV) {
      const float i0 = rcpf_(s0), i1 = rcpf_(s1);
      const float xs0 = x0 * i0, xs1 = x1 * i1;
      r0 = carry0 - xs0;
      r1 = carry1 - xs1;
      c0 = xs0;
      c1 = xs1;
      dd0 = xs0;
      dd1 = xs1;
      dos0 = (x0 * x0 - 1.0f) * i0 * sg0;
      dos1 = (x1 * x1 - 1.0f) * i1 * sg1;
    } else {
      // w0 != 0: this candidate's loss also holds -w0 * q_0 with q_0 = -0.5|x|^2 - logabsdet_F(x) evaluated on the
      // forward pass itself (inverse_0(F_0(x)) == x): d/dx_t = w0 x_t, d/ds_t = w0 / s_t.
      const float D0 = gin[c][2 * t] + carry0;
      const float D1 = gin[c][2 * t + 1] + carry1;
      r0 = fmaf(D0, s0, w0 * x0);
      r1 = fmaf(D1, s1, w0 * x1);
      c0 = D0;
      c1 = D1;
      dd0 = D0;
      dd1 = D1;
      dos0 = (D0 * x0 + w0 * rcpf_(s0)) * sg0;
      dos1 = (D1 * x1 + w0 * rcpf_(s1)) * sg1;
    }
    // static-index scatter of (r0, r1) into res[2t], res[2t+1]
#pragma unroll
    for (int tt = 1; tt < T; ++tt) {
      res[2 * tt] = tt == t ? r0 : res[2 * tt];
      res[2 * tt + 1] = tt == t ? r1 : res[2 * tt + 1];
    }
    const float4* tp = tape + (t - 1) * TAPE_STEP_F4;
    // ---- head adjoint: da1 = relu'(a1) * W2^T do ----
    const float4 w2t = bw[0];
    const float4 a1s0 = tape_ld(tp + 20 * 64), a1s1 = tape_ld(tp + 21 * 64);
    // Streamed operands are issued well ahead of their MFMAs (explicit register ring): the L2 round trip
    // (~700 cycles) is longer than the 16 MFMAs (512 cycles) one float4 feeds.  The contraction is one
    // sequence of 8 (W1^T, B = da1) + 48 (W_hh^T, B = dgh_{t+1}) steps; at t = T-1 only the first 8 exist.
    // The first 16 entries (W1^T and the first 8 of W_hh^T) are kernel-resident registers (rw); the ring streams
    // entries 16..55 and is first filled here, two resident bodies (~2k MFMA cycles) before its first use.  The
    // stream is not latency- but throughput-limited: all four waves of a CU miss L1 on every operand row (4 x 57 KB
    // per step) and the refill *issue* back-pressures the in-order instruction stream, stalling the MFMAs behind it
    // (a 16-deep ring changed nothing, removing the refills gave -20 %), so the cure is fewer bytes: -33 % here.
    constexpr int RING = 8;
    const int nsteps = t == T - 1 ? 8 : 56;
    float4 wb[RING];
    if (nsteps > 16) {
#pragma unroll
      for (int j = 0; j < RING; ++j) wb[j] = bw[(17 + j) * 64];
    }
    const float bdo = q == 0 ? dd0 : (q == 1 ? dd1 : (q == 2 ? dos0 : dos1));
    const f32x4 da0 = mfma(w2t.x, bdo, zero4());
    const f32x4 da1 = mfma(w2t.y, bdo, zero4());
    dgl[80 * 64] = a1s0.x > 0.f ? da0[0] : 0.f;
    dgl[81 * 64] = a1s0.y > 0.f ? da0[1] : 0.f;
    dgl[82 * 64] = a1s0.z > 0.f ? da0[2] : 0.f;
    dgl[83 * 64] = a1s0.w > 0.f ? da0[3] : 0.f;
    dgl[84 * 64] = a1s1.x > 0.f ? da1[0] : 0.f;
    dgl[85 * 64] = a1s1.y > 0.f ? da1[1] : 0.f;
    dgl[86 * 64] = a1s1.z > 0.f ? da1[2] : 0.f;
    dgl[87 * 64] = a1s1.w > 0.f ? da1[3] : 0.f;
    // ---- dh_t = W1^T da1_t + W_hh^T dgh_{t+1} + dh'_{t+1} z_{t+1} ----
    f32x4 acc0 = zero4(), acc1 = zero4(), acc2 = zero4(), acc3 = zero4();
    PSTOP(0);
    PSTART();
    // Two straight-line bodies (with / without ring refills) instead of a per-entry `if`: the uniform branches cut
    // the loop into one basic block per entry, and each block then read its LDS operand and waited for it
    // (lgkmcnt(0) right after the ds_read, 56 exposed LDS round trips per step).
    auto ring_body = [&](int s0, bool refill) __attribute__((always_inline)) {
      float bv[RING];
#pragma unroll
      for (int j = 0; j < RING; ++j) {
        bv[j] = dgl[(s0 - 8 + j) * 64];  // B operand slot of entry s0 + j >= 16: dgh
      }
#pragma unroll
      for (int j = 0; j < RING; ++j) {
        // read the entry straight out of its ring registers, refill after: a by-value copy of the float4 made the
        // compiler funnel every A operand through ONE scratch register (v_mov + s_nop before each MFMA, and the
        // write-after-read on that register against the MFMA in flight serialised the pipe: ~70 cycles per MFMA)
        acc0 = mfma(wb[j].x, bv[j], acc0);
        acc1 = mfma(wb[j].y, bv[j], acc1);
        acc2 = mfma(wb[j].z, bv[j], acc2);
        acc3 = mfma(wb[j].w, bv[j], acc3);
        if (refill) wb[j] = bw[(1 + s0 + RING + j) * 64];
      }
      // Keep each refill next to the entry it replaces: left alone, the scheduler clusters the 8 refill loads at the
      // end of the body and the next iteration waits for all of them (vmcnt(0)) after two entries' worth of MFMAs --
      // the "8-deep" ring then covers ~300 cycles instead of ~1000.
      __builtin_amdgcn_sched_group_barrier(0x100, RING, 0);  // the LDS operand reads first
#pragma unroll
      for (int j = 0; j < RING; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        if (refill) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    };
    auto resident_body = [&](int r0, int slot0) __attribute__((always_inline)) {
      float bv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[j] = dgl[(slot0 + j) * 64];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc0 = mfma(rw[r0 + j].x, bv[j], acc0);
        acc1 = mfma(rw[r0 + j].y, bv[j], acc1);
        acc2 = mfma(rw[r0 + j].z, bv[j], acc2);
        acc3 = mfma(rw[r0 + j].w, bv[j], acc3);
      }
    };
    // the GRU adjoint's tape rows (tg) are requested once no further operand refill has to queue behind them (vmcnt
    // retires in order), i.e. under the last 8 entries' MFMAs instead of at their first use
    float4 tg[20];
    if (nsteps <= 16) {
#pragma unroll
      for (int r = 0; r < 20; ++r) tg[r] = tape_ld(tp + r * 64);
    }
    resident_body(0, 80);  // entries 0..7: W1^T, B = da1
    if (nsteps > 16) {
      resident_body(8, 0);  // entries 8..15: W_hh^T rows of units 0..7 of d pre_r
#pragma unroll 1
      for (int s0 = 16; s0 + RING < nsteps; s0 += RING) ring_body(s0, true);
#pragma unroll
      for (int r = 0; r < 20; ++r) tg[r] = tape_ld(tp + r * 64);
      ring_body(nsteps - RING, false);
    }
    // ---- GRUCell adjoint, lane-local in the H layout ----
    PSTOP(1);
    PSTART();
    // two units per instruction (v_pk_mul_f32 / v_pk_fma_f32): VALU work of a wave does not overlap its own MFMAs
    // (tools/micro/mfma_mix.hip), so every VALU instruction saved here is matrix-pipe time gained.  Same operations
    // in the same order per unit as the scalar form (mul, mul, mul; no new contraction).
    using f2 = __attribute__((ext_vector_type(2))) float;
    const f32x4 accs[4] = {acc0, acc1, acc2, acc3};
#pragma unroll
    for (int up = 0; up < 4; ++up) {
      const float4 hp = tg[up * 5 + 0], rr = tg[up * 5 + 1], zz = tg[up * 5 + 2], nn = tg[up * 5 + 3];
      const float4 gh = tg[up * 5 + 4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // unit pairs (0,1), (2,3) of the tile
        const int i = up * 4 + 2 * h;
        const f2 hp2 = h ? f2{hp.z, hp.w} : f2{hp.x, hp.y};
        const f2 rr2 = h ? f2{rr.z, rr.w} : f2{rr.x, rr.y};
        const f2 zz2 = h ? f2{zz.z, zz.w} : f2{zz.x, zz.y};
        const f2 nn2 = h ? f2{nn.z, nn.w} : f2{nn.x, nn.y};
        const f2 gh2 = h ? f2{gh.z, gh.w} : f2{gh.x, gh.y};
        const f2 one = {1.0f, 1.0f};
        const f2 dh = f2{accs[up][2 * h], accs[up][2 * h + 1]} + f2{dgl[(64 + i) * 64], dgl[(65 + i) * 64]};
        const f2 dn = dh * (one - zz2);
        const f2 dzg = dh * (hp2 - nn2);
        const f2 dhz = dh * zz2;
        const f2 dp = dn * (one - nn2 * nn2);
        const f2 dr = dp * gh2;
        const f2 dgn = dp * rr2;
        const f2 dpr = dr * rr2 * (one - rr2);
        const f2 dpz = dzg * zz2 * (one - zz2);
        dgl[(64 + i) * 64] = dhz.x;
        dgl[(65 + i) * 64] = dhz.y;
        dgl[(48 + i) * 64] = dp.x;   // d pre_n
        dgl[(49 + i) * 64] = dp.y;
        dgl[(32 + i) * 64] = dgn.x;  // d gh_n
        dgl[(33 + i) * 64] = dgn.y;
        dgl[i * 64] = dpr.x;         // d pre_r
        dgl[(1 + i) * 64] = dpr.y;
        dgl[(16 + i) * 64] = dpz.x;  // d pre_z
        dgl[(17 + i) * 64] = dpz.y;
      }
    }
    // ---- du = W_ih^T (dpr, dpz, dpn): rows m <-> input dim m & 1 ----
    PSTOP(2);
    PSTART();
    f32x4 dua = zero4(), dub = zero4();
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      const int sl = 4 * g < 32 ? 4 * g : 4 * g + 16;
      const float4 wq = wiht4[g * 64];
      dua = mfma(wq.x, dgl[(sl + 0) * 64], dua);
      dub = mfma(wq.y, dgl[(sl + 1) * 64], dub);
      dua = mfma(wq.z, dgl[(sl + 2) * 64], dua);
      dub = mfma(wq.w, dgl[(sl + 3) * 64], dub);
    }
    carry0 = c0 + (dua[0] + dub[0]);
    carry1 = c1 + (dua[1] + dub[1]);
    PSTOP(3);
  }
  // ---- t = 0: coupling only ----
  {
    const float x0 = st[0][0][c], x1 = st[0][1][c], s0 = st[0][2][c], s1 = st[0][3][c];
    if (mode == MODE_INV) {
      res[0] = carry0 - x0 * rcpf_(s0);
      res[1] = carry1 - x1 * rcpf_(s1);
    } else {
      res[0] = fmaf(gin[c][0] + carry0, s0, w0 * x0);
      res[1] = fmaf(gin[c][1] + carry1, s1, w0 * x1);
    }
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void search_mfma_kernel(SearchArgs a, const float* __restrict__ mw_all,
                                                              float4* __restrict__ tape_all) {
  __shared__ MShared sh;
  const int tid = threadIdx.x, lane = tid & 63;
  const int c = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K;  // == NW
  const int blocks_per_obs = a.N / CB;
  const int b = blockIdx.x / blocks_per_obs;
  const int n0 = (blockIdx.x - b * blocks_per_obs) * CB;
  const int k = wave;
  const float* mwk = mw_all + (size_t)(a.k0 + k) * MW_SIZE;
  MW W;
  load_mw(W, mwk, lane);
  const float4* bw = reinterpret_cast<const float4*>(mwk + MWF_FLOATS) + lane;
  float4 rw[16];  // kernel-resident transposed operands: entries 0..15 of the adjoint contraction (see pass_backward)
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    rw[i] = bw[(1 + i) * 64];
    // opaque to the optimiser: otherwise the (constant) rows are simply re-loaded at every use
    asm volatile("" : "+v"(rw[i].x), "+v"(rw[i].y), "+v"(rw[i].z), "+v"(rw[i].w));
  }
#pragma unroll
  for (int g = 0; g < 12; ++g) sh.wiht[wave][g * 64 + lane] = bw[(57 + g) * 64];  // own model, read only by this wave
  const float4* wiht = sh.wiht[wave] + lane;
  float* dgl = sh.dg[wave] + lane;
  float4* tape_fwd = tape_all + ((size_t)blockIdx.x * (K + 1) + 0) * TAPE_SLOT_F4 + lane;
  float4* tape_inv = tape_all + ((size_t)blockIdx.x * (K + 1) + 1 + k) * TAPE_SLOT_F4 + lane;

  if (a.goal != nullptr)
    for (int i = tid; i < 2 * a.G; i += NW * 64) sh.goal[i] = a.goal[(size_t)b * a.G * 2 + i];
  const float* goal = a.goal != nullptr ? sh.goal : nullptr;

  // ---- shared prefix: step 0 from h_0 = z_k (same for all 16 candidates), y_0 = 0 ----
  MPrefix pre;
  {
    float H[16];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) H[u * 4 + r] = a.z[((size_t)k * a.B + b) * 64 + 16 * u + 4 * q + r];
    float o[4];
    fwd_step<false>(W, H, 0.f, 0.f, q, nullptr, o);
#pragma unroll
    for (int i = 0; i < 16; ++i) pre.H1[i] = H[i];
    pre.dloc0 = o[0];
    pre.dloc1 = o[1];
    pre.s0 = softplusf_(o[2]) + 1e-3f;
    pre.s1 = softplusf_(o[3]) + 1e-3f;
    pre.lad = __logf(pre.s0 * pre.s1);
  }

  // Adam state: lane (c, q) owns latent coordinates 2q, 2q+1 of candidate c (wave 0 only)
  const size_t row = (size_t)b * a.N + n0 + c;
  float x0v = a.x0[row * 8 + 2 * q], x1v = a.x0[row * 8 + 2 * q + 1];
  float m0 = 0.f, m1 = 0.f, v0 = 0.f, v1 = 0.f;
  float xb0 = x0v, xb1 = x1v;
  float loss_best = 1000.0f;
  double b1p = 1.0, b2p = 1.0;
  __syncthreads();

#pragma unroll 1
  for (int step = 0; step <= a.num_steps; ++step) {
    const bool final_pass = step == a.num_steps;
    if (wave == 0) {
      sh.xbuf[c][2 * q] = final_pass ? xb0 : x0v;
      sh.xbuf[c][2 * q + 1] = final_pass ? xb1 : x1v;
    }
    __syncthreads();
    // ---------------- forward phases: 0 = F_0 (wave 0), 1 = inverses (all waves) ----------------
    const int nph = final_pass ? 0 : 1;
#pragma unroll 1
    for (int ph = 0; ph <= nph; ++ph) {
      if (ph == 1 || wave == 0) {
        const int mode = ph == 0 ? MODE_FWD : MODE_INV;
        const PassOut po = pass_forward(mode, W, pre, ph == 0 ? sh.xbuf : sh.ybuf, sh.ybuf,
                                        sh.stape[ph == 0 ? 0 : 1 + k], ph == 0 ? tape_fwd : tape_inv, c, q);
        if (ph == 0) {
          float gl = 0.f, g0 = 0.f, g1 = 0.f;
          if (goal != nullptr && !final_pass) {
            __builtin_amdgcn_wave_barrier();
            gl = goal_ll(goal, a.G, a.epsilon, sh.ybuf[c][6], sh.ybuf[c][7], &g0, &g1);
          }
          if (q == 0) {
            sh.gl[0][c] = gl;
            sh.gl[1][c] = g0;
            sh.gl[2][c] = g1;
          }
        } else if (q == 0) {
          sh.q[k][c] = (-0.5f * po.sq - 4.0f * LOG_2PI) - po.lad;  // rip/agent.py:111-112
        }
      }
      __syncthreads();
    }
    if (final_pass) break;

    // ---------------- aggregate over the K models, per candidate (rip/agent.py:121-127) ----------------
    const float gl = sh.gl[0][c];
    int ksel = 0;
    float qsel = sh.q[0][c], qmean = sh.q[0][c];
    for (int kk = 1; kk < K; ++kk) {
      const float qk = sh.q[kk][c];
      qmean += qk;
      const bool take = a.algorithm == ALGO_WCM ? (qk > qsel) : (qk < qsel);
      if (take) {
        qsel = qk;
        ksel = kk;
      }
    }
    qmean /= (float)K;
    const bool mean_mode = a.algorithm == ALGO_MA;
    const float loss = -((mean_mode ? qmean : qsel) + gl);

    // ---------------- adjoint phases: 1 = inverses, 0 = F_0 ----------------
#pragma unroll 1
    for (int ph = 1; ph >= 0; --ph) {
      if (ph == 1) {
        const float wk = mean_mode ? 1.0f / (float)K : (ksel == k ? 1.0f : 0.0f);
        if (__any(wk != 0.f)) {
          float res[8];
          pass_backward(MODE_INV, bw, wiht, nullptr, sh.stape[1 + k], tape_inv, dgl, c, q, res, rw);
          sh.gk[k][c][2 * q] = wk * (q == 0 ? res[0] : q == 1 ? res[2] : q == 2 ? res[4] : res[6]);
          sh.gk[k][c][2 * q + 1] = wk * (q == 0 ? res[1] : q == 1 ? res[3] : q == 2 ? res[5] : res[7]);
        } else {
          sh.gk[k][c][2 * q] = 0.f;
          sh.gk[k][c][2 * q + 1] = 0.f;
        }
      } else if (wave == 0) {
        // dLoss/dy = -(sum_k w_k dq_k/dy + d gl/dy_T)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int e = 2 * q + j;
          float g = 0.f;
          for (int kk = 0; kk < K; ++kk) g += sh.gk[kk][c][e];
          if (e >= 6) g += sh.gl[e - 5][c];
          sh.gsum[c][e] = -g * a.grad_scale;
        }
        __builtin_amdgcn_wave_barrier();
        float res[8];
        pass_backward(MODE_FWD, bw, wiht, sh.gsum, sh.stape[0], tape_fwd, dgl, c, q, res, rw);
        const float g0 = q == 0 ? res[0] : q == 1 ? res[2] : q == 2 ? res[4] : res[6];
        const float g1 = q == 0 ? res[1] : q == 1 ? res[3] : q == 2 ? res[5] : res[7];
        // ---- Adam (torch.optim.Adam defaults) + bookkeeping ----
        b1p *= 0.9;
        b2p *= 0.999;
        const float step_size = (float)((double)a.lr / (1.0 - b1p));
        const float bc2s = (float)sqrt(1.0 - b2p);
        m0 = m0 + (g0 - m0) * 0.1f;
        m1 = m1 + (g1 - m1) * 0.1f;
        v0 = v0 * 0.999f + 0.001f * g0 * g0;
        v1 = v1 * 0.999f + 0.001f * g1 * g1;
        x0v = x0v - step_size * (m0 / (sqrtf(v0) / bc2s + 1e-8f));
        x1v = x1v - step_size * (m1 / (sqrtf(v1) / bc2s + 1e-8f));
        if (loss < loss_best) {  // post-step x vs pre-step loss (rip/agent.py:131-135)
          xb0 = x0v;
          xb1 = x1v;
          loss_best = loss;
        }
      }
      __syncthreads();
    }
  }
  // plan = F_0(x_best) is in ybuf (rip/agent.py:137)
  if (wave == 0) {
    if (a.plans != nullptr) {
      a.plans[row * 8 + 2 * q] = sh.ybuf[c][2 * q];
      a.plans[row * 8 + 2 * q + 1] = sh.ybuf[c][2 * q + 1];
    }
    if (a.loss_best != nullptr && q == 0) a.loss_best[row] = loss_best;
  }
}

// --------------------------------------------------------------------------------------------------------
// Software-pipelined variant: 32 candidates (blocks A, B of 16) per workgroup.  Wave 0 is the forward wave of
// model 0: F_0 and its adjoint only — its own posterior q_0 = -0.5|x|^2 - logabsdet_F needs no inverse pass
// because inverse_0(F_0(x)) == x (same weights, same teacher-forced inputs), and its gradient is folded into the
// F_0 adjoint (pass_backward's w0 terms).  Waves 1..K-1 run the inverses of models 1..K-1 and their adjoints.
// The two blocks are staggered by half an Adam step so both wave groups always have work:
//     tick      wave 0                 waves k >= 1
//     4i        F(A, i)                adjoint-inverse(B, i-1)
//     4i+1      adjoint-F(B, i-1)+Adam inverse(A, i)
//     4i+2      F(B, i)                adjoint-inverse(A, i)
//     4i+3      adjoint-F(A, i)+Adam   inverse(B, i)
// --------------------------------------------------------------------------------------------------------
struct MShared2 {
  float xbuf[2][CB][8];
  float ybuf[2][CB][8];
  float gsum[2][CB][8];
  float gl[2][3][CB];
  float q[2][4][CB];
  float gk[2][4][CB][8];
  float goal[2 * MAX_GOALS];
  float stape[2][4][T][6][CB];   // [block][0 = forward pass, k = inverse of model k]
  float dg[4][88 * 64];
  float4 wiht[4][12 * 64];
  // point-to-point progress counters between wave 0 (F / adjoint-F + Adam) and the model waves (inverse / adjoint):
  int flagF[2];   // F passes of the block finished            (written by wave 0)
  int cntInv[2];  // inverse passes of the block finished        (one increment per model wave and step)
  int cntAdj[2];  // adjoint-inverse passes of the block finished
};

// LDS operations of one wave execute in program order, so "write data, then bump the counter" / "see the counter,
// then read data" needs no fence on the hardware side -- only the compiler must keep the order (a release fence
// would also wait for the tape stores still in flight).
__device__ __forceinline__ void wait_ge(const int* p, int target) {
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void signal_inc(int* p, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct Agg {
  int ksel;
  float loss, w0;
  bool mean_mode;
};

__device__ __forceinline__ Agg aggregate(const float (*qb)[CB], const float (*glb)[CB], int K, int algorithm, int c,
                                         float grad_scale) {
  Agg g;
  const float gl = glb[0][c];
  int ksel = 0;
  float qsel = qb[0][c], qmean = qb[0][c];
  for (int kk = 1; kk < K; ++kk) {
    const float qk = qb[kk][c];
    qmean += qk;
    const bool take = algorithm == ALGO_WCM ? (qk > qsel) : (qk < qsel);
    if (take) {
      qsel = qk;
      ksel = kk;
    }
  }
  qmean /= (float)K;
  g.mean_mode = algorithm == ALGO_MA;
  g.ksel = ksel;
  g.loss = -((g.mean_mode ? qmean : qsel) + gl);
  g.w0 = (g.mean_mode ? 1.0f / (float)K : (ksel == 0 ? 1.0f : 0.0f)) * grad_scale;
  return g;
}

// TRACE: per-step posteriors / losses / gradients / post-step latents to the SearchArgs trace pointers (debug and
// parity path: golden G6 and the teacher-forced per-step checks run on THIS kernel); the production instantiation
// (TRACE = false) carries none of it.
template <int NW, bool TRACE>
__global__ __launch_bounds__(NW * 64) void search_mfma2_kernel(SearchArgs a, const float* __restrict__ mw_all,
                                                               float4* __restrict__ tape_all) {
  __shared__ MShared2 sh;
  const int tid = threadIdx.x, lane = tid & 63;
  const int c = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K;  // == NW
  const int wgs_per_obs = a.N / (2 * CB);
  const int b = blockIdx.x / wgs_per_obs;
  const int n0 = (blockIdx.x - b * wgs_per_obs) * 2 * CB;
  const int k = wave;
  const float* mwk = mw_all + (size_t)(a.k0 + k) * MW_SIZE;
  // The 251 forward operands are (re)loaded from L2 at the start of every forward / inverse pass and are dead during
  // the adjoint passes: that leaves the accumulation registers free there (no scratch spills next to the tape traffic).
  const float4* bw = reinterpret_cast<const float4*>(mwk + MWF_FLOATS) + lane;
  float4 rw[16];  // kernel-resident transposed operands: entries 0..15 of the adjoint contraction (see pass_backward)
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    rw[i] = bw[(1 + i) * 64];
    // opaque to the optimiser: otherwise the (constant) rows are simply re-loaded at every use
    asm volatile("" : "+v"(rw[i].x), "+v"(rw[i].y), "+v"(rw[i].z), "+v"(rw[i].w));
  }
#pragma unroll
  for (int g = 0; g < 12; ++g) sh.wiht[wave][g * 64 + lane] = bw[(57 + g) * 64];
  const float4* wiht = sh.wiht[wave] + lane;
  float* dgl = sh.dg[wave] + lane;
  // tape slots of this workgroup: [block][wave]  (wave 0: forward pass, wave k: inverse of model k)
  float4* tape_blk[2];
  tape_blk[0] = tape_all + ((size_t)blockIdx.x * 2 * K + 0 * K + wave) * TAPE_SLOT_F4 + lane;
  tape_blk[1] = tape_all + ((size_t)blockIdx.x * 2 * K + 1 * K + wave) * TAPE_SLOT_F4 + lane;

  if (a.goal != nullptr)
    for (int i = tid; i < 2 * a.G; i += NW * 64) sh.goal[i] = a.goal[(size_t)b * a.G * 2 + i];
  const float* goal = a.goal != nullptr ? sh.goal : nullptr;

  MPrefix pre;
  {
    float H[16];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) H[u * 4 + r] = a.z[((size_t)k * a.B + b) * 64 + 16 * u + 4 * q + r];
    float o[4];
    MW W;
    load_mw(W, mwk, lane);
    fwd_step<false>(W, H, 0.f, 0.f, q, nullptr, o);
#pragma unroll
    for (int i = 0; i < 16; ++i) pre.H1[i] = H[i];
    pre.dloc0 = o[0];
    pre.dloc1 = o[1];
    pre.s0 = softplusf_(o[2]) + 1e-3f;
    pre.s1 = softplusf_(o[3]) + 1e-3f;
    pre.lad = __logf(pre.s0 * pre.s1);
  }

  // Adam state of both blocks: lane (c, q) owns coordinates 2q, 2q+1 of candidate c (wave 0 only)
  float xv[2][2], am[2][2], av[2][2], xb[2][2], lbest[2];
  double b1p[2] = {1.0, 1.0}, b2p[2] = {1.0, 1.0};
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const size_t row = (size_t)b * a.N + n0 + blk * CB + c;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      xv[blk][j] = a.x0[row * 8 + 2 * q + j];
      xb[blk][j] = xv[blk][j];
      am[blk][j] = 0.f;
      av[blk][j] = 0.f;
    }
    lbest[blk] = 1000.0f;
  }
  if (tid < 2) {
    sh.flagF[tid] = 0;
    sh.cntInv[tid] = 0;
    sh.cntAdj[tid] = 0;
  }
  __syncthreads();

  const int S = a.num_steps;
#ifdef RIP_PROFILE_TICKS  // tools/search_ticks.py: where a wave's cycles go (block 0 prints at the end)
  long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = 0;
  long long ptk[4] = {0, 0, 0, 0};  // adjoint-F sub-phases: head adjoint + a1 wait, contraction, GRU adjoint, du
#define TSTART() t0_ = clock64()
#define TSTOP(i_) tk[i_] += clock64() - t0_
#else
#define TSTART()
#define TSTOP(i_)
#endif
  // Two blocks of 16 candidates per workgroup, software-pipelined against each other.  Per Adam step and block the
  // chain is  F -> inverse_k -> adjoint-inverse_k -> adjoint-F + Adam  (1 + 1 + 2 + 2 pass units); wave 0 owns the
  // F passes, wave k >= 1 the passes of model k, each 3 units per block-step.  The waves run their own sequences
  //   wave 0:  F(A,i)  adjF(B,i-1)  F(B,i)  adjF(A,i)          wave k:  inv(A,i)  adj(A,i)  inv(B,i)  adj(B,i)
  // and only wait on the counters of the passes they consume, so neither side idles while the other finishes a
  // longer pass (a barrier per phase costs max(1,2) units per phase: 4 units per block-step instead of 3).
  if (wave == 0) {
#pragma unroll 1
    for (int hs = 0; hs < 2 * (S + 1); ++hs) {
      const int i = hs >> 1, blk = hs & 1;
      {
        // ---------------- F(blk, i)  (or the final pass with x_best when i == S) ----------------
        const bool final_pass = i == S;
        sh.xbuf[blk][c][2 * q] = final_pass ? (blk ? xb[1][0] : xb[0][0]) : (blk ? xv[1][0] : xv[0][0]);
        sh.xbuf[blk][c][2 * q + 1] = final_pass ? (blk ? xb[1][1] : xb[0][1]) : (blk ? xv[1][1] : xv[0][1]);
        __builtin_amdgcn_wave_barrier();
        TSTART();
        MW W;
        const float* mwp = mwk;
        asm volatile("" : "+s"(mwp));  // opaque: keeps the operand loads inside the pass (no hoisting out of the loop)
        load_mw(W, mwp, lane);
        const PassOut po = pass_forward(MODE_FWD, W, pre, sh.xbuf[blk], sh.ybuf[blk], sh.stape[blk][0],
                                        blk ? tape_blk[1] : tape_blk[0], c, q);
        TSTOP(0);
        float gl = 0.f, g0 = 0.f, g1 = 0.f;
        if (goal != nullptr && !final_pass) {
          __builtin_amdgcn_wave_barrier();
          gl = goal_ll(goal, a.G, a.epsilon, sh.ybuf[blk][c][6], sh.ybuf[blk][c][7], &g0, &g1);
        }
        if (q == 0) {
          sh.gl[blk][0][c] = gl;
          sh.gl[blk][1][c] = g0;
          sh.gl[blk][2][c] = g1;
          sh.q[blk][0][c] = (-0.5f * po.sq - 4.0f * LOG_2PI) - po.lad;  // model 0's posterior via the shortcut
        }
        signal_inc(&sh.flagF[blk], lane);
      }
      {
        // ---------------- adjoint-F + Adam of the OTHER block: after F(A,i) -> (B, i-1); after F(B,i) -> (A, i) ----
        const int jb = blk ^ 1;
        const int j = blk == 0 ? i - 1 : i;
        if (j >= 0 && j < S) {
          TSTART();
          wait_ge(&sh.cntAdj[jb], (K - 1) * (j + 1));
          TSTOP(1);
          const Agg ag = aggregate(sh.q[jb], sh.gl[jb], K, a.algorithm, c, a.grad_scale);
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int e = 2 * q + jj;
            float g = 0.f;
            for (int kk = 1; kk < K; ++kk) g += sh.gk[jb][kk][c][e];
            if (e >= 6) g += sh.gl[jb][e - 5][c];
            sh.gsum[jb][c][e] = -g * a.grad_scale;
          }
          __builtin_amdgcn_wave_barrier();
          float res[8];
          TSTART();
#ifdef RIP_PROFILE_TICKS
          pass_backward(MODE_FWD, bw, wiht, sh.gsum[jb], sh.stape[jb][0], jb ? tape_blk[1] : tape_blk[0], dgl, c, q,
                        res, rw, ag.w0, ptk);
#else
          pass_backward(MODE_FWD, bw, wiht, sh.gsum[jb], sh.stape[jb][0], jb ? tape_blk[1] : tape_blk[0], dgl, c, q,
                        res, rw, ag.w0);
#endif
          TSTOP(2);
          const float g0 = q == 0 ? res[0] : q == 1 ? res[2] : q == 2 ? res[4] : res[6];
          const float g1 = q == 0 ? res[1] : q == 1 ? res[3] : q == 2 ? res[5] : res[7];
          // select this block's Adam state with static indices
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            if (bb == jb) {
              b1p[bb] *= 0.9;
              b2p[bb] *= 0.999;
              const float step_size = (float)((double)a.lr / (1.0 - b1p[bb]));
              const float bc2s = (float)sqrt(1.0 - b2p[bb]);
              am[bb][0] = am[bb][0] + (g0 - am[bb][0]) * 0.1f;
              am[bb][1] = am[bb][1] + (g1 - am[bb][1]) * 0.1f;
              av[bb][0] = av[bb][0] * 0.999f + 0.001f * g0 * g0;
              av[bb][1] = av[bb][1] * 0.999f + 0.001f * g1 * g1;
              xv[bb][0] = xv[bb][0] - step_size * (am[bb][0] / (sqrtf(av[bb][0]) / bc2s + 1e-8f));
              xv[bb][1] = xv[bb][1] - step_size * (am[bb][1] / (sqrtf(av[bb][1]) / bc2s + 1e-8f));
              if (ag.loss < lbest[bb]) {  // post-step x vs pre-step loss (rip/agent.py:131-135)
                xb[bb][0] = xv[bb][0];
                xb[bb][1] = xv[bb][1];
                lbest[bb] = ag.loss;
              }
              if (TRACE) {
                const size_t n = (size_t)n0 + bb * CB + c, row = (size_t)b * a.N + n;
                const size_t srow = (size_t)j * a.B * a.N + row;
                if (a.trace_grad != nullptr) {
                  a.trace_grad[srow * 8 + 2 * q] = g0;
                  a.trace_grad[srow * 8 + 2 * q + 1] = g1;
                }
                if (a.trace_x != nullptr) {
                  a.trace_x[srow * 8 + 2 * q] = xv[bb][0];
                  a.trace_x[srow * 8 + 2 * q + 1] = xv[bb][1];
                }
                if (a.trace_loss != nullptr && q == 0) a.trace_loss[srow] = ag.loss;
                if (a.trace_post != nullptr && q < K)  // lane (c, q) reports model q: posterior + goal term
                  a.trace_post[(((size_t)j * K + q) * a.B + b) * a.N + n] = sh.q[bb][q][c] + sh.gl[bb][0][c];
              }
            }
          }
        }
      }
    }
  } else {
#pragma unroll 1
    for (int hs = 0; hs < 2 * S; ++hs) {
      const int i = hs >> 1, blk = hs & 1;
      // ---------------- inverse(blk, i) of model k ----------------
      TSTART();
      wait_ge(&sh.flagF[blk], i + 1);
      TSTOP(3);
      TSTART();
      {
        MW W;
        const float* mwp = mwk;
        asm volatile("" : "+s"(mwp));
        load_mw(W, mwp, lane);
        const PassOut po = pass_forward(MODE_INV, W, pre, sh.ybuf[blk], sh.ybuf[blk], sh.stape[blk][k],
                                        blk ? tape_blk[1] : tape_blk[0], c, q);
        if (q == 0) sh.q[blk][k][c] = (-0.5f * po.sq - 4.0f * LOG_2PI) - po.lad;
      }
      TSTOP(4);
      signal_inc(&sh.cntInv[blk], lane);
      // ---------------- adjoint-inverse(blk, i): needs every model's posterior for the aggregation ----------------
      TSTART();
      wait_ge(&sh.cntInv[blk], (K - 1) * (i + 1));
      TSTOP(5);
      TSTART();
      {
        const Agg ag = aggregate(sh.q[blk], sh.gl[blk], K, a.algorithm, c, a.grad_scale);
        const float wk = ag.mean_mode ? 1.0f / (float)K : (ag.ksel == k ? 1.0f : 0.0f);
        if (__any(wk != 0.f)) {
          float res[8];
          pass_backward(MODE_INV, bw, wiht, nullptr, sh.stape[blk][k], blk ? tape_blk[1] : tape_blk[0], dgl, c, q, res,
                        rw);
          sh.gk[blk][k][c][2 * q] = wk * (q == 0 ? res[0] : q == 1 ? res[2] : q == 2 ? res[4] : res[6]);
          sh.gk[blk][k][c][2 * q + 1] = wk * (q == 0 ? res[1] : q == 1 ? res[3] : q == 2 ? res[5] : res[7]);
        } else {
          sh.gk[blk][k][c][2 * q] = 0.f;
          sh.gk[blk][k][c][2 * q + 1] = 0.f;
        }
      }
      TSTOP(6);
      signal_inc(&sh.cntAdj[blk], lane);
    }
  }
#ifdef RIP_PROFILE_TICKS
  if (blockIdx.x == 0 && lane == 0 && wave < 2)
    printf("ticks wave %d: F %lld waitAdj %lld adjF %lld | waitF %lld inv %lld waitInv %lld adj %lld | adjF phases: head %lld "
           "contraction %lld gru %lld du %lld\n", wave, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5], tk[6], ptk[0], ptk[1],
           ptk[2], ptk[3]);
#endif
  __syncthreads();
  // plans = F_0(x_best) of both blocks are in ybuf (rip/agent.py:137)
  if (wave == 0) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const size_t row = (size_t)b * a.N + n0 + blk * CB + c;
      if (a.plans != nullptr) {
        a.plans[row * 8 + 2 * q] = sh.ybuf[blk][c][2 * q];
        a.plans[row * 8 + 2 * q + 1] = sh.ybuf[blk][c][2 * q + 1];
      }
      if (a.loss_best != nullptr && q == 0) a.loss_best[row] = lbest[blk];
    }
  }
}

}  // namespace

// 0 when the kernel can never run for this (K, N): rip_create sizes the handle's tape with (max_batch, max_candidates)
size_t search_mfma_tape_bytes(int B, int N, int K) {
  if (K > 4 || N < CB) return 0;
  return (size_t)B * (N / CB) * (K + 1) * TAPE_SLOT_F4 * sizeof(float4);
}

static bool wants_trace(const SearchArgs& a) {
  return a.trace_post != nullptr || a.trace_x != nullptr || a.trace_loss != nullptr || a.trace_grad != nullptr;
}

// K <= 4 (wave k = model k), N a multiple of 16; the trace outputs exist on the pipelined dual-block kernel only
// (N a multiple of 32).
bool search_mfma_supported(const SearchArgs& a) {
  return a.K >= 1 && a.K <= 4 && a.N % CB == 0 && (!wants_trace(a) || a.N % (2 * CB) == 0);
}

hipError_t launch_search_mfma(const SearchArgs& a, const float* mw_all, void* tape, hipStream_t s) {
  float4* tp = reinterpret_cast<float4*>(tape);
  if (a.N % (2 * CB) == 0) {  // pipelined dual-block kernel
    const dim3 grid2(a.B * (a.N / (2 * CB)));
    const bool tr = wants_trace(a);
#define LAUNCH2(NW_)                                                                                        \
  if (tr)                                                                                                   \
    hipLaunchKernelGGL((search_mfma2_kernel<NW_, true>), grid2, dim3(NW_ * 64), 0, s, a, mw_all, tp);       \
  else                                                                                                      \
    hipLaunchKernelGGL((search_mfma2_kernel<NW_, false>), grid2, dim3(NW_ * 64), 0, s, a, mw_all, tp)
    switch (a.K) {
      case 1:
        LAUNCH2(1);
        break;
      case 2:
        LAUNCH2(2);
        break;
      case 3:
        LAUNCH2(3);
        break;
      default:
        LAUNCH2(4);
        break;
    }
#undef LAUNCH2
    return hipGetLastError();
  }
  const dim3 grid(a.B * (a.N / CB));
  switch (a.K) {
    case 1:
      hipLaunchKernelGGL(search_mfma_kernel<1>, grid, dim3(64), 0, s, a, mw_all, tp);
      break;
    case 2:
      hipLaunchKernelGGL(search_mfma_kernel<2>, grid, dim3(128), 0, s, a, mw_all, tp);
      break;
    case 3:
      hipLaunchKernelGGL(search_mfma_kernel<3>, grid, dim3(192), 0, s, a, mw_all, tp);
      break;
    default:
      hipLaunchKernelGGL(search_mfma_kernel<4>, grid, dim3(256), 0, s, a, mw_all, tp);
      break;
  }
  return hipGetLastError();
}

}  // namespace rip
