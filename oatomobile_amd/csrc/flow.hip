// Autoregressive-flow kernels for gfx950 (CDNA4): forward / inverse / scoring and the fused
// RIP plan search (forward + K inverses + hand-written adjoint + Adam, all steps in one launch).
//
// Mapping: ONE 64-lane wavefront per (candidate plan, model) chain; lane j owns hidden unit j.
//   * GRU W_hh rows (3 x 64 floats per lane) live in registers for the whole launch;
//     h is broadcast lane->SGPR with v_readlane, so W_hh*h is 192 FMAs + 64 readlanes.
//   * the transposed products of the adjoint (W_hh^T dgh, W1^T da1) reuse the SAME row-resident
//     weights: every lane forms its 64 partial products and a 6-stage reduce-scatter
//     (v_permlane32_swap, v_permlane16_swap, then 8/4/2/1-lane shuffles) leaves sum_i in lane i.
//   * the per-step "tape" (h, r, z, n, gh_n, a1 per lane + a few uniform scalars) and the head's
//     W1 rows are staged in LDS; the K waves of a workgroup exchange y / scores / dL/dy through LDS.
//
// What it restates (reference file:line):
//   chain_forward<FWD>  AutoregressiveFlow._forward   torch/networks/sequence.py:95-151
//   chain_forward<INV>  AutoregressiveFlow._inverse   torch/networks/sequence.py:153-216
//   goal term           ImitativeModel._goal_likelihood   baselines/torch/dim/model.py:143-171
//   search_kernel       RIPAgent.__call__ loop        baselines/torch/rip/agent.py:78-137
//                       ImitativeModel.forward loop   baselines/torch/dim/model.py:98-141
//   adam                torch.optim.Adam defaults     rip/agent.py:96,131
#include "flow.h"
#include "flow_math.h"
#include "train.h"

#include <mutex>
#include <map>
#include <set>
#include <utility>

namespace rip {

namespace {

constexpr int T = 4;
constexpr int W1_STRIDE = 68;                 // floats (272 B rows): conflict-free ds_read_b128 across rows
constexpr int W1_LDS = 32 * W1_STRIDE;        // floats per model
constexpr int TAPE_Q = 6;                     // hprev, r, zg, n, ghn, a1
constexpr int TAPE_LANE = T * TAPE_Q * 64;
constexpr int TAPE_UNI = T * 8;               // x0,x1,s0,s1,sg0,sg1,pad,pad
constexpr int TAPE = TAPE_LANE + TAPE_UNI;    // floats per slot

enum { MODE_FWD = 0, MODE_INV = 1 };

// ------------------------------------------------------------------------------------------
// cross-lane helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float rl(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// sum over each row of 16 lanes, result replicated in the row
__device__ __forceinline__ float row_sum16(float x) {
  x += dpp<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp<0x141>(x);  // row_half_mirror
  x += dpp<0x140>(x);  // row_mirror
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
  x = row_sum16(x);
  return (rl(x, 0) + rl(x, 16)) + (rl(x, 32) + rl(x, 48));
}
// x[lane] + x[lane ^ 32]
__device__ __forceinline__ float xor32_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// reduce-scatter stages: `lo`/`hi` are this lane's partial sums for two target indices whose lane ids
// differ in bit D; afterwards the lane holds the pair-sum for the index matching its own bit D.
__device__ __forceinline__ float rs32(float lo, float hi) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float rs16(float lo, float hi) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int D>
__device__ __forceinline__ float xor_lane(float v) {  // v[lane ^ D] for D in {1,2,4,8}, all DPP (no LDS)
  if (D == 1) return dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  if (D == 2) return dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  if (D == 8) return dpp<0x128>(v);  // row_ror:8
  // D == 4: lanes with bit2 clear read lane+4 (row_shl:4, banks 0/2), the others lane-4 (row_shr:4, banks 1/3)
  int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0x5, false);
  r = __builtin_amdgcn_update_dpp(r, __float_as_int(v), 0x114, 0xf, 0xA, false);
  return __int_as_float(r);
}
template <int D>
__device__ __forceinline__ float rs_small(float lo, float hi, int lane) {
  const bool up = (lane & D) != 0;
  const float send = up ? lo : hi;
  const float keep = up ? hi : lo;
  return keep + xor_lane<D>(send);
}

// ------------------------------------------------------------------------------------------
// register-resident weights of one model, as seen by lane j
// ------------------------------------------------------------------------------------------
struct FlowRegs {
  float whh[3][64];  // W_hh[g*64+j][0..63]
  float wih[3][2];   // W_ih[g*64+j][d]
  float bih[3], bhh[3];
  float b1;          // b1[j&31]
  float w2a, w2b;    // W2[2*half+{0,1}][j&31]
  float b2[4];
};

__device__ __forceinline__ void load_flow_regs(FlowRegs& W, const float* __restrict__ blob, int lane) {
  const float4* p = reinterpret_cast<const float4*>(blob + FW_WHH);
#pragma unroll
  for (int g = 0; g < 3; ++g) {
#pragma unroll
    for (int i4 = 0; i4 < 16; ++i4) {
      const float4 v = p[(g * 16 + i4) * 64 + lane];
      W.whh[g][4 * i4 + 0] = v.x;
      W.whh[g][4 * i4 + 1] = v.y;
      W.whh[g][4 * i4 + 2] = v.z;
      W.whh[g][4 * i4 + 3] = v.w;
    }
  }
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    W.wih[g][0] = blob[FW_WIH + (g * 2 + 0) * 64 + lane];
    W.wih[g][1] = blob[FW_WIH + (g * 2 + 1) * 64 + lane];
    W.bih[g] = blob[FW_BIH + g * 64 + lane];
    W.bhh[g] = blob[FW_BHH + g * 64 + lane];
  }
  W.b1 = blob[FW_B1 + lane];
  W.w2a = blob[FW_W2 + lane];
  W.w2b = blob[FW_W2 + 64 + lane];
#pragma unroll
  for (int c = 0; c < 4; ++c) W.b2[c] = blob[FW_B2 + c];
}

// stage W1 [32][64] of one model into LDS rows of W1_STRIDE floats (all `nthreads` threads cooperate)
__device__ __forceinline__ void stage_w1(float* lds_w1, const float* __restrict__ blob, int tid, int nthreads) {
  const float4* src = reinterpret_cast<const float4*>(blob + FW_W1);
  for (int e = tid; e < 32 * 16; e += nthreads) {
    const int m = e >> 4, c = e & 15;
    *reinterpret_cast<float4*>(lds_w1 + m * W1_STRIDE + 4 * c) = src[e];
  }
}

// gh[g] = b_hh[g] + W_hh[g] h   and/or   a1 = b1 + W1[m] h    (one broadcast sweep over h)
template <bool WITH_GH, bool WITH_A1>
__device__ __forceinline__ void matvec(const FlowRegs& W, const float* w1row, float h, float (&gh)[3], float& a1) {
  if (WITH_GH) {
    gh[0] = W.bhh[0];
    gh[1] = W.bhh[1];
    gh[2] = W.bhh[2];
  }
  if (WITH_A1) a1 = W.b1;
#pragma unroll
  for (int i4 = 0; i4 < 16; ++i4) {
    float4 w1v;
    if (WITH_A1) w1v = *reinterpret_cast<const float4*>(w1row + 4 * i4);
    const float w1a[4] = {w1v.x, w1v.y, w1v.z, w1v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 4 * i4 + q;
      const float hi = rl(h, i);
      if (WITH_GH) {
        gh[0] = fmaf(W.whh[0][i], hi, gh[0]);
        gh[1] = fmaf(W.whh[1][i], hi, gh[1]);
        gh[2] = fmaf(W.whh[2][i], hi, gh[2]);
      }
      if (WITH_A1) a1 = fmaf(w1a[q], hi, a1);
    }
  }
}

// lane i <- sum over lanes j of ( W1[m_j][i]*da1h_j + W_hh[.*64+j][i] . (dpr,dpz,dghn)_j )
__device__ __forceinline__ float transposed_matvec(const FlowRegs& W, const float* w1row, float da1h, float dpr,
                                                   float dpz, float dghn, int lane) {
  float v[32];
#pragma unroll
  for (int i4 = 0; i4 < 8; ++i4) {
    const float4 wa = *reinterpret_cast<const float4*>(w1row + 4 * i4);
    const float4 wb = *reinterpret_cast<const float4*>(w1row + 32 + 4 * i4);
    const float a4[4] = {wa.x, wa.y, wa.z, wa.w};
    const float b4[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 4 * i4 + q;
      float lo = a4[q] * da1h;
      lo = fmaf(W.whh[0][i], dpr, lo);
      lo = fmaf(W.whh[1][i], dpz, lo);
      lo = fmaf(W.whh[2][i], dghn, lo);
      float hi = b4[q] * da1h;
      hi = fmaf(W.whh[0][i + 32], dpr, hi);
      hi = fmaf(W.whh[1][i + 32], dpz, hi);
      hi = fmaf(W.whh[2][i + 32], dghn, hi);
      v[i] = rs32(lo, hi);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = rs16(v[i], v[i + 16]);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = rs_small<8>(v[i], v[i + 8], lane);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = rs_small<4>(v[i], v[i + 4], lane);
#pragma unroll
  for (int i = 0; i < 2; ++i) v[i] = rs_small<2>(v[i], v[i + 2], lane);
  return rs_small<1>(v[0], v[1], lane);
}

struct ChainOut {
  float lad;  // sum_t log(s_t0 * s_t1)
  float sq;   // sum x^2 (INV)
};

// Candidate-independent prefix of a chain: with y_0 = 0 and h_0 = z the first GRU step, its head and
// W_hh h_1 depend only on (model, observation) — computed once per launch, not per candidate / Adam step.
struct Prefix {
  float h1;            // hidden state after step 0 (lane j)
  float gh[3];         // b_hh + W_hh h_1 (lane j)
  float dloc0, dloc1;  // head(h_1)
  float s0, s1;
  float lad;           // log(s0 * s1)
};

__device__ __forceinline__ void head_finish(const FlowRegs& W, float a1, float& o0, float& o1, float& o2, float& o3) {
  // head layer 2: lanes <32 own outputs 0,1 (dloc); lanes >=32 own outputs 2,3 (scale)
  const float a = fmaxf(a1, 0.f);
  const float p0 = row_sum16(W.w2a * a);
  const float p1 = row_sum16(W.w2b * a);
  o0 = (rl(p0, 0) + rl(p0, 16)) + W.b2[0];
  o1 = (rl(p1, 0) + rl(p1, 16)) + W.b2[1];
  o2 = (rl(p0, 32) + rl(p0, 48)) + W.b2[2];
  o3 = (rl(p1, 32) + rl(p1, 48)) + W.b2[3];
}

__device__ __forceinline__ Prefix chain_prefix(const FlowRegs& W, const float* w1row, float h0) {
  Prefix p;
  float gh[3], a1 = 0.f;
  matvec<true, false>(W, w1row, h0, gh, a1);
  // GRUCell with input 0 (sequence.py:119-128)
  const float r = sigmoidf_(W.bih[0] + gh[0]);
  const float zg = sigmoidf_(W.bih[1] + gh[1]);
  const float n = tanhf_(fmaf(r, gh[2], W.bih[2]));
  p.h1 = fmaf(zg, h0 - n, n);
  matvec<true, true>(W, w1row, p.h1, p.gh, a1);
  float o0, o1, o2, o3;
  head_finish(W, a1, o0, o1, o2, o3);
  p.dloc0 = o0;
  p.dloc1 = o1;
  p.s0 = softplusf_(o2) + 1e-3f;
  p.s1 = softplusf_(o3) + 1e-3f;
  p.lad = __logf(p.s0 * p.s1);
  return p;
}

// One pass over the T steps of a chain, starting from the shared prefix.
//   MODE_FWD: reads x from `in8` (LDS/any float[8]), writes y to `out8`.
//   MODE_INV: reads y from `in8`, writes x to `out8` (may be null).
// `tape` (LDS, TAPE floats) receives what the adjoint needs when SAVE.
template <bool SAVE>
__device__ __forceinline__ ChainOut chain_forward(int mode, const FlowRegs& W, const float* w1row, const Prefix& pre,
                                                  const float* in8, float* out8, float* tape, int lane) {
  ChainOut o;
  o.lad = pre.lad;
  o.sq = 0.f;
  float yp0, yp1;
  {  // ---- t = 0: only the affine coupling is candidate-specific ----
    float x0, x1;
    if (mode == MODE_FWD) {
      x0 = in8[0];
      x1 = in8[1];
      yp0 = pre.dloc0 + pre.s0 * x0;  // (0 + dloc) + scale * x, sequence.py:136
      yp1 = pre.dloc1 + pre.s1 * x1;
      o.sq = fmaf(x0, x0, x1 * x1);
      if (lane == 0) {
        out8[0] = yp0;
        out8[1] = yp1;
      }
    } else {
      yp0 = in8[0];
      yp1 = in8[1];
      x0 = (yp0 - pre.dloc0) * rcpf_(pre.s0);
      x1 = (yp1 - pre.dloc1) * rcpf_(pre.s1);
      o.sq = fmaf(x0, x0, x1 * x1);
      if (out8 != nullptr && lane == 0) {
        out8[0] = x0;
        out8[1] = x1;
      }
    }
    if (SAVE && lane == 0) {
      float* tu = tape + TAPE_LANE;
      tu[0] = x0;
      tu[1] = x1;
      tu[2] = pre.s0;
      tu[3] = pre.s1;
    }
  }
  float h = pre.h1;
  float gh[3] = {pre.gh[0], pre.gh[1], pre.gh[2]};
  float a1 = 0.f;
#pragma unroll 1
  for (int t = 1; t < T; ++t) {
    // ---- GRUCell (sequence.py:128 / :188), gate order r, z, n ----
    const float gir = fmaf(W.wih[0][1], yp1, fmaf(W.wih[0][0], yp0, W.bih[0]));
    const float giz = fmaf(W.wih[1][1], yp1, fmaf(W.wih[1][0], yp0, W.bih[1]));
    const float gin = fmaf(W.wih[2][1], yp1, fmaf(W.wih[2][0], yp0, W.bih[2]));
    const float r = sigmoidf_(gir + gh[0]);
    const float zg = sigmoidf_(giz + gh[1]);
    const float n = tanhf_(fmaf(r, gh[2], gin));
    const float hn = fmaf(zg, h - n, n);  // (1-z)*n + z*h
    if (SAVE) {
      float* tl = tape + t * TAPE_Q * 64 + lane;
      tl[0 * 64] = h;
      tl[1 * 64] = r;
      tl[2 * 64] = zg;
      tl[3 * 64] = n;
      tl[4 * 64] = gh[2];
    }
    h = hn;
    // ---- head layer 1 on h_t fused with W_hh h_t for the next step ----
    if (t < T - 1) {
      matvec<true, true>(W, w1row, h, gh, a1);
    } else {
      matvec<false, true>(W, w1row, h, gh, a1);
    }
    if (SAVE) tape[t * TAPE_Q * 64 + 5 * 64 + lane] = a1;
    float o0, o1, o2, o3;
    head_finish(W, a1, o0, o1, o2, o3);
    const float s0 = softplusf_(o2) + 1e-3f;  // sequence.py:133
    const float s1 = softplusf_(o3) + 1e-3f;
    float x0, x1, y0, y1;
    if (mode == MODE_FWD) {
      x0 = in8[2 * t];
      x1 = in8[2 * t + 1];
      y0 = (yp0 + o0) + s0 * x0;  // sequence.py:136
      y1 = (yp1 + o1) + s1 * x1;
      o.sq = fmaf(x0, x0, fmaf(x1, x1, o.sq));
      if (lane == 0) {
        out8[2 * t] = y0;
        out8[2 * t + 1] = y1;
      }
    } else {
      y0 = in8[2 * t];
      y1 = in8[2 * t + 1];
      x0 = (y0 - (yp0 + o0)) * rcpf_(s0);  // sequence.py:196
      x1 = (y1 - (yp1 + o1)) * rcpf_(s1);
      o.sq = fmaf(x0, x0, fmaf(x1, x1, o.sq));
      if (out8 != nullptr && lane == 0) {
        out8[2 * t] = x0;
        out8[2 * t + 1] = x1;
      }
    }
    o.lad += __logf(s0 * s1);  // sequence.py:211-214 (the :148-149 variant agrees to rounding)
    if (SAVE && lane == 0) {
      float* tu = tape + TAPE_LANE + t * 8;
      tu[0] = x0;
      tu[1] = x1;
      tu[2] = s0;
      tu[3] = s1;
      tu[4] = softplus_gradf_(o2);
      tu[5] = softplus_gradf_(o3);
    }
    yp0 = y0;
    yp1 = y1;
  }
  return o;
}

// Adjoint of one chain pass.
//   MODE_INV: q = -0.5|x|^2 - logabsdet  ->  writes dq/dy to out8.
//   MODE_FWD: given dL/dy in in8          ->  writes dL/dx to out8.
// Step 0 has no hidden-state dependence on the candidate (see Prefix), so only steps T-1..1 run the
// head / GRU adjoint; nothing flows into h_0 = z or y_0 = 0.
// w0 (MODE_FWD only): weight of model 0's own posterior q_0 = -0.5|x|^2 - logabsdet_F in the loss; because
// inverse_0(F_0(x)) == x it is evaluated on the forward pass itself and its gradient enters here:
// d(-w0 q_0)/dx_t = w0 x_t, d(-w0 q_0)/ds_t = w0 / s_t (s_0 belongs to the candidate-independent prefix).
__device__ __forceinline__ void chain_backward(int mode, const FlowRegs& W, const float* w1row, const float* tape,
                                               const float* in8, float* out8, int lane, float w0 = 0.f) {
  const bool upper = lane >= 32;
  float dhdir = 0.f, dpr = 0.f, dpz = 0.f, dghn = 0.f;
  float carry0 = 0.f, carry1 = 0.f;
#pragma unroll 1
  for (int t = T - 1; t >= 1; --t) {
    const float* tu = tape + TAPE_LANE + t * 8;
    const float x0 = tu[0], x1 = tu[1], s0 = tu[2], s1 = tu[3], sg0 = tu[4], sg1 = tu[5];
    float dd0, dd1, dos0, dos1, c0, c1;
    if (mode == MODE_INV) {
      const float i0 = rcpf_(s0), i1 = rcpf_(s1);
      const float xs0 = x0 * i0, xs1 = x1 * i1;  // x/s
      if (lane == 0) {
        out8[2 * t] = carry0 - xs0;  // dq/dy_t: own -x/s plus what step t+1 sent back
        out8[2 * t + 1] = carry1 - xs1;
      }
      c0 = xs0;  // dq/dy_{t-1} from x_t
      c1 = xs1;
      dd0 = xs0;  // dq/ddloc
      dd1 = xs1;
      dos0 = (x0 * x0 - 1.0f) * i0 * sg0;  // dq/ds = x^2/s - 1/s, through softplus
      dos1 = (x1 * x1 - 1.0f) * i1 * sg1;
    } else {
      const float D0 = in8[2 * t] + carry0;
      const float D1 = in8[2 * t + 1] + carry1;
      if (lane == 0) {
        out8[2 * t] = fmaf(D0, s0, w0 * x0);
        out8[2 * t + 1] = fmaf(D1, s1, w0 * x1);
      }
      c0 = D0;
      c1 = D1;
      dd0 = D0;
      dd1 = D1;
      dos0 = (D0 * x0 + w0 * rcpf_(s0)) * sg0;
      dos1 = (D1 * x1 + w0 * rcpf_(s1)) * sg1;
    }
    // ---- head adjoint ----
    const float* tl = tape + t * TAPE_Q * 64 + lane;
    const float part = W.w2a * (upper ? dos0 : dd0) + W.w2b * (upper ? dos1 : dd1);
    float da1 = xor32_sum(part);
    da1 = tl[5 * 64] > 0.f ? da1 : 0.f;
    const float da1h = upper ? 0.f : da1;  // rows of W1 are duplicated in both halves
    // ---- dh_t = W1^T da1_t + W_hh^T dgh_{t+1} + dh'_{t+1} * z_{t+1} ----
    const float dh = transposed_matvec(W, w1row, da1h, dpr, dpz, dghn, lane) + dhdir;
    // ---- GRUCell adjoint ----
    const float hprev = tl[0 * 64], r = tl[1 * 64], zg = tl[2 * 64], n = tl[3 * 64], ghn = tl[4 * 64];
    const float dn = dh * (1.0f - zg);
    const float dzg = dh * (hprev - n);
    dhdir = dh * zg;
    const float dpn = dn * (1.0f - n * n);
    const float dr = dpn * ghn;
    dghn = dpn * r;
    dpr = dr * r * (1.0f - r);
    dpz = dzg * zg * (1.0f - zg);
    const float du0 = wave_sum(fmaf(W.wih[0][0], dpr, fmaf(W.wih[1][0], dpz, W.wih[2][0] * dpn)));
    const float du1 = wave_sum(fmaf(W.wih[0][1], dpr, fmaf(W.wih[1][1], dpz, W.wih[2][1] * dpn)));
    carry0 = c0 + du0;
    carry1 = c1 + du1;
  }
  // ---- t = 0: coupling only ----
  if (lane == 0) {
    const float* tu = tape + TAPE_LANE;
    if (mode == MODE_INV) {
      out8[0] = carry0 - tu[0] * rcpf_(tu[2]);
      out8[1] = carry1 - tu[1] * rcpf_(tu[3]);
    } else {
      out8[0] = fmaf(in8[0] + carry0, tu[2], w0 * tu[0]);
      out8[1] = fmaf(in8[1] + carry1, tu[3], w0 * tu[1]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// plain flow kernels: one wave per row, grid-stride over rows, weights loaded once per wave
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flow_rows_kernel(int mode, const float* __restrict__ blob,
                                                         const float* __restrict__ in, const float* __restrict__ z,
                                                         int N, int z_rows, float* __restrict__ out,
                                                         float* __restrict__ logp, float* __restrict__ lad) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w1 = smem;                       // W1_LDS
  float* io = smem + W1_LDS;              // per wave: 8 in + 8 out
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  stage_w1(w1, blob, tid, blockDim.x);
  FlowRegs W;
  load_flow_regs(W, blob, lane);
  __syncthreads();
  const float* w1row = w1 + (lane & 31) * W1_STRIDE;
  float* my_in = io + wave * 16;
  float* my_out = my_in + 8;
  Prefix pre;
  bool first = true;
  for (int row = blockIdx.x * nw + wave; row < N; row += gridDim.x * nw) {
    if (lane < 8) my_in[lane] = in[(size_t)row * 8 + lane];
    if (first || z_rows != 1) pre = chain_prefix(W, w1row, z[(size_t)(z_rows == 1 ? 0 : row) * 64 + lane]);
    first = false;
    __builtin_amdgcn_wave_barrier();
    const ChainOut o = chain_forward<false>(mode, W, w1row, pre, my_in, my_out, nullptr, lane);
    __builtin_amdgcn_wave_barrier();
    if (out != nullptr && lane < 8) out[(size_t)row * 8 + lane] = my_out[lane];
    if (lane == 0) {
      if (lad != nullptr) lad[row] = o.lad;
      if (logp != nullptr) logp[row] = -0.5f * o.sq - 4.0f * LOG_2PI;  // MVN(0, I_8).log_prob, sequence.py:208
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void goal_rows_kernel(const float* __restrict__ y, const float* __restrict__ goal, int N, int goal_rows,
                                 int G, float eps, float* __restrict__ rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float* g = goal + (size_t)(goal_rows == 1 ? 0 : i) * G * 2;
  rows[i] = goal_ll(g, G, eps, y[(size_t)i * 8 + 6], y[(size_t)i * 8 + 7], nullptr, nullptr);
}

// S[k,b,n]: blockIdx.y = k (weights loaded once per wave), waves stride over the B*N rows
__global__ __launch_bounds__(256) void score_kernel(const float* __restrict__ flow_w, int k0,
                                                     const float* __restrict__ z, const float* __restrict__ y,
                                                     const float* __restrict__ goal, int B, int N, int G, float eps,
                                                     float* __restrict__ S) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w1 = smem;
  float* io = smem + W1_LDS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int k = blockIdx.y;
  const float* blob = flow_w + (size_t)(k0 + k) * FW_SIZE;
  stage_w1(w1, blob, tid, blockDim.x);
  FlowRegs W;
  load_flow_regs(W, blob, lane);
  __syncthreads();
  const float* w1row = w1 + (lane & 31) * W1_STRIDE;
  float* my_in = io + wave * 8;
  const int rows = B * N;
  Prefix pre;
  int pre_b = -1;
  for (int row = blockIdx.x * nw + wave; row < rows; row += gridDim.x * nw) {
    const int b = row / N;
    if (lane < 8) my_in[lane] = y[(size_t)row * 8 + lane];
    if (b != pre_b) {
      pre = chain_prefix(W, w1row, z[((size_t)k * B + b) * 64 + lane]);
      pre_b = b;
    }
    __builtin_amdgcn_wave_barrier();
    const ChainOut o = chain_forward<false>(MODE_INV, W, w1row, pre, my_in, nullptr, nullptr, lane);
    float s = (-0.5f * o.sq - 4.0f * LOG_2PI) - o.lad;
    if (goal != nullptr) s += goal_ll(goal + (size_t)b * G * 2, G, eps, my_in[6], my_in[7], nullptr, nullptr);
    if (lane == 0) S[((size_t)k * B + b) * N + (row - b * N)] = s;
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------
// fused plan search: one workgroup per (observation b, candidate n); NW waves share the K models
// ------------------------------------------------------------------------------------------
struct SearchShared {
  float xbuf[8];     // latent fed to F_0
  float ybuf[8];     // y = F_0(x)
  float gsum[8];     // dLoss/dy handed to the F_0 adjoint
  float dxbuf[8];    // dLoss/dx
  float gl[4];       // [0] goal log-likelihood, [1..2] its gradient wrt y_T
  float goal[2 * MAX_GOALS];
  float q[MAX_MODELS];       // log_prob - logabsdet per model
  float gk[MAX_MODELS][8];   // dq_k/dy
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void search_kernel(SearchArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K = a.K;
  float* w1_all = smem;                               // K * W1_LDS
  float* tapes = w1_all + K * W1_LDS;                 // (1 + K) * TAPE
  SearchShared& sh = *reinterpret_cast<SearchShared*>(tapes + (1 + K) * TAPE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bn = blockIdx.x;
  const int b = bn / a.N;
  const int R = (K + NW - 1) / NW;  // inverse rounds per step

  for (int k = 0; k < K; ++k) stage_w1(w1_all + k * W1_LDS, a.flow_w + (size_t)(a.k0 + k) * FW_SIZE, tid, NW * 64);

  FlowRegs W;
  int loaded = wave < K ? wave : 0;
  load_flow_regs(W, a.flow_w + (size_t)(a.k0 + loaded) * FW_SIZE, lane);
  __syncthreads();  // W1 rows staged
  Prefix pre = chain_prefix(W, w1_all + loaded * W1_LDS + (lane & 31) * W1_STRIDE,
                            a.z[((size_t)loaded * a.B + b) * 64 + lane]);  // prefix of the resident model
  if (a.goal != nullptr)
    for (int i = tid; i < 2 * a.G; i += NW * 64) sh.goal[i] = a.goal[(size_t)b * a.G * 2 + i];

  // Adam state of the 8 latent coordinates lives in lanes 0..7 of wave 0
  float x = 0.f, am = 0.f, av = 0.f, xbest = 0.f;
  if (lane < 8) x = a.x0[(size_t)bn * 8 + lane];
  xbest = x;
  float loss_best = 1000.0f;  // rip/agent.py:100
  double b1p = 1.0, b2p = 1.0;
  const float* goal = a.goal != nullptr ? sh.goal : nullptr;
  __syncthreads();

#pragma unroll 1
  for (int step = 0; step <= a.num_steps; ++step) {
    const bool final_pass = step == a.num_steps;
    if (wave == 0 && lane < 8) sh.xbuf[lane] = final_pass ? xbest : x;
    __syncthreads();
    // ---------------- forward phases: ph 0 = F_0 (wave 0), ph 1..R = inverses ----------------
    const int nph = final_pass ? 0 : R;
#pragma unroll 1
    for (int ph = 0; ph <= nph; ++ph) {
      int k = -1;
      if (ph == 0) {
        k = wave == 0 ? 0 : -1;
      } else {
        const int kk = wave + (ph - 1) * NW;
        k = (kk < K && kk != 0) ? kk : -1;  // model 0 needs no inverse pass: inverse_0(F_0(x)) == x
      }
      if (k >= 0) {
        if (k != loaded) {
          load_flow_regs(W, a.flow_w + (size_t)(a.k0 + k) * FW_SIZE, lane);
          pre = chain_prefix(W, w1_all + k * W1_LDS + (lane & 31) * W1_STRIDE, a.z[((size_t)k * a.B + b) * 64 + lane]);
          loaded = k;
        }
        const float* w1row = w1_all + k * W1_LDS + (lane & 31) * W1_STRIDE;
        const int mode = ph == 0 ? MODE_FWD : MODE_INV;
        float* tape = tapes + (ph == 0 ? 0 : 1 + k) * TAPE;
        const ChainOut o = chain_forward<true>(mode, W, w1row, pre, ph == 0 ? sh.xbuf : sh.ybuf,
                                               ph == 0 ? sh.ybuf : nullptr, tape, lane);
        if (ph == 0) {
          if (lane == 0) sh.q[0] = (-0.5f * o.sq - 4.0f * LOG_2PI) - o.lad;  // posterior of model 0 (shortcut)
          if (goal != nullptr && !final_pass) {
            float g0, g1;
            __builtin_amdgcn_wave_barrier();
            const float gl = goal_ll(goal, a.G, a.epsilon, sh.ybuf[6], sh.ybuf[7], &g0, &g1);
            if (lane == 0) {
              sh.gl[0] = gl;
              sh.gl[1] = g0;
              sh.gl[2] = g1;
            }
          } else if (lane == 0) {
            sh.gl[0] = 0.f;
            sh.gl[1] = 0.f;
            sh.gl[2] = 0.f;
          }
        } else if (lane == 0) {
          sh.q[k] = (-0.5f * o.sq - 4.0f * LOG_2PI) - o.lad;  // rip/agent.py:111-112
        }
      }
      __syncthreads();
    }
    if (final_pass) break;

    // ---------------- aggregate over the K models (rip/agent.py:121-127, as coded) ----------------
    const float gl = sh.gl[0];
    int ksel = 0;
    float qsel = sh.q[0], qmean = sh.q[0];
    for (int k = 1; k < K; ++k) {
      const float qk = sh.q[k];
      qmean += qk;
      // WCM: min_k(-posterior) = the largest posterior; BCM: max_k(-posterior) = the smallest
      const bool take = a.algorithm == ALGO_WCM ? (qk > qsel) : (qk < qsel);
      if (take) {
        qsel = qk;
        ksel = k;
      }
    }
    qmean /= (float)K;
    const bool mean_mode = a.algorithm == ALGO_MA;
    const float loss = -((mean_mode ? qmean : qsel) + gl);
    if (a.trace_post != nullptr && lane == 0) {
      for (int k = wave; k < K; k += NW)
        a.trace_post[(((size_t)step * K + k) * a.B + b) * a.N + (bn - b * a.N)] = sh.q[k] + gl;
    }

    // ---------------- adjoint phases: inverses (reverse order), then F_0 ----------------
#pragma unroll 1
    for (int ph = R; ph >= 0; --ph) {
      int k = -1;
      if (ph == 0) {
        k = wave == 0 ? 0 : -1;
      } else {
        const int kk = wave + (ph - 1) * NW;
        k = (kk < K && kk != 0) ? kk : -1;
      }
      const bool needed = ph == 0 || mean_mode || k == ksel;
      if (k >= 0 && !needed && lane < 8) sh.gk[k][lane] = 0.f;
      if (k >= 0 && needed) {
        if (k != loaded) {
          load_flow_regs(W, a.flow_w + (size_t)(a.k0 + k) * FW_SIZE, lane);
          pre = chain_prefix(W, w1_all + k * W1_LDS + (lane & 31) * W1_STRIDE, a.z[((size_t)k * a.B + b) * 64 + lane]);
          loaded = k;
        }
        const float* w1row = w1_all + k * W1_LDS + (lane & 31) * W1_STRIDE;
        if (ph == 0) {
          // dLoss/dy = -sum_k w_k dq_k/dy - d gl/dy_T, scaled (ImitativeModel.forward's batch mean)
          if (lane < 8) {
            float g = 0.f;
            if (mean_mode) {
              for (int kk = 1; kk < K; ++kk) g += sh.gk[kk][lane];
              g /= (float)K;
            } else if (ksel != 0) {
              g = sh.gk[ksel][lane];
            }
            if (lane >= 6) g += sh.gl[lane - 5];
            sh.gsum[lane] = -g * a.grad_scale;
          }
          __builtin_amdgcn_wave_barrier();
          const float w0 = (mean_mode ? 1.0f / (float)K : (ksel == 0 ? 1.0f : 0.0f)) * a.grad_scale;
          chain_backward(MODE_FWD, W, w1row, tapes, sh.gsum, sh.dxbuf, lane, w0);
        } else {
          chain_backward(MODE_INV, W, w1row, tapes + (1 + k) * TAPE, nullptr, sh.gk[k], lane);
        }
      }
      __syncthreads();
    }

    // ---------------- Adam (torch.optim.Adam defaults) + bookkeeping, wave 0 ----------------
    if (wave == 0) {
      b1p *= 0.9;
      b2p *= 0.999;
      if (lane < 8) {
        const float g = sh.dxbuf[lane];
        am = am + (g - am) * 0.1f;                       // exp_avg.lerp_(grad, 1-beta1)
        av = av * 0.999f + 0.001f * g * g;               // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
        const float step_size = (float)((double)a.lr / (1.0 - b1p));
        const float bc2s = (float)sqrt(1.0 - b2p);
        const float denom = sqrtf(av) / bc2s + 1e-8f;
        x = x - step_size * (am / denom);
        if (loss < loss_best) xbest = x;                 // post-step x vs pre-step loss (rip/agent.py:131-135)
        if (a.trace_x != nullptr) a.trace_x[((size_t)step * a.B * a.N + bn) * 8 + lane] = x;
        if (a.trace_grad != nullptr) a.trace_grad[((size_t)step * a.B * a.N + bn) * 8 + lane] = g;
      }
      if (loss < loss_best) loss_best = loss;
      if (a.trace_loss != nullptr && lane == 0) a.trace_loss[(size_t)step * a.B * a.N + bn] = loss;
    }
  }
  // plan = F_0(x_best) is in ybuf (rip/agent.py:137)
  if (wave == 0) {
    if (a.plans != nullptr && lane < 8) a.plans[(size_t)bn * 8 + lane] = sh.ybuf[lane];
    if (a.loss_best != nullptr && lane == 0) a.loss_best[bn] = loss_best;
  }
}

// ------------------------------------------------------------------------------------------
// Step-pipelined plan search (2 <= K <= 4, K waves): same algorithm as search_kernel, shorter critical path.
//   * wave 0 runs F_0 only (model 0's posterior comes from the self-inverse shortcut); wave k >= 1 runs the
//     inverse of model k.  Inverse step t needs y_{t-1} for its GRU + head and y_t only for the final
//     x_t = (y_t - mu_t) / s_t, so all waves advance through the T steps together, one barrier per step, and the
//     inverses finish a division after F_0 does.
//   * adjoint: dq_k/dy_t = carry_t - x_t/s_t is known at the START of inverse-adjoint iteration t, so it is
//     published first and the F_0 adjoint of the same step runs concurrently with the rest of that iteration.
//   Per Adam step the critical path is ~(T-1) forward + (T-1) adjoint steps instead of 4 (T-1).
// ------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void search_pipe_kernel(SearchArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int K = NW;
  float* w1_all = smem;                 // K * W1_LDS
  float* tapes = w1_all + K * W1_LDS;   // K * TAPE (wave w's pass)
  SearchShared& sh = *reinterpret_cast<SearchShared*>(tapes + K * TAPE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int k = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave == model
  const bool upper = lane >= 32;
  const int bn = blockIdx.x;
  const int b = bn / a.N;
  for (int kk = 0; kk < K; ++kk) stage_w1(w1_all + kk * W1_LDS, a.flow_w + (size_t)(a.k0 + kk) * FW_SIZE, tid, NW * 64);
  FlowRegs W;
  load_flow_regs(W, a.flow_w + (size_t)(a.k0 + k) * FW_SIZE, lane);
  if (a.goal != nullptr)
    for (int i = tid; i < 2 * a.G; i += NW * 64) sh.goal[i] = a.goal[(size_t)b * a.G * 2 + i];
  __syncthreads();
  const float* w1row = w1_all + k * W1_LDS + (lane & 31) * W1_STRIDE;
  const Prefix pre = chain_prefix(W, w1row, a.z[((size_t)k * a.B + b) * 64 + lane]);
  float* tape = tapes + k * TAPE;
  const float* goal = a.goal != nullptr ? sh.goal : nullptr;

  float x = 0.f, am = 0.f, av = 0.f, xbest = 0.f;
  if (lane < 8) x = a.x0[(size_t)bn * 8 + lane];
  xbest = x;
  float loss_best = 1000.0f;
  double b1p = 1.0, b2p = 1.0;
  const bool mean_mode = a.algorithm == ALGO_MA;

#pragma unroll 1
  for (int step = 0; step <= a.num_steps; ++step) {
    const bool final_pass = step == a.num_steps;
    // =============================== forward: all waves in lock step ===============================
    float sq = 0.f, lad = pre.lad;
    float yp0, yp1;
    if (k == 0) {
      if (lane < 8) sh.xbuf[lane] = final_pass ? xbest : x;
      __builtin_amdgcn_wave_barrier();
      const float x0 = sh.xbuf[0], x1 = sh.xbuf[1];
      yp0 = pre.dloc0 + pre.s0 * x0;
      yp1 = pre.dloc1 + pre.s1 * x1;
      sq = fmaf(x0, x0, x1 * x1);
      if (lane == 0) {
        sh.ybuf[0] = yp0;
        sh.ybuf[1] = yp1;
        float* tu = tape + TAPE_LANE;
        tu[0] = x0;
        tu[1] = x1;
        tu[2] = pre.s0;
        tu[3] = pre.s1;
      }
    }
    __syncthreads();
    if (k != 0) {
      yp0 = sh.ybuf[0];
      yp1 = sh.ybuf[1];
      const float x0 = (yp0 - pre.dloc0) * rcpf_(pre.s0), x1 = (yp1 - pre.dloc1) * rcpf_(pre.s1);
      sq = fmaf(x0, x0, x1 * x1);
      if (lane == 0) {
        float* tu = tape + TAPE_LANE;
        tu[0] = x0;
        tu[1] = x1;
        tu[2] = pre.s0;
        tu[3] = pre.s1;
      }
    }
    const bool inv_active = !final_pass || k == 0;  // the final pass only needs F_0
    float h = pre.h1;
    float gh[3] = {pre.gh[0], pre.gh[1], pre.gh[2]};
    float a1 = 0.f;
#pragma unroll 1
    for (int t = 1; t < T; ++t) {
      float o0 = 0.f, o1 = 0.f, s0 = 1.f, s1 = 1.f, o2 = 0.f, o3 = 0.f;
      if (inv_active) {
        const float gir = fmaf(W.wih[0][1], yp1, fmaf(W.wih[0][0], yp0, W.bih[0]));
        const float giz = fmaf(W.wih[1][1], yp1, fmaf(W.wih[1][0], yp0, W.bih[1]));
        const float gin = fmaf(W.wih[2][1], yp1, fmaf(W.wih[2][0], yp0, W.bih[2]));
        const float r = sigmoidf_(gir + gh[0]);
        const float zg = sigmoidf_(giz + gh[1]);
        const float n = tanhf_(fmaf(r, gh[2], gin));
        const float hn = fmaf(zg, h - n, n);
        float* tl = tape + t * TAPE_Q * 64 + lane;
        tl[0 * 64] = h;
        tl[1 * 64] = r;
        tl[2 * 64] = zg;
        tl[3 * 64] = n;
        tl[4 * 64] = gh[2];
        h = hn;
        if (t < T - 1) {
          matvec<true, true>(W, w1row, h, gh, a1);
        } else {
          matvec<false, true>(W, w1row, h, gh, a1);
        }
        tl[5 * 64] = a1;
        head_finish(W, a1, o0, o1, o2, o3);
        s0 = softplusf_(o2) + 1e-3f;
        s1 = softplusf_(o3) + 1e-3f;
        lad += __logf(s0 * s1);
      }
      float y0 = 0.f, y1 = 0.f;
      if (k == 0) {
        const float x0 = sh.xbuf[2 * t], x1 = sh.xbuf[2 * t + 1];
        y0 = (yp0 + o0) + s0 * x0;
        y1 = (yp1 + o1) + s1 * x1;
        sq = fmaf(x0, x0, fmaf(x1, x1, sq));
        if (lane == 0) {
          sh.ybuf[2 * t] = y0;
          sh.ybuf[2 * t + 1] = y1;
          float* tu = tape + TAPE_LANE + t * 8;
          tu[0] = x0;
          tu[1] = x1;
          tu[2] = s0;
          tu[3] = s1;
          tu[4] = softplus_gradf_(o2);
          tu[5] = softplus_gradf_(o3);
        }
      }
      __syncthreads();
      if (k != 0) {
        y0 = sh.ybuf[2 * t];
        y1 = sh.ybuf[2 * t + 1];
        if (inv_active) {
          const float x0 = (y0 - (yp0 + o0)) * rcpf_(s0), x1 = (y1 - (yp1 + o1)) * rcpf_(s1);
          sq = fmaf(x0, x0, fmaf(x1, x1, sq));
          if (lane == 0) {
            float* tu = tape + TAPE_LANE + t * 8;
            tu[0] = x0;
            tu[1] = x1;
            tu[2] = s0;
            tu[3] = s1;
            tu[4] = softplus_gradf_(o2);
            tu[5] = softplus_gradf_(o3);
          }
        }
      }
      yp0 = y0;
      yp1 = y1;
    }
    if (final_pass) break;
    if (k == 0) {
      float gl = 0.f, g0 = 0.f, g1 = 0.f;
      if (goal != nullptr) gl = goal_ll(goal, a.G, a.epsilon, yp0, yp1, &g0, &g1);
      if (lane == 0) {
        sh.gl[0] = gl;
        sh.gl[1] = g0;
        sh.gl[2] = g1;
      }
    }
    if (lane == 0) sh.q[k] = (-0.5f * sq - 4.0f * LOG_2PI) - lad;
    __syncthreads();

    // =============================== aggregate (rip/agent.py:121-127, as coded) ===============================
    const float gl = sh.gl[0];
    int ksel = 0;
    float qsel = sh.q[0], qmean = sh.q[0];
#pragma unroll
    for (int kk = 1; kk < K; ++kk) {
      const float qk = sh.q[kk];
      qmean += qk;
      const bool take = a.algorithm == ALGO_WCM ? (qk > qsel) : (qk < qsel);
      if (take) {
        qsel = qk;
        ksel = kk;
      }
    }
    qmean /= (float)K;
    const float loss = -((mean_mode ? qmean : qsel) + gl);
    if (a.trace_post != nullptr && lane == 0)
      a.trace_post[(((size_t)step * K + k) * a.B + b) * a.N + (bn - b * a.N)] = sh.q[k] + gl;
    const float wk = mean_mode ? 1.0f / (float)K : (ksel == k ? 1.0f : 0.0f);  // this wave's model in the loss
    const bool active = k == 0 || wk != 0.f;
    const float w0 = k == 0 ? wk * a.grad_scale : 0.f;

    // =============================== adjoint: F_0 trails the inverses by one hand-off ===============================
    float dhdir = 0.f, dpr = 0.f, dpz = 0.f, dghn = 0.f, carry0 = 0.f, carry1 = 0.f;
#pragma unroll 1
    for (int t = T - 1; t >= 0; --t) {
      const float* tu = tape + TAPE_LANE + t * 8;
      const float x0 = tu[0], x1 = tu[1], s0 = tu[2], s1 = tu[3];
      float xs0 = 0.f, xs1 = 0.f, i0 = 0.f, i1 = 0.f;
      if (k != 0) {  // publish dq_k/dy_t = carry - x/s (weighted), known before this iteration's heavy part
        i0 = rcpf_(s0);
        i1 = rcpf_(s1);
        xs0 = x0 * i0;
        xs1 = x1 * i1;
        if (lane == 0) {
          sh.gk[k][2 * t] = active ? wk * (carry0 - xs0) : 0.f;
          sh.gk[k][2 * t + 1] = active ? wk * (carry1 - xs1) : 0.f;
        }
      }
      __syncthreads();
      if (!active) continue;  // barriers above are unconditional; nothing below synchronises
      float dd0, dd1, dos0 = 0.f, dos1 = 0.f, c0, c1;
      if (k == 0) {
        float g0 = 0.f, g1 = 0.f;
#pragma unroll
        for (int kk = 1; kk < K; ++kk) {
          g0 += sh.gk[kk][2 * t];
          g1 += sh.gk[kk][2 * t + 1];
        }
        if (t == T - 1) {
          g0 += sh.gl[1];
          g1 += sh.gl[2];
        }
        const float D0 = -g0 * a.grad_scale + carry0, D1 = -g1 * a.grad_scale + carry1;
        if (lane == 0) {
          sh.dxbuf[2 * t] = fmaf(D0, s0, w0 * x0);
          sh.dxbuf[2 * t + 1] = fmaf(D1, s1, w0 * x1);
        }
        c0 = D0;
        c1 = D1;
        dd0 = D0;
        dd1 = D1;
        if (t > 0) {
          dos0 = (D0 * x0 + w0 * rcpf_(s0)) * tu[4];
          dos1 = (D1 * x1 + w0 * rcpf_(s1)) * tu[5];
        }
      } else {
        c0 = xs0;
        c1 = xs1;
        dd0 = xs0;
        dd1 = xs1;
        if (t > 0) {
          dos0 = (x0 * x0 - 1.0f) * i0 * tu[4];
          dos1 = (x1 * x1 - 1.0f) * i1 * tu[5];
        }
      }
      if (t == 0) break;  // step 0 is the candidate-independent prefix: coupling only
      const float* tl = tape + t * TAPE_Q * 64 + lane;
      const float part = W.w2a * (upper ? dos0 : dd0) + W.w2b * (upper ? dos1 : dd1);
      float da1 = xor32_sum(part);
      da1 = tl[5 * 64] > 0.f ? da1 : 0.f;
      const float da1h = upper ? 0.f : da1;
      const float dh = transposed_matvec(W, w1row, da1h, dpr, dpz, dghn, lane) + dhdir;
      const float hprev = tl[0 * 64], r = tl[1 * 64], zg = tl[2 * 64], n = tl[3 * 64], ghn = tl[4 * 64];
      const float dn = dh * (1.0f - zg);
      const float dzg = dh * (hprev - n);
      dhdir = dh * zg;
      const float dpn = dn * (1.0f - n * n);
      const float dr = dpn * ghn;
      dghn = dpn * r;
      dpr = dr * r * (1.0f - r);
      dpz = dzg * zg * (1.0f - zg);
      const float du0 = wave_sum(fmaf(W.wih[0][0], dpr, fmaf(W.wih[1][0], dpz, W.wih[2][0] * dpn)));
      const float du1 = wave_sum(fmaf(W.wih[0][1], dpr, fmaf(W.wih[1][1], dpz, W.wih[2][1] * dpn)));
      carry0 = c0 + du0;
      carry1 = c1 + du1;
    }

    // =============================== Adam + bookkeeping (wave 0) ===============================
    if (k == 0) {
      b1p *= 0.9;
      b2p *= 0.999;
      __builtin_amdgcn_wave_barrier();
      if (lane < 8) {
        const float g = sh.dxbuf[lane];
        am = am + (g - am) * 0.1f;
        av = av * 0.999f + 0.001f * g * g;
        const float step_size = (float)((double)a.lr / (1.0 - b1p));
        const float bc2s = (float)sqrt(1.0 - b2p);
        x = x - step_size * (am / (sqrtf(av) / bc2s + 1e-8f));
        if (loss < loss_best) xbest = x;  // post-step x vs pre-step loss (rip/agent.py:131-135)
        if (a.trace_x != nullptr) a.trace_x[((size_t)step * a.B * a.N + bn) * 8 + lane] = x;
        if (a.trace_grad != nullptr) a.trace_grad[((size_t)step * a.B * a.N + bn) * 8 + lane] = g;
      }
      if (loss < loss_best) loss_best = loss;
      if (a.trace_loss != nullptr && lane == 0) a.trace_loss[(size_t)step * a.B * a.N + bn] = loss;
    }
    __syncthreads();
  }
  if (k == 0) {
    if (a.plans != nullptr && lane < 8) a.plans[(size_t)bn * 8 + lane] = sh.ybuf[lane];
    if (a.loss_best != nullptr && lane == 0) a.loss_best[bn] = loss_best;
  }
}

// per observation: the candidate with the lowest best-loss wins (first index on ties)
// R11 — rip/agent.py:141-151 (== dim/agent.py:74-84): the [4,2] plan is the value of a piecewise-linear curve at the
// ticks 0, 10, 20, 30 (`player_future_length // T` apart); the agent returns it sampled at ticks 0..29 with z = 0,
// float64 [30,3].  scipy.interpolate.interp1d's linear rule, operation by operation: the segment of tick t is found
// with a left-sided search clipped to [1, T-1] (a tick ON a knot uses the segment that ends there), the slope is the
// float32 difference of the knots promoted to float64 and divided by the knot distance, value = slope * (t - t_lo) +
// y_lo in float64 without contraction.  Lanes 0..29 of a wave write one output row each.
constexpr int PLAN_ROWS = 30;
constexpr int PLAN_INC = 10;
__device__ __forceinline__ void interpolate_rows(const float* __restrict__ p, double* __restrict__ o, int lane) {
#pragma clang fp contract(off)  // numpy rounds the product before the sum; hipcc's default would fuse them into one fma
  if (lane >= PLAN_ROWS) return;
  int hi = (lane + PLAN_INC - 1) / PLAN_INC;  // searchsorted(knots, t, 'left')
  hi = hi < 1 ? 1 : (hi > 3 ? 3 : hi);
  const int lo = hi - 1;
  const double dt = (double)(lane - PLAN_INC * lo);
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const float ylo = p[2 * lo + d], yhi = p[2 * hi + d];
    const float diff = yhi - ylo;
    const double slope = (double)diff / (double)PLAN_INC;
    const double prod = slope * dt;
    o[lane * 3 + d] = prod + (double)ylo;
  }
  o[lane * 3 + 2] = 0.0;
}

__global__ void select_best_kernel(const float* __restrict__ plans, const float* __restrict__ loss_best, int N,
                                   float* __restrict__ plan, int32_t* __restrict__ best, double* __restrict__ interp) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float v = INFINITY;
  int idx = 0x7fffffff;
  for (int n = lane; n < N; n += 64) {
    const float l = loss_best[(size_t)b * N + n];
    if (l < v || (l == v && n < idx)) {
      v = l;
      idx = n;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const float ov = __shfl_xor(v, d, 64);
    const int oi = __shfl_xor(idx, d, 64);
    if (ov < v || (ov == v && oi < idx)) {
      v = ov;
      idx = oi;
    }
  }
  if (idx == 0x7fffffff) idx = 0;  // all-NaN guard
  if (plan != nullptr && lane < 8) plan[(size_t)b * 8 + lane] = plans[((size_t)b * N + idx) * 8 + lane];
  if (best != nullptr && lane == 0) best[b] = idx;
  if (interp != nullptr) interpolate_rows(plans + ((size_t)b * N + idx) * 8, interp + (size_t)b * PLAN_ROWS * 3, lane);
}

// R11 on its own: [B,4,2] fp32 plans -> [B,30,3] float64 (rip_interpolate_plans), one wave per plan
__global__ void interpolate_plans_kernel(const float* __restrict__ plan, int B, double* __restrict__ out) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b < B) interpolate_rows(plan + (size_t)b * 8, out + (size_t)b * PLAN_ROWS * 3, threadIdx.x & 63);
}

// Ensemble aggregation of a gathered score matrix S[K][B][N] (rip/agent.py:121-127 as coded, per plan):
// loss[b][n] = WCM: min_k(-S) | BCM: max_k(-S) | MA: mean_k(-S); best[b] = argmin_n loss (first on ties).
// One wave per observation; the arg-min over candidates is a 6-step __shfl_xor butterfly.
__global__ void aggregate_scores_kernel(const float* __restrict__ S, int K, int B, int N, int algorithm,
                                        float* __restrict__ loss_out, int32_t* __restrict__ best) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float bv = INFINITY;
  int bi = 0x7fffffff;
  for (int n = lane; n < N; n += 64) {
    float lo = -S[((size_t)0 * B + b) * N + n], acc = lo;
    for (int k = 1; k < K; ++k) {
      const float v = -S[((size_t)k * B + b) * N + n];
      acc += v;
      lo = algorithm == ALGO_WCM ? fminf(lo, v) : fmaxf(lo, v);
    }
    const float l = algorithm == ALGO_MA ? acc / (float)K : lo;
    if (loss_out != nullptr) loss_out[(size_t)b * N + n] = l;
    if (l < bv || (l == bv && n < bi)) {
      bv = l;
      bi = n;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const float ov = __shfl_xor(bv, d, 64);
    const int oi = __shfl_xor(bi, d, 64);
    if (ov < bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  if (best != nullptr && lane == 0) best[b] = bi == 0x7fffffff ? 0 : bi;
}

// ImitativeModel.forward bookkeeping (dim/model.py:124-141): the loss is the batch mean, x_best is the
// whole post-step x of the first step that reaches the running minimum; then y = F(x_best).
__global__ __launch_bounds__(64) void dim_select_kernel(const float* __restrict__ blob, const float* __restrict__ z,
                                                         const float* __restrict__ x0,
                                                         const float* __restrict__ trace_loss,
                                                         const float* __restrict__ trace_x, int B, int num_steps,
                                                         float* __restrict__ y, float* __restrict__ trace_mean) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w1 = smem;
  float* io = smem + W1_LDS;
  const int lane = threadIdx.x;
  stage_w1(w1, blob, lane, 64);
  FlowRegs W;
  load_flow_regs(W, blob, lane);
  float best = 1000.0f;
  int best_step = -1;
  for (int s = 0; s < num_steps; ++s) {
    float acc = 0.f;
    for (int i = lane; i < B; i += 64) acc += trace_loss[(size_t)s * B + i];
    const float mean = wave_sum(acc) / (float)B;
    if (trace_mean != nullptr && blockIdx.x == 0 && lane == 0) trace_mean[s] = mean;
    if (mean < best) {
      best = mean;
      best_step = s;
    }
  }
  __syncthreads();
  const float* w1row = w1 + (lane & 31) * W1_STRIDE;
  for (int row = blockIdx.x; row < B; row += gridDim.x) {
    if (lane < 8)
      io[lane] = best_step < 0 ? x0[(size_t)row * 8 + lane] : trace_x[((size_t)best_step * B + row) * 8 + lane];
    const Prefix pre = chain_prefix(W, w1row, z[(size_t)row * 64 + lane]);
    __builtin_amdgcn_wave_barrier();
    chain_forward<false>(MODE_FWD, W, w1row, pre, io, io + 8, nullptr, lane);
    __builtin_amdgcn_wave_barrier();
    if (lane < 8) y[(size_t)row * 8 + lane] = io[8 + lane];
    __builtin_amdgcn_wave_barrier();
  }
}


// ------------------------------------------------------------------------------------------
// Gradient-mode model-parallel search (BASELINE config 4: K = 8 models over 8 GPUs).
// One Adam step of rip/agent.py:102-135 is cut at the point where the ensemble is reduced:
//   mp_local_kernel   (rank r, its models k): y = F_0(x; z_0) redundantly, then inverse_k(y) and its adjoint ->
//                     out[k][b][n] = (q_k, dq_k/dy[8]).  The rank that owns model 0 reports q_0 through the
//                     self-inverse shortcut (inverse_0(F_0(x)) == x) with a zero gradient row: that gradient is
//                     folded into the F_0 adjoint (chain_backward's w0 terms), as in search_kernel.
//   -- ONE all-gather of the [K_local,B,N,9] blocks (torch.distributed / RCCL) --
//   mp_update_kernel  (every rank, redundantly): aggregate over K (rip/agent.py:121-127), dLoss/dy, F_0 adjoint,
//                     Adam and the loss/x_best bookkeeping on rank-replicated state.  Same device functions and the
//                     same operation order as search_kernel, so one rank with all K models reproduces it.
// One wave per (observation, candidate[, local model]); this mode is bound by the collective's latency.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mp_local_kernel(MpArgs a, const float* __restrict__ x, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w1f = smem;                 // W1 of the forward model
  float* w1l = w1f + W1_LDS;         // W1 of the local model
  float* tape = w1l + W1_LDS;        // TAPE
  float* io = tape + TAPE;           // x[8], y[8], g[8]
  const int lane = threadIdx.x;
  const int bn = blockIdx.x, b = bn / a.N;
  const int kl = blockIdx.y;
  const float* blob_f = a.flow_w + (size_t)a.k_fwd * FW_SIZE;
  const float* blob_l = a.flow_w + (size_t)(a.k_begin + kl) * FW_SIZE;
  stage_w1(w1f, blob_f, lane, 64);
  stage_w1(w1l, blob_l, lane, 64);
  if (lane < 8) io[lane] = x[(size_t)bn * 8 + lane];
  FlowRegs W;
  load_flow_regs(W, blob_f, lane);
  __syncthreads();
  const float* w1row_f = w1f + (lane & 31) * W1_STRIDE;
  const Prefix pre_f = chain_prefix(W, w1row_f, a.z_fwd[(size_t)b * 64 + lane]);
  const ChainOut of = chain_forward<false>(MODE_FWD, W, w1row_f, pre_f, io, io + 8, nullptr, lane);
  __builtin_amdgcn_wave_barrier();
  float* o9 = out + (((size_t)kl * a.B + b) * a.N + (bn - b * a.N)) * 9;
  if (a.first_is_fwd && kl == 0) {
    if (lane == 0) o9[0] = (-0.5f * of.sq - 4.0f * LOG_2PI) - of.lad;
    if (lane >= 1 && lane < 9) o9[lane] = 0.f;
    return;
  }
  load_flow_regs(W, blob_l, lane);
  const float* w1row_l = w1l + (lane & 31) * W1_STRIDE;
  const Prefix pre = chain_prefix(W, w1row_l, a.z[((size_t)kl * a.B + b) * 64 + lane]);
  const ChainOut oi = chain_forward<true>(MODE_INV, W, w1row_l, pre, io + 8, nullptr, tape, lane);
  __builtin_amdgcn_wave_barrier();
  chain_backward(MODE_INV, W, w1row_l, tape, nullptr, io + 16, lane);
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) o9[0] = (-0.5f * oi.sq - 4.0f * LOG_2PI) - oi.lad;  // rip/agent.py:111-112
  if (lane >= 1 && lane < 9) o9[lane] = io[16 + lane - 1];
}

__global__ __launch_bounds__(64) void mp_update_kernel(MpArgs a, const float* __restrict__ gathered,
                                                        float* __restrict__ x, float* __restrict__ am_,
                                                        float* __restrict__ av_, float* __restrict__ x_best,
                                                        float* __restrict__ loss_best, float* __restrict__ grad_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w1f = smem;
  float* tape = w1f + W1_LDS;
  float* io = tape + TAPE;  // x[8], y[8], gsum[8], dx[8]
  const int lane = threadIdx.x;
  const int bn = blockIdx.x, b = bn / a.N, n = bn - b * a.N;
  const float* blob_f = a.flow_w + (size_t)a.k_fwd * FW_SIZE;
  stage_w1(w1f, blob_f, lane, 64);
  float xv = 0.f;
  if (lane < 8) {
    xv = x[(size_t)bn * 8 + lane];
    io[lane] = xv;
  }
  FlowRegs W;
  load_flow_regs(W, blob_f, lane);
  __syncthreads();
  const float* w1row = w1f + (lane & 31) * W1_STRIDE;
  const Prefix pre = chain_prefix(W, w1row, a.z_fwd[(size_t)b * 64 + lane]);
  chain_forward<true>(MODE_FWD, W, w1row, pre, io, io + 8, tape, lane);
  __builtin_amdgcn_wave_barrier();
  float gl = 0.f, gg0 = 0.f, gg1 = 0.f;
  if (a.goal != nullptr) gl = goal_ll(a.goal + (size_t)b * a.G * 2, a.G, a.epsilon, io[8 + 6], io[8 + 7], &gg0, &gg1);
  // ---- aggregate over the K models (rip/agent.py:121-127, as coded) ----
  const size_t kstride = (size_t)a.B * a.N * 9;
  const float* g9 = gathered + ((size_t)b * a.N + n) * 9;
  int ksel = 0;
  float qsel = g9[0], qmean = g9[0];
  for (int k = 1; k < a.K; ++k) {
    const float qk = g9[k * kstride];
    qmean += qk;
    const bool take = a.algorithm == ALGO_WCM ? (qk > qsel) : (qk < qsel);
    if (take) {
      qsel = qk;
      ksel = k;
    }
  }
  qmean /= (float)a.K;
  const bool mean_mode = a.algorithm == ALGO_MA;
  const float loss = -((mean_mode ? qmean : qsel) + gl);
  if (lane < 8) {
    float g = 0.f;
    if (mean_mode) {
      for (int k = 1; k < a.K; ++k) g += g9[k * kstride + 1 + lane];
      g /= (float)a.K;
    } else if (ksel != 0) {
      g = g9[ksel * kstride + 1 + lane];
    }
    if (lane == 6) g += gg0;
    if (lane == 7) g += gg1;
    io[16 + lane] = -g;
  }
  __builtin_amdgcn_wave_barrier();
  const float w0 = mean_mode ? 1.0f / (float)a.K : (ksel == 0 ? 1.0f : 0.0f);
  chain_backward(MODE_FWD, W, w1row, tape, io + 16, io + 24, lane, w0);
  __builtin_amdgcn_wave_barrier();
  // ---- Adam (torch.optim.Adam defaults) + bookkeeping ----
  double b1p = 1.0, b2p = 1.0;
  for (int i = 0; i <= a.step; ++i) {
    b1p *= 0.9;
    b2p *= 0.999;
  }
  const float lb = loss_best[bn];
  if (lane < 8) {
    const float g = io[24 + lane];
    float am = am_[(size_t)bn * 8 + lane], av = av_[(size_t)bn * 8 + lane];
    am = am + (g - am) * 0.1f;
    av = av * 0.999f + 0.001f * g * g;
    const float step_size = (float)((double)a.lr / (1.0 - b1p));
    const float bc2s = (float)sqrt(1.0 - b2p);
    xv = xv - step_size * (am / (sqrtf(av) / bc2s + 1e-8f));
    am_[(size_t)bn * 8 + lane] = am;
    av_[(size_t)bn * 8 + lane] = av;
    x[(size_t)bn * 8 + lane] = xv;
    if (loss < lb) x_best[(size_t)bn * 8 + lane] = xv;  // post-step x vs pre-step loss (rip/agent.py:131-135)
    if (grad_out != nullptr) grad_out[(size_t)bn * 8 + lane] = g;
  }
  if (lane == 0 && loss < lb) loss_best[bn] = loss;
}


// ------------------------------------------------------------------------------------------
// DIM training step, flow part (dim/train.py:198-204): teacher-forced inverse of one (y, z) row per
// wave (sequence.py:153-216) with ALL four steps taped (h_0 = z is a function of the encoder here), its adjoint with
// cotangent `cot` (= -1/B: loss = -mean(log_prob - logabsdet)) into dz, and per (row, step) the vectors whose outer
// products are the weight gradients:  dW_ih = dgi^T u,  dW_hh = dgh^T hprev,  dW1 = da1^T h,  dW2 = do^T relu(a1)
// (formed afterwards as GEMMs over the B*T records; the bias gradients are their column sums).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flow_train_kernel(const float* __restrict__ blob, const float* __restrict__ z,
                                                          const float* __restrict__ y, int B, float cot,
                                                          float* __restrict__ q_rows, float* __restrict__ dz,
                                                          float* __restrict__ rec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w1 = smem;                                    // W1_LDS
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  float* tp = smem + W1_LDS + wave * (T * 7 * 64 + T * 8);  // per wave: [t][7][64] lane tape + [t][8] uniforms
  float* tu = tp + T * 7 * 64;
  stage_w1(w1, blob, tid, blockDim.x);
  FlowRegs W;
  load_flow_regs(W, blob, lane);
  __syncthreads();
  const float* w1row = w1 + (lane & 31) * W1_STRIDE;
  const bool upper = lane >= 32;
  for (int row = blockIdx.x * nw + wave; row < B; row += gridDim.x * nw) {
    const float* yr = y + (size_t)row * 8;
    float h = z[(size_t)row * 64 + lane];
    float u0 = 0.f, u1 = 0.f, sq = 0.f, lad = 0.f;
    // ---------------- forward ----------------
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
      float gh[3], a1 = 0.f;
      matvec<true, false>(W, w1row, h, gh, a1);
      const float gir = fmaf(W.wih[0][1], u1, fmaf(W.wih[0][0], u0, W.bih[0]));
      const float giz = fmaf(W.wih[1][1], u1, fmaf(W.wih[1][0], u0, W.bih[1]));
      const float gin = fmaf(W.wih[2][1], u1, fmaf(W.wih[2][0], u0, W.bih[2]));
      const float r = sigmoidf_(gir + gh[0]);
      const float zg = sigmoidf_(giz + gh[1]);
      const float n = tanhf_(fmaf(r, gh[2], gin));
      const float hn = fmaf(zg, h - n, n);
      float* tl = tp + t * 7 * 64 + lane;
      tl[0 * 64] = h;
      tl[1 * 64] = r;
      tl[2 * 64] = zg;
      tl[3 * 64] = n;
      tl[4 * 64] = gh[2];
      h = hn;
      float ghx[3];
      matvec<false, true>(W, w1row, h, ghx, a1);
      tl[5 * 64] = a1;
      tl[6 * 64] = h;
      float o0, o1, o2, o3;
      head_finish(W, a1, o0, o1, o2, o3);
      const float s0 = softplusf_(o2) + 1e-3f, s1 = softplusf_(o3) + 1e-3f;  // sequence.py:193
      const float y0 = yr[2 * t], y1 = yr[2 * t + 1];
      const float x0 = (y0 - (u0 + o0)) * rcpf_(s0), x1 = (y1 - (u1 + o1)) * rcpf_(s1);  // :196
      sq = fmaf(x0, x0, fmaf(x1, x1, sq));
      lad += __logf(s0 * s1);
      if (lane == 0) {
        float* q8 = tu + t * 8;
        q8[0] = x0;
        q8[1] = x1;
        q8[2] = s0;
        q8[3] = s1;
        q8[4] = softplus_gradf_(o2);
        q8[5] = softplus_gradf_(o3);
        q8[6] = u0;
        q8[7] = u1;
      }
      u0 = y0;  // teacher forcing, :201
      u1 = y1;
    }
    if (lane == 0) q_rows[row] = (-0.5f * sq - 4.0f * LOG_2PI) - lad;
    __builtin_amdgcn_wave_barrier();
    // ---------------- adjoint ----------------
    float dhdir = 0.f, dpr = 0.f, dpz = 0.f, dghn = 0.f;
#pragma unroll 1
    for (int t = T - 1; t >= 0; --t) {
      const float* q8 = tu + t * 8;
      const float x0 = q8[0], x1 = q8[1], s0 = q8[2], s1 = q8[3], sg0 = q8[4], sg1 = q8[5];
      const float i0 = rcpf_(s0), i1 = rcpf_(s1);
      // q = -0.5 |x|^2 - sum log s:  dq/ddloc = x / s,  dq/ds = (x^2 - 1) / s  (then through softplus)
      const float dd0 = cot * x0 * i0, dd1 = cot * x1 * i1;
      const float dos0 = cot * (x0 * x0 - 1.0f) * i0 * sg0, dos1 = cot * (x1 * x1 - 1.0f) * i1 * sg1;
      const float* tl = tp + t * 7 * 64 + lane;
      const float a1 = tl[5 * 64];
      const float part = W.w2a * (upper ? dos0 : dd0) + W.w2b * (upper ? dos1 : dd1);
      float da1 = xor32_sum(part);
      da1 = a1 > 0.f ? da1 : 0.f;
      const float da1h = upper ? 0.f : da1;  // rows of W1 are duplicated in both halves
      const float dh = transposed_matvec(W, w1row, da1h, dpr, dpz, dghn, lane) + dhdir;
      const float hprev = tl[0 * 64], r = tl[1 * 64], zg = tl[2 * 64], n = tl[3 * 64], ghn = tl[4 * 64];
      const float dn = dh * (1.0f - zg);
      const float dzg = dh * (hprev - n);
      dhdir = dh * zg;
      const float dpn = dn * (1.0f - n * n);
      const float dr = dpn * ghn;
      dghn = dpn * r;
      dpr = dr * r * (1.0f - r);
      dpz = dzg * zg * (1.0f - zg);
      float* rc = rec + ((size_t)row * T + t) * FLOW_TRAIN_REC;
      rc[0 * 64 + lane] = dpr;   // dgi = (d pre_r, d pre_z, d pre_n)
      rc[1 * 64 + lane] = dpz;
      rc[2 * 64 + lane] = dpn;
      rc[192 + 0 * 64 + lane] = dpr;  // dgh = (d pre_r, d pre_z, d gh_n)
      rc[192 + 1 * 64 + lane] = dpz;
      rc[192 + 2 * 64 + lane] = dghn;
      rc[384 + lane] = hprev;
      if (lane < 2) rc[448 + lane] = q8[6 + lane];
      if (lane < 32) {
        rc[450 + lane] = da1;
        rc[550 + lane] = fmaxf(a1, 0.f);
      }
      rc[482 + lane] = tl[6 * 64];
      if (lane < 4) rc[546 + lane] = lane == 0 ? dd0 : (lane == 1 ? dd1 : (lane == 2 ? dos0 : dos1));
    }
    dz[(size_t)row * 64 + lane] = transposed_matvec(W, w1row, 0.f, dpr, dpz, dghn, lane) + dhdir;
    __builtin_amdgcn_wave_barrier();
  }
}

// reference tensor layout -> the lane-major FW_* blob of flow.h (what fold_and_pack does on the host for inference)
__global__ void flow_relayout_kernel(const float* __restrict__ wih, const float* __restrict__ whh,
                                     const float* __restrict__ bih, const float* __restrict__ bhh,
                                     const float* __restrict__ w1, const float* __restrict__ b1,
                                     const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ blob) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= FW_SIZE) return;
  float v = 0.f;
  if (i < FW_WIH) {  // [(g*16+i4)][64 lanes][4]
    const int q = i & 3, j = (i >> 2) & 63, c = i >> 8, g = c / 16, i4 = c % 16;
    v = whh[(size_t)(g * 64 + j) * 64 + 4 * i4 + q];
  } else if (i < FW_BIH) {
    const int e = i - FW_WIH, j = e & 63, gd = e >> 6, g = gd >> 1, d = gd & 1;
    v = wih[(g * 64 + j) * 2 + d];
  } else if (i < FW_BHH) {
    v = bih[i - FW_BIH];
  } else if (i < FW_B1) {
    v = bhh[i - FW_BHH];
  } else if (i < FW_W2) {
    v = b1[(i - FW_B1) & 31];
  } else if (i < FW_B2) {
    const int e = i - FW_W2, j = e & 63, qq = e >> 6;
    v = w2[(2 * (j >> 5) + qq) * 32 + (j & 31)];
  } else if (i < FW_W1) {
    v = b2[i - FW_B2];
  } else {
    v = w1[i - FW_W1];
  }
  blob[i] = v;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------

// The search kernels need more dynamic LDS than the default limit: raise it ONCE per (kernel, device) to the CU's
// 160 KiB, so that later launches are pure enqueues (no attribute call between the kernels of a captured graph).
hipError_t allow_lds(const void* fn) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({fn, dev})) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) done.insert({fn, dev});
  return e;
}

// compute units of the current device, queried once per device (hipGetDeviceProperties costs ~100 us of host time:
// not something to pay on every eager launch)
int device_cu_count() {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  cus[dev] = n;
  return n;
}

// XCDs the workgroups of a launch are dealt over round-robin (workgroup L -> XCD L % n), or 0 when that is not known:
// an MI355X in SPX mode is 8 XCDs x 32 CUs = 256 CUs; any other CU count (CPX / partitioned modes, another part) turns
// the XCD-aware work lists OFF (results never depend on placement, only L2 reuse does).
int device_xcd_count() { return device_cu_count() == 256 ? 8 : 0; }

static int rows_grid(int rows) {
  int g = (rows + 3) / 4;
  return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}

hipError_t launch_flow_forward(const float* blob, const float* x, const float* z, int N, int z_rows, float* y,
                               float* lad, hipStream_t s) {
  const size_t lds = (W1_LDS + 4 * 16) * sizeof(float);
  hipLaunchKernelGGL(flow_rows_kernel, dim3(rows_grid(N)), dim3(256), lds, s, MODE_FWD, blob, x, z, N, z_rows, y,
                     (float*)nullptr, lad);
  return hipGetLastError();
}

hipError_t launch_flow_inverse(const float* blob, const float* y, const float* z, int N, int z_rows, float* x,
                               float* logp, float* lad, hipStream_t s) {
  const size_t lds = (W1_LDS + 4 * 16) * sizeof(float);
  hipLaunchKernelGGL(flow_rows_kernel, dim3(rows_grid(N)), dim3(256), lds, s, MODE_INV, blob, y, z, N, z_rows, x,
                     logp, lad);
  return hipGetLastError();
}

hipError_t launch_goal_rows(const float* y, const float* goal, int N, int goal_rows, int G, float eps, float* rows,
                            hipStream_t s) {
  hipLaunchKernelGGL(goal_rows_kernel, dim3((N + 255) / 256), dim3(256), 0, s, y, goal, N, goal_rows, G, eps, rows);
  return hipGetLastError();
}

hipError_t launch_score(const float* flow_w, int k0, int K, const float* z, const float* y, const float* goal, int B,
                        int N, int G, float eps, float* S, hipStream_t s) {
  const size_t lds = (W1_LDS + 4 * 8) * sizeof(float);
  hipLaunchKernelGGL(score_kernel, dim3(rows_grid(B * N), K), dim3(256), lds, s, flow_w, k0, z, y, goal, B, N, G, eps,
                     S);
  return hipGetLastError();
}

size_t search_lds_bytes(int K) {
  return (size_t)(K * W1_LDS + (1 + K) * TAPE) * sizeof(float) + sizeof(SearchShared);
}

template <int NW>
static hipError_t launch_pipe(const SearchArgs& a, hipStream_t s) {
  const size_t lds = (size_t)(NW * W1_LDS + NW * TAPE) * sizeof(float) + sizeof(SearchShared);
  hipError_t e = allow_lds(reinterpret_cast<const void*>(search_pipe_kernel<NW>));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(search_pipe_kernel<NW>, dim3(a.B * a.N), dim3(NW * 64), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_search(const SearchArgs& a, hipStream_t s) {
  if (a.K == 2) return launch_pipe<2>(a, s);
  if (a.K == 3) return launch_pipe<3>(a, s);
  if (a.K == 4) return launch_pipe<4>(a, s);
  const size_t lds = search_lds_bytes(a.K);
  const dim3 grid(a.B * a.N);
  const int nw = a.K >= 4 ? 4 : a.K;
  hipError_t e = hipSuccess;
  switch (nw) {
    case 1:
      e = allow_lds(reinterpret_cast<const void*>(search_kernel<1>));
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(search_kernel<1>, grid, dim3(64), lds, s, a);
      break;
    case 2:
      e = allow_lds(reinterpret_cast<const void*>(search_kernel<2>));
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(search_kernel<2>, grid, dim3(128), lds, s, a);
      break;
    case 3:
      e = allow_lds(reinterpret_cast<const void*>(search_kernel<3>));
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(search_kernel<3>, grid, dim3(192), lds, s, a);
      break;
    default:
      e = allow_lds(reinterpret_cast<const void*>(search_kernel<4>));
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(search_kernel<4>, grid, dim3(256), lds, s, a);
      break;
  }
  return hipGetLastError();
}

hipError_t launch_select_best(const float* plans, const float* loss_best, int B, int N, float* plan, int32_t* best,
                              double* interp, hipStream_t s) {
  hipLaunchKernelGGL(select_best_kernel, dim3(B), dim3(64), 0, s, plans, loss_best, N, plan, best, interp);
  return hipGetLastError();
}

hipError_t launch_interpolate_plans(const float* plan, int B, double* out, hipStream_t s) {
  hipLaunchKernelGGL(interpolate_plans_kernel, dim3((B + 3) / 4), dim3(256), 0, s, plan, B, out);
  return hipGetLastError();
}

hipError_t launch_aggregate_scores(const float* S, int K, int B, int N, int algorithm, float* loss, int32_t* best,
                                   hipStream_t s) {
  hipLaunchKernelGGL(aggregate_scores_kernel, dim3(B), dim3(64), 0, s, S, K, B, N, algorithm, loss, best);
  return hipGetLastError();
}

hipError_t launch_dim_select(const float* blob, const float* z, const float* x0, const float* trace_loss,
                             const float* trace_x, int B, int num_steps, float* y, float* trace_mean, hipStream_t s) {
  const size_t lds = (W1_LDS + 16) * sizeof(float);
  int g = B < 256 ? B : 256;
  hipLaunchKernelGGL(dim_select_kernel, dim3(g), dim3(64), lds, s, blob, z, x0, trace_loss, trace_x, B, num_steps, y,
                     trace_mean);
  return hipGetLastError();
}

hipError_t launch_mp_local(const MpArgs& a, const float* x, float* out, hipStream_t s) {
  const size_t lds = (size_t)(2 * W1_LDS + TAPE + 32) * sizeof(float);
  hipLaunchKernelGGL(mp_local_kernel, dim3(a.B * a.N, a.k_count), dim3(64), lds, s, a, x, out);
  return hipGetLastError();
}

hipError_t launch_mp_update(const MpArgs& a, const float* gathered, float* x, float* m, float* v, float* x_best,
                            float* loss_best, float* grad_out, hipStream_t s) {
  const size_t lds = (size_t)(W1_LDS + TAPE + 32) * sizeof(float);
  hipLaunchKernelGGL(mp_update_kernel, dim3(a.B * a.N), dim3(64), lds, s, a, gathered, x, m, v, x_best, loss_best,
                     grad_out);
  return hipGetLastError();
}

hipError_t launch_flow_train(const float* wih, const float* whh, const float* bih, const float* bhh, const float* w1,
                             const float* b1, const float* w2, const float* b2, const float* z, const float* y, int B,
                             float* q_rows, float* dz, float* records, hipStream_t s) {
  // the lane-major blob lives in the first FW_SIZE floats after the records of the LAST row (the caller sizes the
  // record buffer for max_batch rows + this blob: see trainer_create)
  float* blob = records + (size_t)B * FLOW_TRAIN_ROW_FLOATS;
  hipLaunchKernelGGL(flow_relayout_kernel, dim3((FW_SIZE + 255) / 256), dim3(256), 0, s, wih, whh, bih, bhh, w1, b1, w2, b2,
                     blob);
  const size_t lds = (size_t)(W1_LDS + 4 * (T * 7 * 64 + T * 8)) * sizeof(float);
  int g = (B + 3) / 4;
  g = g < 1 ? 1 : (g > 2048 ? 2048 : g);
  hipLaunchKernelGGL(flow_train_kernel, dim3(g), dim3(256), lds, s, blob, z, y, B, -1.0f / (float)B, q_rows, dz, records);
  return hipGetLastError();
}

}  // namespace rip
