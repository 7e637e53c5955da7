// Split-f16 plan search for gfx950: the throughput kernel of RIPAgent.__call__ (rip/agent.py:78-137), round 3.
//
// The decomposition is flow_phase.hip's (one wave = one block of 16 candidates running ALL K models in sequence, 8 such
// waves per workgroup sharing the current model's operands in LDS, adjoint tape in global memory); what changes is the
// arithmetic of the contractions.  flow_phase.hip keeps them on v_mfma_f32_16x16x4_f32 (fp32 operands, the fp32 VECTOR
// rate: 5406 MFMAs x 32 cycles per block and Adam step, 0.67 of that pipe's peak — its floor).  Here every GRU / head
// product runs on v_mfma_f32_16x16x32_f16 (16 cycles for 8x the K) with BOTH operands carried as two binary16 terms,
//     x ~= hi + lo' 2^-11,     W x ~= Whi xhi + 2^-11 (Whi xlo' + Wlo' xhi),      fp32 accumulation,
// i.e. three f16 MFMAs per 32-deep K block instead of eight fp32 ones per 32 (flow_split_dev.h): 5.3x fewer matrix-
// pipe cycles at 22 instead of 24 significant operand bits.  The k-steps that multiply the (unbounded) waypoints
// y_{t-1}, the 4-wide head output and its transpose stay on the fp32 MFMA.  Operand rows are the same size as before
// (two halves per weight), so the LDS budget — 63 KB forward + 57 KB transposed rows + 1.5 KB W_ih^T table + 4 KB
// per wave — and the DMA / barrier structure are unchanged.
// Parity: gated by the teacher-forced 1e-4 tests and the G6 traces (tests/test_gpu_parity.py), like every search kernel.
#include <cstdlib>

#include "flow.h"
#include "flow_math.h"
#include "flow_split_dev.h"

namespace rip {

namespace {

using namespace split;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int WPB_MAX = 8;       // (scratch is padded to 8 blocks: the paired shape, flow_pair.hip, shares the layout)
constexpr int PARK_F4 = 3 * 64;  // per 16-candidate block: kept in the scratch layout (round 3-5's 8-wave build parked its Adam state here)
constexpr int F_ROWS = MHF_ROWS;
constexpr int T_ROWS = MHT_ROWS;
#ifndef RIP_PAIR_COST
#define RIP_PAIR_COST 1.27  // time of a round of paired workgroups relative to the 4-wave shape's (measured)
#endif
#ifndef RIP_REGTAPE
#define RIP_REGTAPE 1  // 4- / 2-wave workgroups keep the inverse passes' whole tape in registers
#endif

template <int WPB>
struct PShared {
  uint4 fbuf[F_ROWS * 64];       // forward operand rows of the current model
  uint4 tbuf[T_ROWS * 64];       // transposed operand rows of the current model
  uint4 wihc[MH_TABLE_F4];       // its W_ih^T table: entry ((kb * 2 + term) * 8 + q * 2 + parity)
  float io[WPB][CB][8];          // per wave: x in, y out (in place)
  float gy[WPB][CB][8];          // per wave: dLoss/dy handed to the F_0 adjoint
  float stape[WPB][2][T][6][CB]; // per wave: per-candidate scalars of the F_0 pass [0] and of the current inverse [1]
};

// ---- operand staging: direct global -> LDS DMA, one 1 KB lane-major row per wave instruction ----
template <int WPB>
__device__ __forceinline__ void dma_rows(const uint4* __restrict__ src, uint4* dst, int rows, int wave, int lane) {
  for (int r = wave; r < rows; r += WPB)
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + r * 64 + lane), (lds_ptr_t)(dst + r * 64), 16, 0, 0);
}
template <int WPB>
__device__ __forceinline__ void load_fbuf(PShared<WPB>& sh, const uint32_t* __restrict__ mhk, int wave, int lane) {
  // rows 48, 49, 51 (fp32 k-steps of r, z, gh_n) and 60..62 (fp32 b1 / W2 / b2) belong to round 5's forward step
  // (flow_pair.hip still stages them): this kernel reads row 50 (gi_n, the F_0 adjoint's recompute) and nothing else of them
  const uint4* src = reinterpret_cast<const uint4*>(mhk);
  dma_rows<WPB>(src, sh.fbuf, 48, wave, lane);
  if (wave == 0) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + 50 * 64 + lane), (lds_ptr_t)(sh.fbuf + 50 * 64), 16, 0, 0);
  dma_rows<WPB>(src + 52 * 64, sh.fbuf + 52 * 64, 8, wave, lane);
  dma_rows<WPB>(src + MHF_KS * 64, sh.fbuf + MHF_KS * 64, MHF_ROWS - MHF_KS, wave, lane);
}
template <int WPB>
__device__ __forceinline__ void load_tbuf(PShared<WPB>& sh, const uint32_t* __restrict__ mhk, int wave, int lane, int tid) {
  static_assert(WPB * 64 >= MH_TABLE_F4, "the W_ih^T table is copied by 96 threads");
  const uint4* src = reinterpret_cast<const uint4*>(mhk) + F_ROWS * 64;
  dma_rows<WPB>(src, sh.tbuf, T_ROWS, wave, lane);
  if (tid < MH_TABLE_F4) sh.wihc[tid] = src[T_ROWS * 64 + tid];
}

// prefix of every (model, observation): step 0 from h_0 = z_k, y_0 = 0 is candidate independent.  One wave each, with
// the operands read straight from global memory (L2).  (h_0 = z is the merger's ReLU output, O(1): split unscaled.)
__global__ __launch_bounds__(64) void split_prefix_kernel(SearchArgs a, const uint32_t* __restrict__ mh_all,
                                                          float* __restrict__ pre_out) {
  const int lane = threadIdx.x, q = lane >> 4;
  const int b = blockIdx.x, k = blockIdx.y;
  const uint4* wl = reinterpret_cast<const uint4*>(mh_all + (size_t)(a.k0 + k) * MH_SIZE) + lane;
  float H[16];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) H[u * 4 + r] = a.z[((size_t)k * a.B + b) * 64 + 16 * u + 4 * q + r];
  if (a.range_flag != nullptr) {  // operand-range guard: NaN counts as out of range
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) m = fmaxf(m, H[i] == H[i] ? fabsf(H[i]) : SPLIT_Z_LIMIT);
    if (__any(m >= SPLIT_Z_LIMIT) && lane == 0) atomicOr(a.range_flag, 1u);
  }
  float o[4];
  BSplit hs;
  split16(H, hs);
  fwd_step<SAVE_NONE>(wl, H, hs, 0.f, 0.f, q, (unsigned)lane, nullptr, nullptr, o);
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

// WPB: waves per workgroup = 16-candidate blocks per workgroup (one workgroup per CU holds the operand buffers): 4 = one
// wave per SIMD (full launches), 2 for launches that would otherwise leave CUs idle.  Rounds 3-5 also built WPB = 8 (two
// waves per SIMD at 256 registers each: the tape in global memory, the Adam state parked there): never faster (2.66 vs
// 2.50 ms, DESIGN_HISTORY), never selected by the cost model, and retired in round 6 — its LDS no longer fits beside the
// forward step's new operand rows; the two-waves-per-SIMD experiment of round 6 is the paired shape (flow_pair.hip).
template <bool TRACE, int WPB>
__global__ __launch_bounds__(WPB * 64) void search_split_kernel(SearchArgs a, const uint32_t* __restrict__ mh_all,
                                                                const float* __restrict__ pre_all,
                                                                float4* __restrict__ tape_all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  PShared<WPB>& sh = *reinterpret_cast<PShared<WPB>*>(smem_raw);
  // operand-range guard: some |z| of this launch is beyond what the unscaled binary16 split carries -> the fp32-MFMA
  // kernel behind this launch does the search (the whole launch: one word, read by every workgroup, uniform)
  if (a.range_flag != nullptr && __builtin_nontemporal_load(a.range_flag) != 0u) return;
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
  const uint32_t* mh0 = mh_all + (size_t)a.k0 * MH_SIZE;

  float (*io)[8] = sh.io[wave];
  float (*gy)[8] = sh.gy[wave];
  float (*stF)[6][CB] = sh.stape[wave][0];
  float (*stI)[6][CB] = sh.stape[wave][1];
  const uint4* wl = sh.fbuf + lane;
  const uint4* tw = sh.tbuf + lane;
  const uint4* wq4 = sh.wihc + q * 2 + (c & 1);
  // wave-uniform tape bases (scalar registers): lanes add their own 16-byte column at each access
  float4* tapeF = tape_all + ((size_t)item * 2 + (RIP_ABL == 4 ? 1 : 0)) * TAPE_SLOT_F4;  // ABL 4: aliased tapes
  float4* tapeI = tape_all + ((size_t)item * 2 + 1) * TAPE_SLOT_F4;
  static_assert(WPB <= 4, "one wave per SIMD: the inverse passes' tape lives in registers");

  // Adam state: lane (c, q) owns latent coordinates 2q, 2q+1 of candidate c
  float xv0 = a.x0[row * 8 + 2 * q], xv1 = a.x0[row * 8 + 2 * q + 1];
  // Adjoints nobody needs (WCM / BCM back-propagate through ONE member per candidate, but a block runs an inverse pass's
  // adjoint as soon as one of its 16 candidates needs it: 2.4-2.7 per block and step, `rip_search_stats`): two remedies
  // were built and measured, neither pays at this launch size, neither ships.  Rounds 3 / 4: regrouping a WORKGROUP's 64
  // candidates by selected member between steps — 1.58 adjoints per block, 2.82 vs 2.73 ms: the waves of a workgroup
  // walk the model phases together and among 64 candidates of one observation some block needs every adjoint.  Round 5:
  // one launch per Adam step with ALL candidates of the call binned by selected member in between (1.31 adjoints per
  // block, 1.74 instead of 2.96 adjoint phases per workgroup) — 2.80-2.95 vs 2.56 ms: per block and step the gathers
  // of prefix / goal rows of 16 different observations, the state round trip and the binning atomics cost 14.5 k cycles,
  // the adjoints saved 16.6 k, and every launch ends in a tail (profiles/r5/binned_search_v1.txt).
  const int orig = n0 + c;
  float am0 = 0.f, am1 = 0.f, av0 = 0.f, av1 = 0.f;
  float xb0 = xv0, xb1 = xv1, lbest = 1000.0f;
  double b1p = 1.0, b2p = 1.0;
  const bool mean_mode = a.algorithm == ALGO_MA;
  const float inv_k = 1.0f / (float)K;

  load_fbuf(sh, mh0, wave, lane);
  load_tbuf(sh, K > 1 ? mh0 + MH_SIZE : mh0, wave, lane, tid);

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
    float gsa = 0.f, gsb = 0.f;  // sum_k w_k dq_k/dy, coordinates 2q and 2q+1 of this lane's candidate
    {
      const Prefix16 pre = load_prefix(pre_all + ((size_t)0 * a.B + b) * PRE_FLOATS, q);
      TK_START();
      const PassOut po = pass_forward<MODE_FWD, false, false>(wl, pre, io, stF, tapeF, nullptr, c, q, (unsigned)lane);
      TK_STOP(1);
      __builtin_amdgcn_wave_barrier();
      if (final_pass) break;
      if (goal != nullptr) gl = goal_ll(goal, a.G, a.epsilon, io[c][6], io[c][7], &gg0, &gg1);
      q_sel = (-0.5f * po.sq - 4.0f * LOG_2PI) - po.lad;  // model 0's posterior through the self-inverse shortcut
      if (TRACE && a.trace_post != nullptr && q == 0 && active)
        a.trace_post[(((size_t)step * K + 0) * a.B + b) * a.N + n0 + c] = q_sel + gl;
    }
    float q_sum = q_sel;
    // ================= models 1..K-1: inverse, adjoint, streaming aggregation =================
#pragma unroll 1
    for (int k = 1; k < K; ++k) {
      constexpr bool REGTAPE = RIP_REGTAPE != 0;
      // (Requesting model k + 1's forward operands under model k's adjoint — the adjoint of a register-tape inverse pass
      // never reads the F-buf — was built twice.  Rounds 3 / 4: slower, 2.81 vs 2.77 ms; round 5 found why: the compiler
      // cannot tell an LDS-DMA's destination from any other LDS location and puts s_waitcnt vmcnt(0) in front of the
      // adjoint's first operand read, i.e. the whole transfer was waited for there.  With the DMA issued from inline
      // assembly (invisible to that pass, explicit waits at the barriers) the adjoint runs under the transfer — and
      // the launch takes exactly as long, 2.50 ms: the 4 k cycles per phase at the second barrier are not DMA latency
      // any more than before, they are the waves waiting for the slowest wave's data-dependent adjoint, wherever the
      // barrier stands (profiles/r5/overlap_v1.txt).  Not kept.)
      const uint32_t* mhk = mh_all + (size_t)(a.k0 + k) * MH_SIZE;
      {
        TK_START();
        __syncthreads();  // every wave is done with the F-buf (F_0 or inverse_{k-1}) and the T-buf (adjoint_{k-1})
        TK_STOP(2);
        TK_START();
        if (RIP_ABL != 2) {
          load_fbuf(sh, mhk, wave, lane);
          if (k > 1) load_tbuf(sh, mhk, wave, lane, tid);  // (model 1's T-buf was requested under F_0)
        }
        __syncthreads();  // operands of model k landed
        TK_STOP(3);
      }
      TK_START();
      const Prefix16 pre = load_prefix(pre_all + ((size_t)k * a.B + b) * PRE_FLOATS, q);
      StepTape last[3];
      const PassOut po = pass_forward<MODE_INV, REGTAPE, false>(wl, pre, io, stI, tapeI, last, c, q, (unsigned)lane);
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
        pass_backward<MODE_INV, REGTAPE, (WPB <= 4)>(tw, wq4, wl, io, nullptr, stI, tapeI, last, pre_all + ((size_t)k * a.B + b) * PRE_FLOATS, c, q,
                                res, 0.f);
        if (a.stats != nullptr && lane == 0 && active) atomicAdd(a.stats, 1ull);  // executed inverse-pass adjoints (bench.py)
        TK_STOP(5);
        const float ra = q == 0 ? res[0] : q == 1 ? res[2] : q == 2 ? res[4] : res[6];
        const float rb = q == 0 ? res[1] : q == 1 ? res[3] : q == 2 ? res[5] : res[7];
        if (mean_mode) {
          gsa += inv_k * ra;
          gsb += inv_k * rb;
        } else if (take) {
          gsa = ra;
          gsb = rb;
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
        load_tbuf(sh, mh0, wave, lane, tid);
        load_fbuf(sh, mh0, wave, lane);  // next step's F_0 (and the input rows F_0's adjoint recomputes n from)
      }
    }
    // dLoss/dy = -(sum_k w_k dq_k/dy + d gl/dy_T): lane (c, q) fills coordinates 2q, 2q+1
    {
      float ga = gsa, gb = gsb;
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
    pass_backward<MODE_FWD, false, (WPB <= 4)>(tw, wq4, wl, io, gy, stF, tapeF, nullptr, pre_all + ((size_t)0 * a.B + b) * PRE_FLOATS, c, q, res, w0);
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
      if (step + 1 < S && RIP_ABL != 2) load_tbuf(sh, mh0 + MH_SIZE, wave, lane, tid);
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
    const size_t orow = (size_t)b * a.N + orig;
    if (a.plans != nullptr) {
      a.plans[orow * 8 + 2 * q] = io[c][2 * q];
      a.plans[orow * 8 + 2 * q + 1] = io[c][2 * q + 1];
    }
    if (a.loss_best != nullptr && q == 0) a.loss_best[orow] = lbest;
  }
}

static bool wants_trace(const SearchArgs& a) {
  return a.trace_post != nullptr || a.trace_x != nullptr || a.trace_loss != nullptr || a.trace_grad != nullptr;
}

}  // namespace

bool search_split_supported(const SearchArgs& a) { return a.K >= 1 && a.K <= MAX_MODELS && a.N % CB == 0; }

// scratch of one launch: the prefix table [K][B][PRE_FLOATS] followed by two tape slots per 16-candidate block
size_t search_split_scratch_bytes(int B, int N, int K) {
  if (N < CB) return 0;
  const size_t pre = ((size_t)K * B * PRE_FLOATS * sizeof(float) + 255) / 256 * 256;
  const size_t items = ((size_t)B * (N / CB) + WPB_MAX - 1) / WPB_MAX * WPB_MAX;
  return pre + items * (2 * TAPE_SLOT_F4 + PARK_F4) * sizeof(float4);
}

namespace {
template <int WPB>
hipError_t launch_split_wpb(const SearchArgs& a, const uint32_t* mh_all, const float* pre, float4* tape, int items, hipStream_t s) {
  hipError_t e = allow_lds(reinterpret_cast<const void*>(search_split_kernel<false, WPB>));
  if (e != hipSuccess) return e;
  e = allow_lds(reinterpret_cast<const void*>(search_split_kernel<true, WPB>));
  if (e != hipSuccess) return e;
  const dim3 grid((items + WPB - 1) / WPB);
  if (wants_trace(a))
    hipLaunchKernelGGL((search_split_kernel<true, WPB>), grid, dim3(WPB * 64), sizeof(PShared<WPB>), s, a, mh_all, pre, tape);
  else
    hipLaunchKernelGGL((search_split_kernel<false, WPB>), grid, dim3(WPB * 64), sizeof(PShared<WPB>), s, a, mh_all, pre, tape);
  return hipGetLastError();
}
}  // namespace

// waves per workgroup of a launch over `items` 16-candidate blocks.  Cost model from the measurements (B = 512, K = 4,
// N = 128, 10 Adam steps): the 4-wave workgroup (one wave per SIMD, the whole register file, the inverse passes' tape in
// registers) takes 0.63 ms, the 2-wave one about as long for half the blocks.  A launch is ceil(workgroups / CUs) rounds
// of that.  development: RIP_SPLIT_WPB=4|2|16 in the environment pins the shape (16 = the paired shape).
static int split_pick_wpb(int items, int forced_shape = 0) {
  static const int env_forced = [] {
    const char* e = getenv("RIP_SPLIT_WPB");
    return e != nullptr ? atoi(e) : 0;
  }();
  const int forced = forced_shape != 0 ? forced_shape : env_forced;
  if (forced == 4 || forced == 2 || forced == SPLIT_SHAPE_PAIR) return forced;
  const int cus = device_cu_count();
  auto rounds = [&](int wpb) { return (double)((items + wpb * cus - 1) / (wpb * cus)); };
  // round 6: the paired shape (flow_pair.hip) — four blocks per workgroup like the 4-wave shape, two waves per block:
  // measured 1.27x the 4-wave shape's time (profiles/r6/pair_kernel_v1.txt), so the model never picks it
  const double c4 = rounds(4), c2 = 0.96 * rounds(2), cp = RIP_PAIR_COST * rounds(4);
  if (cp <= c4 && cp <= c2) return SPLIT_SHAPE_PAIR;
  return c4 <= c2 ? 4 : 2;
}

// What a launch executes on the matrix cores, per 16-candidate block (bench.py's executed-flops count; checked against
// rocprofv3 SQ_INSTS_MFMA in profiles/): [0] waves per workgroup, then (f16, fp32) MFMA instructions of [1,2] a
// forward / inverse pass (3 steps of 99 + 0; rounds 3-5: 84 + 27), [3,4] the adjoint of an inverse pass, [5,6] the adjoint of F_0, [7,8]
// the prefix step per (model, observation).  An adjoint step is 12 (W1^T) + 72 (W_hh^T; none at t = T-1) + 18 (W_ih^T)
// f16 and 2 (W2^T) + 4 (gi_n, only when the step comes from the tape) fp32 instructions.
void search_split_info(int B, int N, int K, int out[9], int shape) {
  (void)K;
  const int wpb = split_pick_wpb(B * (N / CB), shape);
  const bool regtape = RIP_REGTAPE != 0;
  out[0] = wpb;
  out[1] = 3 * 99, out[2] = 0;  // round 6: a forward step is 99 f16 MFMAs (84 + 12 k-steps + 3 W2), no fp32 MFMA
  out[3] = 30 + 2 * 102, out[4] = regtape ? 3 * 2 : 2 + 2 * 6;
  out[5] = 30 + 2 * 102, out[6] = 3 * 6;
  out[7] = 99, out[8] = 0;
  if (wpb == SPLIT_SHAPE_PAIR) {
    // the paired shape, both waves of a block together: a forward step is 84 f16 + 2 x (8 + 1 + 5) fp32 MFMAs (the b2
    // k-step and nothing else is issued twice), an adjoint step 2 x (6 + 9 + 36) f16 (none of the 36 at t = 1) + 2 x 2
    // fp32 (W2^T on both waves) + 2 x 2 (gi_n, only when the step comes from the global tape)
    out[1] = 3 * 84, out[2] = 3 * 28;
    out[3] = 30 + 2 * 102, out[4] = 3 * 4;
    out[5] = 30 + 2 * 102, out[6] = 3 * 8;
  }
}

hipError_t launch_search_split(const SearchArgs& a, const uint32_t* mh_all, void* scratch, hipStream_t s) {
  float* pre = reinterpret_cast<float*>(scratch);
  const size_t pre_bytes = ((size_t)a.K * a.B * PRE_FLOATS * sizeof(float) + 255) / 256 * 256;
  float4* tape = reinterpret_cast<float4*>(reinterpret_cast<char*>(scratch) + pre_bytes);
  if (a.range_flag != nullptr) {
    hipError_t e = hipMemsetAsync(a.range_flag, 0, sizeof(unsigned), s);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(split_prefix_kernel, dim3(a.B, a.K), dim3(64), 0, s, a, mh_all, pre);
  const int items = a.B * (a.N / CB);
  switch (split_pick_wpb(items, a.split_shape)) {
    case SPLIT_SHAPE_PAIR: return launch_search_pair(a, mh_all, pre, tape, items, s);
    case 4: return launch_split_wpb<4>(a, mh_all, pre, tape, items, s);
    default: return launch_split_wpb<2>(a, mh_all, pre, tape, items, s);
  }
}

}  // namespace rip
