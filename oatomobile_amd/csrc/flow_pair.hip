// Paired split-f16 plan search for gfx950 (round 6): the throughput shape of RIPAgent.__call__'s search
// (rip/agent.py:78-137) for launches that fill the chip.
//
// flow_split.hip runs a 16-candidate block on one wave at one wave per SIMD, where the wave's time is the SUM of its
// matrix-pipe and vector-issue cycles.  Here a block belongs to a PAIR of waves that split the hidden units
// (flow_pair_dev.h): eight waves per workgroup = four blocks, two waves on every SIMD, so that one wave's MFMAs run
// beside the other's gate math — with each wave holding half of the register tape (the two-waves-per-SIMD build of the
// one-wave kernel had to spill its tape to global memory).  Phase structure (F_0, the K - 1 inverses with their
// data-dependent adjoints, F_0's adjoint + Adam, operands of the current model in LDS by DMA, workgroup barriers between
// model phases), arithmetic (two-term binary16 operands on v_mfma_f32_16x16x32_f16, fp32 accumulate), scratch layout
// and the operand blob are flow_split.hip's.  Same gates: the teacher-forced 1e-4 tests and the G6 traces.
#include <cstdlib>

#include "flow.h"
#include "flow_math.h"
#include "flow_pair_dev.h"

namespace rip {

namespace {

using namespace split;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int PWAVES = 8;   // waves per workgroup
constexpr int PBLOCKS = 4;  // 16-candidate blocks (= pairs) per workgroup
constexpr int F_ROWS = MHF_BASE_ROWS;  // (round 5's forward rows: this kernel keeps the fp32 k-steps; its LDS has no room for more)
constexpr int T_ROWS = MHT_ROWS;

struct PairShared {
  uint4 fbuf[F_ROWS * 64];            // forward operand rows of the current model
  uint4 tbuf[T_ROWS * 64];            // transposed operand rows of the current model
  uint4 wihc[MH_TABLE_F4];            // its W_ih^T table
  float xs[PBLOCKS][CB][8];           // per pair: the latent the F_0 pass maps
  float ys[PBLOCKS][CB][8];           // per pair: y = F_0(x)
  float gy[PBLOCKS][CB][8];           // per pair: dLoss/dy handed to the F_0 adjoint
  float stape[PBLOCKS][2][T][6][CB];  // per pair: per-candidate scalars of the F_0 pass [0] and of the current inverse [1]
  u32x4 xrows[PWAVES][2][64];         // per wave: exchange slot rows
  f32x2 xextra[PWAVES][64];           //           ... and 8 more bytes per lane
  unsigned ctl[PWAVES][2];            // per wave: flag (payloads published), ack (peer payloads consumed)
};
static_assert(sizeof(PairShared) <= 160 * 1024, "the pair kernel's LDS must fit a CU");

__device__ __forceinline__ void dma_rows8(const uint4* __restrict__ src, uint4* dst, int rows, int wave, int lane) {
  for (int r = wave; r < rows; r += PWAVES)
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + r * 64 + lane), (lds_ptr_t)(dst + r * 64), 16, 0, 0);
}
__device__ __forceinline__ void load_fbuf(PairShared& sh, const uint32_t* __restrict__ mhk, int wave, int lane) {
  dma_rows8(reinterpret_cast<const uint4*>(mhk), sh.fbuf, F_ROWS, wave, lane);
}
__device__ __forceinline__ void load_tbuf(PairShared& sh, const uint32_t* __restrict__ mhk, int wave, int lane, int tid) {
  const uint4* src = reinterpret_cast<const uint4*>(mhk) + MHF_ROWS * 64;
  dma_rows8(src, sh.tbuf, T_ROWS, wave, lane);
  if (tid < MH_TABLE_F4) sh.wihc[tid] = src[T_ROWS * 64 + tid];
}

#ifndef RIP_PAIR_SAME_SIMD
#define RIP_PAIR_SAME_SIMD 0  // 1: the two waves of a pair are waves w, w + 4 (one SIMD under round-robin placement)
#endif

#ifdef RIP_PROFILE_TICKS  // development (tools/search_ticks.py): where a wave's cycles go; one workgroup prints at the end
#define TK_DECL() long long tk_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk0_ = 0
#define TK_START() tk0_ = clock64()
#define TK_STOP(i_) tk_[i_] += clock64() - tk0_
#else
#define TK_DECL()
#define TK_START()
#define TK_STOP(i_)
#endif

template <bool TRACE>
__global__ __launch_bounds__(PWAVES * 64) void search_pair_kernel(SearchArgs a, const uint32_t* __restrict__ mh_all,
                                                                  const float* __restrict__ pre_all, float4* __restrict__ tape_all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  PairShared& sh = *reinterpret_cast<PairShared*>(smem_raw);
  if (a.range_flag != nullptr && __builtin_nontemporal_load(a.range_flag) != 0u) return;  // (flow_split.hip: operand-range guard)
  const int tid = threadIdx.x, lane = tid & 63;
  const int c = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = RIP_PAIR_SAME_SIMD ? (wave & 3) : (wave >> 1);
  const int hw = RIP_PAIR_SAME_SIMD ? (wave >> 2) : (wave & 1);
  const int peer = RIP_PAIR_SAME_SIMD ? (wave ^ 4) : (wave ^ 1);
  const int K = a.K;
  const int blocks_per_obs = a.N / CB;
  const int items = a.B * blocks_per_obs;
  const int item = blockIdx.x * PBLOCKS + pair;
  const bool active = item < items;  // a tail workgroup may carry idle pairs (they still serve the DMA)
  const int it = active ? item : items - 1;
  const int b = it / blocks_per_obs;
  const int n0 = (it - b * blocks_per_obs) * CB;
  const size_t row = (size_t)b * a.N + n0 + c;
  const float* goal = a.goal != nullptr ? a.goal + (size_t)b * a.G * 2 : nullptr;
  const uint32_t* mh0 = mh_all + (size_t)a.k0 * MH_SIZE;

  float (*xs)[8] = sh.xs[pair];
  float (*ys)[8] = sh.ys[pair];
  float (*gy)[8] = sh.gy[pair];
  float (*stF)[6][CB] = sh.stape[pair][0];
  float (*stI)[6][CB] = sh.stape[pair][1];
  const uint4* wl = sh.fbuf + lane;
  const uint4* tw = sh.tbuf + lane;
  const uint4* wq4 = sh.wihc + q * 2 + (c & 1);
  float4* tapeF = tape_all + (size_t)item * 2 * TAPE_SLOT_F4;  // (flow_split.hip's scratch layout: two slots per block, one used)

  PairXchg x;
  x.my_rows = (volatile RIP_LDS u32x4*)(sh.xrows[wave][0] + lane);
  x.peer_rows = (const volatile RIP_LDS u32x4*)(sh.xrows[peer][0] + lane);
  x.my_extra = (volatile RIP_LDS f32x2*)(sh.xextra[wave] + lane);
  x.peer_extra = (const volatile RIP_LDS f32x2*)(sh.xextra[peer] + lane);
  x.my_ctl = (volatile RIP_LDS unsigned*)sh.ctl[wave];
  x.peer_ctl = (const volatile RIP_LDS unsigned*)sh.ctl[peer];
  x.seq = 0;
  if (lane < 2) sh.ctl[wave][lane] = 0u;
  if (RIP_PAIR_PRIO == 3 && wave >= 4) __builtin_amdgcn_s_setprio(1);  // the later-dispatched half loses every arbitration otherwise

  // Adam state: lane (c, q) owns latent coordinates 2q, 2q+1 of candidate c — on BOTH waves of the pair (identical)
  float xv0 = a.x0[row * 8 + 2 * q], xv1 = a.x0[row * 8 + 2 * q + 1];
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
    xs[c][2 * q] = final_pass ? xb0 : xv0;
    xs[c][2 * q + 1] = final_pass ? xb1 : xv1;
    TK_START();
    __syncthreads();  // F-buf (and, at step 0, T-buf and the exchange words) landed; xs visible
    TK_STOP(0);
    float q_sel, gl = 0.f, gg0 = 0.f, gg1 = 0.f, w0;
    int ksel = 0;
    float gsa = 0.f, gsb = 0.f;  // sum_k w_k dq_k/dy, coordinates 2q and 2q+1 of this lane's candidate
    {
      const Prefix16 pre = load_prefix(pre_all + ((size_t)0 * a.B + b) * PRE_FLOATS, q);
      TK_START();
      const PassOut po = pass_forward_pair<MODE_FWD>(wl, hw, pre, xs, ys, stF, tapeF, nullptr, c, q, (unsigned)lane, x);
      TK_STOP(1);
      if (final_pass) break;
      if (goal != nullptr) gl = goal_ll(goal, a.G, a.epsilon, ys[c][6], ys[c][7], &gg0, &gg1);
      q_sel = (-0.5f * po.sq - 4.0f * LOG_2PI) - po.lad;  // model 0's posterior through the self-inverse shortcut
      if (TRACE && a.trace_post != nullptr && q == 0 && active && hw == 0)
        a.trace_post[(((size_t)step * K + 0) * a.B + b) * a.N + n0 + c] = q_sel + gl;
    }
    float q_sum = q_sel;
    // ================= models 1..K-1: inverse, adjoint, streaming aggregation =================
#pragma unroll 1
    for (int k = 1; k < K; ++k) {
      const uint32_t* mhk = mh_all + (size_t)(a.k0 + k) * MH_SIZE;
      TK_START();
      __syncthreads();  // every wave is done with the F-buf (F_0 or inverse_{k-1}) and the T-buf (adjoint_{k-1})
      TK_STOP(2);
      TK_START();
      load_fbuf(sh, mhk, wave, lane);
      if (k > 1) load_tbuf(sh, mhk, wave, lane, tid);  // (model 1's T-buf was requested under F_0)
      __syncthreads();  // operands of model k landed
      TK_STOP(3);
      const float* prek = pre_all + ((size_t)k * a.B + b) * PRE_FLOATS;
      const Prefix16 pre = load_prefix(prek, q);
      HalfTape last[3];
      TK_START();
      const PassOut po = pass_forward_pair<MODE_INV>(wl, hw, pre, xs, ys, stI, nullptr, last, c, q, (unsigned)lane, x);
      TK_STOP(4);
      const float qk = (-0.5f * po.sq - 4.0f * LOG_2PI) - po.lad;  // rip/agent.py:111-112
      if (TRACE && a.trace_post != nullptr && q == 0 && active && hw == 0)
        a.trace_post[(((size_t)step * K + k) * a.B + b) * a.N + n0 + c] = qk + gl;
      q_sum += qk;
      // rip/agent.py:121-127 as coded: WCM = min_k(-q) = the largest posterior, BCM = the smallest (first on ties)
      const bool take = a.algorithm == ALGO_WCM ? (qk > q_sel) : (qk < q_sel);
      if (mean_mode || __any(take)) {
        float res[8];
        TK_START();
        pass_backward_pair<MODE_INV>(tw, wq4, wl, hw, ys, nullptr, stI, nullptr, last, prek, c, q, res, 0.f, x);
        TK_STOP(5);
        if (a.stats != nullptr && lane == 0 && active && hw == 0) atomicAdd(a.stats, 1ull);  // executed inverse-pass adjoints (bench.py)
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
      load_tbuf(sh, mh0, wave, lane, tid);
      load_fbuf(sh, mh0, wave, lane);  // next step's F_0 (and the input rows F_0's adjoint recomputes n from)
    }
    // dLoss/dy = -(sum_k w_k dq_k/dy + d gl/dy_T): lane (c, q) fills coordinates 2q, 2q+1 (both waves: the same values)
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
    pass_backward_pair<MODE_FWD>(tw, wq4, wl, hw, ys, gy, stF, tapeF, nullptr, pre_all + ((size_t)0 * a.B + b) * PRE_FLOATS, c, q, res, w0, x);
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
    if (TRACE && active && hw == 0) {
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
      if (step + 1 < S) load_tbuf(sh, mh0 + MH_SIZE, wave, lane, tid);
    }
  }
#ifdef RIP_PROFILE_TICKS
  if (blockIdx.x == 7 && lane == 0)
    printf("ticks wave %d: barrier-top %lld | F %lld | barrier-done %lld barrier-dma %lld | inv %lld adj %lld | "
           "barrier-last %lld barrier-dma0 %lld | adjF %lld | spin ack %lld data %lld\n", wave, tk_[0], tk_[1], tk_[2], tk_[3], tk_[4],
           tk_[5], tk_[6], tk_[7], tk_[8], x.spin_ack, x.spin_data);
#endif
  // plan = F_0(x_best) is in ys (rip/agent.py:137)
  if (active && hw == 0) {
    const size_t orow = (size_t)b * a.N + n0 + c;
    if (a.plans != nullptr) {
      a.plans[orow * 8 + 2 * q] = ys[c][2 * q];
      a.plans[orow * 8 + 2 * q + 1] = ys[c][2 * q + 1];
    }
    if (a.loss_best != nullptr && q == 0) a.loss_best[orow] = lbest;
  }
}

bool wants_trace(const SearchArgs& a) {
  return a.trace_post != nullptr || a.trace_x != nullptr || a.trace_loss != nullptr || a.trace_grad != nullptr;
}

}  // namespace

// launched by launch_search_split (flow_split.hip) behind its prefix kernel, on its scratch layout
hipError_t launch_search_pair(const SearchArgs& a, const uint32_t* mh_all, const float* pre, float4* tape, int items, hipStream_t s) {
  hipError_t e = allow_lds(reinterpret_cast<const void*>(search_pair_kernel<false>));
  if (e != hipSuccess) return e;
  e = allow_lds(reinterpret_cast<const void*>(search_pair_kernel<true>));
  if (e != hipSuccess) return e;
  const dim3 grid((items + PBLOCKS - 1) / PBLOCKS);
  if (wants_trace(a))
    hipLaunchKernelGGL((search_pair_kernel<true>), grid, dim3(PWAVES * 64), sizeof(PairShared), s, a, mh_all, pre, tape);
  else
    hipLaunchKernelGGL((search_pair_kernel<false>), grid, dim3(PWAVES * 64), sizeof(PairShared), s, a, mh_all, pre, tape);
  return hipGetLastError();
}

}  // namespace rip
