// Scalar device helpers shared by the flow kernels (flow.hip, flow_phase.hip, flow_split.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace rip {

constexpr float LOG_2PI = 1.8378770664093453f;

// ------------------------------------------------------------------------------------------
// scalar math (fp32; tolerances of the parity tests are 1e-4 absolute)
// ------------------------------------------------------------------------------------------
// v_rcp_f32 / v_exp_f32 / v_log_f32 are ~1 ulp; the parity budget is 1e-4 absolute.
__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  // 1 - 2/(e^{2x}+1): absolute error ~1e-7, saturates cleanly for large |x|
  return 1.0f - 2.0f * rcpf_(__expf(2.0f * x) + 1.0f);
}
// F.softplus(beta=1, threshold=20) (sequence.py:133,193); the caller adds the 1e-3 floor, so the
// absolute error of log(1+e^x) for very negative x (<= 6e-8) is invisible
__device__ __forceinline__ float softplusf_(float x) { return x > 20.0f ? x : __logf(1.0f + __expf(x)); }
__device__ __forceinline__ float softplus_gradf_(float x) { return x > 20.0f ? 1.0f : sigmoidf_(x); }


// goal log-likelihood of the last waypoint and (optionally) its gradient  (dim/model.py:163-171)
// -|y - g|^2 / (2 eps^2) of one goal, with every operation rounded on its own (no fma contraction): goal_ll evaluates
// it twice per goal (running maximum, then the exponentials) and the two values of the arg-max goal must be THE SAME
// float — at |y| ~ 5e4 one ulp of the quadratic is 128, and a contraction chosen differently in the two loops made the
// largest exponent -128 instead of 0: every term underflowed, log(0) = -inf (found by test_split_kernel_operand_ranges).
__device__ __forceinline__ float goal_arg(const float* __restrict__ goal, int j, float inv2, float y0, float y1, float* d0o,
                                          float* d1o) {
  const float d0 = y0 - goal[2 * j], d1 = y1 - goal[2 * j + 1];
  *d0o = d0;
  *d1o = d1;
  return __fmul_rn(-__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), inv2);
}
__device__ __forceinline__ float goal_ll(const float* __restrict__ goal, int G, float eps, float y0, float y1,
                                         float* g0, float* g1) {
  const float inv2 = 1.0f / (2.0f * eps * eps);
  float m = -INFINITY;
  for (int j = 0; j < G; ++j) {
    float d0, d1;
    m = fmaxf(m, goal_arg(goal, j, inv2, y0, y1, &d0, &d1));
  }
  float se = 0.f, a0 = 0.f, a1 = 0.f;
  for (int j = 0; j < G; ++j) {
    float d0, d1;
    const float e = __expf(goal_arg(goal, j, inv2, y0, y1, &d0, &d1) - m);  // argument <= 0, == 0 for the arg-max goal
    se += e;
    a0 = fmaf(e, -d0, a0);
    a1 = fmaf(e, -d1, a1);
  }
  if (g0 != nullptr) {
    const float sc = 2.0f * inv2 / se;  // d/dy logsumexp = sum_j softmax_j * (g_j - y)/eps^2
    *g0 = a0 * sc;
    *g1 = a1 * sc;
  }
  return m + __logf(se) - 2.0f * logf(eps) - LOG_2PI - logf((float)G);  // se in [1, G]
}


}  // namespace rip
