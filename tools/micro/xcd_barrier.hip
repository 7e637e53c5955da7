// Micro-benchmark 5: a barrier among the workgroups of ONE XCD (one L2) against the device-wide one of
// grid_barrier.hip and the ~4.8 us a dependent launch costs inside a hipGraph (the B = 1 encoder is 55 of them).
//   1. where do the workgroups of a launch land?  (HW_REG_XCC_ID per workgroup: round-robin over the 8 XCDs?)
//   2. rounds of { write a slab; barrier; read the slab another workgroup of the same XCD wrote } with
//        mode 0: device-wide counter, agent-scope release / acquire fences (buffer_wbl2 sc1 / buffer_inv sc1)
//        mode 1: one counter per XCD, no cache maintenance: s_waitcnt vmcnt(0) before the arrival (stores are in the
//                XCD's L2 then: the TCP is write-through), polls are returning atomics (performed in the L2), every
//                round writes a fresh slab (no line that a TCP could still hold is ever rewritten)
//        mode 2: as 1, but the slab is REUSED every round (a consumer's TCP may hold the previous round's line)
//        mode 3: as 2 with buffer_inv sc1 after the wait
//      and the number of wrong values read (modes 0, 1, 3 must read none).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_barrier.hip -o tools/micro/xcd_barrier && tools/micro/xcd_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int SLAB = 256;  // floats per workgroup and round

struct Ctl {
  unsigned xcd_count[8][32];   // arrival counters, one 128-byte line per XCD
  unsigned xcd_rank[8][32];    // rank allocation per XCD
  unsigned global_count[32];
  unsigned timeout[32];
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(63508) & 0xf; }

__device__ __forceinline__ bool spin_until(unsigned* ctr, unsigned target, unsigned* timeout_flag) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_fetch_add(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > 20000000) {  // 0.2 s at 100 MHz: give up instead of hanging the box
      atomicAdd(timeout_flag, 1u);
      return false;
    }
  }
  return true;
}

template <int MODE>
__global__ __launch_bounds__(256) void barrier_kernel(Ctl* ctl, float* buf, int iters, int per_xcd, int* xcc_of, unsigned* bad) {
  const int G = gridDim.x;
  const unsigned x = xcc_id();
  __shared__ unsigned s_rank;
  if (threadIdx.x == 0) {
    xcc_of[blockIdx.x] = (int)x;
    s_rank = __hip_atomic_fetch_add(&ctl->xcd_rank[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const unsigned rank = s_rank;
  if (MODE != 0 && rank >= (unsigned)per_xcd) return;  // more workgroups on this XCD than expected: not part of the group
  unsigned* ctr = MODE == 0 ? ctl->global_count : ctl->xcd_count[x];
  const unsigned members = MODE == 0 ? (unsigned)G : (unsigned)per_xcd;
  const unsigned me = MODE == 0 ? blockIdx.x : rank;
  const unsigned peer = (me + 1) % members;
  // slab index space: [xcd or 0][round (fresh modes)][member]
  const size_t group_base = MODE == 0 ? 0 : (size_t)x * per_xcd;
  unsigned wrong = 0;
  bool alive = true;
  for (int it = 1; it <= iters && alive; ++it) {
    const size_t round_base = (MODE == 1 ? (size_t)(it & 63) * 8 * 64 : 0);  // 64 fresh slabs in rotation (written 64 rounds apart)
    float* mine = buf + (round_base + group_base + me) * SLAB;
    const float* theirs = buf + (round_base + group_base + peer) * SLAB;
    mine[threadIdx.x] = (float)(it * 1024 + (int)me);
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      alive = spin_until(ctr, (unsigned)it * members, ctl->timeout);
      s_rank = alive ? 1u : 0u;
    }
    __syncthreads();
    alive = s_rank != 0u;
    if (MODE == 0 || MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const float got = theirs[threadIdx.x];
    if (alive && got != (float)(it * 1024 + (int)peer)) ++wrong;
    __syncthreads();  // s_rank is rewritten next round
  }
  if (wrong) atomicAdd(bad, wrong);
}

template <int MODE>
void run(int G, int per_xcd, int iters, Ctl* ctl, float* buf, int* xcc_of, unsigned* bad) {
  hipMemset(ctl, 0, sizeof(Ctl));
  hipMemset(bad, 0, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(barrier_kernel<MODE>, dim3(G), dim3(256), 0, 0, ctl, buf, iters, per_xcd, xcc_of, bad);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  Ctl h;
  unsigned hb = 0;
  hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
  hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("mode %d  G = %4d (%3d per XCD): %6.2f us per round, wrong values %u, timeouts %u\n", MODE, G, per_xcd,
         1e3 * ms / iters, hb, h.timeout[0]);
}

int main() {
  Ctl* ctl;
  float* buf;
  int* xcc_of;
  unsigned* bad;
  hipMalloc(&ctl, sizeof(Ctl));
  hipMalloc(&buf, (size_t)64 * 8 * 64 * SLAB * sizeof(float) + (size_t)2048 * SLAB * sizeof(float));
  hipMalloc(&xcc_of, 4096 * sizeof(int));
  hipMalloc(&bad, 4);
  hipMemset(buf, 0, (size_t)64 * 8 * 64 * SLAB * sizeof(float));
  const int iters = 2000;
  // 1. placement
  for (int G : {8, 64, 256, 512}) {
    run<0>(G, G / 8, 10, ctl, buf, xcc_of, bad);
    std::vector<int> h(G);
    hipMemcpy(h.data(), xcc_of, G * sizeof(int), hipMemcpyDeviceToHost);
    int rr = 0;
    for (int i = 0; i < G; ++i) rr += h[i] == (i % 8);
    printf("  placement G = %d: %d of %d workgroups on XCD (blockIdx %% 8); first 16:", G, rr, G);
    for (int i = 0; i < 16 && i < G; ++i) printf(" %d", h[i]);
    printf("\n");
  }
  // 2. barrier cost
  for (int per : {4, 8, 16, 32, 64}) {
    const int G = 8 * per;
    run<0>(G, per, iters, ctl, buf, xcc_of, bad);
    run<1>(G, per, iters, ctl, buf, xcc_of, bad);
    run<2>(G, per, iters, ctl, buf, xcc_of, bad);
    run<3>(G, per, iters, ctl, buf, xcc_of, bad);
  }
  return 0;
}
