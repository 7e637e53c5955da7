// Micro-benchmark 4: what a software grid barrier costs on MI355X, against ~5.6 us per dependent kernel launch (the
// B = 1 encoder is 41 such launches).  G co-resident workgroups of 256 threads do `iters` rounds of
//   { touch a little global memory; barrier }  with a monotonically increasing ticket counter in global memory:
//   arrive: __threadfence(); atomicAdd(counter, 1)   (one thread per workgroup)
//   wait  : spin on an atomic load until counter >= round * G; __threadfence()
//   hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o tools/micro/grid_barrier && tools/micro/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();  // release: this workgroup's writes are visible device-wide before it arrives
    atomicAdd(counter, 1u);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();  // acquire
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counter, float* buf, int iters, long long* cyc) {
  const int G = gridDim.x;
  const long long t0 = wall_clock64();
  float acc = 0.f;
  for (int it = 1; it <= iters; ++it) {
    // a token amount of "layer" work: every workgroup writes a line and reads its neighbour's from the last round
    buf[(size_t)blockIdx.x * 256 + threadIdx.x] = acc + it;
    grid_barrier(counter, (unsigned)it * G);
    acc += buf[(size_t)((blockIdx.x + 1) % G) * 256 + threadIdx.x];
  }
  if (threadIdx.x == 0) cyc[blockIdx.x] = wall_clock64() - t0;
  if (acc == 12345.f) buf[0] = acc;
}

int main() {
  unsigned* counter;
  float* buf;
  long long* cyc;
  hipMalloc(&counter, 4);
  hipMalloc(&buf, 1024 * 256 * sizeof(float));
  hipMalloc(&cyc, 1024 * sizeof(long long));
  const int iters = 2000;
  int wall_khz = 100000;  // wall_clock64 runs at a constant 100 MHz
  hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  for (int G : {4, 16, 32, 64, 128, 256}) {
    hipMemset(counter, 0, 4);
    hipMemset(buf, 0, 1024 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(256), 0, 0, counter, buf, iters, cyc);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("G = %3d workgroups: %.2f us per round (kernel %.2f ms for %d rounds)\n", G, 1e3 * ms / iters, ms, iters);
  }
  return 0;
}
