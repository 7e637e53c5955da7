// Micro-benchmark: how fast do 256 workgroups (one per CU, 256 threads) write 40 KB each — the tile blocks' epilogue burst
// (encoder_bf16_tile.hip) — as a function of the lane -> address pattern?   hipcc --offload-arch=gfx950 -O3 store_burst.hip
//   A: 16 B per lane, lanes contiguous (1 KB per instruction)
//   B: the MFMA accumulator layout as it is: lane (n = lane & 15, q = lane >> 4) writes 8 B at pixel n * 640 B + 8 q (+ 32 ct)
//   C: transposed pair: lane (n' = lane >> 2, j = lane & 3) writes 16 B at pixel n' * 640 B + 16 j (+ 64 ctp)
//   D: like A but 128-byte pieces round-robin over pixels (8 lanes x 16 B = one full line per pixel row)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int PIX = 64, ROWB = 640;  // a workgroup's region: 64 pixels x 640 bytes = 40 KB
template <int MODE>
__global__ __launch_bounds__(256) void burst(unsigned char* out, int reps, size_t rep_stride) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int r = 0; r < reps; ++r) {
    unsigned char* base = out + (size_t)r * rep_stride + (size_t)blockIdx.x * PIX * ROWB;
    if (MODE == 0) {
      for (int i = 0; i < 10; ++i)  // wave w: 10 KB contiguous
        *reinterpret_cast<u32x4*>(base + w * 10240 + i * 1024 + lane * 16) = u32x4{1u, 2u, 3u, (unsigned)r};
    } else if (MODE == 1) {
      const int n = lane & 15, q = lane >> 4;  // wave w: 80 channels (160 B) of every pixel, 4 pixel tiles x 5 channel tiles
      for (int t = 0; t < 4; ++t)
        for (int ct = 0; ct < 5; ++ct)
          *reinterpret_cast<u32x2*>(base + (16 * t + n) * ROWB + w * 160 + ct * 32 + q * 8) = u32x2{1u, (unsigned)r};
    } else if (MODE == 2) {
      const int n2 = lane >> 2, j = lane & 3;
      for (int t = 0; t < 4; ++t) {
        for (int cp = 0; cp < 2; ++cp)
          *reinterpret_cast<u32x4*>(base + (16 * t + n2) * ROWB + w * 160 + cp * 64 + j * 16) = u32x4{1u, 2u, 3u, (unsigned)r};
        *reinterpret_cast<u32x2*>(base + (16 * t + n2) * ROWB + w * 160 + 128 + j * 8) = u32x2{1u, (unsigned)r};
      }
    } else {
      const int n8 = lane >> 3, j = lane & 7;  // 8 pixels x 128 B per instruction; wave w: 128-byte column block w (+ 4: the last 128 B by wave 0)
      for (int t = 0; t < 8; ++t) {
        *reinterpret_cast<u32x4*>(base + (8 * t + n8) * ROWB + w * 128 + j * 16) = u32x4{1u, 2u, 3u, (unsigned)r};
        if (w == 0) *reinterpret_cast<u32x4*>(base + (8 * t + n8) * ROWB + 512 + j * 16) = u32x4{1u, 2u, 3u, (unsigned)r};
      }
    }
  }
}
int main() {
  const int WGS = 256, REPS = 16;
  const size_t rep_stride = (size_t)WGS * PIX * ROWB, bytes = rep_stride * REPS;
  unsigned char* d;
  hipMalloc(&d, bytes);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const char* names[4] = {"A contiguous 16 B/lane", "B accumulator layout 8 B/lane", "C transposed 16 B/lane, 64 B/pixel", "D full 128-B lines"};
  for (int reps : {1, REPS})
    for (int m = 0; m < 4; ++m) {
      float best = 1e9f;
      for (int it = 0; it < 6; ++it) {
        hipMemsetAsync(d, 0, 64, 0);
        hipEventRecord(a, 0);
        if (m == 0) hipLaunchKernelGGL(burst<0>, dim3(WGS), dim3(256), 0, 0, d, reps, rep_stride);
        if (m == 1) hipLaunchKernelGGL(burst<1>, dim3(WGS), dim3(256), 0, 0, d, reps, rep_stride);
        if (m == 2) hipLaunchKernelGGL(burst<2>, dim3(WGS), dim3(256), 0, 0, d, reps, rep_stride);
        if (m == 3) hipLaunchKernelGGL(burst<3>, dim3(WGS), dim3(256), 0, 0, d, reps, rep_stride);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (it > 0 && ms < best) best = ms;
      }
      printf("reps %2d  %-40s %8.1f us  %7.2f TB/s\n", reps, names[m], best * 1e3, rep_stride * (double)reps / (best * 1e-3) / 1e12);
    }
  return 0;
}
