// probe: does the memory-side cache (MALL / Infinity Cache, 256 MB) serve the SECOND sweep over a tensor, and does the sweep DIRECTION decide?
// kernel W writes a buffer front to back; kernel R reads a buffer front to back or back to front (16-byte loads, 1 KB per wavefront and
// step, 2 048 persistent wavefronts striding together -- a sliding window).  Sequences timed: (W, R forward), (W, R reverse), (R forward, R
// forward), (R forward, R reverse), for buffers of 128 MB ... 1.6 GB.
//   hipcc --offload-arch=gfx950 -O3 mall_reuse.hip -o mall_reuse && ./mall_reuse
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void sweep_read(const u32x4* src, unsigned long long nkb, int reverse, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const unsigned long long w = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (unsigned long long)gridDim.x * 4;
  unsigned acc = 0;
  for (unsigned long long k = w; k < nkb; k += nw) {
    const unsigned long long kk = reverse ? nkb - 1 - k : k;
    const u32x4 v = src[kk * 64 + lane];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void sweep_write(u32x4* dst, unsigned long long nkb) {
  const int lane = threadIdx.x & 63;
  const unsigned long long w = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (unsigned long long)gridDim.x * 4;
  for (unsigned long long k = w; k < nkb; k += nw) dst[k * 64 + lane] = u32x4{1u, 2u, 3u, (unsigned)k};
}

int main() {
  const unsigned long long cap = 1664ull << 20;
  u32x4* buf; unsigned* sink;
  if (hipMalloc(&buf, cap) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  hipEvent_t e0, e1, e2;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
  printf("size MB | first kernel | second kernel | second: ms, TB/s\n");
  for (unsigned long long mb : {128ull, 256ull, 410ull, 812ull, 1625ull}) {
    const unsigned long long nkb = mb << 10;
    for (int first = 0; first < 2; ++first)       // 0: write, 1: read forward
      for (int rev = 0; rev < 2; ++rev) {
        float ms1 = 0, ms2 = 0;
        for (int rep = 0; rep < 3; ++rep) {
          (void)hipEventRecord(e0);
          if (first == 0) hipLaunchKernelGGL(sweep_write, dim3(512), dim3(256), 0, 0, buf, nkb);
          else hipLaunchKernelGGL(sweep_read, dim3(512), dim3(256), 0, 0, buf, nkb, 0, sink);
          (void)hipEventRecord(e1);
          hipLaunchKernelGGL(sweep_read, dim3(512), dim3(256), 0, 0, buf, nkb, rev, sink);
          (void)hipEventRecord(e2);
          (void)hipDeviceSynchronize();
          (void)hipEventElapsedTime(&ms1, e0, e1); (void)hipEventElapsedTime(&ms2, e1, e2);
        }
        printf("%5llu | %s %.3f ms (%.2f TB/s) | read %s | %.3f ms, %.2f TB/s\n", mb, first == 0 ? "write" : "read ", ms1, (double)nkb * 1024 / ms1 / 1e9,
               rev ? "reverse" : "forward", ms2, (double)nkb * 1024 / ms2 / 1e9);
      }
  }
  return 0;
}
