// probe: do two kernels on two streams that sweep the SAME tensor front to back at the same time share its HBM reads through the
// memory-side cache (256 MB)?  Timed: one sweep alone; two sweeps one after the other; two concurrent sweeps of the same buffer; two
// concurrent sweeps of different buffers (control); concurrent with the second kernel started late (offset = a fraction of the buffer).
//   hipcc --offload-arch=gfx950 -O3 mall_concurrent.hip -o mall_concurrent && ./mall_concurrent
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void sweep_read(const u32x4* src, unsigned long long nkb, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const unsigned long long w = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (unsigned long long)gridDim.x * 4;
  unsigned acc = 0;
  unsigned long long k = w;
  for (; k + 3 * nw < nkb; k += 4 * nw) {   // four loads in flight per wavefront
    const u32x4 a = src[k * 64 + lane], b = src[(k + nw) * 64 + lane], c = src[(k + 2 * nw) * 64 + lane], d = src[(k + 3 * nw) * 64 + lane];
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  for (; k < nkb; k += nw) acc += src[k * 64 + lane].x;
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const unsigned long long mb = 1632, nkb = mb << 10;
  u32x4 *b1, *b2; unsigned* sink;
  if (hipMalloc(&b1, mb << 20) != hipSuccess || hipMalloc(&b2, mb << 20) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  (void)hipMemset(b1, 1, mb << 20); (void)hipMemset(b2, 2, mb << 20);
  hipStream_t s1, s2;
  (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t e0, e1, ej;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&ej);
  for (int grid : {512, 1024, 2048}) {
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, s1);
        (void)hipStreamWaitEvent(s2, e0, 0);
        if (mode == 0) {
          hipLaunchKernelGGL(sweep_read, dim3(grid), dim3(256), 0, s1, b1, nkb, sink);
        } else if (mode == 1) {
          hipLaunchKernelGGL(sweep_read, dim3(grid), dim3(256), 0, s1, b1, nkb, sink);
          hipLaunchKernelGGL(sweep_read, dim3(grid), dim3(256), 0, s1, b1, nkb, sink);
        } else {
          hipLaunchKernelGGL(sweep_read, dim3(grid / 2), dim3(256), 0, s1, b1, nkb, sink);
          hipLaunchKernelGGL(sweep_read, dim3(grid / 2), dim3(256), 0, s2, mode == 2 ? b1 : b2, nkb, sink);
          (void)hipEventRecord(ej, s2);
          (void)hipStreamWaitEvent(s1, ej, 0);
        }
        (void)hipEventRecord(e1, s1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const char* names[4] = {"one sweep", "two sweeps, one stream", "two concurrent sweeps, same buffer", "two concurrent sweeps, two buffers"};
      const double bytes = (double)(mode == 0 ? 1 : 2) * (double)(mb << 20);
      printf("grid %4d | %-36s | %.3f ms | %.2f TB/s of requested bytes\n", grid, names[mode], best, bytes / best / 1e9);
    }
  }
  return 0;
}
