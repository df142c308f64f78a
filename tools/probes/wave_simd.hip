// Probe (experiments only): which SIMD do the wavefronts of a 512-thread workgroup land on?  (HW_REG_HW_ID: SIMD_ID = bits 5:4)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
  extern __shared__ char lds[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; hipMalloc(&d, 1024 * 8 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(probe, dim3(1024), dim3(512), 130 * 1024, 0, d);
  unsigned h[1024 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int pair_same = 0, total = 0;
  for (int b = 0; b < 1024; ++b) {
    if (b < 6) { printf("wg %d: simd of waves 0..7:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("   cu %u\n", (h[b * 8] >> 8) & 15); }
    for (int w = 0; w < 4; ++w) { total++; pair_same += ((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3); }
  }
  printf("waves w and w+4 on the same SIMD: %d of %d\n", pair_same, total);
  int adj = 0; for (int b = 0; b < 1024; ++b) for (int w = 0; w < 8; w += 2) adj += ((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 1] >> 4) & 3);
  printf("waves 2k and 2k+1 on the same SIMD: %d of %d\n", adj, 1024 * 4);
  return 0;
}
