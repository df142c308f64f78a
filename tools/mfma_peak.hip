// measures the achievable fp32 MFMA rate and the clock held under that load on this device (experiments only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void peak(float* out, int iters, unsigned long long* clk) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0}, a4={0}, a5={0}, a6={0}, a7={0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f + 0.5f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    a4 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a4, 0, 0, 0);
    a5 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a5, 0, 0, 0);
    a6 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a6, 0, 0, 0);
    a7 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a7, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int r = 0; r < 16; ++r) s += a0[r]+a1[r]+a2[r]+a3[r]+a4[r]+a5[r]+a6[r]+a7[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main() {
  float* out; unsigned long long* clk; hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&clk, 16);
  for (int wgs : {256, 512}) {
    const int iters = 20000;
    hipLaunchKernelGGL(peak, dim3(wgs), dim3(256), 0, 0, out, 1000, clk); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(peak, dim3(wgs), dim3(256), 0, 0, out, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double fl = (double)wgs * 4 * iters * 8 * 4096.0;
    printf("wgs=%d  %.3f ms  %.1f TFLOP/s  clock=%.3f GHz (memtime/memrealtime*100MHz)\n", wgs, ms, fl / ms / 1e9, (double)h[0] / (double)h[1] * 0.1);
  }
  return 0;
}
