// measures the achievable bf16 MFMA rate (v_mfma_f32_32x32x16_bf16) and the clock held under that load (experiments only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void peak(float* out, int iters, unsigned long long* clk, float seed) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0}, a4 = {0}, a5 = {0}, a6 = {0}, a7 = {0};
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(seed * (threadIdx.x % 13 + i)); y[i] = (__bf16)(seed * (threadIdx.x % 7 + 2 * i) + 0.5f * seed); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a3, 0, 0, 0);
    a4 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a4, 0, 0, 0);
    a5 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a5, 0, 0, 0);
    a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a6, 0, 0, 0);
    a7 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a7, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int r = 0; r < 16; ++r) s += a0[r]+a1[r]+a2[r]+a3[r]+a4[r]+a5[r]+a6[r]+a7[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int WAVES>
void run(float* out, unsigned long long* clk, float seed, const char* what) {
  const int iters = 20000, wgs = 256;
  hipLaunchKernelGGL(peak<WAVES>, dim3(wgs), dim3(64 * WAVES), 0, 0, out, 1000, clk, seed); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(peak<WAVES>, dim3(wgs), dim3(64 * WAVES), 0, 0, out, iters, clk, seed); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  double fl = (double)wgs * WAVES * iters * 8 * 32768.0;
  printf("%s waves/CU=%d  %.3f ms  %.1f TFLOP/s  s_memtime ticks per MFMA per SIMD=%.1f  ticks/100MHz-tick=%.2f\n", what, WAVES, ms, fl / ms / 1e9,
         (double)h[0] / (iters * 8.0 * (WAVES / 4.0)), (double)h[0] / (double)h[1]);
}
int main() {
  float* out; unsigned long long* clk; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16);
  run<4>(out, clk, 0.f, "zeros ");
  run<4>(out, clk, 0.37f, "random");
  run<8>(out, clk, 0.37f, "random");
  return 0;
}
