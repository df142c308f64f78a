// Probe (experiments only): sustained bf16 MFMA rate under this pool's power cap, 32x32x16 against 16x16x32, operands in
// registers, two wavefronts per SIMD, the same 128 x 64 output tile per wavefront, no back-to-back dependent MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_shapes2.hip -o tools/probes/mfma_shapes2 && tools/probes/mfma_shapes2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void probe(float* out, int iters, unsigned long long* clk, const u32x4* src) {
  const int tid = threadIdx.x, lane = tid & 63;
  u32x4 a[8], b[4];
  for (int i = 0; i < 8; ++i) a[i] = src[(i * 64 + lane) % (6 * 64 * 4)];
  for (int i = 0; i < 4; ++i) b[i] = src[((8 + i) * 64 + lane) % (6 * 64 * 4)];
  float s = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (SHAPE == 0) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {   // one iteration = a k32 step of the 128 x 64 tile: 16 MFMAs, a dependent pair is 8 apart
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[h * 4 + m]), __builtin_bit_cast(bf16x8, b[h * 2 + n]),
                                                                   acc[m * 2 + n], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {   // 32 MFMAs of 16x16x32, all independent inside an iteration
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[m]), __builtin_bit_cast(bf16x8, b[n]), acc[m * 4 + n], 0, 0, 0);
    }
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0 && blockIdx.x == 17) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int SHAPE>
void run(float* out, unsigned long long* clk, const u32x4* src, const char* what) {
  const int iters = 60000, wgs = 256;
  hipLaunchKernelGGL((probe<SHAPE>), dim3(wgs), dim3(512), 0, 0, out, 4000, clk, src);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f, worst = 0.f; unsigned long long h[2] = {0, 0};
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<SHAPE>), dim3(wgs), dim3(512), 0, 0, out, iters, clk, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) { best = ms; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }
    if (ms > worst) worst = ms;
  }
  const double fl = (double)wgs * 8 * iters * 2.0 * 128 * 64 * 32;
  printf("%-22s best %8.3f ms (worst %8.3f)  %7.1f TFLOP/s  ticks per k32 step %.1f\n", what, best, worst, fl / best / 1e9, (double)h[0] / iters);
}

int main() {
  float* out; unsigned long long* clk; u32x4* src;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16); hipMalloc(&src, 6 * 64 * 4 * 16);
  unsigned short hsrc[6 * 64 * 4 * 8];
  unsigned s = 12345u;
  for (auto& v : hsrc) { s = s * 1664525u + 1013904223u; const float f = ((int)(s >> 8) % 2001 - 1000) / 1000.0f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
  for (int round = 0; round < 3; ++round) {
    run<0>(out, clk, src, "32x32x16 registers");
    run<1>(out, clk, src, "16x16x32 registers");
  }
  return 0;
}
