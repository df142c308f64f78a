// Probe (experiments only, not part of the product): sustained rate and held clock of the bf16 MFMA shapes on this pool, bare
// (operands in registers) and with every operand re-read from LDS (ds_read_b128) -- same output tile per wavefront (128 x 64).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_shapes.hip -o tools/probes/mfma_shapes && tools/probes/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// SHAPE 0: 32x32x16 (8 accumulators = 128 x 64 tile), 1: 16x16x32 (32 accumulators), 2: 16x16x16 (32 accumulators, half the k)
template <int SHAPE, bool LDS>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned long long* clk, const u32x4* src) {
  __shared__ u32x4 lds[12 * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 12 * 64 * 4; i += 256) lds[i] = src[i % (6 * 64 * 4)] ^ u32x4{(unsigned)i, 0u, (unsigned)(i * 7), 0u};
  __syncthreads();
  u32x4 a[4], b[2];   // 128 rows x 16 k (32x32x16): 4 A fragments; 64 columns: 2 B fragments (one k16 step); x2 for a k32 step
  u32x4 a2[4], b2[2];
  for (int i = 0; i < 4; ++i) { a[i] = lds[(wave * 12 + i) * 64 + lane]; a2[i] = lds[(wave * 12 + 6 + i) * 64 + lane]; }
  for (int i = 0; i < 2; ++i) { b[i] = lds[(wave * 12 + 4 + i) * 64 + lane]; b2[i] = lds[(wave * 12 + 10 + i) * 64 + lane]; }
  f32x16 acc32[8];
  f32x4 acc16[32];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
  for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc16[i][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {   // one iteration = a k32 step of the 128 x 64 tile = 524288 flop... (2*128*64*32)
    if constexpr (LDS) {
      const int o = (it & 1) * 0;   // same addresses (the data does not matter, the reads do)
      for (int i = 0; i < 4; ++i) { a[i] = lds[(wave * 12 + i) * 64 + lane + o]; a2[i] = lds[(wave * 12 + 6 + i) * 64 + lane + o]; }
      for (int i = 0; i < 2; ++i) { b[i] = lds[(wave * 12 + 4 + i) * 64 + lane + o]; b2[i] = lds[(wave * 12 + 10 + i) * 64 + lane + o]; }
      asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]));
      asm volatile("" : "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(b2[0]), "+v"(b2[1]));
    }
    if constexpr (SHAPE == 0) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          acc32[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m]), __builtin_bit_cast(bf16x8, b[n]), acc32[m * 2 + n], 0, 0, 0);
          acc32[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a2[m]), __builtin_bit_cast(bf16x8, b2[n]), acc32[m * 2 + n], 0, 0, 0);
        }
    } else if constexpr (SHAPE == 1) {   // 8 row blocks of 16 (a, a2 = 8 fragments), 4 column blocks (b, b2)
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc16[m * 4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, m < 4 ? a[m] : a2[m - 4]),
                                                                     __builtin_bit_cast(bf16x8, n < 2 ? b[n] : b2[n - 2]), acc16[m * 4 + n], 0, 0, 0);
    } else {   // 16x16x16: two k16 halves per k32 step
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const u32x4 av = m < 4 ? a[m] : a2[m - 4], bv = n < 2 ? b[n] : b2[n - 2];
          const unsigned alo[2] = {av.x, av.y}, ahi[2] = {av.z, av.w}, blo[2] = {bv.x, bv.y}, bhi[2] = {bv.z, bv.w};
          acc16[m * 4 + n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, alo), __builtin_bit_cast(s16x4, blo), acc16[m * 4 + n], 0, 0, 0);
          acc16[m * 4 + n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, ahi), __builtin_bit_cast(s16x4, bhi), acc16[m * 4 + n], 0, 0, 0);
        }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc32[i][r];
  for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc16[i][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0 && blockIdx.x == 17) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int SHAPE, bool LDS>
void run(float* out, unsigned long long* clk, const u32x4* src, const char* what) {
  const int iters = 40000, wgs = 256;
  hipLaunchKernelGGL((probe<SHAPE, LDS>), dim3(wgs), dim3(256), 0, 0, out, 2000, clk, src);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f; unsigned long long h[2] = {0, 0};
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<SHAPE, LDS>), dim3(wgs), dim3(256), 0, 0, out, iters, clk, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) { best = ms; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }
  }
  const double fl = (double)wgs * 4 * iters * 2.0 * 128 * 64 * 32;
  printf("%-28s %8.3f ms  %7.1f TFLOP/s  shader clock %.3f GHz  cycles per k32 step of 128x64 per wave %.1f (512 = back to back)\n", what, best,
         fl / best / 1e9, (double)h[0] / (double)h[1] * 0.1, (double)h[0] / iters);
}

int main() {
  float* out; unsigned long long* clk; u32x4* src;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 16); hipMalloc(&src, 6 * 64 * 4 * 16);
  unsigned short hsrc[6 * 64 * 4 * 8];
  unsigned s = 12345u;
  for (auto& v : hsrc) { s = s * 1664525u + 1013904223u; const float f = ((int)(s >> 8) % 2001 - 1000) / 1000.0f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
  for (int round = 0; round < 2; ++round) {
    run<0, false>(out, clk, src, "32x32x16 registers");
    run<1, false>(out, clk, src, "16x16x32 registers");
    run<2, false>(out, clk, src, "16x16x16 registers");
    run<0, true>(out, clk, src, "32x32x16 LDS re-read");
    run<1, true>(out, clk, src, "16x16x32 LDS re-read");
    run<2, true>(out, clk, src, "16x16x16 LDS re-read");
  }
  return 0;
}
