// probe: is v_mfma_f32_16x16x32_bf16 symmetric under swapping its operands, bit for bit?  D = A x B (A: 16 x 32, B: 32 x 16) against
// D' = B^T x A^T: element (m, n) of D and element (n, m) of D' are the same 32 products -- does the hardware sum them in the same order?
// (conv_first_fused_bwd.hip recomputes layer 0's output with the pixel as the ROW of the result where the forward pass has it as the column.)
//   hipcc --offload-arch=gfx950 -O3 mfma_swap.hip -o mfma_swap && ./mfma_swap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// a: [16][32] row-major (m, k), b: [16][32] row-major (n, k)  (both "K-major per row"); d[m][n], dt[n][m]
__global__ void probe(const __bf16* a, const __bf16* b, float* d, float* dt, int nk) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  bf16x8 av, bv;
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * g + j;
    av[j] = k < nk ? a[r * 32 + k] : (__bf16)0.f;
    bv[j] = k < nk ? b[r * 32 + k] : (__bf16)0.f;
  }
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, z, 0, 0, 0);    // rows m = 4 g + i (of a), column n = r (of b)
  const f32x4 y = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, av, z, 0, 0, 0);    // rows n = 4 g + i (of b), column m = r (of a)
  for (int i = 0; i < 4; ++i) {
    d[(4 * g + i) * 16 + r] = x[i];
    dt[(4 * g + i) * 16 + r] = y[i];
  }
}
int main() {
  __bf16 ha[512], hb[512], *da, *db;
  float hd[256], hdt[256], *dd, *ddt;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, sizeof(hd)); hipMalloc(&ddt, sizeof(hdt));
  long long diff = 0, total = 0;
  srand(1);
  for (int trial = 0; trial < 2000; ++trial) {
    const int mode = trial % 3;   // 0: uint8 pixels x small weights, 12 taps; 1: random normal-ish, K = 32; 2: wide dynamic range
    for (int i = 0; i < 512; ++i) {
      const float u = (float)rand() / RAND_MAX, v = (float)rand() / RAND_MAX;
      ha[i] = (__bf16)(mode == 0 ? (float)(rand() % 256) : mode == 1 ? (u - 0.5f) * 4.f : (u - 0.5f) * expf(20.f * (v - 0.5f)));
      hb[i] = (__bf16)(mode == 0 ? (v - 0.5f) * 0.7f : mode == 1 ? (v - 0.5f) * 4.f : (v - 0.5f) * expf(20.f * (u - 0.5f)));
    }
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd, ddt, mode == 0 ? 12 : 32);
    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost); hipMemcpy(hdt, ddt, sizeof(hdt), hipMemcpyDeviceToHost);
    for (int m = 0; m < 16; ++m)
      for (int n = 0; n < 16; ++n) {
        ++total;
        if (memcmp(&hd[m * 16 + n], &hdt[n * 16 + m], 4) != 0) ++diff;
      }
  }
  printf("elements compared %lld, different bits %lld\n", total, diff);
  return 0;
}
