// probe: what does a CU's STORE stream cost the LDS-DMA loads issued beside it, and how many wavefronts does it take to saturate
// the chip's write bandwidth?  One workgroup per CU (256): W store wavefronts, each issuing `n_st` 1-KB stores (buffer_store_dwordx4,
// 64 lanes x 16 B contiguous, a private streamed region per wavefront) with a pause of `pace` ticks between stores; L load
// wavefronts, each issuing `n_ld` LDS-DMA pieces (1 KB each) from an L2-resident 64 KB source with a counted vmcnt(8).
//   hipcc --offload-arch=gfx950 -O3 store_rate.hip -o store_rate && ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const char* src, char* dst, unsigned long long bytes, int W, int n_st, int pace, int n_ld, unsigned long long* out) {
  extern __shared__ u32x4 lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < W) {
    const unsigned long long region = (unsigned long long)n_st * 1024ull;
    const unsigned long long a = reinterpret_cast<unsigned long long>(dst) + (((unsigned long long)blockIdx.x * W + wave) * region) % (bytes - region);
    const i32x4 rs = {(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)0x7FFFFFFF, 0x00020000};
    u32x4 v = {1u, 2u, 3u, (unsigned)lane};
    unsigned long long tn = t0;
    for (int it = 0; it < n_st; ++it) {
      if (pace > 0) {
        tn += (unsigned long long)pace;
        while (__builtin_amdgcn_s_memtime() < tn) __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(v), "v"(lane * 16), "s"(rs), "s"((unsigned)it * 1024u) : "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();   // all stores ISSUED
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[(blockIdx.x * nw + wave) * 2] = t1 - t0; out[(blockIdx.x * nw + wave) * 2 + 1] = t2 - t0; }
  } else {
    const unsigned long long a = reinterpret_cast<unsigned long long>(src);
    const i32x4 rs = {(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)0x7FFFFFFF, 0x00020000};
    const int lw = wave - W;
    for (int it = 0; it < n_ld / 8; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" ::"v"(lane * 16), "s"((unsigned)(lw * 8192 + k * 1024)), "s"(rs), "s"((unsigned)(lw * 8192 + k * 1024)) : "memory");
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[(blockIdx.x * nw + wave) * 2] = t1 - t0; out[(blockIdx.x * nw + wave) * 2 + 1] = t1 - t0; }
  }
}

int main() {
  const unsigned long long bytes = 3ull << 30;
  char *src, *dst;
  unsigned long long* out;
  hipMalloc(&src, 1 << 20);
  hipMalloc(&dst, bytes);
  hipMemset(src, 1, 1 << 20);
  hipMalloc(&out, 256 * 16 * 16);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int total_kb_per_cu = 6400;   // 1.6 GB over 256 CUs (the 128-channel stride-2 data gradient writes this much)
  printf("W store waves, pace (ticks between a wave's stores), L load waves | wall ms, store TB/s | ticks until issued / acknowledged per store wave | load waves: ticks per piece and wave, B/tick/CU\n");
  for (int L : {0, 3})
    for (int W : {1, 2, 4})
      for (int pace : {0, 100, 200, 400, 800}) {
        const int n_st = total_kb_per_cu / W;
        const int n_ld = 6400;   // pieces per load wave
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(probe, dim3(256), dim3(64 * (W + L)), 64 * 1024, 0, src, dst, bytes, W, n_st, pace, n_ld, out);
          hipEventRecord(e1);
          hipDeviceSynchronize();
          hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<unsigned long long> h(256 * (W + L) * 2);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double si = 0, sa = 0, ld = 0;
        for (int b = 0; b < 256; ++b)
          for (int w = 0; w < W + L; ++w) {
            if (w < W) { si += (double)h[(b * (W + L) + w) * 2]; sa += (double)h[(b * (W + L) + w) * 2 + 1]; }
            else ld += (double)h[(b * (W + L) + w) * 2];
          }
        si /= 256.0 * W; sa /= 256.0 * W;
        if (L) ld /= 256.0 * L;
        const double store_ms = sa / (si > 0 ? 1 : 1);
        (void)store_ms;
        printf("W=%d pace=%3d L=%d | %.3f ms | issued %9.0f acked %9.0f ticks -> %.2f B/tick/CU", W, pace, L, ms, si, sa, (double)total_kb_per_cu * 1024.0 / sa);
        if (L) printf(" | loads: %.0f ticks per piece and wave, %.1f B/tick/CU", ld / n_ld, (double)L * n_ld * 1024.0 / ld);
        printf("\n");
      }
  return 0;
}
