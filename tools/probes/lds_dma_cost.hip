// probe: what does an LDS-DMA piece (buffer_load_dwordx4 ... lds: 64 lanes x 16 bytes) cost the LDS, compared with the same kilobyte through
// registers (global_load_dwordx4 + ds_write_b128)?  One workgroup of 16 wavefronts per CU: wavefronts 0-7 read LDS (ds_read_b128, sixteen
// independent reads per step), wavefronts 8-15 stream kilobytes from a 64 KB (L2-resident) buffer into another LDS region, sixteen pieces in
// flight each.  Fixed work per role; the kernel is timed with events for: readers alone, writers alone (DMA / registers), both together.
// If the two roles share a bottleneck, "both" approaches the SUM of the two; if they do not, the larger of the two.
//   hipcc --offload-arch=gfx950 -O3 lds_dma_cost.hip -o lds_dma_cost && ./lds_dma_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(i32x4 rs, unsigned lds_addr, int voff) {
  lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(voff), "s"(lds_addr), "s"(rs) : "memory");
}

// roles: bit 0 = readers work, bit 1 = writers work; path: 0 = LDS-DMA, 1 = through registers
__global__ __launch_bounds__(1024) void probe(const u32x4* __restrict__ src, unsigned* sink, int roles, int path, int reader_iters, int writer_iters) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];   // [0, 32 KB): the readers' region; [32 KB, 160 KB): the writers' (16 KB per wavefront)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2048; i += 1024) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  if (wave < 8) {
    if (!(roles & 1)) return;
    u32x4 a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = u32x4{0u, 0u, 0u, 0u};
    for (int it = 0; it < reader_iters; ++it) {
#pragma unroll
      for (int k = 0; k < 16; ++k) a[k] ^= lds[((it + k) & 31) * 64 + lane];
    }
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += a[k].x + a[k].w;
    if (acc == 0x12345u) sink[0] = acc;
  } else {
    if (!(roles & 2)) return;
    const unsigned long long base = reinterpret_cast<unsigned long long>(src + (size_t)blockIdx.x * 4096);   // 64 KB per workgroup
    const i32x4 rs = {__builtin_amdgcn_readfirstlane((int)(unsigned)base), __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xFFFFu)), 65536, 0x00020000};
    const int w = wave - 8;
    const unsigned wbase = 32768u + (unsigned)w * 16384u;
    for (int it = 0; it < writer_iters; ++it) {
      if (path == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) dma16(rs, wbase + (unsigned)(k * 1024), ((((it + w) * 16 + k) & 63) * 1024 + lane * 16));
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      } else {
        u32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = src[(size_t)blockIdx.x * 4096 + (((it + w) * 16 + k) & 63) * 64 + lane];
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[2048 + w * 1024 + k * 64 + lane] = v[k];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lds[2048 + w * 1024 + lane].x == 0x87654321u) sink[1] = 1;
  }
}

static float run(const u32x4* src, unsigned* sink, int roles, int path, int ri, int wi) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 163840, 0, src, sink, roles, path, ri, wi);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  u32x4* src; unsigned* sink;
  if (hipMalloc(&src, (size_t)256 * 65536) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  (void)hipMemset(src, 1, (size_t)256 * 65536);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess) { printf("attr failed\n"); return 1; }
  const int ri = 4000, wi = 1500;
  const double rkb = 8.0 * 16 * ri, wkb = 8.0 * 16 * wi;   // kilobytes per CU
  const float r = run(src, sink, 1, 0, ri, wi);
  printf("readers alone                  %.3f ms  (%.0f KB per CU: %.1f B/cycle/CU at 2.1 GHz)\n", r, rkb, rkb * 1024 / (r * 1e-3 * 2.1e9));
  for (int path = 0; path < 2; ++path) {
    const float w = run(src, sink, 2, path, ri, wi), b = run(src, sink, 3, path, ri, wi);
    printf("%s writers alone  %.3f ms  (%.0f KB per CU: %.1f B/cycle/CU) | both %.3f ms  (sum %.3f, max %.3f)\n", path == 0 ? "LDS-DMA      " : "register-path", w, wkb,
           wkb * 1024 / (w * 1e-3 * 2.1e9), b, r + w, r > w ? r : w);
  }
  return 0;
}
