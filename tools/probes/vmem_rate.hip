// probe: how fast does ONE CU move bytes through its vector-memory path?  W wavefronts per workgroup (one workgroup per CU,
// every CU busy) each issue `iters` x 8 LDS-DMA pieces (buffer_load_dwordx4 ... lds, 1 KB each) from an L2-resident source
// (mode 0), from a streamed source (mode 1: HBM), or 16-byte stores to a streamed destination (mode 2).  Prints bytes per
// cycle and CU from s_memtime.   hipcc --offload-arch=gfx950 -O3 vmem_rate.hip -o vmem_rate && ./vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const char* src, char* dst, unsigned long long bytes, int iters, int mode, unsigned long long* out) {
  extern __shared__ u32x4 lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const unsigned long long a = reinterpret_cast<unsigned long long>(mode == 2 ? dst : src);
  const i32x4 rs = {(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)0x7FFFFFFF, 0x00020000};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // L2-resident: every workgroup re-reads the same 64 KB; streamed: a private 8 KB stride walk per (workgroup, wave, iteration)
  unsigned base = mode == 0 ? (unsigned)(wave * 8192) : (unsigned)(((unsigned long long)blockIdx.x * nw + wave) * (unsigned long long)iters * 8192ull % (bytes - (1u << 20)));
  u32x4 v = {1u, 2u, 3u, (unsigned)lane};
  for (int it = 0; it < iters; ++it) {
    const unsigned so = mode == 0 ? base : base + (unsigned)it * 8192u;
    if (mode != 2) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" ::"v"(lane * 16), "s"((unsigned)(wave * 8192 + k * 1024)), "s"(rs), "s"(so + (unsigned)k * 1024u) : "memory");
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(v), "v"(lane * 16), "s"(rs), "s"(so + (unsigned)k * 1024u) : "memory");
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * nw + wave] = t1 - t0;
}

int main() {
  const unsigned long long bytes = 1ull << 31;
  char *src, *dst;
  unsigned long long* out;
  hipMalloc(&src, bytes);
  hipMalloc(&dst, bytes);
  hipMemset(src, 1, bytes);
  hipMalloc(&out, 256 * 16 * 8);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  const char* names[3] = {"LDS-DMA, L2-resident source", "LDS-DMA, streamed source", "16-byte stores, streamed"};
  for (int mode = 0; mode < 3; ++mode)
    for (int W : {1, 2, 4, 8, 16}) {
      const int iters = 200;
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(64 * W), W * 8192, 0, src, dst, bytes, iters, mode, out);
        hipDeviceSynchronize();
      }
      std::vector<unsigned long long> h(256 * W);
      hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
      double mean = 0;
      for (auto x : h) mean += (double)x;
      mean /= h.size();
      printf("%-32s W=%2d: %8.0f ticks per wave for %d KB -> %.1f B/tick/CU (%.0f ticks per piece and wave)\n", names[mode], W, mean, iters * 8,
             (double)W * iters * 8192.0 / mean, mean / (iters * 8));
    }
  return 0;
}
