// probe: what does a plain streaming kernel reach on this chip for the read : write mixes of the memory-bound layers (2.45 GB per launch)?
// Persistent grid, 16 bytes per lane, four independent loads in flight per wavefront, non-temporal or cached stores.
//   hipcc --offload-arch=gfx950 -O3 stream_mix.hip -o stream_mix && ./stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// reads NR kilobyte-units and writes NW per step (a step = one "pixel group"): NR, NW in {0..3}
template <int NR, int NW, bool NT>
__global__ __launch_bounds__(256) void mix(const u32x4* __restrict__ src, u32x4* __restrict__ dst, unsigned long long nsteps, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const unsigned long long w = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (unsigned long long)gridDim.x * 4;
  unsigned acc = 0;
  for (unsigned long long k = w; k < nsteps; k += nw) {
    u32x4 v[NR > 0 ? NR : 1];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = src[(k * NR + r) * 64 + lane];
    u32x4 o = {(unsigned)k, 1u, 2u, 3u};
#pragma unroll
    for (int r = 0; r < NR; ++r) o += v[r];
    if (NW == 0) acc += o.x ^ o.y ^ o.z ^ o.w;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      if (NT) __builtin_nontemporal_store(o, &dst[(k * NW + q) * 64 + lane]);
      else dst[(k * NW + q) * 64 + lane] = o;
      o.x += 1u;
    }
  }
  if (NW == 0 && acc == 0x12345678u) sink[0] = acc;
}

template <int NR, int NW, bool NT>
void run(const char* name, u32x4* a, u32x4* b, unsigned* sink) {
  const double total = 2.45e9;
  const unsigned long long nsteps = (unsigned long long)(total / (1024.0 * (NR + NW)));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int grid : {512, 1024, 2048}) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL((mix<NR, NW, NT>), dim3(grid), dim3(256), 0, 0, a, b, nsteps, sink);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-28s grid %4d | %.3f ms | %.2f TB/s\n", name, grid, best, (double)nsteps * 1024.0 * (NR + NW) / best / 1e9);
  }
}
int main() {
  u32x4 *a, *b; unsigned* sink;
  if (hipMalloc(&a, 2600ull << 20) != hipSuccess || hipMalloc(&b, 2600ull << 20) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  (void)hipMemset(a, 1, 2600ull << 20);
  run<1, 0, false>("read only", a, b, sink);
  run<0, 1, false>("write only (cached)", a, b, sink);
  run<0, 1, true>("write only (non-temporal)", a, b, sink);
  run<1, 2, false>("read 1 : write 2 (cached)", a, b, sink);
  run<1, 2, true>("read 1 : write 2 (nt)", a, b, sink);
  run<2, 1, false>("read 2 : write 1 (cached)", a, b, sink);
  run<2, 1, true>("read 2 : write 1 (nt)", a, b, sink);
  run<3, 0, false>("read 3 streams", a, b, sink);
  return 0;
}
