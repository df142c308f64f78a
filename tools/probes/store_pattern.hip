// probe: what do the LANE ORDER and the PLANE SPREAD of the direct kernels' output stores cost?  256 x 4 workgroups of 4 wavefronts write
// 1.6 GB in 1-KB stores (buffer_store_dwordx4, 64 lanes x 16 B filling one contiguous kilobyte):
//   order 0: lane j writes bytes [16 j, 16 j + 16)                      (natural)
//   order 1: lane (l31, half) writes unit 2 l31 + half                  (the two half-waves interleave: what conv_bf16_direct.hip stores)
//   order 2: lanes 0-31 write 512 contiguous bytes, lanes 32-63 another 512 in the NEXT plane (the tiled kernel's forward epilogue)
//   order 3: lane j writes 16 bytes every 32 (two stores fill a 2 KB span: the half-filled stores the direct kernels started with)
//   planes P: a wavefront's consecutive stores go round P regions 'plane_stride' apart (the channel-block / row planes of a tile),
//             and the NEXT wavefront continues each region where this one stopped (tiles are consecutive inside a plane)
//   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void probe(char* dst, int order, int planes, unsigned long long plane_stride, int tiles_total) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long a = reinterpret_cast<unsigned long long>(dst);
  const i32x4 rs = {(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)0xFFFFFFFF, 0x00020000};
  const int voff = order == 3 ? lane * 32 : (order == 2 ? l31 * 16 : (order == 0 ? lane : 2 * l31 + half) * 16);
  const u32x4 v = {1u, 2u, 3u, (unsigned)lane};
  const int nw = gridDim.x * 4;
  for (int tile = blockIdx.x * 4 + wave; tile < tiles_total; tile += nw) {
    for (int pl = 0; pl < planes; ++pl) {
      unsigned long long off = (unsigned long long)pl * plane_stride + (unsigned long long)tile * 1024ull;
      int vo = voff;
      if (order == 2) {   // plane pair (pl, pl ^ 1): this store the first 512 bytes of the tile's kilobyte in plane pl (lower half-wave) and in plane pl ^ 1 (upper)
        off = (unsigned long long)(pl & ~1) * plane_stride + (unsigned long long)tile * 1024ull + (pl & 1) * 512ull;
        vo = voff + half * (int)plane_stride;
      }
      if (order == 3) {   // two stores fill 2 KB: tile pair (tile & ~1), this store the units of parity (tile & 1)
        off = (unsigned long long)pl * plane_stride + (unsigned long long)(tile & ~1) * 1024ull + (tile & 1) * 16ull;
      }
      const i32x4 rp = {(int)(unsigned)(a + off), (int)((unsigned)((a + off) >> 32) & 0xFFFFu), (int)0x7FFFFFFF, 0x00020000};
      asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" ::"v"(v), "v"(vo), "s"(rp) : "memory");
    }
  }
  (void)rs;
}

int main() {
  const unsigned long long bytes = 1664ull << 20;
  char* dst;
  hipMalloc(&dst, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int planes : {2, 8, 16})
    for (int order : {0, 1, 2, 3}) {
      const unsigned long long plane_stride = (bytes / planes) & ~1023ull;
      const int tiles_total = (int)(plane_stride / 1024ull);
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, dst, order, planes, plane_stride, tiles_total);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("planes=%2d order=%d | %.3f ms | %.2f TB/s\n", planes, order, ms, (double)planes * tiles_total * 1024.0 / ms / 1e9);
    }
  return 0;
}
