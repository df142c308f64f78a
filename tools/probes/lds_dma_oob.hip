// probe: does an out-of-range lane of `buffer_load_dwordx4 ... lds` write zeros to LDS or leave the old bytes? (experiments only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const u32x4* src, u32x4* out, int nvalid) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[128];
  lds[threadIdx.x] = u32x4{0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu};
  lds[threadIdx.x + 64] = u32x4{0xBBBBBBBBu, 0xBBBBBBBBu, 0xBBBBBBBBu, 0xBBBBBBBBu};
  __syncthreads();
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, nvalid * 16, 0x00020000);
  const int voff = (threadIdx.x & 1) ? (int)0x80000000u : (int)threadIdx.x * 16;  // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  __syncthreads();
  out[threadIdx.x] = lds[threadIdx.x];
  out[threadIdx.x + 64] = lds[threadIdx.x + 64];
}
int main() {
  u32x4 h[64], *d, *o, r[128];
  for (int i = 0; i < 64; ++i) h[i] = u32x4{(unsigned)i + 1, 2, 3, 4};
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, 40);  // lanes >= 40 are beyond num_records as well
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int i = 0; i < 48; i += 1) printf("lane %2d: %08x %08x\n", i, r[i][0], r[i][1]);
  printf("second KiB untouched: %08x\n", r[64][0]);
  return 0;
}
