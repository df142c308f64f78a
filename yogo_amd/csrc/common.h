// Shared helpers for the yogo_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

#define YOGO_OK 0
#define YOGO_ERR_ARG 1
#define YOGO_ERR_HIP 2
#define YOGO_ERR_WORKSPACE 3

// defined in api_common.hip
extern "C" const char* yogo_hip_last_error(void);
void yogo_set_error(const char* fmt, ...);
// launch log (api_common.hip): one line per kernel launch while yogo_hip_launch_log(1) is in effect
bool yogo_launch_log_enabled();
void yogo_launch_log(const char* fmt, ...);

#define YOGO_CHECK_ARG(cond, ...)                 \
  do {                                            \
    if (!(cond)) {                                \
      yogo_set_error(__VA_ARGS__);                \
      return YOGO_ERR_ARG;                        \
    }                                             \
  } while (0)

#define YOGO_CHECK_LAUNCH(name)                                               \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      yogo_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
      return YOGO_ERR_HIP;                                                    \
    }                                                                         \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }
// The kernels divide by run-time constants with n / d = __umulhi(n, m), m = ceil(2^32 / d): exact for every 0 <= n <= nmax iff
// nmax * (m * d - 2^32) < 2^32.  Planners / eligibility checks call this with the largest dividend a launch can produce and leave the
// launch to the tiled kernel otherwise (a very large batch of small images is where it could fail).
static inline bool magic_div_exact(long long nmax, int d) {
  if (d <= 1 || nmax <= 0) return true;
  const unsigned long long m = ((1ull << 32) + (unsigned)d - 1ull) / (unsigned)d;
  const unsigned long long e = m * (unsigned long long)d - (1ull << 32);
  return e == 0 || (unsigned long long)nmax < ((1ull << 32) + e - 1ull) / e;
}

#define ACT_NONE 0
#define ACT_LEAKY 1
#define ACT_SILU 2
#define LEAKY_SLOPE 0.01f

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == ACT_LEAKY) return v > 0.f ? v : LEAKY_SLOPE * v;
  if (act == ACT_SILU) return v / (1.f + __expf(-v));
  return v;
}
// derivative of the activation; `ref` is the block OUTPUT for leaky (sign-preserving), the PRE-activation for silu
__device__ __forceinline__ float act_bwd_factor(float ref, int act) {
  if (act == ACT_LEAKY) return ref > 0.f ? 1.f : LEAKY_SLOPE;
  if (act == ACT_SILU) {
    float s = 1.f / (1.f + __expf(-ref));
    return s * (1.f + ref * (1.f - s));
  }
  return 1.f;
}

// Sum over the 64 lanes with DPP adds (no LDS traffic): quads, half rows, rows, rows 0->1 / 2->3, rows 0..1 -> 3; lane 63
// then holds the total, which is broadcast through a scalar register.  Call it from wave-uniform control flow only.
__device__ __forceinline__ float wave_sum(float v) {
#define YOGO_DPP_ADD(CTRL, ROWMASK) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
  YOGO_DPP_ADD(0xB1, 0xf)   // quad_perm [1,0,3,2]
  YOGO_DPP_ADD(0x4E, 0xf)   // quad_perm [2,3,0,1]
  YOGO_DPP_ADD(0x141, 0xf)  // row_half_mirror
  YOGO_DPP_ADD(0x140, 0xf)  // row_mirror
  YOGO_DPP_ADD(0x142, 0xa)  // row_bcast:15 -> rows 1, 3
  YOGO_DPP_ADD(0x143, 0xc)  // row_bcast:31 -> rows 2, 3
#undef YOGO_DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
