// BatchNorm2d (training + eval) around the conv kernels, and the small deterministic reductions the backward
// pass needs.  Replaces ATen batch_norm / batch_norm_backward behind nn.BatchNorm2d(…) in the reference
// (yogo/model_defns.py:35,55,60).
//
// Training statistics: the conv epilogues write per-workgroup partial (sum, sumsq) rows; bn_finalize adds them
// in fp64 in a fixed order (deterministic), producing mean / invstd and the momentum-0.1 running-stat update
// (biased variance to normalise, unbiased into running_var -- torch semantics).
// All element-wise kernels here are HBM-bound: one read + one write per element, float4 where the plane allows.
#include "common.h"

// ---- finalize: partial rows -> mean, invstd, running stats ---------------------------------------------------
// One workgroup per 8 channels: lane t sums the rows t / 8, t / 8 + 32, ... of channel 8 * blockIdx.x + t % 8 (8 adjacent
// channels = 64 contiguous bytes of a row), then the 32 row slices are folded in a fixed order through LDS.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int rows, int row_stride, int C,
                                                          double count, float eps, float momentum,
                                                          float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          long long* __restrict__ num_batches_tracked) {
  __shared__ double sh[2][32][8];
  const int cl = threadIdx.x & 7, slice = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double s = 0.0, q = 0.0;
  if (c < C) {
    // (eight loads in flight, added in the same order: the chain of dependent load -> add steps was what the 24 us of this
    //  16-workgroup kernel were)
    int r = slice;
    for (; r + 7 * 32 < rows; r += 8 * 32) {
      float2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float2*>(part + ((size_t)(r + j * 32) * row_stride + c) * 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s += (double)v[j].x;
        q += (double)v[j].y;
      }
    }
    for (; r < rows; r += 32) {
      const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)r * row_stride + c) * 2);
      s += (double)v.x;
      q += (double)v.y;
    }
  }
  sh[0][slice][cl] = s;
  sh[1][slice][cl] = q;
  __syncthreads();
  if (threadIdx.x < 8 && c < C) {
    s = 0.0;
    q = 0.0;
    for (int k = 0; k < 32; ++k) {
      s += sh[0][k][cl];
      q += sh[1][k][cl];
    }
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_out[c] = (float)mean;
    invstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean != nullptr) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
    if (c == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;
  }
}

// ---- apply: y = act((z - mean) * (invstd * gamma) + beta) ------------------------------------------------------
// stat_is_var != 0: `invstd_or_var` holds a variance (eval mode: running_var) and invstd is computed here.
template <bool VEC4>
__global__ __launch_bounds__(256) void bn_apply_act_kernel(const float* __restrict__ z, float* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd_or_var,
                                                           int stat_is_var, float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int C, int HW, int act) {
  const int plane = blockIdx.y;
  const int c = plane % C;
  const float mu = mean[c];
  const float is = stat_is_var ? 1.0f / sqrtf(invstd_or_var[c] + eps) : invstd_or_var[c];
  const float sc = is * gamma[c];
  const float sh = beta[c];
  const float* zp = z + (size_t)plane * HW;
  float* yp = y + (size_t)plane * HW;
  if (VEC4) {
    const int n4 = HW >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
      float4 v = reinterpret_cast<const float4*>(zp)[i];
      v.x = act_fwd(fmaf(v.x - mu, sc, sh), act);
      v.y = act_fwd(fmaf(v.y - mu, sc, sh), act);
      v.z = act_fwd(fmaf(v.z - mu, sc, sh), act);
      v.w = act_fwd(fmaf(v.w - mu, sc, sh), act);
      reinterpret_cast<float4*>(yp)[i] = v;
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) yp[i] = act_fwd(fmaf(zp[i] - mu, sc, sh), act);
  }
}

// ---- backward reduce: per (plane chunk) partial sums of g and g * xhat -------------------------------------------
// g is the gradient w.r.t. the block output; when act != none the activation derivative is applied here, from the
// recomputed pre-activation xhat*gamma + beta (so no pre-activation tensor has to be saved, LeakyReLU and SiLU alike).
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int act, float* __restrict__ part, int C, int HW) {
  __shared__ float sh[2][4];
  const int plane = blockIdx.y;  // b*C + c
  const int c = plane % C;
  const int b = plane / C;
  const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
  const float* gp = g + (size_t)plane * HW;
  const float* zp = z + (size_t)plane * HW;
  float s = 0.f, q = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const float xh = (zp[i] - mu) * is;
    const float gv = gp[i] * act_bwd_factor(fmaf(xh, ga, be), act);
    s += gv;
    q += gv * xh;
  }
  s = wave_sum(s);
  q = wave_sum(q);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh[0][wave] = s;
    sh[1][wave] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* dst = part + ((size_t)(b * gridDim.x + blockIdx.x) * C + c) * 2;
    dst[0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    dst[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

// dbeta = sum g, dgamma = sum g*xhat (fp64, fixed order), optionally clamped (the reference's per-parameter hook)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int rows, int C, float clip,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ sum_g, float* __restrict__ sum_gx) {
  __shared__ double sh[2][4];
  const int c = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const float* src = part + ((size_t)r * C + c) * 2;
    s += (double)src[0];
    q += (double)src[1];
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh[0][wave] = s;
    sh[1][wave] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    q = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    sum_g[c] = (float)s;
    sum_gx[c] = (float)q;
    float db = (float)s, dg = (float)q;
    if (clip > 0.f) {
      db = fminf(fmaxf(db, -clip), clip);
      dg = fminf(fmaxf(dg, -clip), clip);
    }
    dbeta[c] = db;
    dgamma[c] = dg;
  }
}

// dz = gamma * invstd * (g - sum_g/n - xhat * sum_gx/n)   (training);  dz = gamma * invstd * g  (frozen stats)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                           float* __restrict__ dz, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int act,
                                                           const float* __restrict__ sum_g, const float* __restrict__ sum_gx,
                                                           float inv_count, int training, int C, int HW) {
  const int plane = blockIdx.y;
  const int c = plane % C;
  const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c], gi = ga * is;
  const float mg = training ? sum_g[c] * inv_count : 0.f;
  const float mgx = training ? sum_gx[c] * inv_count : 0.f;
  const float* gp = g + (size_t)plane * HW;
  const float* zp = z + (size_t)plane * HW;
  float* dp = dz + (size_t)plane * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const float xh = (zp[i] - mu) * is;
    const float gv = gp[i] * act_bwd_factor(fmaf(xh, ga, be), act);
    dp[i] = gi * (gv - mg - xh * mgx);
  }
}

// eval-mode helper: invstd = 1/sqrt(var + eps)
__global__ void bn_invstd_kernel(const float* __restrict__ var, float eps, float* __restrict__ invstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) invstd[c] = 1.0f / sqrtf(var[c] + eps);
}

// ---- generic deterministic reductions --------------------------------------------------------------------------
// out[j] = clamp(sum_r part[r][j]) for j < N  (fp64 accumulate, fixed order)
__global__ __launch_bounds__(256) void partials_reduce_kernel(const float* __restrict__ part, int rows, int N, float clip,
                                                              float* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  double s = 0.0;
  int r = 0;   // (eight loads in flight, added in the same order)
  for (; r + 8 <= rows; r += 8) {
    float v8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v8[k] = part[(size_t)(r + k) * N + j];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += (double)v8[k];
  }
  for (; r < rows; ++r) s += (double)part[(size_t)r * N + j];
  float v = (float)s;
  if (clip > 0.f) v = fminf(fmaxf(v, -clip), clip);
  out[j] = v;
}

// per-channel sum over (b, h, w) of an NCHW tensor (bias gradients): one workgroup per channel, fixed order
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ g, int B, int C, int HW, float clip,
                                                          float* __restrict__ out) {
  __shared__ double sh[4];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int b = 0; b < B; ++b) {
    const float* gp = g + ((size_t)b * C + c) * HW;
    float ps = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) ps += gp[i];
    s += (double)ps;
  }
  s = wave_sum_d(s);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
    if (clip > 0.f) v = fminf(fmaxf(v, -clip), clip);
    out[c] = v;
  }
}

static inline int plane_blocks(int HW, int per_thread) { return max(1, min(64, cdiv(HW, 256 * per_thread))); }

// =========================================================================================================
// C ABI
// =========================================================================================================
extern "C" int yogo_bn_finalize(const float* part, int rows, int row_stride, int C, long long count, float eps,
                                float momentum, float* mean_out, float* invstd_out, float* running_mean,
                                float* running_var, long long* num_batches_tracked, hipStream_t stream) {
  YOGO_CHECK_ARG(part && mean_out && invstd_out && C > 0 && rows > 0 && row_stride >= C && count > 0,
                 "bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 8)), dim3(256), 0, stream, part, rows, row_stride, C, (double)count, eps,
                     momentum, mean_out, invstd_out, running_mean, running_var, num_batches_tracked);
  YOGO_CHECK_LAUNCH("bn_finalize");
  return YOGO_OK;
}

extern "C" int yogo_bn_apply_act(const float* z, float* y, const float* mean, const float* invstd_or_var, int stat_is_var,
                                 float eps, const float* gamma, const float* beta, int B, int C, int HW, int act,
                                 hipStream_t stream) {
  YOGO_CHECK_ARG(z && y && mean && invstd_or_var && gamma && beta && C > 0 && HW > 0 && B >= 0, "bn_apply_act: bad arguments");
  if (B == 0) return YOGO_OK;
  if ((HW & 3) == 0) {
    dim3 grid(plane_blocks(HW, 8), B * C);
    hipLaunchKernelGGL((bn_apply_act_kernel<true>), grid, dim3(256), 0, stream, z, y, mean, invstd_or_var, stat_is_var, eps,
                       gamma, beta, C, HW, act);
  } else {
    dim3 grid(plane_blocks(HW, 4), B * C);
    hipLaunchKernelGGL((bn_apply_act_kernel<false>), grid, dim3(256), 0, stream, z, y, mean, invstd_or_var, stat_is_var, eps,
                       gamma, beta, C, HW, act);
  }
  YOGO_CHECK_LAUNCH("bn_apply_act");
  return YOGO_OK;
}

extern "C" int yogo_bn_invstd(const float* var, float eps, float* invstd, int C, hipStream_t stream) {
  YOGO_CHECK_ARG(var && invstd && C > 0, "bn_invstd: bad arguments");
  hipLaunchKernelGGL(bn_invstd_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, var, eps, invstd, C);
  YOGO_CHECK_LAUNCH("bn_invstd");
  return YOGO_OK;
}

extern "C" int yogo_bn_bwd_rows(int B, int HW, int* rows) {
  YOGO_CHECK_ARG(rows != nullptr, "bn_bwd_rows: null");
  *rows = B * plane_blocks(HW, 8);
  return YOGO_OK;
}

// g: grad w.r.t. BN output; z: saved conv output; part: workspace rows*C*2 floats (rows from yogo_bn_bwd_rows).
// Writes dz (may alias g), dgamma, dbeta (clamped to +-clip when clip > 0).  sums: workspace 2*C floats.
extern "C" int yogo_bn_bwd(const float* g, const float* z, float* dz, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, int act, float* dgamma, float* dbeta, float* part,
                           float* sums, int B, int C, int HW, int training, float clip, hipStream_t stream) {
  YOGO_CHECK_ARG(g && z && dz && mean && invstd && gamma && beta && dgamma && dbeta && part && sums, "bn_bwd: null pointer");
  YOGO_CHECK_ARG(B > 0 && C > 0 && HW > 0, "bn_bwd: bad shape");
  const int nb = plane_blocks(HW, 8);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb, B * C), dim3(256), 0, stream, g, z, mean, invstd, gamma, beta, act, part, C, HW);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, stream, part, B * nb, C, clip, dgamma, dbeta, sums, sums + C);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nb, B * C), dim3(256), 0, stream, g, z, dz, mean, invstd, gamma, beta, act, sums,
                     sums + C, 1.0f / ((float)B * (float)HW), training, C, HW);
  YOGO_CHECK_LAUNCH("bn_bwd");
  return YOGO_OK;
}

// in-place fold: row s (s < S) <- sum of rows s, s+S, s+2S, ...; each element is read and written by one thread only
__global__ __launch_bounds__(256) void partials_fold_kernel(float* __restrict__ part, int rows, int N, int S) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int s = blockIdx.y;
  if (j >= N) return;
  double acc = 0.0;
  int r = s;   // (eight loads in flight, added in the same order)
  for (; r + 7 * S < rows; r += 8 * S) {
    float v8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v8[k] = part[(size_t)(r + k * S) * N + j];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += (double)v8[k];
  }
  for (; r < rows; r += S) acc += (double)part[(size_t)r * N + j];
  part[(size_t)s * N + j] = (float)acc;
}

// out[j] = clamp(sum_r part[r][j]).  NOTE: `part` is used as scratch (folded in place) when rows > 64.
extern "C" int yogo_partials_reduce(float* part, int rows, int N, float clip, float* out, hipStream_t stream) {
  YOGO_CHECK_ARG(part && out && rows > 0 && N > 0, "partials_reduce: bad arguments");
  if (rows > 64) {
    const int S = 64;
    hipLaunchKernelGGL(partials_fold_kernel, dim3(cdiv(N, 256), S), dim3(256), 0, stream, part, rows, N, S);
    rows = S;
  }
  hipLaunchKernelGGL(partials_reduce_kernel, dim3(cdiv(N, 256)), dim3(256), 0, stream, part, rows, N, clip, out);
  YOGO_CHECK_LAUNCH("partials_reduce");
  return YOGO_OK;
}

extern "C" int yogo_channel_sum(const float* g, int B, int C, int HW, float clip, float* out, hipStream_t stream) {
  YOGO_CHECK_ARG(g && out && B > 0 && C > 0 && HW > 0, "channel_sum: bad arguments");
  hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(256), 0, stream, g, B, C, HW, clip, out);
  YOGO_CHECK_LAUNCH("channel_sum");
  return YOGO_OK;
}

// =========================================================================================================
// bf16 NCHW8c variants (training in bf16 storage): one 16-byte unit = 8 consecutive channels of one pixel.
// A "plane" is (image, channel block); statistics stay fp32 / fp64.
// =========================================================================================================
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

struct Bn8cParams {
  float mu[8], sc[8], sh[8], ga[8], is[8];
};

__device__ __forceinline__ void bn8c_load(Bn8cParams& q, int cb, int C, const float* mean, const float* invstd_or_var,
                                          int stat_is_var, float eps, const float* gamma, const float* beta) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = min(cb * 8 + j, C - 1);
    const float is = stat_is_var ? 1.0f / sqrtf(invstd_or_var[c] + eps) : invstd_or_var[c];
    q.mu[j] = mean[c];
    q.is[j] = is;
    q.ga[j] = gamma[c];
    q.sc[j] = is * gamma[c];
    q.sh[j] = beta[c];
  }
}

__global__ __launch_bounds__(256) void bn_apply_act_8c_kernel(const u32x4_t* __restrict__ z, u32x4_t* __restrict__ y,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd_or_var, int stat_is_var, float eps,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int C, int Cb, int HW, int act) {
  const int plane = blockIdx.y;  // b*Cb + cb
  const int cb = plane % Cb;
  Bn8cParams q;
  bn8c_load(q, cb, C, mean, invstd_or_var, stat_is_var, eps, gamma, beta);
  const u32x4_t* zp = z + (size_t)plane * HW;
  u32x4_t* yp = y + (size_t)plane * HW;
  float sh2[8];  // y = z * sc + (beta - mean * sc): one FMA per value (conv_first_mfma_kernel applies the same form)
#pragma unroll
  for (int j = 0; j < 8; ++j) sh2[j] = fmaf(-q.mu[j], q.sc[j], q.sh[j]);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const bf16x8_t v = __builtin_bit_cast(bf16x8_t, zp[i]);
    bf16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float r = act_fwd(fmaf((float)v[j], q.sc[j], sh2[j]), act);
      o[j] = (cb * 8 + j < C) ? (__bf16)r : (__bf16)0.f;
    }
    yp[i] = __builtin_bit_cast(u32x4_t, o);
  }
}

// batch statistics of a stored bf16 tensor: partial (sum, sum of squares) per channel, rows = B * gridDim.x
__global__ __launch_bounds__(256) void bn_stats_8c_kernel(const u32x4_t* __restrict__ z, float* __restrict__ part, int C, int Cb,
                                                          int HW) {
  __shared__ float sh[4][16];
  const int plane = blockIdx.y;
  const int cb = plane % Cb, b = plane / Cb;
  const u32x4_t* zp = z + (size_t)plane * HW;
  float s[8], t[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = t[j] = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const bf16x8_t zv = __builtin_bit_cast(bf16x8_t, zp[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = (float)zv[j];
      s[j] += v;
      t[j] = fmaf(v, v, t[j]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a = wave_sum(s[j]), c = wave_sum(t[j]);
    if (lane == 0) {
      sh[wave][2 * j] = a;
      sh[wave][2 * j + 1] = c;
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int j = threadIdx.x >> 1;
    if (cb * 8 + j < C)
      part[((size_t)(b * gridDim.x + blockIdx.x) * C + cb * 8 + j) * 2 + (threadIdx.x & 1)] =
          sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
  }
}

// partial sums of g' and g' * xhat per channel, g' = g * act'(xhat*gamma + beta)
__global__ __launch_bounds__(256) void bn_bwd_reduce_8c_kernel(const u32x4_t* __restrict__ g, const u32x4_t* __restrict__ z,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               int act, float* __restrict__ part, int C, int Cb, int HW) {
  __shared__ float sh[4][16];
  const int plane = blockIdx.y;
  const int cb = plane % Cb, b = plane / Cb;
  Bn8cParams q;
  bn8c_load(q, cb, C, mean, invstd, 0, 0.f, gamma, beta);
  const u32x4_t* gp = g + (size_t)plane * HW;
  const u32x4_t* zp = z + (size_t)plane * HW;
  float s[8], t[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = t[j] = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const bf16x8_t gv = __builtin_bit_cast(bf16x8_t, gp[i]);
    const bf16x8_t zv = __builtin_bit_cast(bf16x8_t, zp[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = ((float)zv[j] - q.mu[j]) * q.is[j];
      const float ge = (float)gv[j] * act_bwd_factor(fmaf(xh, q.ga[j], q.sh[j]), act);
      s[j] += ge;
      t[j] += ge * xh;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a = wave_sum(s[j]), c = wave_sum(t[j]);
    if (lane == 0) {
      sh[wave][2 * j] = a;
      sh[wave][2 * j + 1] = c;
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int j = threadIdx.x >> 1;
    if (cb * 8 + j < C)
      part[((size_t)(b * gridDim.x + blockIdx.x) * C + cb * 8 + j) * 2 + (threadIdx.x & 1)] =
          sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_8c_kernel(const u32x4_t* __restrict__ g, const u32x4_t* __restrict__ z,
                                                              u32x4_t* __restrict__ dz, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int act,
                                                              const float* __restrict__ sum_g, const float* __restrict__ sum_gx,
                                                              float inv_count, int training, int C, int Cb, int HW) {
  const int plane = blockIdx.y;
  const int cb = plane % Cb;
  Bn8cParams q;
  bn8c_load(q, cb, C, mean, invstd, 0, 0.f, gamma, beta);
  float mg[8], mgx[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = min(cb * 8 + j, C - 1);
    mg[j] = training ? sum_g[c] * inv_count : 0.f;
    mgx[j] = training ? sum_gx[c] * inv_count : 0.f;
  }
  const u32x4_t* gp = g + (size_t)plane * HW;
  const u32x4_t* zp = z + (size_t)plane * HW;
  u32x4_t* dp = dz + (size_t)plane * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const bf16x8_t gv = __builtin_bit_cast(bf16x8_t, gp[i]);
    const bf16x8_t zv = __builtin_bit_cast(bf16x8_t, zp[i]);
    bf16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = ((float)zv[j] - q.mu[j]) * q.is[j];
      const float ge = (float)gv[j] * act_bwd_factor(fmaf(xh, q.ga[j], q.sh[j]), act);
      const float r = q.sc[j] * (ge - mg[j] - xh * mgx[j]);
      o[j] = (cb * 8 + j < C) ? (__bf16)r : (__bf16)0.f;
    }
    dp[i] = __builtin_bit_cast(u32x4_t, o);
  }
}

// fp32 NCHW -> bf16 NCHW8c (Cb channel blocks, padding channels zero) and back
__global__ __launch_bounds__(256) void nchw_f32_to_8c_kernel(const float* __restrict__ in, u32x4_t* __restrict__ out, int C, int Cb,
                                                             int HW) {
  const int plane = blockIdx.y;  // b*Cb + cb
  const int cb = plane % Cb, b = plane / Cb;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    bf16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cb * 8 + j;
      o[j] = (c < C) ? (__bf16)in[((size_t)b * C + c) * HW + i] : (__bf16)0.f;
    }
    out[(size_t)plane * HW + i] = __builtin_bit_cast(u32x4_t, o);
  }
}

__global__ __launch_bounds__(256) void nchw8c_to_f32_kernel(const u32x4_t* __restrict__ in, float* __restrict__ out, int C, int Cb,
                                                            int HW) {
  const int plane = blockIdx.y;
  const int cb = plane % Cb, b = plane / Cb;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const bf16x8_t v = __builtin_bit_cast(bf16x8_t, in[(size_t)plane * HW + i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cb * 8 + j;
      if (c < C) out[((size_t)b * C + c) * HW + i] = (float)v[j];
    }
  }
}

static inline int cb_of(int C) { return ((C + 15) / 16) * 2; }  // channel blocks: C padded to a multiple of 16

extern "C" int yogo_bn_apply_act_bf16(const void* z, void* y, const float* mean, const float* invstd_or_var, int stat_is_var,
                                      float eps, const float* gamma, const float* beta, int B, int C, int HW, int act,
                                      hipStream_t stream) {
  YOGO_CHECK_ARG(z && y && mean && invstd_or_var && gamma && beta && C > 0 && HW > 0 && B >= 0, "bn_apply_act_bf16: bad arguments");
  if (B == 0) return YOGO_OK;
  const int Cb = cb_of(C);
  YOGO_CHECK_ARG(B * Cb <= 65535, "bn_apply_act_bf16: batch * channel blocks exceeds 65535");
  hipLaunchKernelGGL(bn_apply_act_8c_kernel, dim3(plane_blocks(HW, 4), B * Cb), dim3(256), 0, stream,
                     reinterpret_cast<const u32x4_t*>(z), reinterpret_cast<u32x4_t*>(y), mean, invstd_or_var, stat_is_var, eps,
                     gamma, beta, C, Cb, HW, act);
  YOGO_CHECK_LAUNCH("bn_apply_act_bf16");
  return YOGO_OK;
}

extern "C" int yogo_bn_bwd_bf16_rows(int B, int HW, int* rows) {
  YOGO_CHECK_ARG(rows != nullptr, "bn_bwd_bf16_rows: null");
  *rows = B * plane_blocks(HW, 4);
  return YOGO_OK;
}

// bf16 NCHW8c g (gradient w.r.t. the block output), z (saved conv output) -> dz (may alias g), dgamma, dbeta
extern "C" int yogo_bn_bwd_bf16(const void* g, const void* z, void* dz, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, int act, float* dgamma, float* dbeta, float* part,
                                float* sums, int B, int C, int HW, int training, float clip, hipStream_t stream) {
  YOGO_CHECK_ARG(g && z && dz && mean && invstd && gamma && beta && dgamma && dbeta && part && sums, "bn_bwd_bf16: null pointer");
  YOGO_CHECK_ARG(B > 0 && C > 0 && HW > 0, "bn_bwd_bf16: bad shape");
  const int Cb = cb_of(C), nb = plane_blocks(HW, 4);
  YOGO_CHECK_ARG(B * Cb <= 65535, "bn_bwd_bf16: batch * channel blocks exceeds 65535");
  const u32x4_t* g4 = reinterpret_cast<const u32x4_t*>(g);
  const u32x4_t* z4 = reinterpret_cast<const u32x4_t*>(z);
  hipLaunchKernelGGL(bn_bwd_reduce_8c_kernel, dim3(nb, B * Cb), dim3(256), 0, stream, g4, z4, mean, invstd, gamma, beta, act, part, C,
                     Cb, HW);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, stream, part, B * nb, C, clip, dgamma, dbeta, sums, sums + C);
  hipLaunchKernelGGL(bn_bwd_apply_8c_kernel, dim3(nb, B * Cb), dim3(256), 0, stream, g4, z4, reinterpret_cast<u32x4_t*>(dz), mean,
                     invstd, gamma, beta, act, sums, sums + C, 1.0f / ((float)B * (float)HW), training, C, Cb, HW);
  YOGO_CHECK_LAUNCH("bn_bwd_bf16");
  return YOGO_OK;
}

// ---- BatchNorm backward of the block UNDER a 1x1 convolution with P <= 16 outputs (the detection head, yogo/model.py:150-155) without that
// convolution's data gradient in memory.  g[c][pixel] = sum_k gh[k][pixel] * w[k][c] is one v_mfma_f32_16x16x32_bf16 per 16 pixels x 16 channels:
// A = the weights (row = channel, K = the head's outputs), B = the head's output gradient (column = pixel; its 16-byte units are the operand
// as they lie in memory), so lane (c16, g4) gets channels 4 g4 .. 4 g4 + 3 of pixel c16 -- the half of a 16-byte unit of z it then works on.
// Both sweeps compute the same g (rounded to bf16 as the data-gradient kernel's output would have been): the 128-channel gradient tensor is
// never written or read (-0.41 GB written, -0.82 GB read per step of base_model at batch 128).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
struct BnHeadLane {
  bf16x8_t wop;                            // A operand: channel 16 cp + c16, head outputs 8 g4 .. 8 g4 + 7
  float mu[4], is[4], ga[4], sh[4], sc[4];  // channels 16 cp + 4 g4 + i
};
__device__ __forceinline__ void bn_head_lane(BnHeadLane& L, const float* __restrict__ hw, int P, int C, int cp, int c16, int g4, const float* mean,
                                             const float* invstd, const float* gamma, const float* beta) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * g4 + j;
    L.wop[j] = (k < P) ? (__bf16)hw[(size_t)k * C + 16 * cp + c16] : (__bf16)0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 16 * cp + 4 * g4 + i;
    L.mu[i] = mean[c]; L.is[i] = invstd[c]; L.ga[i] = gamma[c]; L.sh[i] = beta[c]; L.sc[i] = invstd[c] * gamma[c];
  }
}
// this lane's four gradients (bf16-rounded, widened) and pre-activations of pixel px (nothing beyond the plane: zeros)
__device__ __forceinline__ void bn_head_fetch(const BnHeadLane& L, const u32x4_t* __restrict__ ghp, const u32x2_t* __restrict__ zp2, int HW, int px, int g4,
                                              float (&gb)[4], float (&zf)[4]) {
  const bool valid = px < HW;
  u32x4_t gop = {0u, 0u, 0u, 0u};
  if (valid && g4 < 2) gop = ghp[(size_t)g4 * HW + px];
  u32x2_t zr = {0u, 0u};
  if (valid) zr = zp2[(size_t)px * 2];
  const f32x4_t d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L.wop, __builtin_bit_cast(bf16x8_t, gop), f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  const bf16x4_t zv = __builtin_bit_cast(bf16x4_t, zr);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    gb[i] = (float)(__bf16)d[i];
    zf[i] = (float)zv[i];
  }
}
__global__ __launch_bounds__(256) void bn_bwd_reduce_head_kernel(const u32x4_t* __restrict__ gh, const float* __restrict__ hw, int P,
                                                                 const u32x4_t* __restrict__ z, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int act, float* __restrict__ part, int C, int Cb, int HW) {
  __shared__ float sh[4][32];
  const int cp = blockIdx.y % (Cb / 2), b = blockIdx.y / (Cb / 2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, g4 = lane >> 4;
  BnHeadLane L;
  bn_head_lane(L, hw, P, C, cp, c16, g4, mean, invstd, gamma, beta);
  const u32x4_t* ghp = gh + (size_t)b * 2 * HW;
  const u32x2_t* zp2 = reinterpret_cast<const u32x2_t*>(z + ((size_t)b * Cb + 2 * cp + (g4 >> 1)) * HW) + (g4 & 1);
  float s[4] = {0.f, 0.f, 0.f, 0.f}, t[4] = {0.f, 0.f, 0.f, 0.f};
  for (int px0 = (blockIdx.x * 4 + wave) * 16; px0 < HW; px0 += gridDim.x * 64) {
    float gb[4], zf[4];
    bn_head_fetch(L, ghp, zp2, HW, px0 + c16, g4, gb, zf);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xh = (zf[i] - L.mu[i]) * L.is[i];
      const float ge = gb[i] * act_bwd_factor(fmaf(xh, L.ga[i], L.sh[i]), act);
      s[i] += ge;
      t[i] += ge * xh;
    }
  }
  // sums over the 16 pixels of a lane group, then the four wavefronts
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = s[i], c = t[i];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      a += __shfl_xor(a, o, 64);
      c += __shfl_xor(c, o, 64);
    }
    if (c16 == 0) {
      sh[wave][2 * (4 * g4 + i)] = a;
      sh[wave][2 * (4 * g4 + i) + 1] = c;
    }
  }
  __syncthreads();
  if (threadIdx.x < 32)
    part[((size_t)(b * gridDim.x + blockIdx.x) * C + 16 * cp + (threadIdx.x >> 1)) * 2 + (threadIdx.x & 1)] =
        sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void bn_bwd_apply_head_kernel(const u32x4_t* __restrict__ gh, const float* __restrict__ hw, int P,
                                                                const u32x4_t* __restrict__ z, u32x4_t* __restrict__ dz, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int act, const float* __restrict__ sum_g,
                                                                const float* __restrict__ sum_gx, float inv_count, int training, int C, int Cb, int HW) {
  const int cp = blockIdx.y % (Cb / 2), b = blockIdx.y / (Cb / 2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, g4 = lane >> 4;
  BnHeadLane L;
  bn_head_lane(L, hw, P, C, cp, c16, g4, mean, invstd, gamma, beta);
  float mg[4], mgx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 16 * cp + 4 * g4 + i;
    mg[i] = training ? sum_g[c] * inv_count : 0.f;
    mgx[i] = training ? sum_gx[c] * inv_count : 0.f;
  }
  const u32x4_t* ghp = gh + (size_t)b * 2 * HW;
  const size_t plane_u = ((size_t)b * Cb + 2 * cp + (g4 >> 1)) * HW;
  const u32x2_t* zp2 = reinterpret_cast<const u32x2_t*>(z + plane_u) + (g4 & 1);
  u32x2_t* dp2 = reinterpret_cast<u32x2_t*>(dz + plane_u) + (g4 & 1);
  for (int px0 = (blockIdx.x * 4 + wave) * 16; px0 < HW; px0 += gridDim.x * 64) {
    const int px = px0 + c16;
    float gb[4], zf[4];
    bn_head_fetch(L, ghp, zp2, HW, px, g4, gb, zf);
    bf16x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xh = (zf[i] - L.mu[i]) * L.is[i];
      const float ge = gb[i] * act_bwd_factor(fmaf(xh, L.ga[i], L.sh[i]), act);
      o[i] = (__bf16)(L.sc[i] * (ge - mg[i] - xh * mgx[i]));
    }
    if (px < HW) dp2[(size_t)px * 2] = __builtin_bit_cast(u32x2_t, o);
  }
}
// yogo_bn_bwd_bf16 with g computed from the head's output gradient gh (bf16 NCHW8c [B][2][HW], channels >= P zero) and the head's weights
// head_w ([P][C] fp32, OIHW of a 1x1 convolution; rounded to bf16 here as the packed operand of the head's data gradient is).  C a multiple of
// 16.  dz must not alias gh.
extern "C" int yogo_bn_bwd_bf16_head(const void* gh, const float* head_w, int P, const void* z, void* dz, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, int act, float* dgamma, float* dbeta, float* part, float* sums,
                                     int B, int C, int HW, int training, float clip, hipStream_t stream) {
  YOGO_CHECK_ARG(gh && head_w && z && dz && mean && invstd && gamma && beta && dgamma && dbeta && part && sums, "bn_bwd_bf16_head: null pointer");
  YOGO_CHECK_ARG(B > 0 && C > 0 && (C % 16) == 0 && HW > 0 && P > 0 && P <= 16, "bn_bwd_bf16_head: bad shape");
  const int Cb = cb_of(C), nb = plane_blocks(HW, 4);
  YOGO_CHECK_ARG(B * Cb <= 65535, "bn_bwd_bf16_head: batch * channel blocks exceeds 65535");
  const u32x4_t* g4 = reinterpret_cast<const u32x4_t*>(gh);
  const u32x4_t* z4 = reinterpret_cast<const u32x4_t*>(z);
  hipLaunchKernelGGL(bn_bwd_reduce_head_kernel, dim3(nb, B * (Cb / 2)), dim3(256), 0, stream, g4, head_w, P, z4, mean, invstd, gamma, beta, act, part, C, Cb,
                     HW);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, stream, part, B * nb, C, clip, dgamma, dbeta, sums, sums + C);
  hipLaunchKernelGGL(bn_bwd_apply_head_kernel, dim3(nb, B * (Cb / 2)), dim3(256), 0, stream, g4, head_w, P, z4, reinterpret_cast<u32x4_t*>(dz), mean, invstd,
                     gamma, beta, act, sums, sums + C, 1.0f / ((float)B * (float)HW), training, C, Cb, HW);
  YOGO_CHECK_LAUNCH("bn_bwd_bf16_head");
  return YOGO_OK;
}

// batch statistics of a bf16 NCHW8c tensor (the stored convolution output) as partial rows for yogo_bn_finalize:
// part [rows][C][2] with rows from yogo_bn_bwd_bf16_rows(B, HW).  Cheaper than carrying the sums through the convolution's
// epilogue when that epilogue is on the critical path of a one-workgroup-per-CU kernel.
extern "C" int yogo_bn_stats_bf16(const void* z, float* part, int B, int C, int HW, hipStream_t stream) {
  YOGO_CHECK_ARG(z && part && B > 0 && C > 0 && HW > 0, "bn_stats_bf16: bad arguments");
  const int Cb = cb_of(C), nb = plane_blocks(HW, 4);
  YOGO_CHECK_ARG(B * Cb <= 65535, "bn_stats_bf16: batch * channel blocks exceeds 65535");
  hipLaunchKernelGGL(bn_stats_8c_kernel, dim3(nb, B * Cb), dim3(256), 0, stream, reinterpret_cast<const u32x4_t*>(z), part, C, Cb, HW);
  YOGO_CHECK_LAUNCH("bn_stats_bf16");
  return YOGO_OK;
}

extern "C" int yogo_nchw_f32_to_bf16_8c(const float* in, void* out, int B, int C, int HW, hipStream_t stream) {
  YOGO_CHECK_ARG(in && out && B > 0 && C > 0 && HW > 0, "nchw_f32_to_bf16_8c: bad arguments");
  const int Cb = cb_of(C);
  hipLaunchKernelGGL(nchw_f32_to_8c_kernel, dim3(plane_blocks(HW, 2), B * Cb), dim3(256), 0, stream, in,
                     reinterpret_cast<u32x4_t*>(out), C, Cb, HW);
  YOGO_CHECK_LAUNCH("nchw_f32_to_bf16_8c");
  return YOGO_OK;
}

extern "C" int yogo_bf16_8c_to_nchw_f32(const void* in, float* out, int B, int C, int HW, hipStream_t stream) {
  YOGO_CHECK_ARG(in && out && B > 0 && C > 0 && HW > 0, "bf16_8c_to_nchw_f32: bad arguments");
  const int Cb = cb_of(C);
  hipLaunchKernelGGL(nchw8c_to_f32_kernel, dim3(plane_blocks(HW, 2), B * Cb), dim3(256), 0, stream,
                     reinterpret_cast<const u32x4_t*>(in), out, C, Cb, HW);
  YOGO_CHECK_LAUNCH("bf16_8c_to_nchw_f32");
  return YOGO_OK;
}
