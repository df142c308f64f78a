// Fused AdamW over one flat fp32 parameter buffer (+ cosine-annealed LR computed by the host per step).
// Replaces torch.optim.AdamW(foreach) of yogo/train.py:213-217, 324: weight decay 5e-2 on ALL
// parameters (incl. BN and bias, as the reference does), betas (0.9, 0.999), eps 1e-8.
// HBM-bound: 4 reads + 3 writes of 541 852 floats = 15.2 MB per step.
#include "common.h"

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float beta1, float beta2,
                                                    float eps, float weight_decay, float bc1, float bc2_sqrt, float grad_scale) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * grad_scale;
    float pi = p[i];
    pi = pi * (1.f - lr * weight_decay);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - (lr / bc1) * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

// step is 1-based (the value AFTER incrementing, as in torch).  grad_scale multiplies the gradient first
// (1/world_size after a sum all-reduce; 1 otherwise).
extern "C" int yogo_adamw_step(float* p, const float* g, float* m, float* v, long long n, int step, float lr, float beta1,
                               float beta2, float eps, float weight_decay, float grad_scale, hipStream_t stream) {
  YOGO_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adamw_step: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int blocks = (int)min((long long)2048, (n + 255) / 256);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                     (float)bc1, (float)sqrt(bc2), grad_scale);
  YOGO_CHECK_LAUNCH("adamw_step");
  return YOGO_OK;
}
