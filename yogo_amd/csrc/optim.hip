// Fused AdamW over one flat fp32 parameter buffer (+ cosine-annealed LR computed by the host per step).
// Replaces torch.optim.AdamW(foreach) of yogo/train.py:213-217, 324: weight decay 5e-2 on ALL
// parameters (incl. BN and bias, as the reference does), betas (0.9, 0.999), eps 1e-8.
// HBM-bound: 4 reads + 3 writes of 541 852 floats = 15.2 MB per step.
// Scalars are formed in double on the host and rounded once to fp32, as torch does with its Python scalars.
#include "common.h"

struct AdamwScalars {
  float decay;        // 1 - lr*wd
  float w1;           // 1 - beta1
  float beta2;
  float w2;           // 1 - beta2
  float step_size;    // lr / (1 - beta1^t)
  float bc2_sqrt;     // sqrt(1 - beta2^t)
  float eps;
  float grad_scale;
};

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, const AdamwScalars s) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * s.grad_scale;
    const float pi = p[i] * s.decay;
    const float m0 = m[i];
    const float mi = m0 + s.w1 * (gi - m0);            // lerp
    const float vi = v[i] * s.beta2 + (s.w2 * gi) * gi;  // mul + addcmul
    const float denom = sqrtf(vi) / s.bc2_sqrt + s.eps;
    p[i] = pi - s.step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

// step is 1-based (the value AFTER incrementing, as in torch).  grad_scale multiplies the gradient first
// (1/world_size after a sum all-reduce; 1 otherwise).
extern "C" int yogo_adamw_step(float* p, const float* g, float* m, float* v, long long n, int step, double lr, double beta1,
                               double beta2, double eps, double weight_decay, double grad_scale, hipStream_t stream) {
  YOGO_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adamw_step: bad arguments");
  AdamwScalars s;
  s.decay = (float)(1.0 - lr * weight_decay);
  s.w1 = (float)(1.0 - beta1);
  s.beta2 = (float)beta2;
  s.w2 = (float)(1.0 - beta2);
  s.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
  s.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
  s.eps = (float)eps;
  s.grad_scale = (float)grad_scale;
  const int blocks = (int)min((long long)2048, (n + 255) / 256);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, n, s);
  YOGO_CHECK_LAUNCH("adamw_step");
  return YOGO_OK;
}
