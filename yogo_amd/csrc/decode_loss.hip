// Box decode (yogo/model.py:277-313) and the grid-cell loss (yogo/yogo_loss.py:38-129), forward + backward.
// SURVEY.md K10, K13-K15.  Both are one pass over [B, 5+C, Sy, Sx] cells: HBM-bound, one lane per cell, channel
// planes read/written coalesced along Sx.  The loss kernel replaces ~30 ATen launches, two boolean-mask gathers
// (each a host sync) and three .item() syncs by one launch: per-cell CIoU / label-smoothed CE / weighted MSE,
// analytic gradients, wavefront-shuffle + LDS block reduction, fixed-order fp64 final sum (deterministic).
//
// This TU is compiled with -ffp-contract=off so that "mul then add" stays two roundings as on the CPU.
#include "common.h"

#define MAX_CLASSES 64

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- decode forward ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_fwd_kernel(const float* __restrict__ raw, float* __restrict__ out,
                                                         const float* __restrict__ cxs, const float* __restrict__ cys,
                                                         float inv_sx, float inv_sy, float anchor_w, float anchor_h,
                                                         float wmul, float hmul, int P, int cells, int inference) {
  const int b = blockIdx.y;
  const int cell = blockIdx.x * 256 + threadIdx.x;
  if (cell >= cells) return;
  const float* r = raw + (size_t)b * P * cells + cell;
  float* o = out + (size_t)b * P * cells + cell;
  const float t0 = r[0], t1 = r[(size_t)cells], t2 = r[(size_t)2 * cells], t3 = r[(size_t)3 * cells], t4 = r[(size_t)4 * cells];
  o[0] = inv_sx * sigmoidf_(t0) + cxs[cell];
  o[(size_t)cells] = inv_sy * sigmoidf_(t1) + cys[cell];
  o[(size_t)2 * cells] = anchor_w * expf(fminf(t2, 80.f)) * wmul;
  o[(size_t)3 * cells] = anchor_h * expf(fminf(t3, 80.f)) * hmul;
  o[(size_t)4 * cells] = sigmoidf_(t4);
  const int C = P - 5;
  if (inference) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, r[(size_t)(5 + c) * cells]);
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(r[(size_t)(5 + c) * cells] - mx);
    for (int c = 0; c < C; ++c) o[(size_t)(5 + c) * cells] = expf(r[(size_t)(5 + c) * cells] - mx) / sum;
  } else {
    for (int c = 0; c < C; ++c) o[(size_t)(5 + c) * cells] = r[(size_t)(5 + c) * cells];
  }
}

// ---- decode backward: graw = gout * d(out)/d(raw) ------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ out,
                                                         const float* __restrict__ gout, float* __restrict__ graw,
                                                         float inv_sx, float inv_sy, int P, int cells, int inference) {
  const int b = blockIdx.y;
  const int cell = blockIdx.x * 256 + threadIdx.x;
  if (cell >= cells) return;
  const size_t base = (size_t)b * P * cells + cell;
  const float* r = raw + base;
  const float* o = out + base;
  const float* g = gout + base;
  float* d = graw + base;
  const float s0 = sigmoidf_(r[0]), s1 = sigmoidf_(r[(size_t)cells]), s4 = o[(size_t)4 * cells];
  d[0] = g[0] * (inv_sx * (s0 * (1.f - s0)));
  d[(size_t)cells] = g[(size_t)cells] * (inv_sy * (s1 * (1.f - s1)));
  d[(size_t)2 * cells] = r[(size_t)2 * cells] <= 80.f ? g[(size_t)2 * cells] * o[(size_t)2 * cells] : 0.f;
  d[(size_t)3 * cells] = r[(size_t)3 * cells] <= 80.f ? g[(size_t)3 * cells] * o[(size_t)3 * cells] : 0.f;
  d[(size_t)4 * cells] = g[(size_t)4 * cells] * (s4 * (1.f - s4));
  const int C = P - 5;
  if (inference) {
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += g[(size_t)(5 + c) * cells] * o[(size_t)(5 + c) * cells];
    for (int c = 0; c < C; ++c) d[(size_t)(5 + c) * cells] = o[(size_t)(5 + c) * cells] * (g[(size_t)(5 + c) * cells] - dot);
  } else {
    for (int c = 0; c < C; ++c) d[(size_t)(5 + c) * cells] = g[(size_t)(5 + c) * cells];
  }
}

// the same with the gradient written as bf16 NCHW8c [B][kb(P)][cells][8] (padding channels zero): what the bf16 backward pass of
// the head convolution reads -- saves the fp32 tensor and the conversion pass
typedef __bf16 dl_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int dl_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void decode_bwd_bf16_kernel(const float* __restrict__ raw, const float* __restrict__ out,
                                                              const float* __restrict__ gout, dl_u32x4* __restrict__ g8, float inv_sx,
                                                              float inv_sy, int P, int Pb, int cells, int inference) {
  const int b = blockIdx.y;
  const int cell = blockIdx.x * 256 + threadIdx.x;
  if (cell >= cells) return;
  const size_t base = (size_t)b * P * cells + cell;
  const float* r = raw + base;
  const float* o = out + base;
  const float* g = gout + base;
  const int C = P - 5;
  float dot = 0.f;
  if (inference)
    for (int c = 0; c < C; ++c) dot += g[(size_t)(5 + c) * cells] * o[(size_t)(5 + c) * cells];
  const float s0 = sigmoidf_(r[0]), s1 = sigmoidf_(r[(size_t)cells]), s4 = o[(size_t)4 * cells];
  for (int kb = 0; kb < Pb; ++kb) {
    dl_bf16x8 u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = kb * 8 + j;
      float v = 0.f;
      if (ch < P) {
        const float gv = g[(size_t)ch * cells];
        if (ch == 0) v = gv * (inv_sx * (s0 * (1.f - s0)));
        else if (ch == 1) v = gv * (inv_sy * (s1 * (1.f - s1)));
        else if (ch == 2 || ch == 3) v = r[(size_t)ch * cells] <= 80.f ? gv * o[(size_t)ch * cells] : 0.f;
        else if (ch == 4) v = gv * (s4 * (1.f - s4));
        else v = inference ? o[(size_t)ch * cells] * (gv - dot) : gv;
      }
      u[j] = (__bf16)v;
    }
    g8[((size_t)b * Pb + kb) * cells + cell] = __builtin_bit_cast(dl_u32x4, u);
  }
}

// ---- loss forward + backward -------------------------------------------------------------------------------------
// d max(a,b)/da as torch's `maximum` backward: 1 if a > b, 0.5 on ties, 0 otherwise (min likewise)
__device__ __forceinline__ float dmax_a(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float dmin_a(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

struct LossParams {
  const float* pred;   // [B][P][cells] decoded boxes + objectness, raw class logits
  const float* label;  // [B][6][cells] mask, x1, y1, x2, y2, class
  float* grad;         // [B][P][cells]  d(total loss)/d(pred)
  float* part;         // [B*gridDim.x][3] per-workgroup partial sums (iou, obj, cls) -- unweighted by 1/B
  int B, P, cells;
  float no_obj_weight, iou_weight, classify_weight, label_smoothing, inv_batch;
};

__global__ __launch_bounds__(256) void yogo_loss_kernel(const LossParams p) {
  __shared__ float sh[3][4];
  const int b = blockIdx.y;
  const int cell = blockIdx.x * 256 + threadIdx.x;
  const int cells = p.cells;
  const int C = p.P - 5;
  float l_iou = 0.f, l_obj = 0.f, l_cls = 0.f;
  if (cell < cells) {
    const float* pr = p.pred + (size_t)b * p.P * cells + cell;
    const float* lb = p.label + (size_t)b * 6 * cells + cell;
    float* gr = p.grad + (size_t)b * p.P * cells + cell;
    const float m = lb[0];
    // objectness: (pred4 - mask)^2 * (mask*(1-w) + w)
    {
      const float po = pr[(size_t)4 * cells];
      const float wgt = m * (1.f - p.no_obj_weight) + p.no_obj_weight;
      const float df = po - m;
      l_obj = df * df * wgt;
      gr[(size_t)4 * cells] = 2.f * df * wgt * p.inv_batch;
    }
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    if (m != 0.f) {
      // ---- CIoU on clamp(xyxy(pred), 0, 1) vs label xyxy -------------------------------------------------
      const float cx = pr[0], cy = pr[(size_t)cells], w = pr[(size_t)2 * cells], h = pr[(size_t)3 * cells];
      const float x1 = cx - 0.5f * w, y1 = cy - 0.5f * h, x2 = cx + 0.5f * w, y2 = cy + 0.5f * h;
      if (x1 != x2 && y1 != y2) {
        const float X1 = fminf(fmaxf(x1, 0.f), 1.f), Y1 = fminf(fmaxf(y1, 0.f), 1.f);
        const float X2 = fminf(fmaxf(x2, 0.f), 1.f), Y2 = fminf(fmaxf(y2, 0.f), 1.f);
        const float c1 = (x1 >= 0.f && x1 <= 1.f) ? 1.f : 0.f, c2 = (y1 >= 0.f && y1 <= 1.f) ? 1.f : 0.f;
        const float c3 = (x2 >= 0.f && x2 <= 1.f) ? 1.f : 0.f, c4 = (y2 >= 0.f && y2 <= 1.f) ? 1.f : 0.f;
        const float x1g = lb[(size_t)cells], y1g = lb[(size_t)2 * cells], x2g = lb[(size_t)3 * cells], y2g = lb[(size_t)4 * cells];
        const float eps = 1e-7f;
        const float xk1 = fmaxf(X1, x1g), yk1 = fmaxf(Y1, y1g), xk2 = fminf(X2, x2g), yk2 = fminf(Y2, y2g);
        const bool has = (yk2 > yk1) && (xk2 > xk1);
        const float iw = xk2 - xk1, ih = yk2 - yk1;
        const float I = has ? iw * ih : 0.f;
        const float wp = X2 - X1, hp = Y2 - Y1, wg = x2g - x1g, hg = y2g - y1g;
        const float U = wp * hp + wg * hg - I;
        const float Ue = U + eps;
        const float iou = I / Ue;
        const float xc1 = fminf(X1, x1g), yc1 = fminf(Y1, y1g), xc2 = fmaxf(X2, x2g), yc2 = fmaxf(Y2, y2g);
        const float ex = xc2 - xc1, ey = yc2 - yc1;
        const float D = ex * ex + ey * ey + eps;
        const float dxc = (X2 + X1) / 2.f - (x1g + x2g) / 2.f, dyc = (Y2 + Y1) / 2.f - (y1g + y2g) / 2.f;
        const float dist = dxc * dxc + dyc * dyc;
        const float kv = 4.f / (3.14159265358979323846f * 3.14159265358979323846f);
        const float th = atanf(wg / hg) - atanf(wp / hp);
        const float v = kv * th * th;
        const float alpha = v / (1.f - iou + v + eps);
        l_iou = 1.f - iou + dist / D + alpha * v;
        // ---- gradient w.r.t. (X1, Y1, X2, Y2) ------------------------------------------------------------
        float dI1 = 0.f, dI2 = 0.f, dI3 = 0.f, dI4 = 0.f;
        if (has) {
          dI1 = -ih * dmax_a(X1, x1g);
          dI2 = -iw * dmax_a(Y1, y1g);
          dI3 = ih * dmin_a(X2, x2g);
          dI4 = iw * dmin_a(Y2, y2g);
        }
        const float dU1 = -hp - dI1, dU2 = -wp - dI2, dU3 = hp - dI3, dU4 = wp - dI4;
        const float iU2 = 1.f / (Ue * Ue);
        const float di1 = (dI1 * Ue - I * dU1) * iU2, di2 = (dI2 * Ue - I * dU2) * iU2;
        const float di3 = (dI3 * Ue - I * dU3) * iU2, di4 = (dI4 * Ue - I * dU4) * iU2;
        const float dD1 = -2.f * ex * dmin_a(X1, x1g), dD2 = -2.f * ey * dmin_a(Y1, y1g);
        const float dD3 = 2.f * ex * dmax_a(X2, x2g), dD4 = 2.f * ey * dmax_a(Y2, y2g);
        const float iD2 = 1.f / (D * D);
        const float dr1 = (dxc * D - dist * dD1) * iD2, dr2 = (dyc * D - dist * dD2) * iD2;
        const float dr3 = (dxc * D - dist * dD3) * iD2, dr4 = (dyc * D - dist * dD4) * iD2;
        const float den = hp * hp + wp * wp;
        const float dv_dw = -2.f * kv * th * hp / den, dv_dh = 2.f * kv * th * wp / den;
        const float gX1 = -di1 + dr1 - alpha * dv_dw, gY1 = -di2 + dr2 - alpha * dv_dh;
        const float gX2 = -di3 + dr3 + alpha * dv_dw, gY2 = -di4 + dr4 + alpha * dv_dh;
        const float sc = p.iou_weight * p.inv_batch;
        g0 = (gX1 * c1 + gX2 * c3) * sc;
        g1 = (gY1 * c2 + gY2 * c4) * sc;
        g2 = 0.5f * (gX2 * c3 - gX1 * c1) * sc;
        g3 = 0.5f * (gY2 * c4 - gY1 * c2) * sc;
      }
    }
    gr[0] = g0;
    gr[(size_t)cells] = g1;
    gr[(size_t)2 * cells] = g2;
    gr[(size_t)3 * cells] = g3;
    // ---- label-smoothed cross entropy, weighted by the mask VALUE (yogo_loss.py:107-114) ----------------------
    if (m != 0.f) {
      const int tgt = (int)lb[(size_t)5 * cells];
      float mx = -INFINITY;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, pr[(size_t)(5 + c) * cells]);
      float sum = 0.f;
      for (int c = 0; c < C; ++c) sum += expf(pr[(size_t)(5 + c) * cells] - mx);
      const float lse = mx + logf(sum);
      float nll_t = 0.f, nll_sum = 0.f;
      for (int c = 0; c < C; ++c) {
        const float lp = pr[(size_t)(5 + c) * cells] - lse;
        nll_sum -= lp;
        if (c == tgt) nll_t = -lp;
      }
      const float ls = p.label_smoothing;
      l_cls = m * ((1.f - ls) * nll_t + (ls / (float)C) * nll_sum);
      const float sc = m * p.classify_weight * p.inv_batch;
      for (int c = 0; c < C; ++c) {
        const float sm = expf(pr[(size_t)(5 + c) * cells] - lse);
        gr[(size_t)(5 + c) * cells] = sc * (sm - (c == tgt ? (1.f - ls) : 0.f) - ls / (float)C);
      }
    } else {
      for (int c = 0; c < C; ++c) gr[(size_t)(5 + c) * cells] = 0.f;
    }
  }
  l_iou = wave_sum(l_iou);
  l_obj = wave_sum(l_obj);
  l_cls = wave_sum(l_cls);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh[0][wave] = l_iou;
    sh[1][wave] = l_obj;
    sh[2][wave] = l_cls;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    p.part[((size_t)b * gridDim.x + blockIdx.x) * 3 + k] = sh[k][0] + sh[k][1] + sh[k][2] + sh[k][3];
  }
}

// ---- training step, fused: decode + loss forward/backward + decode backward in ONE pass over the cells ------------------
// (SURVEY.md K10 + K13-K15 for the trainer: yogo/model.py:277-313 -> yogo/yogo_loss.py:38-129 -> autograd of both.)  The three
// kernels above move 780 MB per 128-image step (decoded prediction and its gradient written and read back as fp32 tensors); a
// cell's decode, loss and both backward steps only need its own P raw values and 6 label values: 166 MB.  The arithmetic is the
// three kernels' own, statement by statement (tests/test_gpu_kernels.py compares bit for bit); training mode only (class logits
// pass through the decode).
struct FusedParams {
  const float* raw;    // [B][P][cells] head output
  const float* label;  // [B][6][cells]
  const float *cxs, *cys;
  dl_u32x4* g8;        // d total / d raw as bf16 NCHW8c [B][Pb][cells][8]
  float* part;         // as LossParams::part
  int B, P, Pb, cells;
  float inv_sx, inv_sy, anchor_w, anchor_h, wmul, hmul;
  float no_obj_weight, iou_weight, classify_weight, label_smoothing, inv_batch;
};

__global__ __launch_bounds__(256) void decode_loss_bwd_bf16_kernel(const FusedParams p) {
  __shared__ float sh[3][4];
  const int b = blockIdx.y;
  const int cell = blockIdx.x * 256 + threadIdx.x;
  const int cells = p.cells;
  const int C = p.P - 5;
  float l_iou = 0.f, l_obj = 0.f, l_cls = 0.f;
  if (cell < cells) {
    const float* r = p.raw + (size_t)b * p.P * cells + cell;
    const float* lb = p.label + (size_t)b * 6 * cells + cell;
    // decode (decode_fwd_kernel)
    const float t0 = r[0], t1 = r[(size_t)cells], t2 = r[(size_t)2 * cells], t3 = r[(size_t)3 * cells], t4 = r[(size_t)4 * cells];
    const float s0 = sigmoidf_(t0), s1 = sigmoidf_(t1);
    const float pcx = p.inv_sx * s0 + p.cxs[cell];
    const float pcy = p.inv_sy * s1 + p.cys[cell];
    const float pw = p.anchor_w * expf(fminf(t2, 80.f)) * p.wmul;
    const float ph = p.anchor_h * expf(fminf(t3, 80.f)) * p.hmul;
    const float po = sigmoidf_(t4);
    // loss + gradient w.r.t. the decoded prediction (yogo_loss_kernel)
    const float m = lb[0];
    float gobj;
    {
      const float wgt = m * (1.f - p.no_obj_weight) + p.no_obj_weight;
      const float df = po - m;
      l_obj = df * df * wgt;
      gobj = 2.f * df * wgt * p.inv_batch;
    }
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    if (m != 0.f) {
      const float cx = pcx, cy = pcy, w = pw, h = ph;
      const float x1 = cx - 0.5f * w, y1 = cy - 0.5f * h, x2 = cx + 0.5f * w, y2 = cy + 0.5f * h;
      if (x1 != x2 && y1 != y2) {
        const float X1 = fminf(fmaxf(x1, 0.f), 1.f), Y1 = fminf(fmaxf(y1, 0.f), 1.f);
        const float X2 = fminf(fmaxf(x2, 0.f), 1.f), Y2 = fminf(fmaxf(y2, 0.f), 1.f);
        const float c1 = (x1 >= 0.f && x1 <= 1.f) ? 1.f : 0.f, c2 = (y1 >= 0.f && y1 <= 1.f) ? 1.f : 0.f;
        const float c3 = (x2 >= 0.f && x2 <= 1.f) ? 1.f : 0.f, c4 = (y2 >= 0.f && y2 <= 1.f) ? 1.f : 0.f;
        const float x1g = lb[(size_t)cells], y1g = lb[(size_t)2 * cells], x2g = lb[(size_t)3 * cells], y2g = lb[(size_t)4 * cells];
        const float eps = 1e-7f;
        const float xk1 = fmaxf(X1, x1g), yk1 = fmaxf(Y1, y1g), xk2 = fminf(X2, x2g), yk2 = fminf(Y2, y2g);
        const bool has = (yk2 > yk1) && (xk2 > xk1);
        const float iw = xk2 - xk1, ih = yk2 - yk1;
        const float I = has ? iw * ih : 0.f;
        const float wp = X2 - X1, hp = Y2 - Y1, wg = x2g - x1g, hg = y2g - y1g;
        const float U = wp * hp + wg * hg - I;
        const float Ue = U + eps;
        const float iou = I / Ue;
        const float xc1 = fminf(X1, x1g), yc1 = fminf(Y1, y1g), xc2 = fmaxf(X2, x2g), yc2 = fmaxf(Y2, y2g);
        const float ex = xc2 - xc1, ey = yc2 - yc1;
        const float D = ex * ex + ey * ey + eps;
        const float dxc = (X2 + X1) / 2.f - (x1g + x2g) / 2.f, dyc = (Y2 + Y1) / 2.f - (y1g + y2g) / 2.f;
        const float dist = dxc * dxc + dyc * dyc;
        const float kv = 4.f / (3.14159265358979323846f * 3.14159265358979323846f);
        const float th = atanf(wg / hg) - atanf(wp / hp);
        const float v = kv * th * th;
        const float alpha = v / (1.f - iou + v + eps);
        l_iou = 1.f - iou + dist / D + alpha * v;
        float dI1 = 0.f, dI2 = 0.f, dI3 = 0.f, dI4 = 0.f;
        if (has) {
          dI1 = -ih * dmax_a(X1, x1g);
          dI2 = -iw * dmax_a(Y1, y1g);
          dI3 = ih * dmin_a(X2, x2g);
          dI4 = iw * dmin_a(Y2, y2g);
        }
        const float dU1 = -hp - dI1, dU2 = -wp - dI2, dU3 = hp - dI3, dU4 = wp - dI4;
        const float iU2 = 1.f / (Ue * Ue);
        const float di1 = (dI1 * Ue - I * dU1) * iU2, di2 = (dI2 * Ue - I * dU2) * iU2;
        const float di3 = (dI3 * Ue - I * dU3) * iU2, di4 = (dI4 * Ue - I * dU4) * iU2;
        const float dD1 = -2.f * ex * dmin_a(X1, x1g), dD2 = -2.f * ey * dmin_a(Y1, y1g);
        const float dD3 = 2.f * ex * dmax_a(X2, x2g), dD4 = 2.f * ey * dmax_a(Y2, y2g);
        const float iD2 = 1.f / (D * D);
        const float dr1 = (dxc * D - dist * dD1) * iD2, dr2 = (dyc * D - dist * dD2) * iD2;
        const float dr3 = (dxc * D - dist * dD3) * iD2, dr4 = (dyc * D - dist * dD4) * iD2;
        const float den = hp * hp + wp * wp;
        const float dv_dw = -2.f * kv * th * hp / den, dv_dh = 2.f * kv * th * wp / den;
        const float gX1 = -di1 + dr1 - alpha * dv_dw, gY1 = -di2 + dr2 - alpha * dv_dh;
        const float gX2 = -di3 + dr3 + alpha * dv_dw, gY2 = -di4 + dr4 + alpha * dv_dh;
        const float sc = p.iou_weight * p.inv_batch;
        g0 = (gX1 * c1 + gX2 * c3) * sc;
        g1 = (gY1 * c2 + gY2 * c4) * sc;
        g2 = 0.5f * (gX2 * c3 - gX1 * c1) * sc;
        g3 = 0.5f * (gY2 * c4 - gY1 * c2) * sc;
      }
    }
    // classification: value now, the per-class gradient in the output loop below (from lse / tgt / scl)
    float lse = 0.f, scl = 0.f;
    int tgt = -1;
    const float ls = p.label_smoothing;
    if (m != 0.f) {
      tgt = (int)lb[(size_t)5 * cells];
      float mx = -INFINITY;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, r[(size_t)(5 + c) * cells]);
      float sum = 0.f;
      for (int c = 0; c < C; ++c) sum += expf(r[(size_t)(5 + c) * cells] - mx);
      lse = mx + logf(sum);
      float nll_t = 0.f, nll_sum = 0.f;
      for (int c = 0; c < C; ++c) {
        const float lp = r[(size_t)(5 + c) * cells] - lse;
        nll_sum -= lp;
        if (c == tgt) nll_t = -lp;
      }
      l_cls = m * ((1.f - ls) * nll_t + (ls / (float)C) * nll_sum);
      scl = m * p.classify_weight * p.inv_batch;
    }
    // decode backward (decode_bwd_bf16_kernel, training mode), written as bf16 NCHW8c units
    for (int kb = 0; kb < p.Pb; ++kb) {
      dl_bf16x8 u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = kb * 8 + j;
        float v = 0.f;
        if (ch < p.P) {
          if (ch == 0) v = g0 * (p.inv_sx * (s0 * (1.f - s0)));
          else if (ch == 1) v = g1 * (p.inv_sy * (s1 * (1.f - s1)));
          else if (ch == 2) v = t2 <= 80.f ? g2 * pw : 0.f;
          else if (ch == 3) v = t3 <= 80.f ? g3 * ph : 0.f;
          else if (ch == 4) v = gobj * (po * (1.f - po));
          else if (m != 0.f) {
            const int c = ch - 5;
            const float sm = expf(r[(size_t)ch * cells] - lse);
            v = scl * (sm - (c == tgt ? (1.f - ls) : 0.f) - ls / (float)C);
          }
        }
        u[j] = (__bf16)v;
      }
      p.g8[((size_t)b * p.Pb + kb) * cells + cell] = __builtin_bit_cast(dl_u32x4, u);
    }
  }
  l_iou = wave_sum(l_iou);
  l_obj = wave_sum(l_obj);
  l_cls = wave_sum(l_cls);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh[0][wave] = l_iou;
    sh[1][wave] = l_obj;
    sh[2][wave] = l_cls;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    p.part[((size_t)b * gridDim.x + blockIdx.x) * 3 + k] = sh[k][0] + sh[k][1] + sh[k][2] + sh[k][3];
  }
}

// out[0] = total, out[1] = iou_loss, out[2] = objectness_loss, out[3] = classification_loss
__global__ __launch_bounds__(256) void yogo_loss_finalize_kernel(const float* __restrict__ part, int rows, float iou_weight,
                                                                 float classify_weight, float inv_batch,
                                                                 float* __restrict__ out) {
  __shared__ double sh[3][4];
  double s[3] = {0.0, 0.0, 0.0};
  for (int r = threadIdx.x; r < rows; r += 256)
    for (int k = 0; k < 3; ++k) s[k] += (double)part[(size_t)r * 3 + k];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < 3; ++k) {
    s[k] = wave_sum_d(s[k]);
    if (lane == 0) sh[k][wave] = s[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float iou = (float)(iou_weight * (sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]) * inv_batch);
    const float obj = (float)((sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]) * inv_batch);
    const float cls = (float)(classify_weight * (sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3]) * inv_batch);
    out[0] = obj + iou + cls;
    out[1] = iou;
    out[2] = obj;
    out[3] = cls;
  }
}

// =========================================================================================================
// C ABI
// =========================================================================================================
extern "C" int yogo_decode_fwd(const float* raw, float* out, const float* cxs, const float* cys, int B, int P, int Sy,
                               int Sx, float anchor_w, float anchor_h, float width_multiplier, float height_multiplier,
                               int inference, hipStream_t stream) {
  YOGO_CHECK_ARG(raw && out && cxs && cys, "decode_fwd: null pointer");
  YOGO_CHECK_ARG(B >= 0 && P > 5 && P - 5 <= MAX_CLASSES && Sy > 0 && Sx > 0 && B <= 65535, "decode_fwd: bad shape");
  if (B == 0) return YOGO_OK;
  const int cells = Sy * Sx;
  yogo_launch_log("decode_fwd_kernel | B=%d cells=%d P=%d inference=%d", B, cells, P, inference);
  hipLaunchKernelGGL(decode_fwd_kernel, dim3(cdiv(cells, 256), B), dim3(256), 0, stream, raw, out, cxs, cys,
                     (float)(1.0 / Sx), (float)(1.0 / Sy), anchor_w, anchor_h, width_multiplier, height_multiplier, P, cells,
                     inference);
  YOGO_CHECK_LAUNCH("decode_fwd");
  return YOGO_OK;
}

extern "C" int yogo_decode_bwd(const float* raw, const float* out, const float* gout, float* graw, int B, int P, int Sy,
                               int Sx, int inference, hipStream_t stream) {
  YOGO_CHECK_ARG(raw && out && gout && graw, "decode_bwd: null pointer");
  YOGO_CHECK_ARG(B >= 0 && P > 5 && Sy > 0 && Sx > 0 && B <= 65535, "decode_bwd: bad shape");
  if (B == 0) return YOGO_OK;
  const int cells = Sy * Sx;
  hipLaunchKernelGGL(decode_bwd_kernel, dim3(cdiv(cells, 256), B), dim3(256), 0, stream, raw, out, gout, graw,
                     (float)(1.0 / Sx), (float)(1.0 / Sy), P, cells, inference);
  YOGO_CHECK_LAUNCH("decode_bwd");
  return YOGO_OK;
}

// graw8c: bf16 NCHW8c [B][2 * ceil(P / 16)][Sy][Sx][8]
extern "C" int yogo_decode_bwd_bf16(const float* raw, const float* out, const float* gout, void* graw8c, int B, int P, int Sy, int Sx,
                                    int inference, hipStream_t stream) {
  YOGO_CHECK_ARG(raw && out && gout && graw8c, "decode_bwd_bf16: null pointer");
  YOGO_CHECK_ARG(B >= 0 && P > 5 && Sy > 0 && Sx > 0 && B <= 65535, "decode_bwd_bf16: bad shape");
  if (B == 0) return YOGO_OK;
  const int cells = Sy * Sx;
  hipLaunchKernelGGL(decode_bwd_bf16_kernel, dim3(cdiv(cells, 256), B), dim3(256), 0, stream, raw, out, gout,
                     reinterpret_cast<dl_u32x4*>(graw8c), (float)(1.0 / Sx), (float)(1.0 / Sy), P, ((P + 15) / 16) * 2, cells, inference);
  YOGO_CHECK_LAUNCH("decode_bwd_bf16");
  return YOGO_OK;
}

extern "C" int yogo_loss_workspace_bytes(int B, int Sy, int Sx, size_t* bytes) {
  YOGO_CHECK_ARG(bytes && B > 0 && Sy > 0 && Sx > 0, "loss_workspace_bytes: bad arguments");
  *bytes = (size_t)B * cdiv(Sy * Sx, 256) * 3 * sizeof(float);
  return YOGO_OK;
}

// loss_out: 4 device floats (total, iou, objectness, classification); grad: d total / d pred, same shape as pred
extern "C" int yogo_loss_fwd_bwd(const float* pred, const float* label, float* grad, float* loss_out, void* workspace,
                                 int B, int P, int Sy, int Sx, float no_obj_weight, float iou_weight, float classify_weight,
                                 float label_smoothing, hipStream_t stream) {
  YOGO_CHECK_ARG(pred && label && grad && loss_out && workspace, "loss_fwd_bwd: null pointer");
  YOGO_CHECK_ARG(B > 0 && B <= 65535 && P > 5 && P - 5 <= MAX_CLASSES && Sy > 0 && Sx > 0, "loss_fwd_bwd: bad shape");
  LossParams p{};
  p.pred = pred; p.label = label; p.grad = grad; p.part = reinterpret_cast<float*>(workspace);
  p.B = B; p.P = P; p.cells = Sy * Sx;
  p.no_obj_weight = no_obj_weight; p.iou_weight = iou_weight; p.classify_weight = classify_weight;
  p.label_smoothing = label_smoothing; p.inv_batch = 1.0f / (float)B;
  const int nb = cdiv(p.cells, 256);
  hipLaunchKernelGGL(yogo_loss_kernel, dim3(nb, B), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(yogo_loss_finalize_kernel, dim3(1), dim3(256), 0, stream, p.part, B * nb, iou_weight, classify_weight,
                     p.inv_batch, loss_out);
  YOGO_CHECK_LAUNCH("loss_fwd_bwd");
  return YOGO_OK;
}

// the trainer's fused form of yogo_decode_fwd + yogo_loss_fwd_bwd + yogo_decode_bwd_bf16 (training mode: class logits pass through
// the decode).  graw8c: bf16 NCHW8c [B][2 * ceil(P / 16)][Sy][Sx][8]; loss_out / workspace as yogo_loss_fwd_bwd.
extern "C" int yogo_decode_loss_bwd_bf16(const float* raw, const float* label, const float* cxs, const float* cys, void* graw8c, float* loss_out,
                                         void* workspace, int B, int P, int Sy, int Sx, float anchor_w, float anchor_h, float width_multiplier,
                                         float height_multiplier, float no_obj_weight, float iou_weight, float classify_weight,
                                         float label_smoothing, hipStream_t stream) {
  YOGO_CHECK_ARG(raw && label && cxs && cys && graw8c && loss_out && workspace, "decode_loss_bwd_bf16: null pointer");
  YOGO_CHECK_ARG(B > 0 && B <= 65535 && P > 5 && P - 5 <= MAX_CLASSES && Sy > 0 && Sx > 0, "decode_loss_bwd_bf16: bad shape");
  FusedParams p{};
  p.raw = raw; p.label = label; p.cxs = cxs; p.cys = cys; p.g8 = reinterpret_cast<dl_u32x4*>(graw8c); p.part = reinterpret_cast<float*>(workspace);
  p.B = B; p.P = P; p.Pb = ((P + 15) / 16) * 2; p.cells = Sy * Sx;
  p.inv_sx = (float)(1.0 / Sx); p.inv_sy = (float)(1.0 / Sy); p.anchor_w = anchor_w; p.anchor_h = anchor_h; p.wmul = width_multiplier; p.hmul = height_multiplier;
  p.no_obj_weight = no_obj_weight; p.iou_weight = iou_weight; p.classify_weight = classify_weight; p.label_smoothing = label_smoothing;
  p.inv_batch = 1.0f / (float)B;
  const int nb = cdiv(p.cells, 256);
  hipLaunchKernelGGL(decode_loss_bwd_bf16_kernel, dim3(nb, B), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(yogo_loss_finalize_kernel, dim3(1), dim3(256), 0, stream, p.part, B * nb, iou_weight, classify_weight, p.inv_batch, loss_out);
  YOGO_CHECK_LAUNCH("decode_loss_bwd_bf16");
  return YOGO_OK;
}
