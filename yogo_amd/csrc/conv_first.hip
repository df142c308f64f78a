// First convolution of the backbone: Cin = 1 (gray) or 3 (rgb), uint8 or fp32 input, 3x3 stride 1|2 pad 1.
// Replaces nn.Conv2d(input_channels, C, 3, stride=2, padding=1, bias=False) + the uint8->float cast
// (yogo/model_defns.py:34, yogo/model.py:272-273).
//
// K = 9*Cin is far too short for the matrix cores and the layer is HBM-bound (0.8 MB in, 12.75 MB out per
// image at 772x1032): a direct VALU kernel, one output pixel per lane (coalesced NCHW stores per channel),
// weights through the scalar cache, BatchNorm partial sums carried in registers across the thread's pixels.
#include "common.h"

#define CF_THREADS 256
#define CF_PPT 4       // pixels per thread
#define CF_COCHUNK 16  // output channels per register pass

typedef __bf16 cf_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int cf_u32x4 __attribute__((ext_vector_type(4)));

struct ConvFirstParams {
  cf_u32x4* out_bf16;      // when set: output goes to bf16 NCHW8c [B][Mb][OH][OW] units instead of `out`
  int Mb;
  const void* in;
  const float* w;          // OIHW [Cout][Cin][3][3]
  const float* bias;       // optional
  float* out;              // [B][Cout][OH][OW]
  float* out_pre;          // optional pre-activation copy
  const float* chan_scale; // optional [B][Cout]
  float* stats_part;       // optional [B*gridDim.x][Cout][2]
  int B, Cin, Cout, IH, IW, OH, OW, stride, act;
};

template <typename TIn, int CIN>
__global__ __launch_bounds__(CF_THREADS) void conv_first_kernel(const ConvFirstParams p) {
  __shared__ float red[4][2 * CF_COCHUNK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int npix = p.OH * p.OW;
  const int pbase = blockIdx.x * (CF_THREADS * CF_PPT);
  const TIn* inb = reinterpret_cast<const TIn*>(p.in) + (size_t)b * CIN * p.IH * p.IW;
  const float* __restrict__ w = p.w;

  for (int co0 = 0; co0 < p.Cout; co0 += CF_COCHUNK) {
    float s[CF_COCHUNK], q[CF_COCHUNK];
#pragma unroll
    for (int c = 0; c < CF_COCHUNK; ++c) s[c] = q[c] = 0.f;
#pragma unroll
    for (int k = 0; k < CF_PPT; ++k) {
      const int pix = pbase + k * CF_THREADS + tid;
      const bool ok = pix < npix;
      const int oy = ok ? pix / p.OW : 0;
      const int ox = ok ? pix - oy * p.OW : 0;
      float x[CIN * 9];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int iy = oy * p.stride + kh - 1, ix = ox * p.stride + kw - 1;
            float v = 0.f;
            if (ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) v = (float)inb[((size_t)ci * p.IH + iy) * p.IW + ix];
            x[(ci * 3 + kh) * 3 + kw] = v;
          }
      float vout[CF_COCHUNK];
#pragma unroll
      for (int c = 0; c < CF_COCHUNK; ++c) {
        const int co = co0 + c;
        vout[c] = 0.f;
        if (co < p.Cout) {  // uniform
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < CIN * 9; ++j) acc = fmaf(w[co * CIN * 9 + j], x[j], acc);
          if (p.bias != nullptr) acc += p.bias[co];
          float v = act_fwd(acc, p.act);
          if (p.chan_scale != nullptr) v *= p.chan_scale[(size_t)b * p.Cout + co];
          vout[c] = v;
          if (ok) {
            s[c] += acc;
            q[c] += acc * acc;
            if (p.out_bf16 == nullptr) {
              const size_t idx = ((size_t)b * p.Cout + co) * npix + pix;
              if (p.out_pre != nullptr) p.out_pre[idx] = acc;
              p.out[idx] = v;
            }
          }
        }
      }
      if (p.out_bf16 != nullptr && ok) {  // two 16-byte units (8 channels each) per pixel and 16-channel pass
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const int cb = (co0 >> 3) + hb;
          if (cb < p.Mb) {
            cf_bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)vout[hb * 8 + j];
            p.out_bf16[((size_t)b * p.Mb + cb) * npix + pix] = __builtin_bit_cast(cf_u32x4, o);
          }
        }
      }
    }
    if (p.stats_part != nullptr) {
#pragma unroll
      for (int c = 0; c < CF_COCHUNK; ++c) {
        const float ss = wave_sum(s[c]), qq = wave_sum(q[c]);
        if (lane == 0) {
          red[wave][2 * c] = ss;
          red[wave][2 * c + 1] = qq;
        }
      }
      __syncthreads();
      if (tid < 2 * CF_COCHUNK) {
        const int c = tid >> 1;
        if (co0 + c < p.Cout) {
          const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
          p.stats_part[((size_t)(b * gridDim.x + blockIdx.x) * p.Cout + co0 + c) * 2 + (tid & 1)] = v;
        }
      }
      __syncthreads();
    }
  }
}

// Training variant with bf16 NCHW8c output: weights (transposed to [tap][16 channels]) and bias sit in LDS and are read as
// broadcast ds_read_b128 -- once per tap for the lane's CF_PPT register-blocked pixels; input reads are branch-free (clamped
// address + select); the BatchNorm sums are formed from the accumulators at the end and reduced with DPP adds.
template <typename TIn, int CIN>
__global__ __launch_bounds__(CF_THREADS) void conv_first_train_bf16_kernel(const ConvFirstParams p) {
  constexpr int NJ = CIN * 9;
  __shared__ __attribute__((aligned(16))) float wsh[NJ][CF_COCHUNK];
  __shared__ __attribute__((aligned(16))) float bsh[CF_COCHUNK], csh[CF_COCHUNK];
  __shared__ float red[4][2 * CF_COCHUNK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int npix = p.OH * p.OW;
  const int pbase = blockIdx.x * (CF_THREADS * CF_PPT);
  const TIn* inb = reinterpret_cast<const TIn*>(p.in) + (size_t)b * CIN * p.IH * p.IW;
  bool okk[CF_PPT];
  int oy[CF_PPT], ox[CF_PPT];
#pragma unroll
  for (int k = 0; k < CF_PPT; ++k) {
    const int pix = pbase + k * CF_THREADS + tid;
    okk[k] = pix < npix;
    const int pc = okk[k] ? pix : 0;
    oy[k] = pc / p.OW;
    ox[k] = pc - oy[k] * p.OW;
  }
  for (int co0 = 0; co0 < p.Mb * 8; co0 += CF_COCHUNK) {
    __syncthreads();
    for (int e = tid; e < NJ * CF_COCHUNK; e += CF_THREADS) {
      const int j = e / CF_COCHUNK, c = e - j * CF_COCHUNK;
      wsh[j][c] = co0 + c < p.Cout ? p.w[(size_t)(co0 + c) * NJ + j] : 0.f;
    }
    if (tid < CF_COCHUNK) {
      const int co = co0 + tid;
      bsh[tid] = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
      csh[tid] = co < p.Cout ? (p.chan_scale != nullptr ? p.chan_scale[(size_t)b * p.Cout + co] : 1.f) : 0.f;
    }
    __syncthreads();
    float acc[CF_PPT][CF_COCHUNK];
#pragma unroll
    for (int c4 = 0; c4 < CF_COCHUNK; c4 += 4) {
      const float4 bv = *reinterpret_cast<const float4*>(&bsh[c4]);
#pragma unroll
      for (int k = 0; k < CF_PPT; ++k) {
        acc[k][c4] = bv.x; acc[k][c4 + 1] = bv.y; acc[k][c4 + 2] = bv.z; acc[k][c4 + 3] = bv.w;
      }
    }
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int j = (ci * 3 + kh) * 3 + kw;
          float x[CF_PPT];
#pragma unroll
          for (int k = 0; k < CF_PPT; ++k) {
            const int iy = oy[k] * p.stride + kh - 1, ix = ox[k] * p.stride + kw - 1;
            const bool in = iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            const float v = (float)inb[in ? ((size_t)ci * p.IH + iy) * p.IW + ix : 0];
            x[k] = in ? v : 0.f;
          }
#pragma unroll
          for (int c4 = 0; c4 < CF_COCHUNK; c4 += 4) {
            const float4 wv = *reinterpret_cast<const float4*>(&wsh[j][c4]);
#pragma unroll
            for (int k = 0; k < CF_PPT; ++k) {
              acc[k][c4] = fmaf(wv.x, x[k], acc[k][c4]);
              acc[k][c4 + 1] = fmaf(wv.y, x[k], acc[k][c4 + 1]);
              acc[k][c4 + 2] = fmaf(wv.z, x[k], acc[k][c4 + 2]);
              acc[k][c4 + 3] = fmaf(wv.w, x[k], acc[k][c4 + 3]);
            }
          }
        }
    // BatchNorm partial sums of the pre-activation (tail pixels masked out)
    if (p.stats_part != nullptr) {
#pragma unroll
      for (int c = 0; c < CF_COCHUNK; ++c) {
        float sc = 0.f, qc = 0.f;
#pragma unroll
        for (int k = 0; k < CF_PPT; ++k) {
          const float vm = okk[k] ? acc[k][c] : 0.f;
          sc += vm;
          qc += vm * vm;
        }
        const float ss = wave_sum(sc), qq = wave_sum(qc);
        if (lane == 0) {
          red[wave][2 * c] = ss;
          red[wave][2 * c + 1] = qq;
        }
      }
    }
    // activation + channel scale + bf16 store
    const float4 c0 = *reinterpret_cast<const float4*>(&csh[0]), c1 = *reinterpret_cast<const float4*>(&csh[4]);
    const float4 c2 = *reinterpret_cast<const float4*>(&csh[8]), c3 = *reinterpret_cast<const float4*>(&csh[12]);
    const float cs[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
#pragma unroll
    for (int k = 0; k < CF_PPT; ++k) {
      const int pix = pbase + k * CF_THREADS + tid;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int cb = (co0 >> 3) + hb;
        if (cb < p.Mb && okk[k]) {
          cf_bf16x8 o;
          if (p.act == ACT_NONE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)(acc[k][hb * 8 + j] * cs[hb * 8 + j]);
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)(act_fwd(acc[k][hb * 8 + j], p.act) * cs[hb * 8 + j]);
          }
          p.out_bf16[((size_t)b * p.Mb + cb) * npix + pix] = __builtin_bit_cast(cf_u32x4, o);
        }
      }
    }
    if (p.stats_part != nullptr) {
      __syncthreads();
      if (tid < 2 * CF_COCHUNK) {
        const int c = tid >> 1;
        if (co0 + c < p.Cout) {
          const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
          p.stats_part[((size_t)(b * gridDim.x + blockIdx.x) * p.Cout + co0 + c) * 2 + (tid & 1)] = v;
        }
      }
    }
  }
}

// dW[co][ci][kh][kw] = sum_{b,pix} dy[b][co][pix] * x[b][ci][pix*stride + tap]; db likewise without x.
// One workgroup per (pixel tile, image); partials [rows][Cout][CIN*9 + 1] are summed by channel_partials_reduce.
struct ConvFirstWgradParams {
  const cf_u32x4* dy_bf16;  // when set: gradient in bf16 NCHW8c [B][Mb][OH][OW] units
  int Mb;
  const void* in;
  const float* dy;    // [B][Cout][OH][OW] (already multiplied by act'/mask)
  float* part;        // [B*gridDim.x][Cout][CIN*9+1]
  int B, Cin, Cout, IH, IW, OH, OW, stride;
};

#define CFW_PPT 32  // pixels per thread of the weight-gradient kernels (accumulators live in registers across them; the
                    // cross-lane reductions at the end cost ~1800 instructions per wavefront, so fewer, longer workgroups)

// Per pass of COC output channels every lane keeps COC x (9*CIN + 1) accumulators over its CFW_PPT pixels, so the cross-lane
// reduction (DPP adds) is paid once per 4096 pixels and pass instead of once per pixel group.
template <typename TIn, int CIN, bool BF16G>
__global__ __launch_bounds__(CF_THREADS) void conv_first_wgrad_kernel(const ConvFirstWgradParams p) {
  constexpr int NJ = CIN * 9 + 1;
  constexpr int COC = CIN == 1 ? 16 : 4;
  __shared__ float red[4][COC * NJ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int npix = p.OH * p.OW;
  const int pbase = blockIdx.x * (CF_THREADS * CFW_PPT);
  const TIn* inb = reinterpret_cast<const TIn*>(p.in) + (size_t)b * CIN * p.IH * p.IW;
  for (int co0 = 0; co0 < p.Cout; co0 += COC) {
    float acc[COC][NJ];
#pragma unroll
    for (int c = 0; c < COC; ++c)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[c][j] = 0.f;
    for (int k = 0; k < CFW_PPT; ++k) {
      const int pix = pbase + k * CF_THREADS + tid;
      const bool ok = pix < npix;
      const int pc = ok ? pix : 0;
      const int oy = pc / p.OW, ox = pc - oy * p.OW;
      float x[CIN * 9];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int iy = oy * p.stride + kh - 1, ix = ox * p.stride + kw - 1;
            const bool in = ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            const float v = (float)inb[in ? ((size_t)ci * p.IH + iy) * p.IW + ix : 0];
            x[(ci * 3 + kh) * 3 + kw] = in ? v : 0.f;
          }
      float g[COC];
      if constexpr (BF16G) {
        if constexpr (COC == 16) {
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) {
            const int cb = min((co0 >> 3) + hb, p.Mb - 1);
            const cf_bf16x8 u = __builtin_bit_cast(cf_bf16x8, p.dy_bf16[((size_t)b * p.Mb + cb) * npix + pc]);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[hb * 8 + j] = (ok && co0 + hb * 8 + j < p.Cout) ? (float)u[j] : 0.f;
          }
        } else {
          const cf_bf16x8 u = __builtin_bit_cast(cf_bf16x8, p.dy_bf16[((size_t)b * p.Mb + (co0 >> 3)) * npix + pc]);
#pragma unroll
          for (int c = 0; c < COC; ++c) {
            float sel = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sel = ((co0 & 7) + c == j) ? (float)u[j] : sel;
            g[c] = (ok && co0 + c < p.Cout) ? sel : 0.f;
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < COC; ++c) {
          const bool cok = ok && co0 + c < p.Cout;
          const float v = p.dy[cok ? ((size_t)b * p.Cout + co0 + c) * npix + pc : 0];
          g[c] = cok ? v : 0.f;
        }
      }
#pragma unroll
      for (int c = 0; c < COC; ++c) {
#pragma unroll
        for (int j = 0; j < CIN * 9; ++j) acc[c][j] = fmaf(g[c], x[j], acc[c][j]);
        acc[c][NJ - 1] += g[c];
      }
    }
#pragma unroll
    for (int c = 0; c < COC; ++c)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float v = wave_sum(acc[c][j]);
        if (lane == 0) red[wave][c * NJ + j] = v;
      }
    __syncthreads();
    if (tid < COC * NJ) {
      const int c = tid / NJ, j = tid - c * NJ;
      if (co0 + c < p.Cout)
        p.part[((size_t)(b * gridDim.x + blockIdx.x) * p.Cout + co0 + c) * NJ + j] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
    __syncthreads();
  }
}

// =========================================================================================================
// Layer-0 backward in ONE pass over (image, gradient, saved conv output): BatchNorm backward + LeakyReLU derivative + weight
// gradient of the first convolution (the layer has no data gradient and, in the reference's architectures, no conv bias).
// With xh = (z - mean) * invstd, gb = g * act'(gamma * xh + beta), c1 = gamma * invstd, N = B*OH*OW:
//   dz = c1 * (gb - S1/N - xh * S2/N),  S1 = sum gb (= dbeta),  S2 = sum gb * xh (= dgamma)      [batch statistics]
//   dW[c][j] = sum dz[c] * patch_j = c1 * (A1[c][j] - S1/N * P[j] - S2/N * A2[c][j])
//   A1 = sum gb * patch_j,  A2 = sum xh * patch_j,  P = sum patch_j
// so every sum is accumulated in the same sweep and dz never goes to memory (replaces bn_bwd_reduce + bn_bwd_apply +
// conv_first_wgrad: 5 tensor passes -> 2).  Partials [rows][Cout*(2*NJ+2) + NJ], finished by yogo_partials_reduce +
// conv_first_bn_wgrad_finalize.
struct ConvFirstBnWgradParams {
  const void* in;
  const cf_u32x4* g;   // gradient w.r.t. the block output, bf16 NCHW8c [B][Mb][OH][OW]
  const cf_u32x4* z;   // saved conv output (pre-BatchNorm), same layout
  const unsigned short* signs;   // ... or (conv_first_bn_wgrad_pk_kernel<true>) the sign map of the BatchNorm output, yogo_conv_first_mfma_signs
  const float *mean, *invstd, *gamma, *beta;
  float* part;
  int B, Cin, Cout, Mb, IH, IW, OH, OW, stride, act;
  int ext_gram;  // GRAM kernels: the caller already has P and G of this batch (yogo_conv_first_gram) -- skip that pass
  unsigned m_ow; // conv_first_bn_wgrad_pk_kernel: ceil(2^32 / OW) when pixel / OW may be taken as a multiply-high (OH * OW * OW < 2^32), else 0
  unsigned m_ow2; // conv_first_bn_wgrad_pk2_kernel: ceil(2^32 / (OW / 2))
};

// GRAM (Cin = 1): z is the bias-free convolution of the patches, so A2[c][j] = invstd_c * (sum_j' W[c][j'] G[j'][j] - mean_c P[j])
// with the 9x9 Gram matrix G = sum patch_j' patch_j of the input -- channel independent: 45 accumulators replace 16 x 9, the
// A2 columns of the partial rows stay unwritten and the finalize kernel forms A2 from G (appended after P).
// FAST (uint8, one channel, stride 2, even image sizes): the three bytes of a patch row come from two aligned 16-bit loads
// and only the top row / left column can fall outside the image -- a third of the per-byte bounds arithmetic.
template <typename TIn, int CIN, int COC, bool GRAM, bool FAST = false>
__global__ __launch_bounds__(CF_THREADS, GRAM ? 3 : 2) void conv_first_bn_wgrad_kernel(const ConvFirstBnWgradParams p) {
  constexpr int NJ = CIN * 9;
  constexpr int PER = 2 * NJ + 2;  // per channel: A1[NJ], A2[NJ], S1, S2
  constexpr int NG = GRAM ? NJ * (NJ + 1) / 2 : 0;
  __shared__ float red[4][(COC * PER + NJ > NJ + NG) ? COC * PER + NJ : NJ + NG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int npix = p.OH * p.OW;
  const int pbase = blockIdx.x * (CF_THREADS * CFW_PPT);
  const int ncol = p.Cout * PER + NJ + (GRAM ? NJ * NJ : 0);
  const TIn* inb = reinterpret_cast<const TIn*>(p.in) + (size_t)b * CIN * p.IH * p.IW;
  float* prow = p.part + (size_t)(b * gridDim.x + blockIdx.x) * ncol;
// this lane's k-th pixel: validity, pixel index and the input patch (zero outside the image / beyond the tail)
#define CFB_PIXEL(K)                                                                                       \
  const int pix = pbase + (K) * CF_THREADS + tid;                                                           \
  const bool ok = pix < npix;                                                                              \
  const int pc = ok ? pix : 0;                                                                             \
  const int oy = pc / p.OW, ox = pc - oy * p.OW;                                                           \
  float x[NJ];                                                                                             \
  if constexpr (FAST) {                                                                                    \
    const unsigned char* ib_ = reinterpret_cast<const unsigned char*>(inb);                                \
    _Pragma("unroll") for (int kh = 0; kh < 3; ++kh) {                                                     \
      const int iy = 2 * oy + kh - 1; /* <= IH - 1 */                                                      \
      const bool rok = ok && iy >= 0;                                                                      \
      const int o_ = rok ? iy * p.IW + 2 * ox : 0; /* bytes o_ - 2 .. o_ + 1 */                            \
      const unsigned hi_ = *reinterpret_cast<const unsigned short*>(ib_ + o_);                             \
      const unsigned lo_ = *reinterpret_cast<const unsigned short*>(ib_ + ((rok && ox > 0) ? o_ - 2 : o_)); \
      x[kh * 3 + 0] = (rok && ox > 0) ? (float)(lo_ >> 8) : 0.f;                                           \
      x[kh * 3 + 1] = rok ? (float)(hi_ & 0xFFu) : 0.f;                                                    \
      x[kh * 3 + 2] = rok ? (float)(hi_ >> 8) : 0.f;                                                       \
    }                                                                                                      \
  } else                                                                                                   \
  _Pragma("unroll") for (int ci = 0; ci < CIN; ++ci)                                                       \
  _Pragma("unroll") for (int kh = 0; kh < 3; ++kh)                                                         \
  _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) {                                                       \
    const int iy = oy * p.stride + kh - 1, ix = ox * p.stride + kw - 1;                                    \
    const bool in = ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;                                    \
    const float v = (float)inb[in ? ((size_t)ci * p.IH + iy) * p.IW + ix : 0];                             \
    x[(ci * 3 + kh) * 3 + kw] = in ? v : 0.f;                                                              \
  }

  if constexpr (GRAM) if (p.ext_gram) {  // (uniform) the P / G columns of the row are zero-filled, the finalize reads the caller's
    for (int e = tid; e < NJ + NJ * NJ; e += CF_THREADS) prow[p.Cout * PER + e] = 0.f;
  }
  if constexpr (GRAM) if (!p.ext_gram) {
    // ---- pass over the image alone: P[j] = sum patch_j, G[j][j2] = sum patch_j * patch_j2 (upper triangle) -------------
    float ps[NJ], gm[NG];
#pragma unroll
    for (int j = 0; j < NJ; ++j) ps[j] = 0.f;
#pragma unroll
    for (int q = 0; q < NG; ++q) gm[q] = 0.f;
    for (int k = 0; k < CFW_PPT; ++k) {
      CFB_PIXEL(k)
      int q = 0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        ps[j] += x[j];
#pragma unroll
        for (int j2 = j; j2 < NJ; ++j2) {
          gm[q] = fmaf(x[j], x[j2], gm[q]);
          ++q;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float v = wave_sum(ps[j]);
      if (lane == 0) red[wave][j] = v;
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const float v = wave_sum(gm[q]);
      if (lane == 0) red[wave][NJ + q] = v;
    }
    __syncthreads();
    for (int e = tid; e < NJ + NG; e += CF_THREADS) {
      const float v = red[0][e] + red[1][e] + red[2][e] + red[3][e];
      if (e < NJ) {
        prow[p.Cout * PER + e] = v;
      } else {  // packed upper triangle -> both halves of the full NJ x NJ matrix
        int q = e - NJ, j = 0;
        while (q >= NJ - j) {
          q -= NJ - j;
          ++j;
        }
        const int j2 = j + q;
        prow[p.Cout * PER + NJ + j * NJ + j2] = v;
        prow[p.Cout * PER + NJ + j2 * NJ + j] = v;
      }
    }
    __syncthreads();
  }
  for (int co0 = 0; co0 < p.Cout; co0 += COC) {
    float a1[COC][NJ], a2[GRAM ? 1 : COC][NJ], s1[COC], s2[COC], ps[GRAM ? 1 : NJ];
    float mu[COC], is[COC], ga[COC], be[COC];
#pragma unroll
    for (int c = 0; c < COC; ++c) {
      const int co = min(co0 + c, p.Cout - 1);
      mu[c] = p.mean[co]; is[c] = p.invstd[co]; ga[c] = p.gamma[co]; be[c] = p.beta[co];
      s1[c] = s2[c] = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        a1[c][j] = 0.f;
        if constexpr (!GRAM) a2[c][j] = 0.f;
      }
    }
    if constexpr (!GRAM) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) ps[j] = 0.f;
    }
    for (int k = 0; k < CFW_PPT; ++k) {
      CFB_PIXEL(k)
      // COC consecutive channels live inside one 8-channel unit (COC divides 8)
      const size_t u = ((size_t)b * p.Mb + (co0 >> 3)) * npix + pc;
      const cf_bf16x8 gu = __builtin_bit_cast(cf_bf16x8, p.g[u]);
      const cf_bf16x8 zu = __builtin_bit_cast(cf_bf16x8, p.z[u]);
#pragma unroll
      for (int c = 0; c < COC; ++c) {
        float gv = 0.f, zv = 0.f;
        if constexpr (COC == 8) {  // a pass = one whole 8-channel unit
          gv = (float)gu[c];
          zv = (float)zu[c];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const bool pick = ((co0 & 7) + c) == j;
            gv = pick ? (float)gu[j] : gv;
            zv = pick ? (float)zu[j] : zv;
          }
        }
        const bool cok = ok && co0 + c < p.Cout;
        const float xh = cok ? (zv - mu[c]) * is[c] : 0.f;
        const float yb = fmaf(ga[c], xh, be[c]);
        float gb = cok ? gv : 0.f;
        if (p.act == ACT_LEAKY) gb *= yb > 0.f ? 1.f : LEAKY_SLOPE;
        else if (p.act == ACT_SILU) gb *= act_bwd_factor(yb, ACT_SILU);
        s1[c] += gb;
        s2[c] = fmaf(gb, xh, s2[c]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          a1[c][j] = fmaf(gb, x[j], a1[c][j]);
          if constexpr (!GRAM) a2[c][j] = fmaf(xh, x[j], a2[c][j]);
        }
      }
      if constexpr (!GRAM) {
        if (co0 == 0) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) ps[j] += x[j];
        }
      }
    }
    // cross-lane sums (DPP), then the four wavefronts through LDS
#pragma unroll
    for (int c = 0; c < COC; ++c) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float v1 = wave_sum(a1[c][j]);
        if (lane == 0) red[wave][c * PER + j] = v1;
        if constexpr (!GRAM) {
          const float v2 = wave_sum(a2[c][j]);
          if (lane == 0) red[wave][c * PER + NJ + j] = v2;
        }
      }
      const float t1 = wave_sum(s1[c]), t2 = wave_sum(s2[c]);
      if (lane == 0) {
        red[wave][c * PER + 2 * NJ] = t1;
        red[wave][c * PER + 2 * NJ + 1] = t2;
      }
    }
    const bool with_p = !GRAM && co0 == 0;
    if (with_p) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float v = wave_sum(ps[GRAM ? 0 : j]);
        if (lane == 0) red[wave][COC * PER + j] = v;
      }
    }
    __syncthreads();
    for (int e = tid; e < COC * PER + (with_p ? NJ : 0); e += CF_THREADS) {
      const float v = red[0][e] + red[1][e] + red[2][e] + red[3][e];
      if (e < COC * PER) {
        const int c = e / PER, k = e - c * PER;
        if (co0 + c < p.Cout && !(GRAM && k >= NJ && k < 2 * NJ)) prow[(co0 + c) * PER + k] = v;
      } else {
        prow[p.Cout * PER + (e - COC * PER)] = v;
      }
    }
    __syncthreads();
  }
#undef CFB_PIXEL
}

// The sweep above for the shape the bf16 training step runs (uint8 one-channel image, stride 2, even sizes, caller-held Gram
// matrix, Cout a multiple of 8, no activation or LeakyReLU), written for the PACKED fp32 pipe.  The sweep is bound by vector-
// instruction issue (0.77 of the SIMD cycles, profiles/r03_issue_util.txt; a wave64 instruction holds its SIMD for 4 cycles), and
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do two lanes' worth per instruction: every quantity lives as a PAIR of adjacent
// channels -- (z, g) unpacked from the bf16 pairs of the unit, xh, the activation factor, gb, S1, S2 -- and
// A1[c..c+1][j] += gb[c..c+1] * patch_j is one packed FMA with the patch value broadcast.  Same arithmetic per element, same
// summation order, so the sums are bit-identical to conv_first_bn_wgrad_kernel's.
//
// SIGNS = true: no z at all.  z enters the sweep twice: through the sign of the BatchNorm output (LeakyReLU derivative) -- the forward
// pass wrote that as one bit per value -- and through S2 = sum gb * xh, which is LINEAR in z = W . patch:
//   sum gb * z = sum_j W[c][j] * A1[c][j]   =>   S2 = invstd * (W[c] . A1[c] - mean * S1)      (the finalize kernel, derive_s2)
// so the layer's conv output is neither written by the forward pass nor read here (0.82 GB each way at 128 x 772 x 1032).
typedef float cf_f32x2 __attribute__((ext_vector_type(2)));
template <bool SIGNS>
__global__ __launch_bounds__(CF_THREADS, 3) void conv_first_bn_wgrad_pk_kernel(const ConvFirstBnWgradParams p) {
  constexpr int NJ = 9, PER = 2 * NJ + 2, COC = 8, NP = COC / 2;
  __shared__ float red[4][COC * PER];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int npix = p.OH * p.OW;
  const int pbase = blockIdx.x * (CF_THREADS * CFW_PPT);
  const int ncol = p.Cout * PER + NJ + NJ * NJ;
  const unsigned char* ib = reinterpret_cast<const unsigned char*>(p.in) + (size_t)b * p.IH * p.IW;
  float* prow = p.part + (size_t)(b * gridDim.x + blockIdx.x) * ncol;
  for (int e = tid; e < NJ + NJ * NJ; e += CF_THREADS) prow[p.Cout * PER + e] = 0.f;   // P / G: the finalize reads the caller's
  const bool leaky = p.act == ACT_LEAKY;   // (uniform; the launcher takes ACT_NONE / ACT_LEAKY only)
  for (int co0 = 0; co0 < p.Cout; co0 += COC) {
    cf_f32x2 a1[NP][NJ], s1[NP], s2[NP], mu[NP], is[NP], ga[NP], be[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int c = co0 + 2 * q;
      mu[q] = cf_f32x2{p.mean[c], p.mean[c + 1]};
      is[q] = cf_f32x2{p.invstd[c], p.invstd[c + 1]};
      ga[q] = cf_f32x2{p.gamma[c], p.gamma[c + 1]};
      be[q] = cf_f32x2{p.beta[c], p.beta[c + 1]};
      s1[q] = s2[q] = cf_f32x2{0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; ++j) a1[q][j] = cf_f32x2{0.f, 0.f};
    }
    for (int k = 0; k < CFW_PPT; ++k) {
      const int pix = pbase + k * CF_THREADS + tid;
      const bool ok = pix < npix;
      const int pc = ok ? pix : 0;
      // (one multiply-high where it is exact.  Measured on this sweep and dropped, round 4: one 2-byte-aligned dword per patch row in
      //  place of the two 16-bit loads -- 356 -> 611 us, misaligned dwords are slow; the loads of pixel k + 1 issued before the
      //  arithmetic of pixel k -- 356 -> 426 us)
      const int oy = p.m_ow ? (int)__umulhi((unsigned)pc, p.m_ow) : pc / p.OW, ox = pc - oy * p.OW;
      float x[NJ];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int iy = 2 * oy + kh - 1;  // <= IH - 1 (even sizes)
        const bool rok = ok && iy >= 0;
        const int o_ = rok ? iy * p.IW + 2 * ox : 0;  // bytes o_ - 2 .. o_ + 1
        const unsigned hi_ = *reinterpret_cast<const unsigned short*>(ib + o_);
        const unsigned lo_ = *reinterpret_cast<const unsigned short*>(ib + ((rok && ox > 0) ? o_ - 2 : o_));
        x[kh * 3 + 0] = (rok && ox > 0) ? (float)(lo_ >> 8) : 0.f;
        x[kh * 3 + 1] = rok ? (float)(hi_ & 0xFFu) : 0.f;
        x[kh * 3 + 2] = rok ? (float)(hi_ >> 8) : 0.f;
      }
      const size_t u = ((size_t)b * p.Mb + (co0 >> 3)) * npix + pc;
      cf_u32x4 gw = p.g[u];
      cf_u32x4 zw = {0u, 0u, 0u, 0u};
      unsigned sgn = 0;
      if constexpr (SIGNS) {
        // the pixel's 16 bits: channel c < 4 or >= 12 at bit c, 4..7 at c + 4, 8..11 at c - 4 (two bytes in the forward kernel's lane order)
        if (leaky) sgn = (unsigned)p.signs[(size_t)b * npix + pc] >> (co0 ? 4 : 0);
      } else {
        zw = p.z[u];
      }
      if (!ok) gw = cf_u32x4{0u, 0u, 0u, 0u};   // a pixel beyond the tail contributes gb = 0 to every sum
      const unsigned gws[4] = {gw.x, gw.y, gw.z, gw.w}, zws[4] = {zw.x, zw.y, zw.z, zw.w};
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        // the two bf16 of a dword, widened: low half << 16, high half masked
        const cf_f32x2 gv = {__builtin_bit_cast(float, gws[q] << 16), __builtin_bit_cast(float, gws[q] & 0xFFFF0000u)};
        cf_f32x2 gb = gv;
        if constexpr (SIGNS) {
          if (leaky) {
            const int pos = q < 2 ? 2 * q : 8 + 2 * (q - 2);   // (a constant of the unrolled loop)
            const cf_f32x2 f = {(sgn >> pos) & 1u ? 1.f : LEAKY_SLOPE, (sgn >> (pos + 1)) & 1u ? 1.f : LEAKY_SLOPE};
            gb = gv * f;
          }
          s1[q] += gb;
        } else {
          const cf_f32x2 zv = {__builtin_bit_cast(float, zws[q] << 16), __builtin_bit_cast(float, zws[q] & 0xFFFF0000u)};
          const cf_f32x2 xh = (zv - mu[q]) * is[q];
          if (leaky) {
            const cf_f32x2 yb = __builtin_elementwise_fma(ga[q], xh, be[q]);
            const cf_f32x2 f = {yb.x > 0.f ? 1.f : LEAKY_SLOPE, yb.y > 0.f ? 1.f : LEAKY_SLOPE};
            gb = gv * f;
          }
          s1[q] += gb;
          s2[q] = __builtin_elementwise_fma(gb, xh, s2[q]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) a1[q][j] = __builtin_elementwise_fma(gb, cf_f32x2{x[j], x[j]}, a1[q][j]);
      }
    }
    // cross-lane sums (DPP), then the four wavefronts through LDS -- the column layout of conv_first_bn_wgrad_kernel
#pragma unroll
    for (int q = 0; q < NP; ++q) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = 2 * q + h;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float v1 = wave_sum(h ? a1[q][j].y : a1[q][j].x);
          if (lane == 0) red[wave][c * PER + j] = v1;
        }
        const float t1 = wave_sum(h ? s1[q].y : s1[q].x), t2 = wave_sum(h ? s2[q].y : s2[q].x);
        if (lane == 0) {
          red[wave][c * PER + 2 * NJ] = t1;
          red[wave][c * PER + 2 * NJ + 1] = t2;
        }
      }
    }
    __syncthreads();
    for (int e = tid; e < COC * PER; e += CF_THREADS) {
      const int c = e / PER, kk = e - c * PER;
      if (!(kk >= NJ && kk < 2 * NJ)) prow[(co0 + c) * PER + kk] = red[0][e] + red[1][e] + red[2][e] + red[3][e];   // (the A2 columns come from the Gram matrix)
    }
    __syncthreads();
  }
}

// The sign-map sweep with TWO horizontally adjacent output pixels per lane and step (output width even).  The sweep above issues 8
// vector-memory instructions per pixel and pass (six 16-bit image loads, the gradient unit, the sign word) and waits on the CU's
// vector-memory path for most of its time (wait_any 0.55, issuing 0.29: profiles/r04_mfma_util.txt).  A pixel pair (2m, 2m + 1) reads
// image columns 4m - 1 .. 4m + 3 of three rows: two ALIGNED dwords per row, [4m - 4, 4m) for its last byte and [4m, 4m + 4) -- 6 + 2 + 1
// instructions per pair instead of 16.  Same arithmetic per element; a lane's pixels enter its sums in another order than above
// (the sums agree to fp32 rounding, not bit for bit).
__global__ __launch_bounds__(CF_THREADS, 3) void conv_first_bn_wgrad_pk2_kernel(const ConvFirstBnWgradParams p) {
  constexpr int NJ = 9, PER = 2 * NJ + 2, COC = 8, NP = COC / 2;
  __shared__ float red[4][COC * PER];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int npix = p.OH * p.OW, npair = npix >> 1, ow2 = p.OW >> 1;
  const int qbase = blockIdx.x * (CF_THREADS * CFW_PPT / 2);
  const int ncol = p.Cout * PER + NJ + NJ * NJ;
  const unsigned char* ib = reinterpret_cast<const unsigned char*>(p.in) + (size_t)b * p.IH * p.IW;
  const unsigned* sgp = reinterpret_cast<const unsigned*>(p.signs) + (size_t)b * npair;
  float* prow = p.part + (size_t)(b * gridDim.x + blockIdx.x) * ncol;
  for (int e = tid; e < NJ + NJ * NJ; e += CF_THREADS) prow[p.Cout * PER + e] = 0.f;   // P / G: the finalize reads the caller's
  const bool leaky = p.act == ACT_LEAKY;   // (uniform)
  for (int co0 = 0; co0 < p.Cout; co0 += COC) {
    cf_f32x2 a1[NP][NJ], s1[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      s1[q] = cf_f32x2{0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; ++j) a1[q][j] = cf_f32x2{0.f, 0.f};
    }
    for (int k = 0; k < CFW_PPT / 2; ++k) {
      const int pr = qbase + k * CF_THREADS + tid;
      const bool ok = pr < npair;
      const int pc = ok ? pr : 0;
      const int oy = (int)__umulhi((unsigned)pc, p.m_ow2), m = pc - oy * ow2;   // pixel pair (oy, 2m), (oy, 2m + 1)
      float x[2][NJ];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int iy = 2 * oy + kh - 1;  // <= IH - 1 (even sizes)
        const bool rok = ok && iy >= 0;
        const int o_ = rok ? iy * p.IW + 4 * m : 0;   // bytes o_ - 1 .. o_ + 3
        const unsigned wb = *reinterpret_cast<const unsigned*>(ib + o_);
        unsigned wa = *reinterpret_cast<const unsigned*>(ib + ((rok && m > 0) ? o_ - 4 : o_));
        wa = (rok && m > 0) ? wa >> 24 : 0u;   // column 4m - 1 (zero padding in the first column)
        const unsigned w = rok ? wb : 0u;
        x[0][kh * 3 + 0] = (float)wa;
        x[0][kh * 3 + 1] = (float)(w & 0xFFu);
        x[0][kh * 3 + 2] = (float)((w >> 8) & 0xFFu);
        x[1][kh * 3 + 0] = (float)((w >> 8) & 0xFFu);
        x[1][kh * 3 + 1] = (float)((w >> 16) & 0xFFu);
        x[1][kh * 3 + 2] = (float)(w >> 24);
      }
      const size_t u = ((size_t)b * p.Mb + (co0 >> 3)) * npix + 2 * (size_t)pc;
      cf_u32x4 gw[2] = {p.g[u], p.g[u + 1]};
      // a pixel's 16 sign bits: channel c < 4 or >= 12 at bit c, 4..7 at c + 4, 8..11 at c - 4; the pair's two words are one dword
      unsigned sg2 = leaky ? sgp[pc] >> (co0 ? 4 : 0) : 0u;
      if (!ok) gw[0] = gw[1] = cf_u32x4{0u, 0u, 0u, 0u};   // a pair beyond the tail contributes gb = 0 to every sum
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned gws[4] = {gw[e].x, gw[e].y, gw[e].z, gw[e].w};
        const unsigned sgn = e ? sg2 >> 16 : sg2;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const cf_f32x2 gv = {__builtin_bit_cast(float, gws[q] << 16), __builtin_bit_cast(float, gws[q] & 0xFFFF0000u)};
          cf_f32x2 gb = gv;
          if (leaky) {
            const int pos = q < 2 ? 2 * q : 8 + 2 * (q - 2);   // (a constant of the unrolled loop)
            const cf_f32x2 f = {(sgn >> pos) & 1u ? 1.f : LEAKY_SLOPE, (sgn >> (pos + 1)) & 1u ? 1.f : LEAKY_SLOPE};
            gb = gv * f;
          }
          s1[q] += gb;
#pragma unroll
          for (int j = 0; j < NJ; ++j) a1[q][j] = __builtin_elementwise_fma(gb, cf_f32x2{x[e][j], x[e][j]}, a1[q][j]);
        }
      }
    }
    // cross-lane sums (DPP), then the four wavefronts through LDS -- the column layout of conv_first_bn_wgrad_kernel
#pragma unroll
    for (int q = 0; q < NP; ++q) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = 2 * q + h;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float v1 = wave_sum(h ? a1[q][j].y : a1[q][j].x);
          if (lane == 0) red[wave][c * PER + j] = v1;
        }
        const float t1 = wave_sum(h ? s1[q].y : s1[q].x);
        if (lane == 0) {
          red[wave][c * PER + 2 * NJ] = t1;
          red[wave][c * PER + 2 * NJ + 1] = 0.f;   // (S2: derived by the finalize kernel)
        }
      }
    }
    __syncthreads();
    for (int e = tid; e < COC * PER; e += CF_THREADS) {
      const int c = e / PER, kk = e - c * PER;
      if (!(kk >= NJ && kk < 2 * NJ)) prow[(co0 + c) * PER + kk] = red[0][e] + red[1][e] + red[2][e] + red[3][e];   // (the A2 columns come from the Gram matrix)
    }
    __syncthreads();
  }
}

// sums [Cout*(2*NJ+2) + NJ (+ NJ*NJ)] -> dW (OIHW), dgamma, dbeta, each clamped to +-clip when clip > 0
__global__ void conv_first_bn_wgrad_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ mean,
                                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                    const float* __restrict__ w, float* __restrict__ dw,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int Cout, int NJ,
                                                    float inv_count, int training, int gram, float clip,
                                                    const float* __restrict__ ext_pg, int derive_s2) {
  const int PER = 2 * NJ + 2;
  const float* pg = ext_pg != nullptr ? ext_pg : sums + Cout * PER;  // P[NJ] then G[NJ][NJ]
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Cout * (NJ + 2)) return;
  const int c = e / (NJ + 2), j = e - c * (NJ + 2);
  const float* sc = sums + c * PER;
  const float S1 = sc[2 * NJ];
  float S2 = sc[2 * NJ + 1];
  if (derive_s2) {   // the sweep saw no z: sum gb * xh = invstd * (sum_j W[c][j] * A1[c][j] - mean * S1)   (z = W . patch, no conv bias)
    // (in double: for a uint8 image W . A1 and mean * S1 are both ~ pixel mean x |W| x S1 and cancel down to the covariance part)
    double t = 0.0;
    for (int k = 0; k < NJ; ++k) t = fma((double)w[c * NJ + k], (double)sc[k], t);
    S2 = (float)((double)invstd[c] * (t - (double)mean[c] * (double)S1));
  }
  float v;
  if (j < NJ) {
    const float c1 = gamma[c] * invstd[c];
    if (training) {
      const float Pj = pg[j];
      float a2;
      if (gram) {  // sum xh * patch_j = invstd * (sum_j' W[c][j'] G[j'][j] - mean * P[j])
        const float* G = pg + NJ;
        float t = 0.f;
        for (int k = 0; k < NJ; ++k) t = fmaf(w[c * NJ + k], G[k * NJ + j], t);
        a2 = invstd[c] * (t - mean[c] * Pj);
      } else {
        a2 = sc[NJ + j];
      }
      v = c1 * (sc[j] - S1 * inv_count * Pj - S2 * inv_count * a2);
    } else {
      v = c1 * sc[j];
    }
  } else {
    v = j == NJ ? S2 : S1;
  }
  if (clip > 0.f) v = fminf(fmaxf(v, -clip), clip);
  if (j < NJ) dw[c * NJ + j] = v;
  else if (j == NJ) dgamma[c] = v;
  else dbeta[c] = v;
}

static int first_tiles(int OH, int OW) { return cdiv(OH * OW, CF_THREADS * CF_PPT); }
static int first_wgrad_tiles(int OH, int OW) { return cdiv(OH * OW, CF_THREADS * CFW_PPT); }

// rows of the partial buffer the weight-gradient kernels fill
extern "C" int yogo_conv_first_wgrad_rows(int B, int IH, int IW, int stride, int* rows) {
  YOGO_CHECK_ARG(rows && (stride == 1 || stride == 2), "conv_first_wgrad_rows: bad arguments");
  const int OH = (IH - 1) / stride + 1, OW = (IW - 1) / stride + 1;
  *rows = B * first_wgrad_tiles(OH, OW);
  return YOGO_OK;
}

extern "C" int yogo_conv_first_stats_rows(int B, int IH, int IW, int stride, int* rows) {
  YOGO_CHECK_ARG(rows && (stride == 1 || stride == 2), "conv_first_stats_rows: bad arguments");
  const int OH = (IH - 1) / stride + 1, OW = (IW - 1) / stride + 1;
  *rows = B * first_tiles(OH, OW);
  return YOGO_OK;
}

static int conv_first_fwd_impl(const void* in, int in_dtype, const float* w, const float* bias, float* out, void* out_bf16,
                               float* out_pre, const float* chan_scale, float* stats_part, int B, int Cin, int Cout, int IH,
                               int IW, int stride, int act, hipStream_t stream);

// in_dtype: 0 = uint8, 1 = float32
extern "C" int yogo_conv_first_fwd(const void* in, int in_dtype, const float* w, const float* bias, float* out,
                                   float* out_pre, const float* chan_scale, float* stats_part, int B, int Cin, int Cout,
                                   int IH, int IW, int stride, int act, hipStream_t stream) {
  YOGO_CHECK_ARG(out != nullptr, "conv_first_fwd: null pointer");
  return conv_first_fwd_impl(in, in_dtype, w, bias, out, nullptr, out_pre, chan_scale, stats_part, B, Cin, Cout, IH, IW, stride, act,
                             stream);
}

// training variant with bf16 NCHW8c output (+ fp32 BatchNorm partial sums): out = chan_scale * act(conv + bias)
extern "C" int yogo_conv_first_fwd_train_bf16(const void* in, int in_dtype, const float* w, const float* bias, void* out_bf16,
                                              const float* chan_scale, float* stats_part, int B, int Cin, int Cout, int IH,
                                              int IW, int stride, int act, hipStream_t stream) {
  YOGO_CHECK_ARG(out_bf16 != nullptr, "conv_first_fwd_train_bf16: null pointer");
  return conv_first_fwd_impl(in, in_dtype, w, bias, nullptr, out_bf16, nullptr, chan_scale, stats_part, B, Cin, Cout, IH, IW, stride,
                             act, stream);
}

static int conv_first_fwd_impl(const void* in, int in_dtype, const float* w, const float* bias, float* out, void* out_bf16,
                               float* out_pre, const float* chan_scale, float* stats_part, int B, int Cin, int Cout, int IH,
                               int IW, int stride, int act, hipStream_t stream) {
  YOGO_CHECK_ARG(in && w, "conv_first_fwd: null pointer");
  YOGO_CHECK_ARG((Cin == 1 || Cin == 3) && Cout > 0 && IH > 0 && IW > 0 && (stride == 1 || stride == 2) && B >= 0,
                 "conv_first_fwd: unsupported shape Cin=%d Cout=%d stride=%d", Cin, Cout, stride);
  YOGO_CHECK_ARG(in_dtype == 0 || in_dtype == 1, "conv_first_fwd: in_dtype must be 0 (uint8) or 1 (float32)");
  ConvFirstParams p{};
  p.out_bf16 = reinterpret_cast<cf_u32x4*>(out_bf16); p.Mb = ((Cout + 15) / 16) * 2;
  p.in = in; p.w = w; p.bias = bias; p.out = out; p.out_pre = out_pre; p.chan_scale = chan_scale; p.stats_part = stats_part;
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.IH = IH; p.IW = IW; p.stride = stride; p.act = act;
  p.OH = (IH - 1) / stride + 1; p.OW = (IW - 1) / stride + 1;
  if (B == 0) return YOGO_OK;
  dim3 grid(first_tiles(p.OH, p.OW), B);
  if (out_bf16 != nullptr) {  // training, bf16 NCHW8c output
    if (in_dtype == 0 && Cin == 1) hipLaunchKernelGGL((conv_first_train_bf16_kernel<uint8_t, 1>), grid, dim3(CF_THREADS), 0, stream, p);
    else if (in_dtype == 0) hipLaunchKernelGGL((conv_first_train_bf16_kernel<uint8_t, 3>), grid, dim3(CF_THREADS), 0, stream, p);
    else if (Cin == 1) hipLaunchKernelGGL((conv_first_train_bf16_kernel<float, 1>), grid, dim3(CF_THREADS), 0, stream, p);
    else hipLaunchKernelGGL((conv_first_train_bf16_kernel<float, 3>), grid, dim3(CF_THREADS), 0, stream, p);
    YOGO_CHECK_LAUNCH("conv_first_fwd");
    return YOGO_OK;
  }
  if (in_dtype == 0 && Cin == 1) hipLaunchKernelGGL((conv_first_kernel<uint8_t, 1>), grid, dim3(CF_THREADS), 0, stream, p);
  else if (in_dtype == 0) hipLaunchKernelGGL((conv_first_kernel<uint8_t, 3>), grid, dim3(CF_THREADS), 0, stream, p);
  else if (Cin == 1) hipLaunchKernelGGL((conv_first_kernel<float, 1>), grid, dim3(CF_THREADS), 0, stream, p);
  else hipLaunchKernelGGL((conv_first_kernel<float, 3>), grid, dim3(CF_THREADS), 0, stream, p);
  YOGO_CHECK_LAUNCH("conv_first_fwd");
  return YOGO_OK;
}

static int conv_first_wgrad_impl(const void* in, int in_dtype, const float* dy, const void* dy_bf16, float* part, int B, int Cin,
                                 int Cout, int IH, int IW, int stride, hipStream_t stream);

// partial weight/bias gradients; part must hold rows*Cout*(Cin*9+1) floats with rows from yogo_conv_first_wgrad_rows
extern "C" int yogo_conv_first_wgrad(const void* in, int in_dtype, const float* dy, float* part, int B, int Cin, int Cout,
                                     int IH, int IW, int stride, hipStream_t stream) {
  YOGO_CHECK_ARG(dy != nullptr, "conv_first_wgrad: null pointer");
  return conv_first_wgrad_impl(in, in_dtype, dy, nullptr, part, B, Cin, Cout, IH, IW, stride, stream);
}

// same with the gradient in bf16 NCHW8c
extern "C" int yogo_conv_first_wgrad_bf16g(const void* in, int in_dtype, const void* dy_bf16, float* part, int B, int Cin,
                                           int Cout, int IH, int IW, int stride, hipStream_t stream) {
  YOGO_CHECK_ARG(dy_bf16 != nullptr, "conv_first_wgrad_bf16g: null pointer");
  return conv_first_wgrad_impl(in, in_dtype, nullptr, dy_bf16, part, B, Cin, Cout, IH, IW, stride, stream);
}

static int conv_first_wgrad_impl(const void* in, int in_dtype, const float* dy, const void* dy_bf16, float* part, int B, int Cin,
                                 int Cout, int IH, int IW, int stride, hipStream_t stream) {
  YOGO_CHECK_ARG(in && part, "conv_first_wgrad: null pointer");
  YOGO_CHECK_ARG((Cin == 1 || Cin == 3) && Cout > 0 && (stride == 1 || stride == 2), "conv_first_wgrad: unsupported shape");
  ConvFirstWgradParams p{};
  p.dy_bf16 = reinterpret_cast<const cf_u32x4*>(dy_bf16); p.Mb = ((Cout + 15) / 16) * 2;
  p.in = in; p.dy = dy; p.part = part; p.B = B; p.Cin = Cin; p.Cout = Cout; p.IH = IH; p.IW = IW; p.stride = stride;
  p.OH = (IH - 1) / stride + 1; p.OW = (IW - 1) / stride + 1;
  if (B == 0) return YOGO_OK;
  dim3 grid(first_wgrad_tiles(p.OH, p.OW), B);
#define CFW_LAUNCH(TIN, CIN_)                                                                                          \
  do {                                                                                                                 \
    if (dy_bf16 != nullptr) hipLaunchKernelGGL((conv_first_wgrad_kernel<TIN, CIN_, true>), grid, dim3(CF_THREADS), 0, stream, p); \
    else hipLaunchKernelGGL((conv_first_wgrad_kernel<TIN, CIN_, false>), grid, dim3(CF_THREADS), 0, stream, p);        \
  } while (0)
  if (in_dtype == 0 && Cin == 1) CFW_LAUNCH(uint8_t, 1);
  else if (in_dtype == 0) CFW_LAUNCH(uint8_t, 3);
  else if (Cin == 1) CFW_LAUNCH(float, 1);
  else CFW_LAUNCH(float, 3);
#undef CFW_LAUNCH
  YOGO_CHECK_LAUNCH("conv_first_wgrad");
  return YOGO_OK;
}

// ---- fused layer-0 backward (BatchNorm + activation + first-conv weight gradient), bf16 NCHW8c g / z ----------------------
extern "C" int yogo_conv_first_bn_wgrad_cols(int Cin, int Cout, int* cols) {
  YOGO_CHECK_ARG(cols && (Cin == 1 || Cin == 3) && Cout > 0, "conv_first_bn_wgrad_cols: bad arguments");
  *cols = Cout * (2 * Cin * 9 + 2) + Cin * 9 + (Cin == 1 ? 81 : 0);  // Cin = 1: + the 9x9 Gram matrix of the patches
  return YOGO_OK;
}

// part: rows (yogo_conv_first_wgrad_rows) x cols floats.  Follow with yogo_partials_reduce(part, rows, cols, 0, sums) and
// yogo_conv_first_bn_wgrad_finalize.
static int conv_first_bn_wgrad_impl(const void* in, int in_dtype, const void* g, const void* z, const void* signs, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta, float* part, int B, int Cin, int Cout,
                                    int IH, int IW, int stride, int act, int ext_gram, hipStream_t stream);
extern "C" int yogo_conv_first_bn_wgrad_bf16(const void* in, int in_dtype, const void* g, const void* z, const float* mean,
                                             const float* invstd, const float* gamma, const float* beta, float* part, int B,
                                             int Cin, int Cout, int IH, int IW, int stride, int act, hipStream_t stream) {
  return conv_first_bn_wgrad_impl(in, in_dtype, g, z, nullptr, mean, invstd, gamma, beta, part, B, Cin, Cout, IH, IW, stride, act, 0, stream);
}
// the same when the caller holds P and G of this batch already (yogo_conv_first_gram in the forward pass): no Gram pass;
// finish with yogo_conv_first_bn_wgrad_finalize_xg
extern "C" int yogo_conv_first_bn_wgrad_bf16_xg(const void* in, int in_dtype, const void* g, const void* z, const float* mean,
                                                const float* invstd, const float* gamma, const float* beta, float* part, int B,
                                                int Cin, int Cout, int IH, int IW, int stride, int act, hipStream_t stream) {
  YOGO_CHECK_ARG(Cin == 1, "conv_first_bn_wgrad_bf16_xg: one input channel only");
  return conv_first_bn_wgrad_impl(in, in_dtype, g, z, nullptr, mean, invstd, gamma, beta, part, B, Cin, Cout, IH, IW, stride, act, 1, stream);
}
extern "C" int yogo_conv_first_mfma_supported(int in_dtype, int Cin, int Cout, int IH, int IW, int stride);   // conv_first_mfma.hip
// the sign-map sweep takes two pixels per lane where the output width is even; a switch only in the test-hooks / diagnostic builds
#if defined(YOGO_TEST_HOOKS) || defined(YOGO_DIAG)
static bool g_cf_pairs = true;
extern "C" int yogo_hook_conv_first_bn_wgrad_pairs(int on) {
  g_cf_pairs = on != 0;
  return YOGO_OK;
}
#else
static constexpr bool g_cf_pairs = true;
#endif
// 1 when the sweep can run WITHOUT the saved conv output: from the sign map of yogo_conv_first_mfma_signs (uint8 one-channel image,
// stride 2, even sizes, Cout = 8 or 16, no activation or LeakyReLU, no conv bias, caller-held Gram matrix)
static bool conv_first_fast_shape(int in_dtype, int Cin, int IH, int IW, int stride) {
  return in_dtype == 0 && Cin == 1 && stride == 2 && IH % 2 == 0 && IW % 2 == 0 && (long long)IH * IW < (1ll << 31);
}
extern "C" int yogo_conv_first_bn_wgrad_xs_supported(int in_dtype, int Cin, int Cout, int IH, int IW, int stride, int act) {
  return conv_first_fast_shape(in_dtype, Cin, IH, IW, stride) && (Cout == 8 || Cout == 16) && (act == ACT_NONE || act == ACT_LEAKY) &&
         yogo_conv_first_mfma_supported(in_dtype, Cin, Cout, IH, IW, stride);
}
// yogo_conv_first_bn_wgrad_bf16_xg without z: signs = the sign map of the forward pass (may be NULL when act is ACT_NONE); the S2
// column of the partial rows stays zero -- finish with yogo_conv_first_bn_wgrad_finalize_xs, which derives it (see the kernel)
extern "C" int yogo_conv_first_bn_wgrad_bf16_xs(const void* in, int in_dtype, const void* g, const void* signs, const float* mean,
                                                const float* invstd, const float* gamma, const float* beta, float* part, int B,
                                                int Cin, int Cout, int IH, int IW, int stride, int act, hipStream_t stream) {
  YOGO_CHECK_ARG(yogo_conv_first_bn_wgrad_xs_supported(in_dtype, Cin, Cout, IH, IW, stride, act), "conv_first_bn_wgrad_bf16_xs: unsupported shape");
  YOGO_CHECK_ARG(signs != nullptr || act == ACT_NONE, "conv_first_bn_wgrad_bf16_xs: LeakyReLU needs the sign map");
  return conv_first_bn_wgrad_impl(in, in_dtype, g, nullptr, signs != nullptr ? signs : g, mean, invstd, gamma, beta, part, B, Cin, Cout, IH, IW, stride, act, 1, stream);
}
static int conv_first_bn_wgrad_impl(const void* in, int in_dtype, const void* g, const void* z, const void* signs, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta, float* part, int B, int Cin, int Cout,
                                    int IH, int IW, int stride, int act, int ext_gram, hipStream_t stream) {
  YOGO_CHECK_ARG(in && g && (z || signs) && mean && invstd && gamma && beta && part, "conv_first_bn_wgrad_bf16: null pointer");
  YOGO_CHECK_ARG((Cin == 1 || Cin == 3) && Cout > 0 && (stride == 1 || stride == 2) && (in_dtype == 0 || in_dtype == 1),
                 "conv_first_bn_wgrad_bf16: unsupported shape");
  ConvFirstBnWgradParams p{};
  p.in = in; p.g = reinterpret_cast<const cf_u32x4*>(g); p.z = reinterpret_cast<const cf_u32x4*>(z); p.signs = reinterpret_cast<const unsigned short*>(signs);
  p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta; p.part = part;
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.Mb = ((Cout + 15) / 16) * 2; p.IH = IH; p.IW = IW; p.stride = stride; p.act = act;
  p.OH = (IH - 1) / stride + 1; p.OW = (IW - 1) / stride + 1; p.ext_gram = ext_gram;
  p.m_ow = ((long long)p.OH * p.OW * p.OW < (1ll << 32) && p.OW > 1) ? (unsigned)(((1ull << 32) + (unsigned)p.OW - 1ull) / (unsigned)p.OW) : 0u;
  if (B == 0) return YOGO_OK;
  dim3 grid(first_wgrad_tiles(p.OH, p.OW), B);
  const bool fast = conv_first_fast_shape(in_dtype, Cin, IH, IW, stride);
  p.m_ow2 = p.OW >= 4 ? (unsigned)(((1ull << 32) + (unsigned)(p.OW / 2) - 1ull) / (unsigned)(p.OW / 2)) : 0u;
  if (z == nullptr && p.OW % 2 == 0 && p.OW >= 4 && p.m_ow != 0 && g_cf_pairs)   // (yogo_conv_first_bn_wgrad_bf16_xs checked the shape)
    hipLaunchKernelGGL(conv_first_bn_wgrad_pk2_kernel, grid, dim3(CF_THREADS), 0, stream, p);
  else if (z == nullptr)
    hipLaunchKernelGGL(conv_first_bn_wgrad_pk_kernel<true>, grid, dim3(CF_THREADS), 0, stream, p);
  else if (fast && ext_gram && Cout % 8 == 0 && (act == ACT_NONE || act == ACT_LEAKY))
    hipLaunchKernelGGL(conv_first_bn_wgrad_pk_kernel<false>, grid, dim3(CF_THREADS), 0, stream, p);
  else if (fast)
    hipLaunchKernelGGL((conv_first_bn_wgrad_kernel<uint8_t, 1, 8, true, true>), grid, dim3(CF_THREADS), 0, stream, p);
  else if (in_dtype == 0 && Cin == 1) hipLaunchKernelGGL((conv_first_bn_wgrad_kernel<uint8_t, 1, 8, true>), grid, dim3(CF_THREADS), 0, stream, p);
  else if (in_dtype == 0) hipLaunchKernelGGL((conv_first_bn_wgrad_kernel<uint8_t, 3, 2, false>), grid, dim3(CF_THREADS), 0, stream, p);
  else if (Cin == 1) hipLaunchKernelGGL((conv_first_bn_wgrad_kernel<float, 1, 8, true>), grid, dim3(CF_THREADS), 0, stream, p);
  else hipLaunchKernelGGL((conv_first_bn_wgrad_kernel<float, 3, 2, false>), grid, dim3(CF_THREADS), 0, stream, p);
  YOGO_CHECK_LAUNCH("conv_first_bn_wgrad_bf16");
  return YOGO_OK;
}

static int conv_first_bn_wgrad_finalize_impl(const float* sums, const float* ext_pg, const float* mean, const float* invstd,
                                             const float* gamma, const float* w_oihw, float* dw, float* dgamma, float* dbeta, int B,
                                             int Cin, int Cout, int IH, int IW, int stride, int training, float clip, int derive_s2,
                                             hipStream_t stream);
extern "C" int yogo_conv_first_bn_wgrad_finalize(const float* sums, const float* mean, const float* invstd, const float* gamma,
                                                 const float* w_oihw, float* dw, float* dgamma, float* dbeta, int B, int Cin,
                                                 int Cout, int IH, int IW, int stride, int training, float clip,
                                                 hipStream_t stream) {
  return conv_first_bn_wgrad_finalize_impl(sums, nullptr, mean, invstd, gamma, w_oihw, dw, dgamma, dbeta, B, Cin, Cout, IH, IW, stride,
                                           training, clip, 0, stream);
}
// gram: float[90] = P[9] then G[9][9] of the batch (yogo_conv_first_gram)
extern "C" int yogo_conv_first_bn_wgrad_finalize_xg(const float* sums, const float* gram, const float* mean, const float* invstd,
                                                    const float* gamma, const float* w_oihw, float* dw, float* dgamma, float* dbeta,
                                                    int B, int Cin, int Cout, int IH, int IW, int stride, int training, float clip,
                                                    hipStream_t stream) {
  YOGO_CHECK_ARG(gram != nullptr && Cin == 1, "conv_first_bn_wgrad_finalize_xg: bad arguments");
  return conv_first_bn_wgrad_finalize_impl(sums, gram, mean, invstd, gamma, w_oihw, dw, dgamma, dbeta, B, Cin, Cout, IH, IW, stride,
                                           training, clip, 0, stream);
}
// ... behind yogo_conv_first_bn_wgrad_bf16_xs: S2 = sum gb * xh is derived from A1 and S1; w_oihw = the weights the FORWARD pass
// multiplied with (rounded to bf16)
extern "C" int yogo_conv_first_bn_wgrad_finalize_xs(const float* sums, const float* gram, const float* mean, const float* invstd,
                                                    const float* gamma, const float* w_oihw, float* dw, float* dgamma, float* dbeta,
                                                    int B, int Cin, int Cout, int IH, int IW, int stride, int training, float clip,
                                                    hipStream_t stream) {
  YOGO_CHECK_ARG(gram != nullptr && Cin == 1, "conv_first_bn_wgrad_finalize_xs: bad arguments");
  return conv_first_bn_wgrad_finalize_impl(sums, gram, mean, invstd, gamma, w_oihw, dw, dgamma, dbeta, B, Cin, Cout, IH, IW, stride,
                                           training, clip, 1, stream);
}
static int conv_first_bn_wgrad_finalize_impl(const float* sums, const float* ext_pg, const float* mean, const float* invstd,
                                             const float* gamma, const float* w_oihw, float* dw, float* dgamma, float* dbeta, int B,
                                             int Cin, int Cout, int IH, int IW, int stride, int training, float clip, int derive_s2,
                                             hipStream_t stream) {
  YOGO_CHECK_ARG(sums && mean && invstd && gamma && w_oihw && dw && dgamma && dbeta, "conv_first_bn_wgrad_finalize: null pointer");
  const int OH = (IH - 1) / stride + 1, OW = (IW - 1) / stride + 1, NJ = Cin * 9;
  const int n = Cout * (NJ + 2);
  hipLaunchKernelGGL(conv_first_bn_wgrad_finalize_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, sums, mean, invstd, gamma,
                     w_oihw, dw, dgamma, dbeta, Cout, NJ, 1.0f / ((float)B * (float)OH * (float)OW), training, Cin == 1 ? 1 : 0, clip, ext_pg, derive_s2);
  YOGO_CHECK_LAUNCH("conv_first_bn_wgrad_finalize");
  return YOGO_OK;
}
