// Weight (and bias) gradient of the 3x3 / 1x1 convolutions on the fp32 matrix cores, gfx950.
// Replaces cuDNN's convolution-backward-filter behind loss.backward() in the reference training step
// (yogo/train.py:322) and the 26 per-parameter clamp hooks of yogo/model.py:76-77 (K12): the
// clamp is applied by the deterministic slab reduction that finishes the gradient.
//
// GEMM view:  dW[t][co][ci] = sum_{b, pixel} g[b][co][pixel] * x[b][ci][pixel*stride + tap t]
//   M axis = co (A operand, lane&31), N axis = ci (B operand, lane&31), K axis = output pixel, two consecutive
//   pixels of one output row per v_mfma_f32_32x32x2_f32 (lane>>5).
// The pixel axis is the long one (B*OH*OW ~ 8e5 .. 1.3e7): split-K over workgroups, each accumulating a full
// [T][32*MBW][32*NBW] slab in registers over its share of (image, row, column-chunk) units, staged through LDS with
// odd pitches (conflict-free channel-strided reads).  Slabs are summed in a fixed order by a second kernel
// (bitwise reproducible; no float atomics), which also writes OIHW and clamps to +-clip.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int wg_u32x4 __attribute__((ext_vector_type(4)));

#define WG_MAX_TAPS 9
#define WGRAD_LDS_BUDGET (80 * 1024)

struct WgradParams {
  const float* x;   // [B][N][IH][IW]  layer input
  const float* g;   // [B][M][OH][OW]  grad w.r.t. conv output
  float* slab;      // [nsplit*KS][T][Mpad][Npad]
  float* bias_part; // optional [nsplit][Mpad]
  int B, N, M, Npad, Mpad, IH, IW, OH, OW, pad;
  int nchunk_w, base_w, rem_w, wce;   // column chunks per row (balanced), staged (even, zero padded) chunk width
  int xw, per_ch;                     // staged input columns per line, elements per channel (xrows * xw)
  unsigned inv_wce, inv_xw, inv_pc;   // ceil(2^32 / d) magic numbers for the flattened tile indexing
  int units, units_per_split;
  int gp, xp;                         // LDS pitches (odd: conflict-free channel-strided reads)
  int x_off;                          // float offset of the x tile in LDS
  int lds_dummy;                      // float offset of a scratch word (stores of padding lanes)
  int Nb, Mbk;                        // BF: channel blocks of the bf16 NCHW8c x / g tensors
};

// One unit = one output row segment (<= WC pixels) of one image.  Per unit the workgroup stages g[MBW*32][wce] and
// x[NBW*32][3][xw] in LDS and every wavefront runs wce/2 k-steps of T MFMAs on its (co-block, ci-block) pair.
// The global loads of unit u+1 are issued into registers BEFORE the MFMA loop of unit u and committed to LDS after it
// (async-STAGE split), so HBM/L2 latency hides under ~10k cycles of matrix work.
// Wavefront jobs: MBW co-blocks x NBW ci-blocks x KS pixel-splits x TG tap rows (TG = 3 for a 3x3 kernel: each wavefront
// owns the three taps of one kernel row, i.e. 3 accumulator tiles = 48 VGPRs, which leaves room for the prefetch registers
// and lets 3 wavefronts share each SIMD).  MBW*NBW*KS = 4, so a workgroup is 12 wavefronts (3x3) or 4 (1x1).
// BF: x and g are bf16 NCHW8c tensors (16-byte units of 8 channels); they are widened to fp32 while being committed to LDS,
// the matrix work stays exact fp32 MFMA.
template <int MBW, int NBW, int KS, int T, int S, bool BF = false>
__global__ __launch_bounds__(64 * MBW * NBW * KS * (T == 1 ? 1 : 3)) void wgrad_f32_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TG = (T == 1) ? 1 : 3;                         // tap groups (kernel rows)
  constexpr int TT = T / TG;                                   // taps per wavefront
  constexpr int NT = 64 * MBW * NBW * KS * TG;                 // threads per workgroup
  constexpr int XR = (T == 1) ? 1 : 3;
  constexpr int WC = (S == 1) ? 64 : 32;                       // max staged chunk width
  constexpr int XWMAX = (WC - 1) * S + ((T == 1) ? 1 : 3);
  constexpr int NGQ = (MBW * 32 * WC + NT - 1) / NT;           // g elements per lane
  constexpr int NXQ = (NBW * 32 * XR * XWMAX + NT - 1) / NT;   // x elements per lane
  float* ldsG = smem;            // [MBW*32][gp]
  float* ldsX = smem + p.x_off;  // [NBW*32][XR][xp]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave % TG;
  const int ks = (wave / TG) % KS;
  const int nb = (wave / (TG * KS)) % NBW;
  const int mb = wave / (TG * KS * NBW);
  const int split = blockIdx.x;
  const int n0 = blockIdx.y * (NBW * 32);
  const int m0 = blockIdx.z * (MBW * 32);
  const int xcs = XR * p.xp;  // channel stride in the x tile

  f32x16 acc[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;  // bias partial of row (m0 + tid), tid < MBW*32

  const int u_begin = split * p.units_per_split;
  const int u_end = min(p.units, u_begin + p.units_per_split);
  const size_t gplane = (size_t)p.OH * p.OW, xplane = (size_t)p.IH * p.IW;
  const int trow = tg * p.xp;  // this wavefront's kernel row inside the staged x tile; its taps are columns 0..TT-1

  const int gtotal = MBW * 32 * p.wce;
  const int xtotal = NBW * 32 * p.per_ch;
  float gq[BF ? 1 : NGQ], xq[BF ? 1 : NXQ];
  // BF mode: units of 8 channels
  constexpr int NGU = (MBW * 4 * WC + NT - 1) / NT;
  constexpr int NXU = (NBW * 4 * XR * XWMAX + NT - 1) / NT;
  wg_u32x4 gu[BF ? NGU : 1], xu[BF ? NXU : 1];
  const int gtotal_u = MBW * 4 * p.wce;
  const int xtotal_u = NBW * 4 * p.per_ch;

  // unit -> (image, output row, first column, real width)
#define WG_UNIT(U, B_, OY_, OX0_, WCR_)                                   \
  const int cwu_ = (U) % p.nchunk_w;                                      \
  const int rowidu_ = (U) / p.nchunk_w;                                   \
  const int OY_ = rowidu_ % p.OH;                                         \
  const int B_ = rowidu_ / p.OH;                                          \
  const int OX0_ = cwu_ * p.base_w + min(cwu_, p.rem_w);                  \
  const int WCR_ = p.base_w + (cwu_ < p.rem_w ? 1 : 0);

#define WG_ISSUE(U)                                                                                         \
  {                                                                                                         \
    WG_UNIT(U, b_, oy_, ox0_, wc_)                                                                          \
    const float* gb_ = p.g + ((size_t)b_ * p.M) * gplane + (size_t)oy_ * p.OW + ox0_;                       \
    _Pragma("unroll") for (int i = 0; i < NGQ; ++i) {                                                       \
      const int e_ = min(tid + NT * i, gtotal - 1);                                                        \
      const int r_ = __umulhi((unsigned)e_, p.inv_wce);                                                     \
      const int c_ = e_ - r_ * p.wce;                                                                       \
      const bool ok_ = (m0 + r_ < p.M) && (c_ < wc_);                                                       \
      gq[i] = gb_[ok_ ? (m0 + r_) * (int)gplane + c_ : 0];                                               \
    }                                                                                                       \
    const float* xb_ = p.x + ((size_t)b_ * p.N) * xplane;                                                   \
    const int iy0_ = oy_ * S - p.pad, ix0_ = ox0_ * S - p.pad;                                              \
    _Pragma("unroll") for (int i = 0; i < NXQ; ++i) {                                                       \
      const int e_ = min(tid + NT * i, xtotal - 1);                                                        \
      const int ch_ = __umulhi((unsigned)e_, p.inv_pc);                                                     \
      const int rm_ = e_ - ch_ * p.per_ch;                                                                  \
      const int r_ = __umulhi((unsigned)rm_, p.inv_xw);                                                     \
      const int c_ = rm_ - r_ * p.xw;                                                                       \
      const int n_ = n0 + ch_, iy_ = iy0_ + r_, ix_ = ix0_ + c_;                                            \
      const bool ok_ = (n_ < p.N) && (iy_ >= 0) && (iy_ < p.IH) && (ix_ >= 0) && (ix_ < p.IW);              \
      xq[i] = xb_[ok_ ? (n_ * p.IH + iy_) * p.IW + ix_ : 0];                                \
    }                                                                                                       \
  }

#define WG_COMMIT(U)                                                                                        \
  {                                                                                                         \
    WG_UNIT(U, b_, oy_, ox0_, wc_)                                                                          \
    (void)b_;                                                                                               \
    _Pragma("unroll") for (int i = 0; i < NGQ; ++i) {                                                       \
      const int e_ = tid + NT * i;                                                                         \
      const int ec_ = min(e_, gtotal - 1);                                                                  \
      const int r_ = __umulhi((unsigned)ec_, p.inv_wce);                                                    \
      const int c_ = ec_ - r_ * p.wce;                                                                      \
      const bool ok_ = (m0 + r_ < p.M) && (c_ < wc_);                                                       \
      smem[e_ < gtotal ? r_ * p.gp + c_ : p.lds_dummy] = ok_ ? gq[i] : 0.f;                                 \
    }                                                                                                       \
    const int iy0_ = oy_ * S - p.pad, ix0_ = ox0_ * S - p.pad;                                              \
    _Pragma("unroll") for (int i = 0; i < NXQ; ++i) {                                                       \
      const int e_ = tid + NT * i;                                                                         \
      const int ec_ = min(e_, xtotal - 1);                                                                  \
      const int ch_ = __umulhi((unsigned)ec_, p.inv_pc);                                                    \
      const int rm_ = ec_ - ch_ * p.per_ch;                                                                 \
      const int r_ = __umulhi((unsigned)rm_, p.inv_xw);                                                     \
      const int c_ = rm_ - r_ * p.xw;                                                                       \
      const int n_ = n0 + ch_, iy_ = iy0_ + r_, ix_ = ix0_ + c_;                                            \
      const bool ok_ = (n_ < p.N) && (iy_ >= 0) && (iy_ < p.IH) && (ix_ >= 0) && (ix_ < p.IW);              \
      smem[e_ < xtotal ? p.x_off + ch_ * xcs + r_ * p.xp + c_ : p.lds_dummy] = ok_ ? xq[i] : 0.f;           \
    }                                                                                                       \
  }

#define WG_ISSUE_BF(U)                                                                                      \
  {                                                                                                         \
    WG_UNIT(U, b_, oy_, ox0_, wc_)                                                                          \
    const wg_u32x4* gb_ = reinterpret_cast<const wg_u32x4*>(p.g) + ((size_t)b_ * p.Mbk) * gplane + (size_t)oy_ * p.OW + ox0_; \
    _Pragma("unroll") for (int i = 0; i < NGU; ++i) {                                                       \
      const int e_ = min(tid + NT * i, gtotal_u - 1);                                                       \
      const int cb_ = __umulhi((unsigned)e_, p.inv_wce);                                                    \
      const int c_ = e_ - cb_ * p.wce;                                                                      \
      const bool ok_ = ((m0 >> 3) + cb_ < p.Mbk) && (c_ < wc_);                                             \
      gu[i] = gb_[ok_ ? ((m0 >> 3) + cb_) * (int)gplane + c_ : 0];                                          \
    }                                                                                                       \
    const wg_u32x4* xb_ = reinterpret_cast<const wg_u32x4*>(p.x) + ((size_t)b_ * p.Nb) * xplane;            \
    const int iy0_ = oy_ * S - p.pad, ix0_ = ox0_ * S - p.pad;                                              \
    _Pragma("unroll") for (int i = 0; i < NXU; ++i) {                                                       \
      const int e_ = min(tid + NT * i, xtotal_u - 1);                                                       \
      const int cb_ = __umulhi((unsigned)e_, p.inv_pc);                                                     \
      const int rm_ = e_ - cb_ * p.per_ch;                                                                  \
      const int r_ = __umulhi((unsigned)rm_, p.inv_xw);                                                     \
      const int c_ = rm_ - r_ * p.xw;                                                                       \
      const int nb_ = (n0 >> 3) + cb_, iy_ = iy0_ + r_, ix_ = ix0_ + c_;                                    \
      const bool ok_ = (nb_ < p.Nb) && (iy_ >= 0) && (iy_ < p.IH) && (ix_ >= 0) && (ix_ < p.IW);            \
      xu[i] = xb_[ok_ ? (nb_ * p.IH + iy_) * p.IW + ix_ : 0];                                               \
    }                                                                                                       \
  }
#define WG_COMMIT_BF(U)                                                                                     \
  {                                                                                                         \
    WG_UNIT(U, b_, oy_, ox0_, wc_)                                                                          \
    (void)b_;                                                                                               \
    _Pragma("unroll") for (int i = 0; i < NGU; ++i) {                                                       \
      const int e_ = tid + NT * i;                                                                          \
      const int ec_ = min(e_, gtotal_u - 1);                                                                \
      const int cb_ = __umulhi((unsigned)ec_, p.inv_wce);                                                   \
      const int c_ = ec_ - cb_ * p.wce;                                                                     \
      const bool ok_ = ((m0 >> 3) + cb_ < p.Mbk) && (c_ < wc_);                                             \
      const wg_bf16x8 v_ = __builtin_bit_cast(wg_bf16x8, gu[i]);                                            \
      _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                         \
        smem[e_ < gtotal_u ? (cb_ * 8 + j) * p.gp + c_ : p.lds_dummy] = ok_ ? (float)v_[j] : 0.f;           \
    }                                                                                                       \
    const int iy0_ = oy_ * S - p.pad, ix0_ = ox0_ * S - p.pad;                                              \
    _Pragma("unroll") for (int i = 0; i < NXU; ++i) {                                                       \
      const int e_ = tid + NT * i;                                                                          \
      const int ec_ = min(e_, xtotal_u - 1);                                                                \
      const int cb_ = __umulhi((unsigned)ec_, p.inv_pc);                                                    \
      const int rm_ = ec_ - cb_ * p.per_ch;                                                                 \
      const int r_ = __umulhi((unsigned)rm_, p.inv_xw);                                                     \
      const int c_ = rm_ - r_ * p.xw;                                                                       \
      const int nb_ = (n0 >> 3) + cb_, iy_ = iy0_ + r_, ix_ = ix0_ + c_;                                    \
      const bool ok_ = (nb_ < p.Nb) && (iy_ >= 0) && (iy_ < p.IH) && (ix_ >= 0) && (ix_ < p.IW);            \
      const wg_bf16x8 v_ = __builtin_bit_cast(wg_bf16x8, xu[i]);                                            \
      _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                         \
        smem[e_ < xtotal_u ? p.x_off + (cb_ * 8 + j) * xcs + r_ * p.xp + c_ : p.lds_dummy] = ok_ ? (float)v_[j] : 0.f; \
    }                                                                                                       \
  }
#define WG_ISSUE_ANY(U)   \
  if constexpr (BF) {     \
    WG_ISSUE_BF(U)        \
  } else {                \
    WG_ISSUE(U)           \
  }
#define WG_COMMIT_ANY(U)  \
  if constexpr (BF) {     \
    WG_COMMIT_BF(U)       \
  } else {                \
    WG_COMMIT(U)          \
  }

  const int npair = p.wce >> 1;
  const int cnt = (npair - ks + KS - 1) / KS;  // pairs ks, ks+KS, ... of this wavefront
  const float* ga = ldsG + (mb * 32 + l31) * p.gp + half;
  const float* xa = ldsX + (nb * 32 + l31) * xcs + half * S;
#define WG_LOAD(AV, BV, I)                                                        \
  {                                                                               \
    const int kp_ = ks + min((I), cnt - 1) * KS;                                  \
    AV = ga[2 * kp_];                                                             \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) BV[t] = xa[trow + t + 2 * kp_ * S];  \
  }
#define WG_MFMA(AV, BV) \
  _Pragma("unroll") for (int t = 0; t < TT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BV[t], acc[t], 0, 0, 0);

  if (u_begin < u_end) {
    WG_ISSUE_ANY(u_begin);
    WG_COMMIT_ANY(u_begin);
  }
  __syncthreads();
  for (int u = u_begin; u < u_end; ++u) {
    const bool more = u + 1 < u_end;
    if (more) WG_ISSUE_ANY(u + 1);
    // ---- bias partial: row sums of the staged g tile (padding columns hold zeros) ------------------------------
    if (p.bias_part != nullptr && blockIdx.y == 0 && tid < MBW * 32) {
      float s = 0.f;
      for (int c = 0; c < p.wce; ++c) s += ldsG[tid * p.gp + c];
      bsum += s;
    }
    // ---- MFMA over pixel pairs, two operand sets: the LDS reads of the next pair precede the T MFMAs of this one ----
    if (cnt > 0) {
      float a0, a1, b0[TT], b1[TT];
      WG_LOAD(a0, b0, 0);
      int i = 0;
      for (; i + 1 < cnt; i += 2) {
        WG_LOAD(a1, b1, i + 1);
        __builtin_amdgcn_sched_barrier(0);
        WG_MFMA(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        WG_LOAD(a0, b0, i + 2);
        __builtin_amdgcn_sched_barrier(0);
        WG_MFMA(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i < cnt) WG_MFMA(a0, b0);
    }
    __syncthreads();
    if (more) {
      WG_COMMIT_ANY(u + 1);
      __syncthreads();
    }
  }
#undef WG_UNIT
#undef WG_ISSUE
#undef WG_ISSUE_BF
#undef WG_COMMIT_BF
#undef WG_ISSUE_ANY
#undef WG_COMMIT_ANY
#undef WG_COMMIT
#undef WG_LOAD
#undef WG_MFMA

  // ---- write the slab: [split*KS + ks][t][m][n] ----------------------------------------------------------------
  float* sl = p.slab + (size_t)(split * KS + ks) * T * p.Mpad * p.Npad;
#pragma unroll
  for (int t = 0; t < TT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int n = n0 + nb * 32 + l31;
      sl[((size_t)(tg * TT + t) * p.Mpad + m) * p.Npad + n] = acc[t][r];
    }
  }
  if (p.bias_part != nullptr && blockIdx.y == 0 && tid < MBW * 32) p.bias_part[(size_t)split * p.Mpad + m0 + tid] = bsum;
}

// dW[co][ci][t] = clamp(sum_s slab[s][t][co][ci]);  db[co] = clamp(sum_s bias_part[s][co])
// 256 consecutive slab elements per workgroup, four per lane (16-byte loads, eight slabs in flight per lane); the four
// wavefronts each add a fixed quarter of the slabs in fp64, then combine in a fixed order -> bitwise reproducible.
// NWAVE = 16 (many slabs): sixteen wavefronts share the slabs of a workgroup's 256 elements -- a lane's chain of dependent
// 8-load batches is what the kernel waits for (24 us per launch, seven launches per step, with a quarter of the CUs busy), and it
// is a quarter as long.
// (the body: workgroup `bid` of one reduction; NWAVE of the launch's wavefronts take part -- the multi-layer launch below runs 16 and
//  gives a reduction with few slabs the 4 its own launch would have, so that its fixed summation order is the same)
template <int MAXW>
__device__ __forceinline__ void wgrad_reduce_body(double (&sh)[MAXW][256], int bid, int NWAVE, const float* __restrict__ slab, int nslab, int T, int M,
                                                  int N, int Mpad, int Npad, float clip, float* __restrict__ dw, const float* __restrict__ bias_part,
                                                  int nbias, float* __restrict__ db) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t stride = (size_t)T * Mpad * Npad;  // Npad is a multiple of 32, Mpad*Npad of 1024
  const int nw = (int)((stride + 255) / 256);
  if (bid < nw) {
    const size_t e0 = (size_t)bid * 256 + lane * 4;
    const bool in = e0 < stride && wave < NWAVE;  // stride is a multiple of 4
    const int per = (nslab + NWAVE - 1) / NWAVE;
    const int k0 = wave * per, k1 = min(nslab, k0 + per);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (in) {
      const float4* src = reinterpret_cast<const float4*>(slab + e0);
      const size_t st4 = stride / 4;
      int k = k0;
      for (; k + 8 <= k1; k += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(k + j) * st4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          s0 += (double)v[j].x; s1 += (double)v[j].y; s2 += (double)v[j].z; s3 += (double)v[j].w;
        }
      }
      for (; k < k1; ++k) {
        const float4 v = src[(size_t)k * st4];
        s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
      }
    }
    sh[wave][lane * 4 + 0] = s0; sh[wave][lane * 4 + 1] = s1; sh[wave][lane * 4 + 2] = s2; sh[wave][lane * 4 + 3] = s3;
    __syncthreads();
    const size_t e = (size_t)bid * 256 + threadIdx.x;
    if (threadIdx.x < 256 && e < stride) {
      const int n = (int)(e % Npad);
      const int m = (int)((e / Npad) % Mpad);
      const int t = (int)(e / ((size_t)Npad * Mpad));
      if (m < M && n < N) {
        const int i = threadIdx.x;
        double acc = 0.0;   // (fixed order, wavefront 0 first)
        for (int w = 0; w < NWAVE; ++w) acc += sh[w][i];
        float v = (float)acc;
        if (clip > 0.f) v = fminf(fmaxf(v, -clip), clip);
        dw[((size_t)m * N + n) * T + t] = v;
      }
    }
  } else if (db != nullptr) {
    // bias: 16 channels per workgroup, 16 lanes per channel each adding every 16th partial row (independent loads), then a
    // fixed-order combine
    double* shb = &sh[0][0];  // [16 parts][16 channels]
    const int c = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int e = (bid - nw) * 16 + c;
    double s = 0.0;
    if (threadIdx.x < 256) {
      if (e < M)
        for (int k = part; k < nbias; k += 16) s += (double)bias_part[(size_t)k * Mpad + e];
      shb[part * 16 + c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 16 && e < M) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += shb[q * 16 + c];
      float v = (float)t;
      if (clip > 0.f) v = fminf(fmaxf(v, -clip), clip);
      db[e] = v;
    }
  }
}
template <int NWAVE>
__global__ __launch_bounds__(64 * NWAVE) void wgrad_reduce_kernel(const float* __restrict__ slab, int nslab, int T, int M, int N,
                                                                  int Mpad, int Npad, float clip, float* __restrict__ dw,
                                                                  const float* __restrict__ bias_part, int nbias,
                                                                  float* __restrict__ db) {
  __shared__ double sh[NWAVE][256];
  wgrad_reduce_body<NWAVE>(sh, (int)blockIdx.x, NWAVE, slab, nslab, T, M, N, Mpad, Npad, clip, dw, bias_part, nbias, db);
}

// ---- the split-K reductions of SEVERAL weight gradients in one launch (a training step: one per convolution, ~21 us each of
// which most is launch + tail latency of a quarter-filled chip, seven launches per step)
#define WRQ_MAX 16
struct WgradReduceDesc {
  const float* slab; float* dw; const float* bias_part; float* db;
  int nslab, T, M, N, Mpad, Npad, nbias, nwave, first_block;
  float clip;
};
struct WgradReduceMulti {
  WgradReduceDesc d[WRQ_MAX];
  int n;
};
__global__ __launch_bounds__(1024) void wgrad_reduce_multi_kernel(const WgradReduceMulti q) {
  __shared__ double sh[16][256];
  int i = 0;   // (uniform) the reduction this workgroup belongs to
  while (i + 1 < q.n && (int)blockIdx.x >= q.d[i + 1].first_block) ++i;
  const WgradReduceDesc& r = q.d[i];
  wgrad_reduce_body<16>(sh, (int)blockIdx.x - r.first_block, r.nwave, r.slab, r.nslab, r.T, r.M, r.N, r.Mpad, r.Npad, r.clip, r.dw, r.bias_part,
                        r.nbias, r.db);
}
struct WgradReduceQueue {
  WgradReduceMulti m;
  int blocks;
};

// slab reduction shared with wgrad_bf16.hip (kernels cannot be launched across translation units without RDC)
extern "C" int yogo_internal_wgrad_reduce(const float* slab, int nslab, int T, int M, int N, int Mpad, int Npad, float clip, float* dw,
                                          const float* bias_part, int nbias, float* db, hipStream_t stream);
static int wgrad_reduce_flush(WgradReduceQueue* q, hipStream_t stream) {
  if (q->m.n == 0) return YOGO_OK;
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(q->blocks), dim3(1024), 0, stream, q->m);
  if (yogo_launch_log_enabled()) yogo_launch_log("wgrad_reduce_multi_kernel | reductions=%d blocks=%d", q->m.n, q->blocks);
  q->m.n = 0;
  q->blocks = 0;
  YOGO_CHECK_LAUNCH("wgrad_reduce_multi");
  return YOGO_OK;
}
// queue != NULL: the reduction is only recorded (yogo_wgrad_reduce_flush launches everything recorded in one kernel)
extern "C" int yogo_internal_wgrad_reduce_q(void* queue, const float* slab, int nslab, int T, int M, int N, int Mpad, int Npad, float clip, float* dw,
                                            const float* bias_part, int nbias, float* db, hipStream_t stream) {
  if (queue == nullptr) return yogo_internal_wgrad_reduce(slab, nslab, T, M, N, Mpad, Npad, clip, dw, bias_part, nbias, db, stream);
  WgradReduceQueue* q = reinterpret_cast<WgradReduceQueue*>(queue);
  if (q->m.n == WRQ_MAX)
    if (int e = wgrad_reduce_flush(q, stream)) return e;
  WgradReduceDesc& r = q->m.d[q->m.n++];
  r.slab = slab; r.dw = dw; r.bias_part = db ? bias_part : nullptr; r.db = db;
  r.nslab = nslab; r.T = T; r.M = M; r.N = N; r.Mpad = Mpad; r.Npad = Npad; r.nbias = nbias; r.clip = clip;
  r.nwave = nslab >= 64 ? 16 : 4;   // (as yogo_internal_wgrad_reduce picks its kernel: same summation order, same bits)
  r.first_block = q->blocks;
  q->blocks += (int)(((size_t)T * Mpad * Npad + 255) / 256) + (db ? cdiv(M, 16) : 0);
  return YOGO_OK;
}
extern "C" int yogo_wgrad_reduce_queue_create(void** queue_out) {
  YOGO_CHECK_ARG(queue_out != nullptr, "wgrad_reduce_queue_create: null pointer");
  WgradReduceQueue* q = new WgradReduceQueue();
  q->m.n = 0;
  q->blocks = 0;
  *queue_out = q;
  return YOGO_OK;
}
// forget whatever is recorded (a backward pass that failed half way leaves reductions whose buffers are gone)
extern "C" int yogo_wgrad_reduce_queue_reset(void* queue) {
  YOGO_CHECK_ARG(queue != nullptr, "wgrad_reduce_queue_reset: null queue");
  WgradReduceQueue* q = reinterpret_cast<WgradReduceQueue*>(queue);
  q->m.n = 0;
  q->blocks = 0;
  return YOGO_OK;
}
extern "C" int yogo_wgrad_reduce_flush(void* queue, hipStream_t stream) {
  YOGO_CHECK_ARG(queue != nullptr, "wgrad_reduce_flush: null queue");
  return wgrad_reduce_flush(reinterpret_cast<WgradReduceQueue*>(queue), stream);
}
extern "C" int yogo_wgrad_reduce_queue_destroy(void* queue) {
  delete reinterpret_cast<WgradReduceQueue*>(queue);
  return YOGO_OK;
}

extern "C" int yogo_internal_wgrad_reduce(const float* slab, int nslab, int T, int M, int N, int Mpad, int Npad, float clip, float* dw,
                                          const float* bias_part, int nbias, float* db, hipStream_t stream) {
  const int nw = (int)(((size_t)T * Mpad * Npad + 255) / 256);
  if (nslab >= 64)
    hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3(nw + (db ? cdiv(M, 16) : 0)), dim3(1024), 0, stream, slab, nslab, T, M, N, Mpad, Npad,
                       clip, dw, bias_part, nbias, db);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3(nw + (db ? cdiv(M, 16) : 0)), dim3(256), 0, stream, slab, nslab, T, M, N, Mpad, Npad,
                       clip, dw, bias_part, nbias, db);
  if (yogo_launch_log_enabled()) yogo_launch_log("wgrad_reduce_kernel<%d> | nslab=%d T=%d M=%d N=%d", nslab >= 64 ? 16 : 4, nslab, T, M, N);
  YOGO_CHECK_LAUNCH("wgrad_reduce");
  return YOGO_OK;
}

namespace {

struct WgradPlan {
  int MBW, NBW, KS, Mpad, Npad, nchunk_w, base_w, rem_w, wce, xw, per_ch, units, nsplit, units_per_split, gp, xp, x_off,
      lds_dummy, lds_bytes;
  dim3 grid;
};

unsigned magic_u32(int d) { return (unsigned)(((1ull << 32) + (unsigned long long)d - 1ull) / (unsigned long long)d); }

bool make_plan(int B, int N, int M, int IH, int IW, int ks, int stride, WgradPlan* pl) {
  const int pad = ks == 3 ? 1 : 0;
  const int OH = (IH + 2 * pad - ks) / stride + 1, OW = (IW + 2 * pad - ks) / stride + 1;
  const int mblocks = cdiv(M, 32), nblocks = cdiv(N, 32);
  int MBW = min(4, mblocks);
  if (MBW == 3) MBW = 2;
  int NBW = min(4 / MBW, nblocks);
  if (NBW == 3) NBW = 2;
  const int KS = 4 / (MBW * NBW);
  pl->MBW = MBW; pl->NBW = NBW; pl->KS = KS;
  pl->Mpad = round_up(M, 32 * MBW);
  pl->Npad = round_up(N, 32 * NBW);
  const int xrows = ks == 3 ? 3 : 1;
  const int WC = stride == 1 ? 64 : 32;        // must match the kernel's compile-time WC
  pl->nchunk_w = cdiv(OW, WC - 1);             // real chunk widths are <= WC - 1, so the even padded width fits WC
  pl->base_w = OW / pl->nchunk_w;
  pl->rem_w = OW - pl->base_w * pl->nchunk_w;
  const int wmax = pl->base_w + (pl->rem_w > 0 ? 1 : 0);
  pl->wce = (wmax + 1) & ~1;
  pl->xw = (pl->wce - 1) * stride + (ks == 3 ? 3 : 1);
  pl->per_ch = xrows * pl->xw;
  pl->gp = (pl->wce + 1) | 1;
  pl->xp = pl->xw | 1;
  pl->x_off = round_up(MBW * 32 * pl->gp, 4);
  pl->lds_dummy = pl->x_off + NBW * 32 * xrows * pl->xp;
  pl->lds_bytes = (pl->lds_dummy + 4) * 4;
  if (pl->lds_bytes > WGRAD_LDS_BUDGET || pl->wce > WC) return false;
  pl->units = B * OH * pl->nchunk_w;
  const int gy = pl->Npad / (32 * NBW), gz = pl->Mpad / (32 * MBW);
  int nsplit = max(1, min(pl->units, 256 / max(1, gy * gz)));  // one (12-wavefront) workgroup per CU
  pl->units_per_split = cdiv(pl->units, nsplit);
  nsplit = cdiv(pl->units, pl->units_per_split);
  pl->nsplit = nsplit;
  pl->grid = dim3(nsplit, gy, gz);
  return true;
}

template <int MBW, int NBW, int KS, int T, int S, bool BF>
void launch_one(const WgradParams& p, const WgradPlan& pl, hipStream_t stream) {
  static bool s = false;
  if (!s) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f32_kernel<MBW, NBW, KS, T, S, BF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD_LDS_BUDGET);
    s = true;
  }
  hipLaunchKernelGGL((wgrad_f32_kernel<MBW, NBW, KS, T, S, BF>), pl.grid, dim3(64 * MBW * NBW * KS * (T == 1 ? 1 : 3)), pl.lds_bytes,
                     stream, p);
}

template <int MBW, int NBW, int KS>
void launch_wgrad(const WgradParams& p, const WgradPlan& pl, int T, int stride, bool bf, hipStream_t stream) {
  if (bf) {
    if (T == 1) launch_one<MBW, NBW, KS, 1, 1, true>(p, pl, stream);
    else if (stride == 1) launch_one<MBW, NBW, KS, 9, 1, true>(p, pl, stream);
    else launch_one<MBW, NBW, KS, 9, 2, true>(p, pl, stream);
  } else {
    if (T == 1) launch_one<MBW, NBW, KS, 1, 1, false>(p, pl, stream);
    else if (stride == 1) launch_one<MBW, NBW, KS, 9, 1, false>(p, pl, stream);
    else launch_one<MBW, NBW, KS, 9, 2, false>(p, pl, stream);
  }
}

}  // namespace

extern "C" int yogo_conv2d_wgrad_workspace_bytes(int B, int Cin, int Cout, int IH, int IW, int ks, int stride, size_t* bytes) {
  YOGO_CHECK_ARG(bytes && B > 0 && Cin > 0 && Cout > 0 && IH > 0 && IW > 0 && (ks == 1 || ks == 3) && (stride == 1 || stride == 2),
                 "wgrad_workspace_bytes: bad arguments");
  WgradPlan pl;
  YOGO_CHECK_ARG(make_plan(B, Cin, Cout, IH, IW, ks, stride, &pl), "wgrad: no LDS plan");
  *bytes = ((size_t)pl.nsplit * pl.KS * ks * ks * pl.Mpad * pl.Npad + (size_t)pl.nsplit * pl.Mpad) * sizeof(float);
  return YOGO_OK;
}

// dw (OIHW) and optional db, both clamped to +-clip when clip > 0.  x: layer input, g: grad w.r.t. conv output.
static int wgrad_impl(const void* x, const void* g, bool bf, float* dw, float* db, void* workspace, int B, int Cin, int Cout, int IH,
                      int IW, int ks, int stride, float clip, hipStream_t stream);

extern "C" int yogo_conv2d_wgrad_f32(const float* x, const float* g, float* dw, float* db, void* workspace, int B, int Cin,
                                     int Cout, int IH, int IW, int ks, int stride, float clip, hipStream_t stream) {
  return wgrad_impl(x, g, false, dw, db, workspace, B, Cin, Cout, IH, IW, ks, stride, clip, stream);
}

// x and g in bf16 NCHW8c (channels padded to 16); fp32 gradients out.  Same workspace size as the fp32 entry point.
extern "C" int yogo_conv2d_wgrad_bf16in(const void* x, const void* g, float* dw, float* db, void* workspace, int B, int Cin,
                                        int Cout, int IH, int IW, int ks, int stride, float clip, hipStream_t stream) {
  return wgrad_impl(x, g, true, dw, db, workspace, B, Cin, Cout, IH, IW, ks, stride, clip, stream);
}

static int wgrad_impl(const void* x_, const void* g_, bool bf, float* dw, float* db, void* workspace, int B, int Cin, int Cout, int IH,
                      int IW, int ks, int stride, float clip, hipStream_t stream) {
  const float* x = reinterpret_cast<const float*>(x_);
  const float* g = reinterpret_cast<const float*>(g_);
  YOGO_CHECK_ARG(x && g && dw && workspace, "conv2d_wgrad: null pointer");
  YOGO_CHECK_ARG(B > 0 && Cin > 0 && Cout > 0 && IH > 0 && IW > 0 && (ks == 1 || ks == 3) && (stride == 1 || stride == 2) &&
                     !(ks == 1 && stride != 1), "conv2d_wgrad: bad shape");
  WgradPlan pl;
  YOGO_CHECK_ARG(make_plan(B, Cin, Cout, IH, IW, ks, stride, &pl), "wgrad: no LDS plan");
  const int pad = ks == 3 ? 1 : 0, T = ks * ks;
  WgradParams p{};
  p.x = x; p.g = g; p.slab = reinterpret_cast<float*>(workspace);
  p.bias_part = db ? p.slab + (size_t)pl.nsplit * pl.KS * T * pl.Mpad * pl.Npad : nullptr;
  p.B = B; p.N = Cin; p.M = Cout; p.Npad = pl.Npad; p.Mpad = pl.Mpad; p.IH = IH; p.IW = IW;
  p.OH = (IH + 2 * pad - ks) / stride + 1; p.OW = (IW + 2 * pad - ks) / stride + 1; p.pad = pad;
  p.nchunk_w = pl.nchunk_w; p.base_w = pl.base_w; p.rem_w = pl.rem_w; p.wce = pl.wce; p.xw = pl.xw; p.per_ch = pl.per_ch;
  p.inv_wce = magic_u32(pl.wce); p.inv_xw = magic_u32(pl.xw); p.inv_pc = magic_u32(pl.per_ch);
  p.units = pl.units; p.units_per_split = pl.units_per_split;
  p.gp = pl.gp; p.xp = pl.xp; p.x_off = pl.x_off; p.lds_dummy = pl.lds_dummy;
  p.Nb = ((Cin + 15) / 16) * 2; p.Mbk = ((Cout + 15) / 16) * 2;
  const int cfg = pl.MBW * 100 + pl.NBW * 10 + pl.KS;
  switch (cfg) {
    case 411: launch_wgrad<4, 1, 1>(p, pl, T, stride, bf, stream); break;
    case 221: launch_wgrad<2, 2, 1>(p, pl, T, stride, bf, stream); break;
    case 212: launch_wgrad<2, 1, 2>(p, pl, T, stride, bf, stream); break;
    case 141: launch_wgrad<1, 4, 1>(p, pl, T, stride, bf, stream); break;
    case 122: launch_wgrad<1, 2, 2>(p, pl, T, stride, bf, stream); break;
    case 114: launch_wgrad<1, 1, 4>(p, pl, T, stride, bf, stream); break;
    default:
      yogo_set_error("wgrad: unsupported wave layout %d", cfg);
      return YOGO_ERR_ARG;
  }
  const int nw = (int)(((size_t)T * pl.Mpad * pl.Npad) / 64);
  hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3(nw + (db ? cdiv(Cout, 256) : 0)), dim3(256), 0, stream, p.slab,
                     pl.nsplit * pl.KS, T, Cout, Cin, pl.Mpad, pl.Npad, clip, dw, p.bias_part, pl.nsplit, db);
  YOGO_CHECK_LAUNCH("conv2d_wgrad");
  return YOGO_OK;
}
