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

#define WG_MAX_TAPS 9
#define WGRAD_LDS_BUDGET (80 * 1024)

struct WgradParams {
  const float* x;   // [B][N][IH][IW]  layer input
  const float* g;   // [B][M][OH][OW]  grad w.r.t. conv output
  float* slab;      // [nsplit*KS][T][Mpad][Npad]
  float* bias_part; // optional [nsplit][Mpad]
  int B, N, M, Npad, Mpad, IH, IW, OH, OW, stride, T;
  int dy[WG_MAX_TAPS], dx[WG_MAX_TAPS];  // tap offsets relative to oy*stride, ox*stride (i.e. kh - pad)
  int nchunk_w, Wc;       // column chunks per row, max chunk width (even)
  int units, units_per_split;
  int gp, xp, xrows;      // LDS pitches (odd) and staged input rows per unit (3 or 1)
  int x_off;              // float offset of the x tile in LDS
};

// MBW co-blocks x NBW ci-blocks x KS pixel-splits = 4 waves
template <int MBW, int NBW, int KS, int T>
__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ldsG = smem;            // [MBW*32][gp]
  float* ldsX = smem + p.x_off;  // [NBW*32][xrows][xp]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wave % KS;
  const int nb = (wave / KS) % NBW;
  const int mb = wave / (KS * NBW);
  const int split = blockIdx.x;
  const int n0 = blockIdx.y * (NBW * 32);
  const int m0 = blockIdx.z * (MBW * 32);
  const int xcs = p.xrows * p.xp;  // channel stride in the x tile

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;  // bias partial of row (m0 + tid), tid < MBW*32

  const int u_begin = split * p.units_per_split;
  const int u_end = min(p.units, u_begin + p.units_per_split);
  const size_t gplane = (size_t)p.OH * p.OW, xplane = (size_t)p.IH * p.IW;
  const int pad_y = -p.dy[0], pad_x = -p.dx[0];  // taps are ordered kh,kw ascending: dy[0] = -pad
  int toff[T];
#pragma unroll
  for (int t = 0; t < T; ++t) toff[t] = (p.dy[t] + pad_y) * p.xp + (p.dx[t] + pad_x);

  for (int u = u_begin; u < u_end; ++u) {
    const int cw = u % p.nchunk_w;
    const int rowid = u / p.nchunk_w;
    const int oy = rowid % p.OH;
    const int b = rowid / p.OH;
    // balanced column chunks
    const int base_w = p.OW / p.nchunk_w, rem = p.OW - base_w * p.nchunk_w;
    const int ox0 = cw * base_w + min(cw, rem);
    const int wc = base_w + (cw < rem ? 1 : 0);
    const int wce = (wc + 1) & ~1;  // even number of pixels (zero padded)
    const int ix0 = ox0 * p.stride - pad_x;
    const int xw = (wce - 1) * p.stride + (T == 1 ? 1 : 3);
    const int iy0 = oy * p.stride - pad_y;

    __syncthreads();
    // ---- stage g[MBW*32][wce]: one row segment (<= 64 floats) per wavefront instruction, 8 loads in flight ------
    {
      const float* gb = p.g + ((size_t)b * p.M) * gplane + (size_t)oy * p.OW + ox0;
      const bool cok = lane < wc;       // real pixel (else zero padding up to wce)
      const bool cwr = lane < wce;
      for (int r8 = wave * 8; r8 < MBW * 32; r8 += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int m = m0 + r8 + u;
          const bool ok = cok && (m < p.M);
          const float ld = gb[ok ? (size_t)m * gplane + lane : 0];
          v[u] = ok ? ld : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (cwr) ldsG[(r8 + u) * p.gp + lane] = v[u];
      }
    }
    // ---- stage x[NBW*32][xrows][xw]: (channel,row) lines, 4 per wavefront pass ---------------------------------
    {
      const float* xb = p.x + ((size_t)b * p.N) * xplane;
      const int nlines = NBW * 32 * p.xrows;
      for (int l4 = wave * 4; l4 < nlines; l4 += 16) {
        int src[4], dst[4];
        bool lok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ln = l4 + u;                 // nlines is a multiple of 16: always in range
          const int ch = ln / p.xrows;
          const int r = ln - ch * p.xrows;
          const int n = n0 + ch, iy = iy0 + r;
          lok[u] = (n < p.N) && (iy >= 0) && (iy < p.IH);
          src[u] = lok[u] ? (n * p.IH + iy) * p.IW + ix0 : 0;
          dst[u] = ch * xcs + r * p.xp;
        }
        for (int c = lane; c < xw; c += 64) {
          const int ix = ix0 + c;
          const bool xok = ix >= 0 && ix < p.IW;
          float v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool ok = lok[u] && xok;
            const float ld = xb[ok ? src[u] + c : 0];
            v[u] = ok ? ld : 0.f;
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; ++u) ldsX[dst[u] + c] = v[u];
        }
      }
    }
    __syncthreads();
    // ---- bias partial: row sums of the staged g tile --------------------------------------------------------
    if (p.bias_part != nullptr && blockIdx.y == 0 && tid < MBW * 32) {
      float s = 0.f;
      for (int c = 0; c < wc; ++c) s += ldsG[tid * p.gp + c];
      bsum += s;
    }
    // ---- MFMA over pixel pairs, two operand sets: the LDS reads of the next pair are issued before the T MFMAs
    //      of the current one ---------------------------------------------------------------------------------------
    {
      const int npair = wce >> 1;
      const int cnt = (npair - ks + KS - 1) / KS;  // pairs ks, ks+KS, ...
      const float* ga = ldsG + (mb * 32 + l31) * p.gp + half;
      const float* xa = ldsX + (nb * 32 + l31) * xcs + half * p.stride;
#define WG_LOAD(AV, BV, I)                                                              \
  {                                                                                     \
    const int kp_ = ks + min((I), cnt - 1) * KS;                                        \
    AV = ga[2 * kp_];                                                                   \
    _Pragma("unroll") for (int t = 0; t < T; ++t) BV[t] = xa[toff[t] + 2 * kp_ * p.stride]; \
  }
#define WG_MFMA(AV, BV) \
  _Pragma("unroll") for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BV[t], acc[t], 0, 0, 0);
      if (cnt > 0) {
        float a0, a1, b0[T], b1[T];
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
#undef WG_LOAD
#undef WG_MFMA
    }
  }

  // ---- write the slab: [split*KS + ks][t][m][n] ----------------------------------------------------------------
  float* sl = p.slab + (size_t)(split * KS + ks) * T * p.Mpad * p.Npad;
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int n = n0 + nb * 32 + l31;
      sl[((size_t)t * p.Mpad + m) * p.Npad + n] = acc[t][r];
    }
  }
  if (p.bias_part != nullptr && blockIdx.y == 0 && tid < MBW * 32) p.bias_part[(size_t)split * p.Mpad + m0 + tid] = bsum;
}

// dW[co][ci][t] = clamp(sum_s slab[s][t][co][ci]);  db[co] = clamp(sum_s bias_part[s][co])
// 64 consecutive slab elements per workgroup (coalesced 256-B reads); the four wavefronts each add a fixed
// quarter of the slabs in fp64, then combine in a fixed order -> bitwise reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int nslab, int T, int M, int N,
                                                           int Mpad, int Npad, float clip, float* __restrict__ dw,
                                                           const float* __restrict__ bias_part, int nbias,
                                                           float* __restrict__ db) {
  __shared__ double sh[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t stride = (size_t)T * Mpad * Npad;
  const int nw = (int)(stride / 64);  // Npad is a multiple of 32, Mpad*Npad of 64
  if ((int)blockIdx.x < nw) {
    const size_t e = (size_t)blockIdx.x * 64 + lane;
    const int per = (nslab + 3) / 4;
    const int k0 = wave * per, k1 = min(nslab, k0 + per);
    double s = 0.0;
    for (int k = k0; k < k1; ++k) s += (double)slab[(size_t)k * stride + e];
    sh[wave][lane] = s;
    __syncthreads();
    if (wave == 0) {
      const int n = (int)(e % Npad);
      const int m = (int)((e / Npad) % Mpad);
      const int t = (int)(e / ((size_t)Npad * Mpad));
      if (m < M && n < N) {
        float v = (float)(sh[0][lane] + sh[1][lane] + sh[2][lane] + sh[3][lane]);
        if (clip > 0.f) v = fminf(fmaxf(v, -clip), clip);
        dw[((size_t)m * N + n) * T + t] = v;
      }
    }
  } else if (db != nullptr) {
    const int e = ((int)blockIdx.x - nw) * 256 + threadIdx.x;
    if (e < M) {
      double s = 0.0;
      for (int k = 0; k < nbias; ++k) s += (double)bias_part[(size_t)k * Mpad + e];
      float v = (float)s;
      if (clip > 0.f) v = fminf(fmaxf(v, -clip), clip);
      db[e] = v;
    }
  }
}

namespace {

struct WgradPlan {
  int MBW, NBW, KS, Mpad, Npad, nchunk_w, Wc, units, nsplit, units_per_split, gp, xp, xrows, x_off, lds_bytes;
  dim3 grid;
};

bool make_plan(int B, int N, int M, int IH, int IW, int ks, int stride, WgradPlan* pl) {
  const int pad = ks == 3 ? 1 : 0;
  const int OH = (IH + 2 * pad - ks) / stride + 1, OW = (IW + 2 * pad - ks) / stride + 1;
  const int mblocks = cdiv(M, 32), nblocks = cdiv(N, 32);
  int MBW = min(4, mblocks);
  if (MBW == 3) MBW = 2;
  int NBW = min(4 / MBW, nblocks);
  if (NBW == 3) NBW = 2;
  int KS = 4 / (MBW * NBW);
  pl->MBW = MBW; pl->NBW = NBW; pl->KS = KS;
  pl->Mpad = round_up(M, 32 * MBW);
  pl->Npad = round_up(N, 32 * NBW);
  pl->xrows = ks == 3 ? 3 : 1;
  int Wc = 0;
  for (int cand : {64, 48, 32, 16, 8}) {
    const int gp = (cand + 1) | 1;
    const int xw = (cand - 1) * stride + (ks == 3 ? 3 : 1);
    const int xp = xw | 1;
    const int x_off = round_up(MBW * 32 * gp, 4);
    const int bytes = (x_off + NBW * 32 * pl->xrows * xp) * 4;
    if (bytes <= WGRAD_LDS_BUDGET) {
      Wc = cand; pl->gp = gp; pl->xp = xp; pl->x_off = x_off; pl->lds_bytes = bytes;
      break;
    }
  }
  if (Wc == 0) return false;
  pl->Wc = Wc;
  pl->nchunk_w = cdiv(OW, Wc - 1);  // chunk widths are <= Wc - 1 before even padding
  pl->units = B * OH * pl->nchunk_w;
  const int gy = pl->Npad / (32 * NBW), gz = pl->Mpad / (32 * MBW);
  int nsplit = max(1, min(pl->units, (2 * 256) / max(1, gy * gz)));
  pl->units_per_split = cdiv(pl->units, nsplit);
  nsplit = cdiv(pl->units, pl->units_per_split);
  pl->nsplit = nsplit;
  pl->grid = dim3(nsplit, gy, gz);
  return true;
}

template <int MBW, int NBW, int KS>
void launch_wgrad(const WgradParams& p, const WgradPlan& pl, int T, hipStream_t stream) {
  if (T == 9) {
    static bool s = false;
    if (!s) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f32_kernel<MBW, NBW, KS, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD_LDS_BUDGET); s = true; }
    hipLaunchKernelGGL((wgrad_f32_kernel<MBW, NBW, KS, 9>), pl.grid, dim3(256), pl.lds_bytes, stream, p);
  } else {
    static bool s = false;
    if (!s) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f32_kernel<MBW, NBW, KS, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD_LDS_BUDGET); s = true; }
    hipLaunchKernelGGL((wgrad_f32_kernel<MBW, NBW, KS, 1>), pl.grid, dim3(256), pl.lds_bytes, stream, p);
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
extern "C" int yogo_conv2d_wgrad_f32(const float* x, const float* g, float* dw, float* db, void* workspace, int B, int Cin,
                                     int Cout, int IH, int IW, int ks, int stride, float clip, hipStream_t stream) {
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
  p.OH = (IH + 2 * pad - ks) / stride + 1; p.OW = (IW + 2 * pad - ks) / stride + 1; p.stride = stride; p.T = T;
  for (int t = 0; t < T; ++t) { p.dy[t] = t / ks - pad; p.dx[t] = t % ks - pad; }
  p.nchunk_w = pl.nchunk_w; p.Wc = pl.Wc; p.units = pl.units; p.units_per_split = pl.units_per_split;
  p.gp = pl.gp; p.xp = pl.xp; p.xrows = pl.xrows; p.x_off = pl.x_off;
  const int cfg = pl.MBW * 100 + pl.NBW * 10 + pl.KS;
  switch (cfg) {
    case 411: launch_wgrad<4, 1, 1>(p, pl, T, stream); break;
    case 221: launch_wgrad<2, 2, 1>(p, pl, T, stream); break;
    case 212: launch_wgrad<2, 1, 2>(p, pl, T, stream); break;
    case 141: launch_wgrad<1, 4, 1>(p, pl, T, stream); break;
    case 122: launch_wgrad<1, 2, 2>(p, pl, T, stream); break;
    case 114: launch_wgrad<1, 1, 4>(p, pl, T, stream); break;
    default:
      yogo_set_error("wgrad: unsupported wave layout %d", cfg);
      return YOGO_ERR_ARG;
  }
  const int nw = (int)(((size_t)T * pl.Mpad * pl.Npad) / 64);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nw + (db ? cdiv(Cout, 256) : 0)), dim3(256), 0, stream, p.slab,
                     pl.nsplit * pl.KS, T, Cout, Cin, pl.Mpad, pl.Npad, clip, dw, p.bias_part, pl.nsplit, db);
  YOGO_CHECK_LAUNCH("conv2d_wgrad");
  return YOGO_OK;
}
