// Weight (+bias) gradient on the bf16 matrix cores: v_mfma_f32_32x32x16_bf16 with the contraction over OUTPUT PIXELS.
// bf16 training counterpart of wgrad_f32.hip (cuDNN backward-filter behind loss.backward(), yogo/train.py:322, under --half).
//
//   dW[t][co][ci] = sum_{b, pixel} g[b][co][pixel] * x[b][ci][pixel*stride + tap t]
//
// Both operands arrive in NCHW8c (channel-fastest 16-byte units) but the MFMA wants, per lane, 8 consecutive PIXELS of one
// channel.  The transpose is done by the LDS hardware: tiles are staged as [pixel][32 channels] bf16 (64 B per pixel, plain
// 16-byte copies of the global units) and read with ds_read_b64_tr_b16, which hands each lane 4 pixels of "its" channel per
// instruction; tap shifts move the address by whole 64-byte pixels, so every read stays aligned.
// Work decomposition as in the fp32 kernel: 12 wavefronts = MBW co-blocks x NBW ci-blocks x KS pixel-splits x 3 kernel rows,
// each owning 3 accumulator tiles; split-K slabs + the same fixed-order reduction (clamp, OIHW, bias) finish the gradient.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define WGB_LDS_BUDGET (150 * 1024)
#define WGB_ROWS(S) ((S) == 1 ? 4 : 2)  // output rows per unit (stride 2 stages a 2x larger input tile per row)

struct WgradBf16Params {
  const u32x4* x;   // [B][Nb][IH][IW] units
  const u32x4* g;   // [B][Mbk][OH][OW] units
  float* slab;      // [nsplit*KS][T][Mpad][Npad]
  float* bias_part; // optional [nsplit][Mpad]
  int B, Nb, Mbk, Npad, Mpad, IH, IW, OH, OW, pad;
  int nchunk_w, base_w, rem_w, wce;   // column chunks per row (balanced), staged chunk width (multiple of 16, zero padded)
  int nrowg;                          // row groups per image (WGB_ROWS output rows each)
  int xw, xrows;                      // staged input columns / rows per unit
  unsigned inv_wce, inv_grow, inv_xw, inv_xrow;  // magic numbers: / wce, / (WGB_ROWS*wce), / xw, / (xrows*xw)
  int units, units_per_split;
  int x_off;                          // byte offset of the x tile in LDS
  int lds_dummy;                      // byte offset of a scratch unit
};

__device__ __forceinline__ bf16x8 lds_tr8(const unsigned char* base, int off0, int off1) {
  typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(base + off0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(base + off1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int MBW, int NBW, int KS, int T, int S>
__global__ __launch_bounds__(64 * MBW * NBW * KS * (T == 1 ? 1 : 3)) void wgrad_bf16_kernel(const WgradBf16Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  constexpr int TG = (T == 1) ? 1 : 3;
  constexpr int TT = T / TG;
  constexpr int NT = 64 * MBW * NBW * KS * TG;
  constexpr int R = WGB_ROWS(S);
  constexpr int WC = 64;  // max staged chunk width
  constexpr int XR = (T == 1) ? R : (R - 1) * S + 3;
  constexpr int XWMAX = (WC - 1) * S + ((T == 1) ? 1 : 3);
  constexpr int NGU = (MBW * 4 * R * WC + NT - 1) / NT;    // g units per lane
  constexpr int NXU = (NBW * 4 * XR * XWMAX + NT - 1) / NT;  // x units per lane
  unsigned char* ldsG = smem_b;            // [MBW][R][wce][32 ch] bf16
  unsigned char* ldsX = smem_b + p.x_off;  // [NBW][xrows][xw][32 ch] bf16
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave % TG;
  const int ks = (wave / TG) % KS;
  const int nb = (wave / (TG * KS)) % NBW;
  const int mb = wave / (TG * KS * NBW);
  const int split = blockIdx.x;
  const int n0b = blockIdx.y * (NBW * 4);  // first channel BLOCK (of 8) of this workgroup's ci range
  const int m0b = blockIdx.z * (MBW * 4);

  f32x16 acc[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const int u_begin = split * p.units_per_split;
  const int u_end = min(p.units, u_begin + p.units_per_split);
  const int gplane = p.OH * p.OW, xplane = p.IH * p.IW;
  const int growsz = R * p.wce;          // g units per 8-channel block and unit
  const int xrowsz = p.xrows * p.xw;     // x units per 8-channel block and unit
  const int gtotal = MBW * 4 * growsz;
  const int xtotal = NBW * 4 * xrowsz;
  u32x4 gu[NGU], xu[NXU];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

#define WB_UNIT(U, B_, OY0_, OX0_, WCR_)                                  \
  const int cwu_ = (U) % p.nchunk_w;                                      \
  const int rgu_ = ((U) / p.nchunk_w) % p.nrowg;                          \
  const int B_ = (U) / (p.nchunk_w * p.nrowg);                            \
  const int OY0_ = rgu_ * R;                                              \
  const int OX0_ = cwu_ * p.base_w + min(cwu_, p.rem_w);                  \
  const int WCR_ = p.base_w + (cwu_ < p.rem_w ? 1 : 0);

  // element e of the g tile -> (channel block cb, row r, column c); of the x tile likewise
#define WB_GDEC(E)                                                        \
  const int cb_ = __umulhi((unsigned)(E), p.inv_grow);                    \
  const int rm_ = (E) - cb_ * growsz;                                     \
  const int r_ = __umulhi((unsigned)rm_, p.inv_wce);                      \
  const int c_ = rm_ - r_ * p.wce;
#define WB_XDEC(E)                                                        \
  const int cb_ = __umulhi((unsigned)(E), p.inv_xrow);                    \
  const int rm_ = (E) - cb_ * xrowsz;                                     \
  const int r_ = __umulhi((unsigned)rm_, p.inv_xw);                       \
  const int c_ = rm_ - r_ * p.xw;

#define WB_ISSUE(U)                                                                                         \
  {                                                                                                         \
    WB_UNIT(U, b_, oy0_, ox0_, wc_)                                                                         \
    const u32x4* gb_ = p.g + (size_t)b_ * p.Mbk * gplane;                                                   \
    _Pragma("unroll") for (int i = 0; i < NGU; ++i) {                                                       \
      const int e_ = min(tid + NT * i, gtotal - 1);                                                         \
      WB_GDEC(e_)                                                                                           \
      const int oy_ = oy0_ + r_;                                                                            \
      const bool ok_ = (m0b + cb_ < p.Mbk) && (c_ < wc_) && (oy_ < p.OH);                                   \
      gu[i] = gb_[ok_ ? (m0b + cb_) * gplane + oy_ * p.OW + ox0_ + c_ : 0];                                 \
    }                                                                                                       \
    const u32x4* xb_ = p.x + (size_t)b_ * p.Nb * xplane;                                                    \
    const int iy0_ = oy0_ * S - p.pad, ix0_ = ox0_ * S - p.pad;                                             \
    _Pragma("unroll") for (int i = 0; i < NXU; ++i) {                                                       \
      const int e_ = min(tid + NT * i, xtotal - 1);                                                         \
      WB_XDEC(e_)                                                                                           \
      const int iy_ = iy0_ + r_, ix_ = ix0_ + c_;                                                           \
      const bool ok_ = (n0b + cb_ < p.Nb) && (iy_ >= 0) && (iy_ < p.IH) && (ix_ >= 0) && (ix_ < p.IW);      \
      xu[i] = xb_[ok_ ? (n0b + cb_) * xplane + iy_ * p.IW + ix_ : 0];                                       \
    }                                                                                                       \
  }
  // commit: unit (cb, r, c) -> LDS [cb/4][r][c][32 ch], 16 bytes at channel offset (cb%4)*8; out-of-range units are zero
#define WB_COMMIT(U)                                                                                        \
  {                                                                                                         \
    WB_UNIT(U, b_, oy0_, ox0_, wc_)                                                                         \
    (void)b_;                                                                                               \
    _Pragma("unroll") for (int i = 0; i < NGU; ++i) {                                                       \
      const int e_ = tid + NT * i;                                                                          \
      const int ec_ = min(e_, gtotal - 1);                                                                  \
      WB_GDEC(ec_)                                                                                          \
      const bool ok_ = (m0b + cb_ < p.Mbk) && (c_ < wc_) && (oy0_ + r_ < p.OH);                             \
      const int off_ = (((cb_ >> 2) * R + r_) * p.wce + c_) * 64 + (cb_ & 3) * 16;                          \
      *reinterpret_cast<u32x4*>(smem_b + (e_ < gtotal ? off_ : p.lds_dummy)) = ok_ ? gu[i] : zero4;         \
    }                                                                                                       \
    const int iy0_ = oy0_ * S - p.pad, ix0_ = ox0_ * S - p.pad;                                             \
    _Pragma("unroll") for (int i = 0; i < NXU; ++i) {                                                       \
      const int e_ = tid + NT * i;                                                                          \
      const int ec_ = min(e_, xtotal - 1);                                                                  \
      WB_XDEC(ec_)                                                                                          \
      const int iy_ = iy0_ + r_, ix_ = ix0_ + c_;                                                           \
      const bool ok_ = (n0b + cb_ < p.Nb) && (iy_ >= 0) && (iy_ < p.IH) && (ix_ >= 0) && (ix_ < p.IW);      \
      const int off_ = p.x_off + (((cb_ >> 2) * p.xrows + r_) * p.xw + c_) * 64 + (cb_ & 3) * 16;           \
      *reinterpret_cast<u32x4*>(smem_b + (e_ < xtotal ? off_ : p.lds_dummy)) = ok_ ? xu[i] : zero4;         \
    }                                                                                                       \
  }

  // ---- per-lane operand addressing for the transposed reads ----------------------------------------------------------
  // 16-lane group gi: channel half (gi & 1) of the 32-channel block, pixel half (gi >> 1) of the 16-pixel k-step;
  // inside the group lane 4q + pc supplies the address of pixel row q, channels 4pc .. 4pc+3.
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, pc = lane & 3;
  const int lane_ch_off = ((gi & 1) * 16 + pc * 4) * 2;   // bytes inside the 64-byte pixel
  const int lane_px = (gi >> 1) * 8 + q4;                 // pixel inside the k-step (second read: +4)
  const int ksteps_row = p.wce >> 4;
  const int nsteps = R * ksteps_row;
  const int cnt = (nsteps - ks + KS - 1) / KS;
  const int gbase = mb * (R * p.wce * 64) + lane_ch_off;
  const int xbase = nb * (p.xrows * p.xw * 64) + lane_ch_off;

#define WB_LOAD(AV, BV, I)                                                                                      \
  {                                                                                                             \
    const int st_ = ks + min((I), cnt - 1) * KS;                                                                \
    const int r_ = st_ / ksteps_row;                                                                            \
    const int px_ = (st_ - r_ * ksteps_row) * 16 + lane_px;                                                     \
    const int ga_ = gbase + (r_ * p.wce + px_) * 64;                                                            \
    AV = lds_tr8(ldsG, ga_, ga_ + 4 * 64);                                                                      \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) {                                                            \
      const int xa_ = xbase + (((r_ * S + (T == 1 ? 0 : tg)) * p.xw) + px_ * S + t) * 64;                       \
      BV[t] = lds_tr8(ldsX, xa_, xa_ + 4 * S * 64);                                                             \
    }                                                                                                           \
  }
#define WB_MFMA(AV, BV) \
  _Pragma("unroll") for (int t = 0; t < TT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AV, BV[t], acc[t], 0, 0, 0);

  if (u_begin < u_end) {
    WB_ISSUE(u_begin);
    WB_COMMIT(u_begin);
  }
  __syncthreads();
  for (int u = u_begin; u < u_end; ++u) {
    const bool more = u + 1 < u_end;
    if (more) WB_ISSUE(u + 1);
    // bias partial: sum over the staged pixels of channel tid (padding pixels / rows hold zeros)
    if (p.bias_part != nullptr && blockIdx.y == 0 && tid < MBW * 32) {
      const __bf16* gp = reinterpret_cast<const __bf16*>(ldsG) + (tid >> 5) * (R * p.wce * 32) + (tid & 31);
      float s = 0.f;
      for (int px = 0; px < R * p.wce; ++px) s += (float)gp[px * 32];
      bsum += s;
    }
    if (cnt > 0) {
      bf16x8 a0, a1, b0[TT], b1[TT];
      WB_LOAD(a0, b0, 0);
      int i = 0;
      for (; i + 1 < cnt; i += 2) {
        WB_LOAD(a1, b1, i + 1);
        __builtin_amdgcn_sched_barrier(0);
        WB_MFMA(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        WB_LOAD(a0, b0, i + 2);
        __builtin_amdgcn_sched_barrier(0);
        WB_MFMA(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i < cnt) WB_MFMA(a0, b0);
    }
    __syncthreads();
    if (more) {
      WB_COMMIT(u + 1);
      __syncthreads();
    }
  }
#undef WB_UNIT
#undef WB_GDEC
#undef WB_XDEC
#undef WB_ISSUE
#undef WB_COMMIT
#undef WB_LOAD
#undef WB_MFMA

  // ---- write the slab: [split*KS + ks][t][m][n] ----------------------------------------------------------------------
  float* sl = p.slab + (size_t)(split * KS + ks) * T * p.Mpad * p.Npad;
  const int m0 = m0b * 8, n0 = n0b * 8;
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

// defined in wgrad_f32.hip (the fixed-order slab reduction is shared)
extern "C" int yogo_internal_wgrad_reduce(const float* slab, int nslab, int T, int M, int N, int Mpad, int Npad, float clip, float* dw,
                                          const float* bias_part, int nbias, float* db, hipStream_t stream);

namespace {

struct WbPlan {
  int MBW, NBW, KS, Mpad, Npad, nchunk_w, base_w, rem_w, wce, nrowg, xw, xrows, units, nsplit, units_per_split, x_off, lds_dummy,
      lds_bytes;
  dim3 grid;
};

unsigned wb_magic(int d) { return (unsigned)(((1ull << 32) + (unsigned long long)d - 1ull) / (unsigned long long)d); }

bool wb_plan(int B, int N, int M, int IH, int IW, int ks, int stride, WbPlan* pl) {
  const int pad = ks == 3 ? 1 : 0;
  const int OH = (IH + 2 * pad - ks) / stride + 1, OW = (IW + 2 * pad - ks) / stride + 1;
  const int mblocks = cdiv(M, 32), nblocks = cdiv(N, 32);
  int MBW = min(4, mblocks);
  if (MBW == 3) MBW = 2;
  int NBW = min(4 / MBW, nblocks);
  if (NBW == 3) NBW = 2;
  pl->MBW = MBW; pl->NBW = NBW; pl->KS = 4 / (MBW * NBW);
  pl->Mpad = round_up(M, 32 * MBW);
  pl->Npad = round_up(N, 32 * NBW);
  // column chunks: <= 64 staged pixels, multiple of 16; pick the count with the least zero padding
  int best_nc = cdiv(OW, 64), best_waste = 1 << 30;
  for (int nc = cdiv(OW, 64); nc <= cdiv(OW, 64) + 4 && nc <= OW; ++nc) {
    const int w = round_up(cdiv(OW, nc), 16);
    if (w > 64) continue;
    const int waste = nc * w - OW;
    if (waste < best_waste) { best_waste = waste; best_nc = nc; }
  }
  pl->nchunk_w = best_nc;
  pl->base_w = OW / best_nc;
  pl->rem_w = OW - pl->base_w * best_nc;
  pl->wce = round_up(pl->base_w + (pl->rem_w > 0 ? 1 : 0), 16);
  if (pl->wce > 64) return false;
  const int R = WGB_ROWS(stride);
  pl->nrowg = cdiv(OH, R);
  pl->xw = (pl->wce - 1) * stride + (ks == 3 ? 3 : 1);
  pl->xrows = ks == 3 ? (R - 1) * stride + 3 : R;
  pl->x_off = MBW * R * pl->wce * 64;
  pl->lds_dummy = pl->x_off + NBW * pl->xrows * pl->xw * 64;
  pl->lds_bytes = pl->lds_dummy + 16;
  if (pl->lds_bytes > WGB_LDS_BUDGET) return false;
  pl->units = B * pl->nrowg * pl->nchunk_w;
  const int gy = pl->Npad / (32 * NBW), gz = pl->Mpad / (32 * MBW);
  int nsplit = max(1, min(pl->units, 256 / max(1, gy * gz)));
  pl->units_per_split = cdiv(pl->units, nsplit);
  pl->nsplit = cdiv(pl->units, pl->units_per_split);
  pl->grid = dim3(pl->nsplit, gy, gz);
  return true;
}

template <int MBW, int NBW, int KS, int T, int S>
void wb_launch_one(const WgradBf16Params& p, const WbPlan& pl, hipStream_t stream) {
  static bool s = false;
  if (!s) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<MBW, NBW, KS, T, S>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, WGB_LDS_BUDGET);
    s = true;
  }
  hipLaunchKernelGGL((wgrad_bf16_kernel<MBW, NBW, KS, T, S>), pl.grid, dim3(64 * MBW * NBW * KS * (T == 1 ? 1 : 3)), pl.lds_bytes,
                     stream, p);
}

template <int MBW, int NBW, int KS>
void wb_launch(const WgradBf16Params& p, const WbPlan& pl, int T, int stride, hipStream_t stream) {
  if (T == 1) wb_launch_one<MBW, NBW, KS, 1, 1>(p, pl, stream);
  else if (stride == 1) wb_launch_one<MBW, NBW, KS, 9, 1>(p, pl, stream);
  else wb_launch_one<MBW, NBW, KS, 9, 2>(p, pl, stream);
}

}  // namespace

extern "C" int yogo_conv2d_wgrad_bf16_workspace_bytes(int B, int Cin, int Cout, int IH, int IW, int ks, int stride, size_t* bytes) {
  YOGO_CHECK_ARG(bytes && B > 0 && Cin > 0 && Cout > 0 && IH > 0 && IW > 0 && (ks == 1 || ks == 3) && (stride == 1 || stride == 2),
                 "wgrad_bf16_workspace_bytes: bad arguments");
  WbPlan pl;
  YOGO_CHECK_ARG(wb_plan(B, Cin, Cout, IH, IW, ks, stride, &pl), "wgrad_bf16: no LDS plan");
  *bytes = ((size_t)pl.nsplit * pl.KS * ks * ks * pl.Mpad * pl.Npad + (size_t)pl.nsplit * pl.Mpad) * sizeof(float);
  return YOGO_OK;
}

// fp32 dw (OIHW) / db from bf16 NCHW8c x and g on the bf16 matrix cores (fp32 accumulation); clamped to +-clip when clip > 0
extern "C" int yogo_conv2d_wgrad_bf16(const void* x, const void* g, float* dw, float* db, void* workspace, int B, int Cin, int Cout,
                                      int IH, int IW, int ks, int stride, float clip, hipStream_t stream) {
  YOGO_CHECK_ARG(x && g && dw && workspace, "conv2d_wgrad_bf16: null pointer");
  YOGO_CHECK_ARG(B > 0 && Cin > 0 && Cout > 0 && IH > 0 && IW > 0 && (ks == 1 || ks == 3) && (stride == 1 || stride == 2) &&
                     !(ks == 1 && stride != 1), "conv2d_wgrad_bf16: bad shape");
  WbPlan pl;
  YOGO_CHECK_ARG(wb_plan(B, Cin, Cout, IH, IW, ks, stride, &pl), "wgrad_bf16: no LDS plan");
  const int pad = ks == 3 ? 1 : 0, T = ks * ks;
  WgradBf16Params p{};
  p.x = reinterpret_cast<const u32x4*>(x); p.g = reinterpret_cast<const u32x4*>(g);
  p.slab = reinterpret_cast<float*>(workspace);
  p.bias_part = db ? p.slab + (size_t)pl.nsplit * pl.KS * T * pl.Mpad * pl.Npad : nullptr;
  p.B = B; p.Nb = ((Cin + 15) / 16) * 2; p.Mbk = ((Cout + 15) / 16) * 2; p.Npad = pl.Npad; p.Mpad = pl.Mpad;
  p.IH = IH; p.IW = IW; p.OH = (IH + 2 * pad - ks) / stride + 1; p.OW = (IW + 2 * pad - ks) / stride + 1; p.pad = pad;
  p.nchunk_w = pl.nchunk_w; p.base_w = pl.base_w; p.rem_w = pl.rem_w; p.wce = pl.wce; p.nrowg = pl.nrowg;
  p.xw = pl.xw; p.xrows = pl.xrows;
  p.inv_wce = wb_magic(pl.wce); p.inv_grow = wb_magic(WGB_ROWS(stride) * pl.wce); p.inv_xw = wb_magic(pl.xw);
  p.inv_xrow = wb_magic(pl.xrows * pl.xw);
  p.units = pl.units; p.units_per_split = pl.units_per_split; p.x_off = pl.x_off; p.lds_dummy = pl.lds_dummy;
  const int cfg = pl.MBW * 100 + pl.NBW * 10 + pl.KS;
  switch (cfg) {
    case 411: wb_launch<4, 1, 1>(p, pl, T, stride, stream); break;
    case 221: wb_launch<2, 2, 1>(p, pl, T, stride, stream); break;
    case 212: wb_launch<2, 1, 2>(p, pl, T, stride, stream); break;
    case 141: wb_launch<1, 4, 1>(p, pl, T, stride, stream); break;
    case 122: wb_launch<1, 2, 2>(p, pl, T, stride, stream); break;
    case 114: wb_launch<1, 1, 4>(p, pl, T, stride, stream); break;
    default:
      yogo_set_error("wgrad_bf16: unsupported wave layout %d", cfg);
      return YOGO_ERR_ARG;
  }
  YOGO_CHECK_LAUNCH("conv2d_wgrad_bf16");
  return yogo_internal_wgrad_reduce(p.slab, pl.nsplit * pl.KS, T, Cout, Cin, pl.Mpad, pl.Npad, clip, dw, p.bias_part, pl.nsplit, db,
                                    stream);
}
