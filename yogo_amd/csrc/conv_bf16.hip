// bf16 inference convolution path: v_mfma_f32_32x32x16_bf16 implicit GEMM on channel-blocked activations.
//
// Replaces the bf16-autocast forward of `yogo infer` (yogo/infer.py:313-317 -> yogo/model.py:275, blocks of
// yogo/model_defns.py:30-77) for eval-mode models: BatchNorm is folded into the packed weights / bias on the host side
// (eval statistics are constants), bias + LeakyReLU/SiLU are fused into the epilogue, the head writes fp32 NCHW for the
// decode + NMS kernels.
//
// Layout "NCHW8c": [B][C/8][H][W][8] bf16 -- one 16-byte unit holds 8 consecutive channels of one pixel.  With it
//   * the MFMA B operand (8 channels of one pixel per lane) is ONE aligned ds_read_b128 for every tap (a tap shifts the
//     address by whole units), the A operand (8 input channels of one output channel) likewise;
//   * global <-> LDS staging moves 16 bytes per lane, coalesced along W;
//   * the epilogue stores 8 bytes per lane and the two half-waves complete each other's 16-byte units (512 contiguous
//     bytes per store instruction).
// At bf16 every layer of the network is HBM-bound on MI355X (ridge ~312 FLOP/B, the widest layer offers ~230), so the
// kernel is built around few, wide memory operations rather than around MFMA issue.
#include "common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// native vector types (HIP's u32x4/u32x2 are structs and get spilled to scratch when selected/stored through pointers)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define BF_MAX_TAPS 9
#define BF_LDS_BUDGET (80 * 1024)      // two workgroups per CU
#define BF_LDS_MAX (150 * 1024)        // fallback: one workgroup per CU

struct ConvBf16Params {
  const u32x4* in;    // [B][Kb][IH][IW] units
  const u32x4* wp;    // [T][Kb][Mpad] units (8 input channels of one output channel each)
  const float* bias;  // [M] fp32 or null
  u32x2* out;         // bf16 8c viewed as 8-byte halves: [B][Mb][OH][OW][2]
  float* out_f32;     // OUT_F32: fp32 NCHW [B][M][OH][OW]
  const u32x2* act_ref;     // training dgrad: multiply by act'(ref); ref = bf16 tensor shaped like `out` (8-byte halves)
  const float* chan_scale;  // optional [B][M] Dropout2d channel mask (already scaled)
  float* stats_part;        // optional BatchNorm partial sums [B*gridDim.x][Mpad][2] of the fp32 pre-activation
  int ref_act, ups;         // ups = 1: the input is read as if zero-upsampled by 2 (stride-2 dgrad; no HBM cost)
  int B, Kb, M, Mpad, Mb;
  int IH, IW, OH, OW, a, T;
  int toff[BF_MAX_TAPS];
  int dy_min, dx_min, span_y, span_x;
  int ncb, TW, tiles_per_band;
  int CKb, nchunk, rows_max, LWp, chs, ldsw_off, lds_dummy;
  int act;
};

// NWV wavefronts per workgroup, each owning NW 32-pixel groups x all MW channel blocks: NWV = 8 shares one staged weight
// slice between 512 output pixels (the weight slice is the larger part of the LDS traffic at 128 channels).
template <int MW, int NW, bool OUT_F32, int NWV>
__global__ __launch_bounds__(64 * NWV, (NWV == 8 ? 2 : 2)) void conv_bf16_kernel(const ConvBf16Params p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
  constexpr int BM = 32 * MW;
  constexpr int NT = 64 * NWV;
  constexpr int PT = NWV * NW * 32;
  u32x4* ldsI = smem4;
  u32x4* ldsW = smem4 + p.ldsw_off;

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware bijective remap: contiguous runs of (image, tile) per XCD so halo rows hit the same L2
  const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
  const unsigned widx = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
  const int bx = widx % gridDim.x;
  const int by = (widx / gridDim.x) % gridDim.y;
  const int b = widx / (gridDim.x * gridDim.y);
  const int m0 = by * BM;
  const int cb = bx / p.tiles_per_band;
  const int tb = bx - cb * p.tiles_per_band;
  const int j0 = cb * p.TW;
  const int bw = min(p.TW, p.OW - j0);
  const int NPb = p.OH * bw;
  const int p0 = tb * PT;
  if (p0 >= NPb) {
    if (p.stats_part != nullptr && tid < BM) {
      float* dst = p.stats_part + (((size_t)b * gridDim.x + bx) * p.Mpad + m0 + tid) * 2;
      dst[0] = 0.f;
      dst[1] = 0.f;
    }
    return;
  }
  const int p1 = min(p0 + PT, NPb);
  const int i_lo = p0 / bw, i_hi = (p1 - 1) / bw;
  const int rows_in = (i_hi - i_lo) * p.a + p.span_y;
  const int iy0 = i_lo * p.a + p.dy_min;
  const int ix0 = j0 * p.a + p.dx_min;
  const int lw = (bw - 1) * p.a + p.span_x;

  int boff[NW], opix[NW];
  bool pvalid[NW];
#pragma unroll
  for (int n = 0; n < NW; ++n) {
    const int pp = p0 + (wave * NW + n) * 32 + l31;
    pvalid[n] = pp < p1;
    const int pc = pvalid[n] ? pp : (p1 - 1);
    const int i = pc / bw, j = pc - i * bw;
    boff[n] = ((i - i_lo) * p.a) * p.LWp + j * p.a;
    opix[n] = i * p.OW + j0 + j;
  }

  f32x16 acc[MW][NW];
#pragma unroll
  for (int mb = 0; mb < MW; ++mb)
#pragma unroll
    for (int n = 0; n < NW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][n][r] = 0.f;

  const u32x4* inb = p.in + (size_t)b * p.Kb * p.IH * p.IW;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const int toff_lane = p.toff[min(lane, BF_MAX_TAPS - 1)];
  // ceil(2^32 / d) magic numbers for the flattened tile indexing (d = 1 is special-cased where they are used)
  const unsigned inv_lw = (unsigned)(((1ull << 32) + (unsigned)lw - 1ull) / (unsigned)lw);
  const unsigned inv_perkb = (unsigned)(((1ull << 32) + (unsigned)(rows_in * lw) - 1ull) / (unsigned)(rows_in * lw));
  const int hk = p.CKb >> 1;  // k-steps (16 channels) per tap and chunk
  const int nsteps = p.T * hk;

#define BF_LOAD(AV, BV, S)                                                                        \
  {                                                                                               \
    const int s_ = min((S), nsteps - 1);                                                          \
    const int t_ = s_ / hk;                                                                       \
    const int kb_ = 2 * (s_ - t_ * hk) + half;                                                    \
    const u32x4* wI_ = ldsI + kb_ * p.chs + __builtin_amdgcn_readlane(toff_lane, t_);             \
    const u32x4* wW_ = ldsW + (t_ * p.CKb + kb_) * BM + l31;                                      \
    _Pragma("unroll") for (int mb = 0; mb < MW; ++mb) AV[mb] = wW_[mb * 32];                      \
    _Pragma("unroll") for (int n = 0; n < NW; ++n) BV[n] = wI_[boff[n]];                          \
  }
#define BF_MFMA(AV, BV)                                                                           \
  _Pragma("unroll") for (int mb = 0; mb < MW; ++mb)                                               \
  _Pragma("unroll") for (int n = 0; n < NW; ++n)                                                  \
    acc[mb][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AV[mb]),      \
                                                         __builtin_bit_cast(bf16x8, BV[n]), acc[mb][n], 0, 0, 0);

  for (int c = 0; c < p.nchunk; ++c) {
    const int kb0 = c * p.CKb;
    __syncthreads();
    // ---- stage the input tile: CKb channel blocks x rows_in rows x lw units, zero padded.  Flattened over all lanes
    //      (magic-number division by the tile's row length), four independent 16-byte loads in flight per lane. -----------
    {
      const int per_kb = rows_in * lw;
      const int itotal = p.CKb * per_kb;
#define BI_LOAD(V, OK, E)                                                                          \
      {                                                                                            \
        const int e_ = min((E), itotal - 1);                                                       \
        const int kc_ = per_kb == 1 ? e_ : (int)__umulhi((unsigned)e_, inv_perkb);                 \
        const int rm_ = e_ - kc_ * per_kb;                                                         \
        const int r_ = lw == 1 ? rm_ : (int)__umulhi((unsigned)rm_, inv_lw);                       \
        const int x_ = rm_ - r_ * lw;                                                              \
        const int kb_ = kb0 + kc_, iy_ = iy0 + r_, ix_ = ix0 + x_;                                 \
        const int ry_ = p.ups ? (iy_ >> 1) : iy_, rx_ = p.ups ? (ix_ >> 1) : ix_;                  \
        OK = (kb_ < p.Kb) && (iy_ >= 0) && (ry_ < p.IH) && (ix_ >= 0) && (rx_ < p.IW) &&           \
             !(p.ups && ((iy_ | ix_) & 1));                                                        \
        V = inb[OK ? (kb_ * p.IH + ry_) * p.IW + rx_ : 0];                                         \
      }
#define BI_STORE(V, OK, E)                                                                         \
      {                                                                                            \
        const int e_ = min((E), itotal - 1);                                                       \
        const int kc_ = per_kb == 1 ? e_ : (int)__umulhi((unsigned)e_, inv_perkb);                 \
        const int rm_ = e_ - kc_ * per_kb;                                                         \
        const int r_ = lw == 1 ? rm_ : (int)__umulhi((unsigned)rm_, inv_lw);                       \
        const int x_ = rm_ - r_ * lw;                                                              \
        smem4[(E) < itotal ? kc_ * p.chs + r_ * p.LWp + x_ : p.lds_dummy] = OK ? V : zero4;        \
      }
      for (int e0 = tid; e0 < itotal; e0 += NT * 4) {
        u32x4 l0, l1, l2, l3;
        bool k0, k1, k2, k3;
        BI_LOAD(l0, k0, e0) BI_LOAD(l1, k1, e0 + NT) BI_LOAD(l2, k2, e0 + 2 * NT) BI_LOAD(l3, k3, e0 + 3 * NT)
        __builtin_amdgcn_sched_barrier(0);
        BI_STORE(l0, k0, e0) BI_STORE(l1, k1, e0 + NT) BI_STORE(l2, k2, e0 + 2 * NT) BI_STORE(l3, k3, e0 + 3 * NT)
      }
#undef BI_LOAD
#undef BI_STORE
    }
    // ---- stage the weight slice [T][CKb][BM] units: rows of BM units ------------------------------------------------------
    {
      constexpr int RPP = NT / BM;  // rows per pass
      const int m_ = tid % BM, r0 = tid / BM;
      const int nrows = p.T * p.CKb;
      const u32x4* wbase = p.wp + (size_t)kb0 * p.Mpad + m0 + m_;
      const size_t tstride = (size_t)p.Kb * p.Mpad;
#define BW_SRC(R) (wbase + (size_t)((R) / p.CKb) * tstride + (size_t)((R) % p.CKb) * p.Mpad)
#define BW_DST(R) (smem4 + ((R) < nrows ? p.ldsw_off + (R) * BM + m_ : p.lds_dummy))
      for (int rb = r0; rb < nrows; rb += RPP * 4) {
        const int ra = rb, rb1 = rb + RPP, rc = rb + 2 * RPP, rd = rb + 3 * RPP;
        const u32x4 va = *BW_SRC(min(ra, nrows - 1));
        const u32x4 vb = *BW_SRC(min(rb1, nrows - 1));
        const u32x4 vc = *BW_SRC(min(rc, nrows - 1));
        const u32x4 vd = *BW_SRC(min(rd, nrows - 1));
        __builtin_amdgcn_sched_barrier(0);
        *BW_DST(ra) = va;
        *BW_DST(rb1) = vb;
        *BW_DST(rc) = vc;
        *BW_DST(rd) = vd;
      }
#undef BW_SRC
#undef BW_DST
    }
    __syncthreads();
    // ---- MFMA over (tap, 16-channel step), operands of step s+1 read before the MFMAs of step s --------------------------
    {
      u32x4 a0[MW], b0[NW], a1[MW], b1[NW];
      BF_LOAD(a0, b0, 0);
      int s = 0;
      for (; s + 1 < nsteps; s += 2) {
        BF_LOAD(a1, b1, s + 1);
        __builtin_amdgcn_sched_barrier(0);
        BF_MFMA(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        BF_LOAD(a0, b0, s + 2);
        __builtin_amdgcn_sched_barrier(0);
        BF_MFMA(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (s < nsteps) BF_MFMA(a0, b0);
    }
  }
#undef BF_LOAD
#undef BF_MFMA

  // ---- epilogue: bias (+ BatchNorm partial sums of the fp32 pre-activation) + activation [or act'(ref)] + channel mask,
  //      then bf16 NCHW8c (8 bytes per lane) or fp32 NCHW ---------------------------------------------------------------
  const size_t plane = (size_t)p.OH * p.OW;
  const bool do_stats = p.stats_part != nullptr;
  if (do_stats) __syncthreads();  // LDS is reused for the cross-wave reduction
  float* red = reinterpret_cast<float*>(smem4);  // [NWV waves][BM][2]
#pragma unroll
  for (int mb = 0; mb < MW; ++mb) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cl = mb * 32 + 8 * g + 4 * half;  // local channel of this lane's 4 consecutive output channels
      const int cbase = m0 + cl;
      float bs[4], cs[4], s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bs[i] = (p.bias != nullptr && cbase + i < p.M) ? p.bias[cbase + i] : 0.f;
        cs[i] = (p.chan_scale != nullptr && cbase + i < p.M) ? p.chan_scale[(size_t)b * p.M + cbase + i] : 1.f;
      }
#pragma unroll
      for (int n = 0; n < NW; ++n) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[mb][n][4 * g + i] + bs[i];
        if (do_stats && pvalid[n]) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s4[i] += v[i];
            q4[i] += v[i] * v[i];
          }
        }
        if (pvalid[n]) {
          if constexpr (OUT_F32) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (cbase + i < p.M) p.out_f32[((size_t)b * p.M + cbase + i) * plane + opix[n]] = act_fwd(v[i], p.act) * cs[i];
          } else {
            const int cblk = cbase >> 3;
            if (cblk < p.Mb) {
              const size_t hidx = (((size_t)b * p.Mb + cblk) * plane + opix[n]) * 2 + half;
              if (p.act_ref != nullptr) {
                const bf16x4 rf = __builtin_bit_cast(bf16x4, p.act_ref[hidx]);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] *= act_bwd_factor((float)rf[i], p.ref_act);
              } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = act_fwd(v[i], p.act);
              }
              bf16x4 o;
#pragma unroll
              for (int i = 0; i < 4; ++i) o[i] = (cbase + i < p.M) ? (__bf16)(v[i] * cs[i]) : (__bf16)0.f;
              p.out[hidx] = __builtin_bit_cast(u32x2, o);
            }
          }
        }
      }
      if (do_stats) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float s = s4[i], q = q4[i];
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) {
            s += __shfl_xor(s, o, 64);
            q += __shfl_xor(q, o, 64);
          }
          if (l31 == 0) {
            red[(wave * BM + cl + i) * 2 + 0] = s;
            red[(wave * BM + cl + i) * 2 + 1] = q;
          }
        }
      }
    }
  }
  if (do_stats) {
    __syncthreads();
    if (tid < BM) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; ++w) {
        s += red[(w * BM + tid) * 2 + 0];
        q += red[(w * BM + tid) * 2 + 1];
      }
      float* dst = p.stats_part + (((size_t)b * gridDim.x + bx) * p.Mpad + m0 + tid) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

// ---- weight packing: OIHW fp32 (x optional per-output-channel scale = folded BatchNorm) -> [T][Kb][Mpad] units ----------
// dgrad = 1: GEMM roles swapped (k = co, m = ci) and the kernel flipped, so the data gradient is a plain stride-1 conv
__global__ void conv_bf16_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, u32x4* __restrict__ wp,
                                      int Cin, int Cout, int ks, int Kb, int Mpad, int dgrad) {
  const int T = ks * ks;
  const int total = T * Kb * Mpad;
  const int Kc = dgrad ? Cout : Cin, Mc = dgrad ? Cin : Cout;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int m = e % Mpad;
    const int kb = (e / Mpad) % Kb;
    const int t = e / (Mpad * Kb);
    const int tt = dgrad ? (T - 1 - t) : t;  // flipped tap
    bf16x8 o;
    const float sc = (!dgrad && m < Cout && scale != nullptr) ? scale[m] : 1.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kb * 8 + j;
      float v = 0.f;
      if (m < Mc && k < Kc) {
        const int co = dgrad ? k : m, ci = dgrad ? m : k;
        v = w[((size_t)(co * Cin + ci) * ks + tt / ks) * ks + tt % ks] * sc;
      }
      o[j] = (__bf16)v;
    }
    wp[e] = __builtin_bit_cast(u32x4, o);
  }
}

// ---- first convolution, bf16 NCHW8c output: Cin = 1|3, uint8 or fp32 input (weights fp32, BatchNorm pre-folded) -----------
struct ConvFirstBf16Params {
  const void* in;
  const float* w;     // [Cout][Cin*9] (already scaled)
  const float* bias;  // [Cout] or null
  u32x4* out;         // [B][Mb][OH][OW] units
  int B, Cin, Cout, Mb, IH, IW, OH, OW, stride, act;
};

template <typename TIn, int CIN>
__global__ __launch_bounds__(256) void conv_first_bf16_kernel(const ConvFirstBf16Params p) {
  const int b = blockIdx.y;
  const int npix = p.OH * p.OW;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= npix) return;
  const int oy = pix / p.OW, ox = pix - oy * p.OW;
  const TIn* inb = reinterpret_cast<const TIn*>(p.in) + (size_t)b * CIN * p.IH * p.IW;
  const float* __restrict__ w = p.w;
  float x[CIN * 9];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iy = oy * p.stride + kh - 1, ix = ox * p.stride + kw - 1;
        float v = 0.f;
        if (iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) v = (float)inb[((size_t)ci * p.IH + iy) * p.IW + ix];
        x[(ci * 3 + kh) * 3 + kw] = v;
      }
  for (int cb = 0; cb < p.Mb; ++cb) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int co = cb * 8 + j;
      float acc = 0.f;
      if (co < p.Cout) {  // uniform
#pragma unroll
        for (int q = 0; q < CIN * 9; ++q) acc = fmaf(w[co * CIN * 9 + q], x[q], acc);
        if (p.bias != nullptr) acc += p.bias[co];
        acc = act_fwd(acc, p.act);
      }
      o[j] = (__bf16)acc;
    }
    p.out[((size_t)b * p.Mb + cb) * npix + pix] = __builtin_bit_cast(u32x4, o);
  }
}

namespace {

int bf_pick_mw(int M) { return M <= 32 ? 1 : (M <= 64 ? 2 : 4); }
int bf_pick_nw(int mw) { return mw == 1 ? 4 : 2; }
int bf_kb_of(int K) { return round_up(K, 16) / 8; }
int bf_mpad_of(int M) { return round_up(M, 32 * bf_pick_mw(M)); }

struct BfTiling {
  int ncb, TW, tiles_per_band, CKb, rows_max, LWp, chs, ldsw_off, lds_dummy, lds_bytes;
};

bool bf_plan(int OH, int OW, int a, int T, int span, int Kb, int MW, int NW, int NWV, BfTiling* out, int budget = BF_LDS_BUDGET) {
  const int BM = 32 * MW, PT = 32 * NWV * NW;
  BfTiling best{};
  long long best_score = -1;
  for (int ncb = 1; ncb <= 16 && ncb <= OW; ++ncb) {
    const int TW = cdiv(OW, ncb);
    const int bw_min = OW - (cdiv(OW, TW) - 1) * TW;
    if (cdiv(OW, TW) != ncb || bw_min <= 0) continue;
    const int LW = (TW - 1) * a + span;
    const int nrow_lat = min(OH, 1 + cdiv(PT - 1, bw_min));
    const int rows_max = (nrow_lat - 1) * a + span;
    const int chs = rows_max * LW;
    for (int CKb : {8, 4, 2}) {
      if (Kb % CKb) continue;
      const int ldsw_off = CKb * chs;
      const int dummy = ldsw_off + T * CKb * BM;
      const int bytes = (dummy + 1) * 16;
      if (bytes > budget) continue;
      // deep chunks first, then the least staged input over the whole image (halo overhead), then fewer bands
      const long long staged = (long long)ncb * cdiv(OH * TW, PT) * rows_max * LW;  // units per channel block and image
      const long long score = (long long)CKb * 100000000000LL - staged * 100 - ncb;
      if (best_score < 0 || score > best_score) {
        best_score = score;
        best = BfTiling{ncb, TW, cdiv(OH * TW, PT), CKb, rows_max, LW, chs, ldsw_off, dummy, bytes};
      }
    }
  }
  if (best_score < 0) return budget < BF_LDS_MAX ? bf_plan(OH, OW, a, T, span, Kb, MW, NW, NWV, out, BF_LDS_MAX) : false;
  *out = best;
  return true;
}

}  // namespace

// =========================================================================================================
// C ABI
// =========================================================================================================
// mode 0: forward (k = ci, m = co); mode 1: dgrad (k = co, m = ci, flipped taps)
extern "C" int yogo_conv_bf16_packed_bytes(int Cin, int Cout, int ks, int mode, size_t* bytes) {
  YOGO_CHECK_ARG(bytes && Cin > 0 && Cout > 0 && (ks == 1 || ks == 3), "conv_bf16_packed_bytes: bad arguments");
  const int K = mode ? Cout : Cin, M = mode ? Cin : Cout;
  *bytes = (size_t)ks * ks * bf_kb_of(K) * bf_mpad_of(M) * 16;
  return YOGO_OK;
}

// scale (optional, [Cout]): per-output-channel factor folded into the weights (eval-mode BatchNorm: gamma / sqrt(var + eps))
extern "C" int yogo_conv_bf16_pack(const float* w_oihw, const float* scale, void* packed, int Cin, int Cout, int ks, int mode,
                                   hipStream_t stream) {
  YOGO_CHECK_ARG(w_oihw && packed && Cin > 0 && Cout > 0 && (ks == 1 || ks == 3), "conv_bf16_pack: bad arguments");
  const int K = mode ? Cout : Cin, M = mode ? Cin : Cout;
  const int Kb = bf_kb_of(K), Mpad = bf_mpad_of(M);
  const int total = ks * ks * Kb * Mpad;
  hipLaunchKernelGGL(conv_bf16_pack_kernel, dim3(min(1024, cdiv(total, 256))), dim3(256), 0, stream, w_oihw, scale,
                     reinterpret_cast<u32x4*>(packed), Cin, Cout, ks, Kb, Mpad, mode);
  YOGO_CHECK_LAUNCH("conv_bf16_pack");
  return YOGO_OK;
}

// channel blocks of a bf16 NCHW8c tensor with C channels AS THE NEXT LAYER READS IT (padded to 16 channels = 2 blocks)
extern "C" int yogo_bf16_channel_blocks(int C) { return bf_kb_of(C); }

namespace {

// One launcher for forward and data-gradient.  (K, M) are the GEMM contraction / output channel counts, (IH, IW) the
// physical input dims, (OH, OW) the output dims, `a` the input step per output pixel, ups = 1 reads the input as if
// zero-upsampled by 2.
int launch_conv_bf16(const void* in, const void* packed, const float* bias, void* out, float* out_f32, const void* act_ref,
                     int ref_act, const float* chan_scale, float* stats_part, int B, int K, int M, int IH, int IW, int OH,
                     int OW, int ks, int a, int ups, int act, hipStream_t stream, int* stats_rows, int* stats_mpad) {
  const int T = ks * ks, pad = ks == 3 ? 1 : 0;
  const int MW = bf_pick_mw(M), NW = bf_pick_nw(MW);
  const int NWV = (MW == 4) ? 8 : 4;  // 128-channel tiles: 8 wavefronts share the staged weight slice
  const int Kb = bf_kb_of(K), Mpad = bf_mpad_of(M);
  BfTiling tl;
  if (!bf_plan(OH, OW, a, T, ks, Kb, MW, NW, NWV, &tl, NWV == 8 ? BF_LDS_MAX : BF_LDS_BUDGET)) {
    yogo_set_error("conv_bf16: no LDS tiling fits (K=%d M=%d OW=%d a=%d)", K, M, OW, a);
    return YOGO_ERR_ARG;
  }
  dim3 grid(tl.ncb * tl.tiles_per_band, Mpad / (32 * MW), B);
  if (stats_rows) *stats_rows = B * (int)grid.x;
  if (stats_mpad) *stats_mpad = Mpad;
  if (in == nullptr) return YOGO_OK;  // shape query only
  ConvBf16Params p{};
  p.in = reinterpret_cast<const u32x4*>(in); p.wp = reinterpret_cast<const u32x4*>(packed); p.bias = bias;
  p.out = reinterpret_cast<u32x2*>(out); p.out_f32 = out_f32;
  p.act_ref = reinterpret_cast<const u32x2*>(act_ref); p.ref_act = ref_act; p.chan_scale = chan_scale; p.stats_part = stats_part;
  p.ups = ups;
  p.B = B; p.Kb = Kb; p.M = M; p.Mpad = Mpad; p.Mb = bf_kb_of(M);
  p.IH = IH; p.IW = IW; p.OH = OH; p.OW = OW; p.a = a; p.T = T;
  p.dy_min = -pad; p.dx_min = -pad; p.span_y = ks; p.span_x = ks;
  for (int t = 0; t < T; ++t) p.toff[t] = (t / ks) * tl.LWp + (t % ks);
  p.ncb = tl.ncb; p.TW = tl.TW; p.tiles_per_band = tl.tiles_per_band;
  p.CKb = tl.CKb; p.nchunk = Kb / tl.CKb; p.rows_max = tl.rows_max; p.LWp = tl.LWp; p.chs = tl.chs;
  p.ldsw_off = tl.ldsw_off; p.lds_dummy = tl.lds_dummy; p.act = act;
  if (B == 0) return YOGO_OK;
  const int lds_bytes = max(tl.lds_bytes, NWV * 32 * MW * 2 * 4);
  {
    static int verbose = -1;
    if (verbose < 0) verbose = getenv("YOGO_IGEMM_VERBOSE") ? 1 : 0;
    if (verbose)
      fprintf(stderr, "[bf16] K=%d M=%d in=%dx%d a=%d ups=%d T=%d | MW=%d NW=%d ncb=%d TW=%d CKb=%d rows=%d LW=%d lds=%d grid=%ux%ux%u\n",
              K, M, IH, IW, a, ups, T, MW, NW, tl.ncb, tl.TW, tl.CKb, tl.rows_max, tl.LWp, lds_bytes, grid.x, grid.y, grid.z);
  }
#define BFLAUNCH(MW_, NW_, F32_, NWV_)                                                                                 \
  do {                                                                                                                 \
    static bool attr_set = false;                                                                                      \
    if (!attr_set) {                                                                                                   \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<MW_, NW_, F32_, NWV_>),                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, BF_LDS_MAX);                               \
      attr_set = true;                                                                                                 \
    }                                                                                                                  \
    hipLaunchKernelGGL((conv_bf16_kernel<MW_, NW_, F32_, NWV_>), grid, dim3(64 * NWV_), lds_bytes, stream, p);         \
  } while (0)
  if (out_f32 != nullptr) {
    if (MW == 4) BFLAUNCH(4, 2, true, 8);
    else if (MW == 2) BFLAUNCH(2, 2, true, 4);
    else BFLAUNCH(1, 4, true, 4);
  } else {
    if (MW == 4) BFLAUNCH(4, 2, false, 8);
    else if (MW == 2) BFLAUNCH(2, 2, false, 4);
    else BFLAUNCH(1, 4, false, 4);
  }
#undef BFLAUNCH
  YOGO_CHECK_LAUNCH("conv_bf16");
  return YOGO_OK;
}

int check_bf16_conv(int B, int Cin, int Cout, int IH, int IW, int ks, int stride) {
  YOGO_CHECK_ARG((ks == 3 || ks == 1) && (stride == 1 || stride == 2) && !(ks == 1 && stride != 1) && B >= 0 && Cin > 0 &&
                     Cout > 0 && IH > 0 && IW > 0, "conv_bf16: unsupported shape");
  return YOGO_OK;
}

}  // namespace

// rows / row stride of the BatchNorm partial-sum buffer a forward launch fills when stats_part != NULL
extern "C" int yogo_conv2d_fwd_bf16_stats_shape(int B, int Cin, int Cout, int IH, int IW, int ks, int stride, int* rows, int* mpad) {
  if (int e = check_bf16_conv(B, Cin, Cout, IH, IW, ks, stride)) return e;
  const int pad = ks == 3 ? 1 : 0;
  return launch_conv_bf16(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, B, Cin, Cout, IH, IW,
                          (IH + 2 * pad - ks) / stride + 1, (IW + 2 * pad - ks) / stride + 1, ks, stride, 0, 0, nullptr, rows, mpad);
}

// in: bf16 NCHW8c [B][kb(Cin)][IH][IW][8]; out: bf16 NCHW8c [B][kb(Cout)][OH][OW][8], or fp32 NCHW when out_f32 != NULL.
// y = chan_scale * act(conv(x, packed) + bias); stats_part (optional) receives (sum, sumsq) of conv + bias per channel.
extern "C" int yogo_conv2d_fwd_bf16(const void* in, const void* packed, const float* bias, void* out, float* out_f32,
                                    const float* chan_scale, float* stats_part, int B, int Cin, int Cout, int IH, int IW,
                                    int ks, int stride, int act, hipStream_t stream) {
  YOGO_CHECK_ARG(in && packed && (out || out_f32), "conv2d_fwd_bf16: null pointer");
  if (int e = check_bf16_conv(B, Cin, Cout, IH, IW, ks, stride)) return e;
  const int pad = ks == 3 ? 1 : 0;
  return launch_conv_bf16(in, packed, bias, out, out_f32, nullptr, 0, chan_scale, stats_part, B, Cin, Cout, IH, IW,
                          (IH + 2 * pad - ks) / stride + 1, (IW + 2 * pad - ks) / stride + 1, ks, stride, 0, act, stream,
                          nullptr, nullptr);
}

// dx = conv_transpose(dy) * act'(act_ref) * chan_scale, all bf16 NCHW8c; (IH, IW) = the forward conv's INPUT dims.
// Stride 2 reads dy as if zero-upsampled (the zeros are produced while staging into LDS, they cost MFMA slots only).
extern "C" int yogo_conv2d_dgrad_bf16(const void* dy, const void* packed_dgrad, void* dx, const void* act_ref, int ref_act,
                                      const float* chan_scale, int B, int Cin, int Cout, int IH, int IW, int ks, int stride,
                                      hipStream_t stream) {
  YOGO_CHECK_ARG(dy && packed_dgrad && dx, "conv2d_dgrad_bf16: null pointer");
  if (int e = check_bf16_conv(B, Cin, Cout, IH, IW, ks, stride)) return e;
  const int pad = ks == 3 ? 1 : 0;
  const int OHf = (IH + 2 * pad - ks) / stride + 1, OWf = (IW + 2 * pad - ks) / stride + 1;
  return launch_conv_bf16(dy, packed_dgrad, nullptr, dx, nullptr, act_ref, ref_act, chan_scale, nullptr, B, Cout, Cin, OHf, OWf,
                          IH, IW, ks, 1, stride == 2 ? 1 : 0, ACT_NONE, stream, nullptr, nullptr);
}

// first conv (Cin 1|3; in_dtype 0 = uint8, 1 = float32), fp32 weights [Cout][Cin][3][3] with BatchNorm already folded;
// out: bf16 NCHW8c with kb(Cout) channel blocks (padding channels are written as zeros)
extern "C" int yogo_conv_first_fwd_bf16(const void* in, int in_dtype, const float* w, const float* bias, void* out, int B,
                                        int Cin, int Cout, int IH, int IW, int stride, int act, hipStream_t stream) {
  YOGO_CHECK_ARG(in && w && out, "conv_first_fwd_bf16: null pointer");
  YOGO_CHECK_ARG((Cin == 1 || Cin == 3) && Cout > 0 && (stride == 1 || stride == 2) && (in_dtype == 0 || in_dtype == 1) && B >= 0,
                 "conv_first_fwd_bf16: unsupported shape");
  ConvFirstBf16Params p{};
  p.in = in; p.w = w; p.bias = bias; p.out = reinterpret_cast<u32x4*>(out);
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.Mb = bf_kb_of(Cout); p.IH = IH; p.IW = IW; p.stride = stride; p.act = act;
  p.OH = (IH - 1) / stride + 1; p.OW = (IW - 1) / stride + 1;
  if (B == 0) return YOGO_OK;
  dim3 grid(cdiv(p.OH * p.OW, 256), B);
  if (in_dtype == 0 && Cin == 1) hipLaunchKernelGGL((conv_first_bf16_kernel<uint8_t, 1>), grid, dim3(256), 0, stream, p);
  else if (in_dtype == 0) hipLaunchKernelGGL((conv_first_bf16_kernel<uint8_t, 3>), grid, dim3(256), 0, stream, p);
  else if (Cin == 1) hipLaunchKernelGGL((conv_first_bf16_kernel<float, 1>), grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((conv_first_bf16_kernel<float, 3>), grid, dim3(256), 0, stream, p);
  YOGO_CHECK_LAUNCH("conv_first_fwd_bf16");
  return YOGO_OK;
}
