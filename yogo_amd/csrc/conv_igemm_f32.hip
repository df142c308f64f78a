// Implicit-GEMM 3x3 / 1x1 convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces the cuDNN conv2d forward + dgrad calls behind nn.Conv2d in the reference
// (yogo/model_defns.py:30-77, called at yogo/model.py:275).
//
// GEMM view (per image):  out[m][pix] = sum_{tap t, k} Wp[t][k][m] * in[k][pix shifted by tap t]
//   M axis  = output channel   (MFMA A operand, lane&31)
//   N axis  = output pixel     (MFMA B operand, lane&31; 32 consecutive pixels of a flat tile)
//   K axis  = (tap, input channel) -- two input channels per MFMA (lane>>5)
// The input tile (NCHW rows incl. halo, zero padded) and the weight slice of CK channels are staged in
// LDS; a lane's B address is base(pixel) + tap offset + channel stride, so one staged tile serves all taps.
// The same kernel runs dgrad: stride-1 dgrad is a conv with flipped/transposed packed weights; stride-2
// dgrad is four parity-class launches (1/2/2/4 taps) writing interleaved output lattices -- no zero
// insertion, no wasted MFMAs.
//
// fp32 MFMA is an exact k-ordered fmaf chain (MI355X guide), so results differ from the CPU reference only
// by summation order.
#include "common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MAX_TAPS 9
#define IGEMM_LDS_BUDGET (80 * 1024)

struct IgemmParams {
  const float* in;
  const float* wp;          // packed [T][Kpad][Mpad]
  const float* bias;        // [M] or null
  float* out;               // [B][M][OH][OW]
  float* out_pre;           // optional pre-activation copy (SiLU training)
  const float* act_ref;     // dgrad epilogue: multiply by act'(ref) (same indexing as out)
  const float* chan_scale;  // optional [B][M] (Dropout2d channel mask, already scaled)
  float* stats_part;        // optional BN partial sums [B*gridDim.x][Mpad][2]
  int B, K, Kpad, M, Mpad;
  int IH, IW, OH, OW;
  int OHt, OWt, oy0, ox0, osy, osx, a;
  int T;
  int toff[MAX_TAPS];       // LDS offset of tap t relative to tile origin
  int dy_min, dx_min, span_y, span_x;
  int ncb, TW, tiles_per_band;
  int CK, ck_shift, nchunk, rows_max, LWp, chs, ldsw_off, lds_dummy;
  int act;                  // forward activation fused in epilogue
  int ref_act;              // dgrad: activation whose derivative multiplies the result
  int pf;                   // 1: register-prefetch pipeline (shape fits its capacity), 0: direct staging
};

// FUSED_S2: stride-2 dgrad with the four output parity classes computed from ONE staged dy tile.  The wavefront's
// "pixel group" is 32 lattice positions (i, j); accumulator column n is the parity class (ph, pw) = (n>>1, n&1), i.e. the
// output pixel (2i+ph, 2j+pw).  Each of the 9 taps feeds exactly one class, so the MFMA count equals a 3x3 conv over the
// lattice: no zero insertion, no wasted matrix work, one launch instead of four.
template <int MW, int NW, bool FUSED_S2 = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32_kernel(const IgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(!FUSED_S2 || NW == 4, "fused stride-2 dgrad: NW counts the four parity classes");
  constexpr int BM = 32 * MW;
  constexpr int NWP = FUSED_S2 ? 1 : NW;  // 32-pixel groups per wavefront
  constexpr int PT = 4 * NWP * 32;
  float* ldsI = smem;
  float* ldsW = smem + p.ldsw_off;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int half = lane >> 5;
  // XCD-aware block remap (bijective): hardware deals consecutive block ids round-robin over the 8 XCDs; give each XCD a
  // contiguous run of (image, tile) work so that vertically adjacent tiles -- which share halo rows -- hit the same L2.
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
  const int bw = min(p.TW, p.OWt - j0);
  const int NPb = p.OHt * bw;
  const int p0 = tb * PT;
  const size_t part_row = (size_t)(b * gridDim.x + bx);

  if (p0 >= NPb) {  // tile beyond a narrow last band: contributes nothing (uniform branch)
    if (p.stats_part != nullptr && tid < BM) {
      float* dst = p.stats_part + (part_row * p.Mpad + m0 + tid) * 2;
      dst[0] = 0.f;
      dst[1] = 0.f;
    }
    return;
  }
  const int p1 = min(p0 + PT, NPb);
  const int i_lo = p0 / bw;
  const int i_hi = (p1 - 1) / bw;
  const int rows_in = (i_hi - i_lo) * p.a + p.span_y;
  const int iy0 = i_lo * p.a + p.dy_min;
  const int ix0 = j0 * p.a + p.dx_min;
  const int lw = (bw - 1) * p.a + p.span_x;

  int boff[NWP];
  int opix[NWP];
  bool pvalid[NWP];
  int lat_i = 0, lat_j = 0;  // FUSED_S2: lattice position of this lane's pixel
#pragma unroll
  for (int n = 0; n < NWP; ++n) {
    const int pp = p0 + (wave * NWP + n) * 32 + l31;
    pvalid[n] = pp < p1;
    const int pc = pvalid[n] ? pp : (p1 - 1);
    const int i = pc / bw;
    const int j = pc - i * bw;
    boff[n] = half * p.chs + ((i - i_lo) * p.a) * p.LWp + j * p.a;
    opix[n] = (p.oy0 + p.osy * i) * p.OW + p.ox0 + p.osx * (j0 + j);
    lat_i = i;
    lat_j = j0 + j;
  }

  f32x16 acc[MW][NW];
#pragma unroll
  for (int mb = 0; mb < MW; ++mb)
#pragma unroll
    for (int n = 0; n < NW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][n][r] = 0.f;

  const float* inb = p.in + (size_t)b * p.K * p.IH * p.IW;
  const int aoff = half * BM + l31;

  // tap offsets live in one VGPR (lane t holds toff[t]); v_readlane turns them into scalars without touching lgkmcnt
  const int toff_lane = p.toff[min(lane, MAX_TAPS - 1)];
  const int hs_shift = p.ck_shift - 1;          // log2(k-steps per tap)
  const int hs_mask = (p.CK >> 1) - 1;
  const int nsteps = p.T * (p.CK >> 1);

  // operands of k-step s: A = weights[tap][2 channels][BM], B = input tile shifted by the tap
#define IGEMM_LOAD(AV, BV, S)                                                                     \
  {                                                                                               \
    const int s_ = min((S), nsteps - 1);                                                          \
    const int t_ = s_ >> hs_shift;                                                                \
    const int kk_ = (s_ & hs_mask) << 1;                                                          \
    const float* wI_ = ldsI + __builtin_amdgcn_readlane(toff_lane, t_) + kk_ * p.chs;             \
    const float* wW_ = ldsW + (t_ * p.CK + kk_) * BM + aoff;                                      \
    _Pragma("unroll") for (int mb = 0; mb < MW; ++mb) AV[mb] = wW_[mb * 32];                      \
    _Pragma("unroll") for (int n = 0; n < NWP; ++n) BV[n] = wI_[boff[n]];                         \
  }
#define IGEMM_MFMA(AV, BV)                                                                        \
  _Pragma("unroll") for (int mb = 0; mb < MW; ++mb)                                               \
  _Pragma("unroll") for (int n = 0; n < NWP; ++n)                                                 \
    acc[mb][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[mb], BV[n], acc[mb][n], 0, 0, 0);

  // ---- global -> register prefetch of one K-chunk (async-STAGE split): issued before the MFMA loop of the previous
  //      chunk, committed to LDS after it, so the global-memory latency hides under ~9k cycles of matrix work.
  //      Capacity per wavefront: PFL (channel,row) lines x NXI 64-wide column groups; per lane: NQ weight float4.
  constexpr int PFL = 8, NXI = (MW == 4 ? 2 : (FUSED_S2 ? 3 : 5)), NQ = 5;  // fewer spare VGPRs -> smaller capacity
  constexpr int Q = BM / 4, RPP = 256 / Q;
  const int m4 = tid % Q, r0 = tid / Q;
  const int nrows = p.T * p.CK;
  const size_t tstride = (size_t)p.Kpad * p.Mpad;
  const int ncr = p.CK * rows_in;
  const int nlw = (ncr - wave + 3) >> 2;  // lines owned by this wavefront: wave, wave+4, ...
  const int nxi = (lw + 63) >> 6;
  float pin[PFL][NXI];
  float4 pw0, pw1, pw2, pw3, pw4;  // named (not an array): keeps them in registers
  static_assert(NQ == 5, "PF_W_* macros are written for five weight registers");
#define PF_W_LOAD(V, QI)                                                                              \
  {                                                                                                   \
    const int rw_ = min(r0 + (QI) * RPP, nrows - 1);                                                  \
    V = *reinterpret_cast<const float4*>(wbase_ + (size_t)(rw_ >> p.ck_shift) * tstride +             \
                                         (size_t)(rw_ & (p.CK - 1)) * p.Mpad);                        \
  }
#define PF_W_STORE(V, QI)                                                                             \
  {                                                                                                   \
    const int rw_ = r0 + (QI) * RPP;                                                                  \
    *reinterpret_cast<float4*>(smem + (rw_ < nrows ? p.ldsw_off + (rw_ * Q + m4) * 4 : p.lds_dummy)) = V; \
  }
#define PF_ISSUE(K0)                                                                                              \
  {                                                                                                               \
    const float* wbase_ = p.wp + (size_t)(K0) * p.Mpad + m0 + m4 * 4;                                             \
    PF_W_LOAD(pw0, 0) PF_W_LOAD(pw1, 1) PF_W_LOAD(pw2, 2) PF_W_LOAD(pw3, 3) PF_W_LOAD(pw4, 4)                     \
    _Pragma("unroll") for (int l = 0; l < PFL; ++l) {                                                             \
      const int line_ = min(wave + 4 * l, ncr - 1);                                                               \
      const int kc_ = line_ / rows_in;                                                                            \
      const int r_ = line_ - kc_ * rows_in;                                                                       \
      const int k_ = (K0) + kc_, iy_ = iy0 + r_;                                                                  \
      const bool rok_ = (l < nlw) && (k_ < p.K) && (iy_ >= 0) && (iy_ < p.IH);                                    \
      const int src_ = ((rok_ ? k_ : 0) * p.IH + (rok_ ? iy_ : 0)) * p.IW + ix0;                                  \
      _Pragma("unroll") for (int xi = 0; xi < NXI; ++xi) {                                                        \
        const int x_ = lane + 64 * xi;                                                                            \
        const int ix_ = ix0 + x_;                                                                                 \
        const bool ok_ = rok_ && (x_ < lw) && (ix_ >= 0) && (ix_ < p.IW);                                         \
        pin[l][xi] = inb[ok_ ? src_ + x_ : 0];  /* raw value; zero padding is applied at commit time */          \
      }                                                                                                           \
    }                                                                                                             \
  }
#define PF_COMMIT(K0)                                                                                             \
  {                                                                                                               \
    PF_W_STORE(pw0, 0) PF_W_STORE(pw1, 1) PF_W_STORE(pw2, 2) PF_W_STORE(pw3, 3) PF_W_STORE(pw4, 4)                \
    _Pragma("unroll") for (int l = 0; l < PFL; ++l) {                                                             \
      const int line_ = min(wave + 4 * l, ncr - 1);                                                               \
      const int kc_ = line_ / rows_in;                                                                            \
      const int r_ = line_ - kc_ * rows_in;                                                                       \
      const int k_ = (K0) + kc_, iy_ = iy0 + r_;                                                                  \
      const bool rok_ = (k_ < p.K) && (iy_ >= 0) && (iy_ < p.IH);                                                 \
      const int dst_ = kc_ * p.chs + r_ * p.LWp;                                                                  \
      _Pragma("unroll") for (int xi = 0; xi < NXI; ++xi) {                                                        \
        const int x_ = lane + 64 * xi;                                                                            \
        const int ix_ = ix0 + x_;                                                                                 \
        const bool ok_ = rok_ && (ix_ >= 0) && (ix_ < p.IW);                                                      \
        smem[(l < nlw && x_ < lw) ? dst_ + x_ : p.lds_dummy] = ok_ ? pin[l][xi] : 0.f;                            \
      }                                                                                                           \
    }                                                                                                             \
  }
#define IGEMM_COMPUTE()                                     \
  if constexpr (FUSED_S2) {                                 \
    IGEMM_COMPUTE_S2();                                     \
  } else {                                                  \
    float a0[MW], b0[NWP], a1[MW], b1[NWP];                 \
    IGEMM_LOAD(a0, b0, 0);                                  \
    int s = 0;                                              \
    for (; s + 1 < nsteps; s += 2) {                        \
      IGEMM_LOAD(a1, b1, s + 1);                            \
      __builtin_amdgcn_sched_barrier(0);                    \
      IGEMM_MFMA(a0, b0);                                   \
      __builtin_amdgcn_sched_barrier(0);                    \
      IGEMM_LOAD(a0, b0, s + 2);                            \
      __builtin_amdgcn_sched_barrier(0);                    \
      IGEMM_MFMA(a1, b1);                                   \
      __builtin_amdgcn_sched_barrier(0);                    \
    }                                                       \
    if (s < nsteps) IGEMM_MFMA(a0, b0);                     \
  }
  // fused stride-2 dgrad: per channel pair, 4 shifted views of the dy tile (dy,dx in {0,1}) serve all 9 taps; tap u of the
  // class-major packing (see s2_class_taps) feeds class S2_CLS[u] from view S2_VIEW[u] = 2*dy + dx
#define S2_LOAD(AV, BV, KK)                                                                        \
  {                                                                                                \
    const int kk_ = min((KK), p.CK - 2);                                                           \
    const float* wI_ = ldsI + kk_ * p.chs + boff[0];                                               \
    BV[0] = wI_[0];                                                                                \
    BV[1] = wI_[1];                                                                                \
    BV[2] = wI_[p.LWp];                                                                            \
    BV[3] = wI_[p.LWp + 1];                                                                        \
    _Pragma("unroll") for (int u = 0; u < 9; ++u)                                                  \
    _Pragma("unroll") for (int mb = 0; mb < MW; ++mb) AV[u][mb] = ldsW[(u * p.CK + kk_) * BM + aoff + mb * 32]; \
  }
#define S2_MFMA(AV, BV)                                                                            \
  _Pragma("unroll") for (int u = 0; u < 9; ++u)                                                    \
  _Pragma("unroll") for (int mb = 0; mb < MW; ++mb)                                                \
    acc[mb][S2_CLS[u]] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[u][mb], BV[S2_VIEW[u]], acc[mb][S2_CLS[u]], 0, 0, 0);
#define IGEMM_COMPUTE_S2()                                  \
  {                                                         \
    constexpr int S2_CLS[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};  \
    constexpr int S2_VIEW[9] = {0, 1, 0, 2, 0, 3, 2, 1, 0}; \
    for (int kk = 0; kk < p.CK; kk += 2) {                  \
      float a0[9][MW], b0[4];                               \
      S2_LOAD(a0, b0, kk);                                  \
      S2_MFMA(a0, b0);                                      \
    }                                                       \
  }

  if (p.pf) {
    PF_ISSUE(0);
    PF_COMMIT(0);
    __syncthreads();
    for (int c = 0; c < p.nchunk; ++c) {
      const bool more = c + 1 < p.nchunk;
      if (more) PF_ISSUE((c + 1) * p.CK);
      IGEMM_COMPUTE();
      __syncthreads();
      if (more) {
        PF_COMMIT((c + 1) * p.CK);
        __syncthreads();
      }
    }
  } else
  for (int c = 0; c < p.nchunk; ++c) {
    const int k0 = c * p.CK;
    __syncthreads();
    {
    // ---- stage the input tile: CK channels x rows_in rows x lw columns, zero padded.  Each wavefront takes four
    //      (channel,row) lines per pass; loads are unconditional (clamped address + select) and issued together,
    //      stores of out-of-range lines go to a dummy LDS word: no branches, four loads in flight per lane. ---------
    for (int cb0 = wave * 4; cb0 < ncr; cb0 += 16) {
      int src[4], dst[4];
      bool rok[4], wr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cr = cb0 + u;
        const int crc = min(cr, ncr - 1);
        const int kc = crc / rows_in;
        const int r = crc - kc * rows_in;
        const int k = k0 + kc;
        const int iy = iy0 + r;
        wr[u] = cr < ncr;
        rok[u] = wr[u] && (k < p.K) && (iy >= 0) && (iy < p.IH);
        src[u] = ((rok[u] ? k : 0) * p.IH + (rok[u] ? iy : 0)) * p.IW + ix0;
        dst[u] = kc * p.chs + r * p.LWp;
      }
      for (int x = lane; x < lw; x += 64) {
        const int ix = ix0 + x;
        const bool xok = ix >= 0 && ix < p.IW;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = rok[u] && xok;
          const float ld = inb[ok ? src[u] + x : 0];
          v[u] = ok ? ld : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) smem[wr[u] ? dst[u] + x : p.lds_dummy] = v[u];
      }
    }
    // ---- stage the weight slice [T][CK][BM]: rows of BM floats, four float4 loads in flight per lane --------------
    {
      const float* wbase = p.wp + (size_t)k0 * p.Mpad + m0 + m4 * 4;
#define W_SRC(R) (wbase + (size_t)((R) >> p.ck_shift) * tstride + (size_t)((R) & (p.CK - 1)) * p.Mpad)
#define W_DST(R) (smem + ((R) < nrows ? p.ldsw_off + ((R) * Q + m4) * 4 : p.lds_dummy))
      for (int rb = r0; rb < nrows; rb += RPP * 4) {
        const int ra = rb, rb1 = rb + RPP, rc = rb + 2 * RPP, rd = rb + 3 * RPP;
        const float4 va = *reinterpret_cast<const float4*>(W_SRC(min(ra, nrows - 1)));
        const float4 vb = *reinterpret_cast<const float4*>(W_SRC(min(rb1, nrows - 1)));
        const float4 vc = *reinterpret_cast<const float4*>(W_SRC(min(rc, nrows - 1)));
        const float4 vd = *reinterpret_cast<const float4*>(W_SRC(min(rd, nrows - 1)));
        __builtin_amdgcn_sched_barrier(0);
        *reinterpret_cast<float4*>(W_DST(ra)) = va;
        *reinterpret_cast<float4*>(W_DST(rb1)) = vb;
        *reinterpret_cast<float4*>(W_DST(rc)) = vc;
        *reinterpret_cast<float4*>(W_DST(rd)) = vd;
      }
#undef W_SRC
#undef W_DST
    }
    }
    __syncthreads();
    // ---- MFMA over k-steps (tap x channel pair), software pipelined with two operand sets: the LDS reads of step
    //      s+1 are issued before the MW*NW MFMAs (>= 256 cycles) of step s, so LDS latency never stalls the pipe. ----
    IGEMM_COMPUTE();
  }
#undef PF_ISSUE
#undef PF_W_LOAD
#undef PF_W_STORE
#undef PF_COMMIT
#undef IGEMM_COMPUTE
#undef IGEMM_COMPUTE_S2
#undef S2_LOAD
#undef S2_MFMA
#undef IGEMM_LOAD
#undef IGEMM_MFMA

  // ---- epilogue -----------------------------------------------------------------------------------------
  const size_t plane = (size_t)p.OH * p.OW;
  const bool do_stats = p.stats_part != nullptr;
  if (do_stats) __syncthreads();  // LDS is reused for the cross-wave reduction
  float* red = smem;              // [4 waves][BM][2]
#pragma unroll
  for (int mb = 0; mb < MW; ++mb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int m = m0 + ml;
      const bool mok = m < p.M;
      const float bsv = (p.bias != nullptr && mok) ? p.bias[m] : 0.f;
      const float cs = (p.chan_scale != nullptr && mok) ? p.chan_scale[(size_t)b * p.M + m] : 1.f;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int n = 0; n < NW; ++n) {
        float v = acc[mb][n][r] + bsv;
        bool ok;
        int op;
        if constexpr (FUSED_S2) {
          const int oy = 2 * lat_i + (n >> 1), ox = 2 * lat_j + (n & 1);
          ok = pvalid[0] && mok && (oy < p.OH) && (ox < p.OW);
          op = oy * p.OW + ox;
        } else {
          ok = pvalid[n] && mok;
          op = opix[n];
        }
        if (do_stats && ok) {
          s += v;
          q += v * v;
        }
        if (ok) {
          const size_t idx = ((size_t)b * p.M + m) * plane + op;
          if (p.out_pre != nullptr) p.out_pre[idx] = v;
          if (p.act_ref != nullptr) {
            v *= act_bwd_factor(p.act_ref[idx], p.ref_act);
          } else {
            v = act_fwd(v, p.act);
          }
          p.out[idx] = v * cs;
        }
      }
      if (do_stats) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          s += __shfl_xor(s, o, 64);
          q += __shfl_xor(q, o, 64);
        }
        if (l31 == 0) {
          red[(wave * BM + ml) * 2 + 0] = s;
          red[(wave * BM + ml) * 2 + 1] = q;
        }
      }
    }
  }
  if (do_stats) {
    __syncthreads();
    if (tid < BM) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s += red[(w * BM + tid) * 2 + 0];
        q += red[(w * BM + tid) * 2 + 1];
      }
      float* dst = p.stats_part + (part_row * p.Mpad + m0 + tid) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// weight packing: OIHW -> [T][Kpad][Mpad] (zero padded).  transpose=0: k = ci, m = co (forward);
// transpose=1: k = co, m = ci (dgrad).
// ---------------------------------------------------------------------------------------------------------
struct PackParams {
  const float* w;
  float* wp;
  int Cin, Cout, ks, T, Kpad, Mpad, transpose;
  int kh[MAX_TAPS], kw[MAX_TAPS];
};

__global__ void conv_pack_kernel(const PackParams p) {
  const int total = p.T * p.Kpad * p.Mpad;
  const int K = p.transpose ? p.Cout : p.Cin;
  const int M = p.transpose ? p.Cin : p.Cout;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int m = e % p.Mpad;
    const int k = (e / p.Mpad) % p.Kpad;
    const int t = e / (p.Mpad * p.Kpad);
    float v = 0.f;
    if (k < K && m < M) {
      const int co = p.transpose ? k : m;
      const int ci = p.transpose ? m : k;
      v = p.w[((co * p.Cin + ci) * p.ks + p.kh[t]) * p.ks + p.kw[t]];
    }
    p.wp[e] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
namespace {

struct Geom {
  int B, K, M, IH, IW, OH, OW, OHt, OWt, oy0, ox0, osy, osx, a, T;
  int dy[MAX_TAPS], dx[MAX_TAPS];
};

int pick_mw(int M) { return M <= 32 ? 1 : (M <= 64 ? 2 : 4); }
int pick_nw(int mw, int M) { (void)M; return mw == 1 ? 4 : 2; }
int kpad_of(int K) { return K >= 16 ? round_up(K, 16) : round_up(K, 2); }
int mpad_of(int M) { return round_up(M, 32 * pick_mw(M)); }

struct Tiling {
  int ncb, TW, tiles_per_band, CK, rows_max, LWp, chs, ldsw_off, lds_bytes, pf;
};

bool plan_tiling(const Geom& g, int Kpad, int MW, int NW, Tiling* out, bool fused_s2 = false) {
  const int BM = 32 * MW, PT = 128 * NW;
  int dy_min = 1 << 30, dy_max = -(1 << 30), dx_min = 1 << 30, dx_max = -(1 << 30);
  for (int t = 0; t < g.T; ++t) {
    dy_min = min(dy_min, g.dy[t]);
    dy_max = max(dy_max, g.dy[t]);
    dx_min = min(dx_min, g.dx[t]);
    dx_max = max(dx_max, g.dx[t]);
  }
  const int span_y = dy_max - dy_min + 1, span_x = dx_max - dx_min + 1;
  // prefetch capacity of the kernel (PFL lines x NXI column groups per wavefront, NQ weight float4 per lane)
  const int PFL = 8, NXI = (MW == 4 ? 2 : (fused_s2 ? 3 : 5)), NQ = 5, RPP = 256 / (BM / 4);
  Tiling best{};
  int best_score = -1;
  for (int ncb = 1; ncb <= 16 && ncb <= g.OWt; ++ncb) {
    const int TW = cdiv(g.OWt, ncb);
    const int bw_min = g.OWt - (cdiv(g.OWt, TW) - 1) * TW;
    if (cdiv(g.OWt, TW) != ncb || bw_min <= 0) continue;
    const int LW = (TW - 1) * g.a + span_x;
    const int LWp = LW;
    const int nrow_lat = min(g.OHt, 1 + cdiv(PT - 1, bw_min));
    const int rows_max = (nrow_lat - 1) * g.a + span_y;
    const int chs = rows_max * LWp;
    for (int CK : {8, 4, 2}) {
      if (Kpad % CK) continue;
      const int ldsw_off = round_up(CK * chs, 4);
      const int bytes = max((ldsw_off + g.T * CK * BM + 4) * 4, 4 * BM * 2 * 4);
      if (bytes > IGEMM_LDS_BUDGET) continue;
      const bool pf = cdiv(CK * rows_max, 4) <= PFL && cdiv(LW, 64) <= NXI && cdiv(g.T * CK, RPP) <= NQ;
      // score: prefetch-capable first, then deeper chunks, then fewer column bands (less halo)
      const int score = (pf ? 1000 : 0) + CK * 10 - ncb;
      if (score > best_score) {
        best_score = score;
        best = Tiling{ncb, TW, cdiv(g.OHt * TW, PT), CK, rows_max, LWp, chs, ldsw_off, bytes, pf ? 1 : 0};
      }
    }
  }
  if (best_score < 0) return false;
  *out = best;
  return true;
}

int launch_igemm(const Geom& g, const float* in, const float* wp, const float* bias, float* out, float* out_pre,
                 const float* act_ref, int ref_act, const float* chan_scale, float* stats_part, int act,
                 hipStream_t stream, int* stats_rows_out, bool fused_s2 = false) {
  // fused stride-2 dgrad: 64 (or 32) output channels x 32 lattice positions x 4 parity classes per wavefront
  const int MW = fused_s2 ? (g.M > 32 ? 2 : 1) : pick_mw(g.M);
  const int NW = fused_s2 ? 1 : pick_nw(MW, g.M);
  const int Kpad = kpad_of(g.K), Mpad = mpad_of(g.M);
  Tiling tl;
  if (!plan_tiling(g, Kpad, MW, NW, &tl, fused_s2)) {
    yogo_set_error("conv_igemm: no LDS tiling fits (K=%d M=%d OWt=%d a=%d)", g.K, g.M, g.OWt, g.a);
    return YOGO_ERR_ARG;
  }
  IgemmParams p{};
  p.in = in; p.wp = wp; p.bias = bias; p.out = out; p.out_pre = out_pre; p.act_ref = act_ref;
  p.chan_scale = chan_scale; p.stats_part = stats_part;
  p.B = g.B; p.K = g.K; p.Kpad = Kpad; p.M = g.M; p.Mpad = Mpad;
  p.IH = g.IH; p.IW = g.IW; p.OH = g.OH; p.OW = g.OW;
  p.OHt = g.OHt; p.OWt = g.OWt; p.oy0 = g.oy0; p.ox0 = g.ox0; p.osy = g.osy; p.osx = g.osx; p.a = g.a;
  p.T = g.T;
  int dy_min = 1 << 30, dy_max = -(1 << 30), dx_min = 1 << 30, dx_max = -(1 << 30);
  for (int t = 0; t < g.T; ++t) {
    dy_min = min(dy_min, g.dy[t]); dy_max = max(dy_max, g.dy[t]);
    dx_min = min(dx_min, g.dx[t]); dx_max = max(dx_max, g.dx[t]);
  }
  p.dy_min = dy_min; p.dx_min = dx_min; p.span_y = dy_max - dy_min + 1; p.span_x = dx_max - dx_min + 1;
  p.ncb = tl.ncb; p.TW = tl.TW; p.tiles_per_band = tl.tiles_per_band;
  p.CK = tl.CK; p.ck_shift = tl.CK == 8 ? 3 : (tl.CK == 4 ? 2 : 1); p.nchunk = Kpad / tl.CK; p.rows_max = tl.rows_max; p.LWp = tl.LWp; p.chs = tl.chs;
  p.ldsw_off = tl.ldsw_off;
  p.lds_dummy = tl.ldsw_off + g.T * tl.CK * 32 * MW;
  for (int t = 0; t < g.T; ++t) p.toff[t] = (g.dy[t] - dy_min) * tl.LWp + (g.dx[t] - dx_min);
  p.act = act; p.ref_act = ref_act; p.pf = tl.pf;

  dim3 grid(tl.ncb * tl.tiles_per_band, Mpad / (32 * MW), g.B);
  char plan_txt[224] = "";
  if (yogo_launch_log_enabled())
    snprintf(plan_txt, sizeof(plan_txt), "K=%d M=%d in=%dx%d lat=%dx%d a=%d T=%d ncb=%d TW=%d CK=%d rows=%d LW=%d lds=%d pf=%d grid=%ux%ux%u",
             g.K, g.M, g.IH, g.IW, g.OHt, g.OWt, g.a, g.T, tl.ncb, tl.TW, tl.CK, tl.rows_max, tl.LWp, tl.lds_bytes, tl.pf, grid.x, grid.y, grid.z);
  if (stats_rows_out) *stats_rows_out = g.B * (int)grid.x;
  if (g.B == 0 || g.OHt <= 0 || g.OWt <= 0) return YOGO_OK;
#define LAUNCH(MW_, NW_, FU_)                                                                               \
  do {                                                                                                      \
    static bool attr_set = false;                                                                           \
    if (!attr_set) {                                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32_kernel<MW_, NW_, FU_>),       \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);                          \
      attr_set = true;                                                                                      \
    }                                                                                                       \
    hipLaunchKernelGGL((conv_igemm_f32_kernel<MW_, NW_, FU_>), grid, dim3(256), tl.lds_bytes, stream, p);   \
    yogo_launch_log("conv_igemm_f32_kernel<" #MW_ ", " #NW_ ", " #FU_ "> | %s", plan_txt);                  \
  } while (0)
  if (fused_s2) {
    if (MW == 2) LAUNCH(2, 4, true);
    else LAUNCH(1, 4, true);
  } else if (MW == 4) LAUNCH(4, 2, false);
  else if (MW == 2) LAUNCH(2, 2, false);
  else LAUNCH(1, 4, false);
#undef LAUNCH
  YOGO_CHECK_LAUNCH("conv_igemm_f32");
  return YOGO_OK;
}

// taps of one stride-2 dgrad parity class (ph, pw): returns T and fills kh/kw/dy/dx
int s2_class_taps(int ph, int pw, int* kh, int* kw, int* dy, int* dx) {
  int T = 0;
  for (int a = 0; a < 3; ++a) {
    if (((ph + 1 - a) & 1) != 0) continue;
    for (int c = 0; c < 3; ++c) {
      if (((pw + 1 - c) & 1) != 0) continue;
      kh[T] = a; kw[T] = c; dy[T] = (ph + 1 - a) / 2; dx[T] = (pw + 1 - c) / 2;
      ++T;
    }
  }
  return T;
}

size_t packed_floats(int K, int M, int T) { return (size_t)T * kpad_of(K) * mpad_of(M); }

int check_conv_args(int ks, int stride) {
  YOGO_CHECK_ARG((ks == 3 || ks == 1) && (stride == 1 || stride == 2) && !(ks == 1 && stride != 1),
                 "conv: only 3x3 (stride 1|2, pad 1) and 1x1 (stride 1) are supported, got ks=%d stride=%d", ks, stride);
  return YOGO_OK;
}

}  // namespace

// =========================================================================================================
// C ABI
// =========================================================================================================
// mode 0: forward packing; mode 1: dgrad packing (stride 2: four parity-class blocks back to back)
extern "C" int yogo_conv_packed_bytes(int Cin, int Cout, int ks, int stride, int mode, size_t* bytes) {
  if (int e = check_conv_args(ks, stride)) return e;
  YOGO_CHECK_ARG(bytes != nullptr && Cin > 0 && Cout > 0, "conv_packed_bytes: bad arguments");
  if (mode == 0) {
    *bytes = packed_floats(Cin, Cout, ks * ks) * 4;
  } else if (stride == 1) {
    *bytes = packed_floats(Cout, Cin, ks * ks) * 4;
  } else {
    *bytes = packed_floats(Cout, Cin, 9) * 4;  // 1 + 2 + 2 + 4 taps over the four classes
  }
  return YOGO_OK;
}

extern "C" int yogo_conv_pack_f32(const float* w, float* packed, int Cin, int Cout, int ks, int stride, int mode,
                                  hipStream_t stream) {
  if (int e = check_conv_args(ks, stride)) return e;
  YOGO_CHECK_ARG(w && packed && Cin > 0 && Cout > 0, "conv_pack: null pointer / bad shape");
  auto launch = [&](PackParams& pp) {
    const int total = pp.T * pp.Kpad * pp.Mpad;
    hipLaunchKernelGGL(conv_pack_kernel, dim3(min(1024, cdiv(total, 256))), dim3(256), 0, stream, pp);
  };
  PackParams pp{};
  pp.w = w; pp.Cin = Cin; pp.Cout = Cout; pp.ks = ks;
  if (mode == 0) {
    pp.wp = packed; pp.T = ks * ks; pp.Kpad = kpad_of(Cin); pp.Mpad = mpad_of(Cout); pp.transpose = 0;
    for (int t = 0; t < pp.T; ++t) { pp.kh[t] = t / ks; pp.kw[t] = t % ks; }
    launch(pp);
  } else if (stride == 1) {
    pp.wp = packed; pp.T = ks * ks; pp.Kpad = kpad_of(Cout); pp.Mpad = mpad_of(Cin); pp.transpose = 1;
    for (int t = 0; t < pp.T; ++t) { pp.kh[t] = t / ks; pp.kw[t] = t % ks; }
    launch(pp);
  } else {
    size_t off = 0;
    for (int cls = 0; cls < 4; ++cls) {
      int dy[MAX_TAPS], dx[MAX_TAPS];
      pp.T = s2_class_taps(cls >> 1, cls & 1, pp.kh, pp.kw, dy, dx);
      pp.wp = packed + off; pp.Kpad = kpad_of(Cout); pp.Mpad = mpad_of(Cin); pp.transpose = 1;
      launch(pp);
      off += packed_floats(Cout, Cin, pp.T);
    }
  }
  YOGO_CHECK_LAUNCH("conv_pack");
  return YOGO_OK;
}

// rows of the BN partial-sum buffer a forward launch writes (each row = Mpad x {sum, sumsq})
extern "C" int yogo_conv2d_fwd_stats_shape(int B, int Cin, int Cout, int IH, int IW, int ks, int stride, int* rows,
                                           int* mpad) {
  if (int e = check_conv_args(ks, stride)) return e;
  Geom g{};
  const int pad = ks == 3 ? 1 : 0;
  g.B = B; g.K = Cin; g.M = Cout; g.IH = IH; g.IW = IW;
  g.OH = (IH + 2 * pad - ks) / stride + 1; g.OW = (IW + 2 * pad - ks) / stride + 1;
  g.OHt = g.OH; g.OWt = g.OW; g.osy = g.osx = 1; g.a = stride; g.T = ks * ks;
  for (int t = 0; t < g.T; ++t) { g.dy[t] = t / ks - pad; g.dx[t] = t % ks - pad; }
  const int MW = pick_mw(Cout), NW = pick_nw(MW, Cout);
  Tiling tl;
  if (!plan_tiling(g, kpad_of(Cin), MW, NW, &tl)) {
    yogo_set_error("conv2d_fwd_stats_shape: no tiling");
    return YOGO_ERR_ARG;
  }
  *rows = B * tl.ncb * tl.tiles_per_band;
  *mpad = mpad_of(Cout);
  return YOGO_OK;
}

// out = chan_scale * act(conv(in, w) + bias); optional pre-activation copy and BN partial sums.
extern "C" int yogo_conv2d_fwd_f32(const float* in, const float* packed, const float* bias, float* out, float* out_pre,
                                   const float* chan_scale, float* stats_part, int B, int Cin, int Cout, int IH, int IW,
                                   int ks, int stride, int act, hipStream_t stream) {
  if (int e = check_conv_args(ks, stride)) return e;
  YOGO_CHECK_ARG(in && packed && out, "conv2d_fwd: null pointer");
  YOGO_CHECK_ARG(B >= 0 && Cin > 0 && Cout > 0 && IH > 0 && IW > 0, "conv2d_fwd: bad shape");
  YOGO_CHECK_ARG(act >= 0 && act <= 2, "conv2d_fwd: bad activation %d", act);
  Geom g{};
  const int pad = ks == 3 ? 1 : 0;
  g.B = B; g.K = Cin; g.M = Cout; g.IH = IH; g.IW = IW;
  g.OH = (IH + 2 * pad - ks) / stride + 1; g.OW = (IW + 2 * pad - ks) / stride + 1;
  g.OHt = g.OH; g.OWt = g.OW; g.oy0 = g.ox0 = 0; g.osy = g.osx = 1; g.a = stride; g.T = ks * ks;
  for (int t = 0; t < g.T; ++t) { g.dy[t] = t / ks - pad; g.dx[t] = t % ks - pad; }
  return launch_igemm(g, in, packed, bias, out, out_pre, nullptr, ACT_NONE, chan_scale, stats_part, act, stream, nullptr);
}

// dx = conv_transpose(dy, w) [* act'(act_ref)] [* chan_scale]; (IH, IW) are the forward conv's INPUT dims.
extern "C" int yogo_conv2d_dgrad_f32(const float* dy, const float* packed_dgrad, float* dx, const float* act_ref,
                                     int ref_act, const float* chan_scale, int B, int Cin, int Cout, int IH, int IW,
                                     int ks, int stride, hipStream_t stream) {
  if (int e = check_conv_args(ks, stride)) return e;
  YOGO_CHECK_ARG(dy && packed_dgrad && dx, "conv2d_dgrad: null pointer");
  YOGO_CHECK_ARG(B >= 0 && Cin > 0 && Cout > 0 && IH > 0 && IW > 0, "conv2d_dgrad: bad shape");
  const int pad = ks == 3 ? 1 : 0;
  const int OH = (IH + 2 * pad - ks) / stride + 1, OW = (IW + 2 * pad - ks) / stride + 1;
  Geom g{};
  g.B = B; g.K = Cout; g.M = Cin; g.IH = OH; g.IW = OW; g.OH = IH; g.OW = IW; g.a = 1;
  if (stride == 1) {
    g.OHt = IH; g.OWt = IW; g.osy = g.osx = 1; g.T = ks * ks;
    for (int t = 0; t < g.T; ++t) { g.dy[t] = pad - t / ks; g.dx[t] = pad - t % ks; }
    return launch_igemm(g, dy, packed_dgrad, nullptr, dx, nullptr, act_ref, ref_act, chan_scale, nullptr, ACT_NONE,
                        stream, nullptr);
  }
  // stride 2: one fused launch over the lattice (i, j) = (oy>>1, ox>>1); taps in the class-major order of the packing
  g.T = 0;
  for (int cls = 0; cls < 4; ++cls) {
    int kh[MAX_TAPS], kw[MAX_TAPS], dyc[MAX_TAPS], dxc[MAX_TAPS];
    const int tc = s2_class_taps(cls >> 1, cls & 1, kh, kw, dyc, dxc);
    for (int t = 0; t < tc; ++t) {
      g.dy[g.T] = dyc[t];
      g.dx[g.T] = dxc[t];
      ++g.T;
    }
  }
  g.OHt = (IH + 1) / 2; g.OWt = (IW + 1) / 2;
  g.oy0 = g.ox0 = 0; g.osy = g.osx = 2;
  return launch_igemm(g, dy, packed_dgrad, nullptr, dx, nullptr, act_ref, ref_act, chan_scale, nullptr, ACT_NONE, stream,
                      nullptr, /*fused_s2=*/true);
}

