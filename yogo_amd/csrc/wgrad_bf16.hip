// Weight (+bias) gradient on the bf16 matrix cores: v_mfma_f32_32x32x16_bf16 with the contraction over OUTPUT PIXELS.
// bf16 training counterpart of wgrad_f32.hip (cuDNN backward-filter behind loss.backward(), yogo/train.py:322, under --half).
//
//   dW[t][co][ci] = sum_{b, pixel} g[b][co][pixel] * x[b][ci][pixel*stride + tap t]
//
// Both operands arrive in NCHW8c (channel-fastest 16-byte units) but the MFMA wants, per lane, 8 consecutive PIXELS of one
// channel.  The transpose is done by the LDS hardware: tiles are staged as [pixel][32 channels] bf16 (64 B per pixel, plain
// 16-byte copies of the global units) and read with ds_read_b64_tr_b16, which hands each lane 4 pixels of "its" channel per
// instruction; tap shifts move the address by whole 64-byte pixels, so every read stays aligned.
//
// Work decomposition: a "unit" is R output rows x one column chunk of one image.  A workgroup owns MBW co-blocks x
// (NBW * NPW) ci-blocks of 32 channels and walks its share of the units (split-K over pixels); its wavefronts are
// MBW x NBW x KS pixel-splits x 3 kernel rows, each holding NPW x 3 accumulator tiles.  Units stream through two LDS buffers
// by LDS-DMA (buffer_load ... lds): every lane owns a fixed set of 16-byte elements of the tile image, their source offsets
// are decoded once, per unit only the unit's origin is added and the image / chunk borders are turned into out-of-range
// offsets (which the buffer unit writes as zeros).  One barrier per unit, up to two units in flight ahead of the MFMAs.  Split-K slabs + the fixed-order reduction of
// wgrad_f32.hip (clamp, OIHW, bias) finish the gradient.
#include "common.h"
#include <mutex>
#include <type_traits>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define WGB_LDS_MAX (160 * 1024)
#ifndef WGB_BLK4
#define WGB_BLK4 1   // (A/B variant builds: 0 = 16 consecutive pixels of a row per k-step on the 128 x 64 tiling too)
#endif
#ifndef WGB_BLK4_W
#define WGB_BLK4_W 36  // BLK4: widest column chunk: 36 = 9 four-column steps, ring of two (the default); 20 = 5 steps, piece-tight buffers in a ring of FOUR -- built, parity-green and 8-13 % slower on layer 5 (gpurun_out/r6_wg_ab1.log): the kernel is not waiting for memory
#endif
#define WGB_GSLOTS 4  // LDS-DMA slots per lane and unit: gradient tile
#define WGB_XSLOTS 6  // ... input tile

struct WgradBf16Params {
  const u32x4* x;   // [B][Nb][IH][IW] units
  const u32x4* g;   // [B][Mbk][OH][OW] units
  float* slab;      // [nsplit*KS][T][Mpad][Npad]
  float* bias_part; // optional [nsplit*KS][Mpad]
  int B, Nb, Mbk, Npad, Mpad, IH, IW, OH, OW, pad;
  int nchunk_w, base_w, rem_w, wce;   // column chunks per row (balanced), staged chunk width (multiple of 16, zero padded)
  int nrowg;                          // row groups per image (R output rows each)
  int xw;                             // staged input columns per unit
  int units, units_per_split;
  int ngs, nxs, bufu;                 // g / x slots in use, 16-byte units per LDS buffer
  int xoff;                           // first unit of the x image inside an LDS buffer (ngs * threads, or the g image rounded up to a whole piece)
  int nsb;                            // BLK4: four-column steps of an ordinary chunk (8 or 4; a wide chunk takes one more)
  int xcb;                            // channel blocks per pixel of the staged x image: 4, or 2 for 16-channel inputs
  int depth;                          // LDS buffers in the ring (2 or 3): units in flight ahead of the MFMAs = depth - 1
#ifdef YOGO_DIAG
  int diag;                           // diagnostic build: experiment bits for in-process A/B runs (tools/ab_wgrad_bf16.py)
  unsigned long long* stamps;         // diagnostic build: [workgroup][4] s_memtime sums of wave 0: wait + barrier, DMA issue, step loop, whole loop
#endif
};
#ifdef YOGO_DIAG
#define WB_DIAG(BIT) (p.diag & (BIT))
#define WB_SC S
static int g_wb_diag = 0;
static unsigned long long* g_wb_stamps = nullptr;
static size_t g_wb_stamps_bytes = 0;
extern "C" int yogo_diag_wgrad_bf16(int bits) { g_wb_diag = bits; return YOGO_OK; }
extern "C" int yogo_diag_wgrad_bf16_stamps(void* buf, size_t bytes) { g_wb_stamps = reinterpret_cast<unsigned long long*>(buf); g_wb_stamps_bytes = bytes; return YOGO_OK; }
#define WB_T() (p.stamps ? __builtin_amdgcn_s_memtime() : 0ull)
#define WB_STAMP(ACC) if (p.stamps) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ACC += t_ - tk_; tk_ = t_; __builtin_amdgcn_sched_barrier(0); }
#else
// compile-time ablations of the product build (build.sh variant TAG wgrad_bf16 -DWGB_ABL=bits + tools/ab_variants.py; timings only, the
// results are wrong): 1 = no MFMAs, 2 = no operand reads, 4 = no LDS-DMA, 8 = every DMA piece out of range (zero fill, no memory traffic),
// 128 = the x operand's column addressing as if the stride were 1 (no bank conflicts of the stride-2 reads: -2 % / 0 % on layers 4 / 2),
// 16 = the second ci-half workgroup zero-fills its gradient tile (what sharing it would save), 32 = the halo rows of the input tile are
// zero-filled (what a row ring would save), 64 = BLK4: the next unit's DMA in one block behind the barrier instead of one slot per step
#ifndef WGB_ABL
#define WGB_ABL 0
#endif
#define WB_DIAG(BIT) ((WGB_ABL & (BIT)) != 0)
#define WB_SC ((WGB_ABL & 128) ? 1 : S)   // (bit 128: the x operand's COLUMN addressing as if the stride were 1: conflict-free reads, wrong results)
#define WB_T() 0ull
#define WB_STAMP(ACC)
#endif

typedef int i32x4 __attribute__((ext_vector_type(4)));

// One LDS-DMA piece: 64 lanes x 16 bytes from (descriptor, per-lane byte offset) to LDS bytes [lds_addr, lds_addr + 1024).
// Issued from inline asm on purpose: hipcc (ROCm 7.2) makes every later LDS read wait for ALL of its own LDS-DMA loads
// (vmcnt(0)), which would serialise the ring; these loads are invisible to it and are retired by wait_dma() below.
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds_addr, int voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(lds_addr), "s"(rsrc)
               : "memory");
}
// descriptor of `bytes` bytes at `ptr` (raw buffer, no swizzle)
__device__ __forceinline__ i32x4 make_rsrc(const void* ptr, int bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), bytes, 0x00020000};
}

// counted wait for this wavefront's LDS-DMA: at most n of its newest loads may still be in flight
__device__ __forceinline__ void wait_dma(int n) {
  switch (n) {
#define WD_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    WD_CASE(1) WD_CASE(2) WD_CASE(3) WD_CASE(4) WD_CASE(5) WD_CASE(6) WD_CASE(7) WD_CASE(8) WD_CASE(9) WD_CASE(10) WD_CASE(11) WD_CASE(12) WD_CASE(13) WD_CASE(14) WD_CASE(15) WD_CASE(16)
#undef WD_CASE
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

__device__ __forceinline__ bf16x8 lds_tr8(const unsigned char* base, int off0, int off1) {
  typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(base + off0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(base + off1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a compile-time unrolled loop whose index can feed constexpr
template <class F, int... I>
__device__ __forceinline__ void wb_static_for(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

// MPW: co-blocks per wavefront (the wavefront grid is MBW x NBW x KS x kernel rows; a workgroup covers MBW * MPW co-blocks and
// NBW * NPW ci-blocks).  MPW = 2, NPW = 1 reads 4 + 6 transposed operand halves per 6 MFMAs where MPW = 1, NPW = 2 reads 2 + 12.
// PACK2 (16 input channels): the 32 output columns of an MFMA hold TWO taps -- lanes of columns 16-31 read the input one pixel
// further instead of re-reading channels that do not exist -- so a kernel row costs 2 MFMAs and 4 B reads instead of 3 and 6.
// BLK4 (the 128 x 64 tiling, stride 1): a k-step is a BLOCK of 4 rows x 4 columns of the unit (R = 4 rows, staged width 36 = 9 steps) instead of
// 16 consecutive pixels of a row.  A row of 129 (= 8 x 16 + 1) or 258 output columns cannot be cut into 16-column steps without 10 - 12 %
// zero padding whatever the chunking; cut into 4-column blocks the chunks are 32 or 33 columns wide (8 steps, a ninth for the wide ones) and
// the padding is 2.3 % in the columns + 1.6 - 3 % in the rows: 7.4 - 7.9 % fewer MFMAs on layers 3 / 5 / 6.
template <int MBW, int NBW, int NPW, int KS, int T, int S, int R, int MPW = 1, bool ROT = false, bool PACK2 = false, bool BLK4 = false>
__global__ __launch_bounds__(64 * MBW * NBW * KS * (T == 1 ? 1 : 3)) void wgrad_bf16_kernel(const WgradBf16Params p) {
  static_assert(!BLK4 || (KS == 1 && !PACK2 && !ROT && S == 1 && T == 9 && R == 4 && NPW == 1), "BLK4: the lean 3x3 stride-1 loop only");
  // LEAN: the instruction-lean step loop below (one pixel split per wavefront grid, 64-byte pixels in the staged x image; the
  // launcher only instantiates KS = 1 tilings with xcb = 4)
  constexpr bool LEAN = KS == 1 && !PACK2 && !ROT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  constexpr int TG = (T == 1) ? 1 : 3;
  constexpr int TT = T / TG;
  constexpr int TM = PACK2 ? (TT + 1) / 2 : TT;  // MFMAs (B operands, accumulator tiles) per k-step and (co, ci) block pair
  constexpr int NT = 64 * MBW * NBW * KS * TG;
  constexpr int XR = (T == 1) ? R : (R - 1) * S + 3;
  constexpr unsigned OOB = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave % TG;
  const int ks = (wave / TG) % KS;
  const int nb = (wave / (TG * KS)) % NBW;
  const int mb = wave / (TG * KS * NBW);
  const int split = blockIdx.x;
  const int n0b = blockIdx.y * (NBW * NPW * 4);  // first channel BLOCK (of 8) of this workgroup's ci range
  const int m0b = blockIdx.z * (MBW * MPW * 4);

  f32x16 acc[MPW][NPW][TM];
#pragma unroll
  for (int m = 0; m < MPW; ++m)
#pragma unroll
    for (int q = 0; q < NPW; ++q)
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][q][t][r] = 0.f;
  float bsum[MPW];
#pragma unroll
  for (int m = 0; m < MPW; ++m) bsum[m] = 0.f;
  // bias gradient = row sums of the gradient operand.  LEAN: every wavefront of the first ci-block workgroup takes part (the
  // NBW * TG wavefronts that hold the same operand sum alternate steps, each writes its own partial row); otherwise one does.
  const bool do_bias = p.bias_part != nullptr && blockIdx.y == 0 && (LEAN || (tg == 0 && nb == 0));
  [[maybe_unused]] int bias_turn = nb * TG + tg;   // LEAN: steps until this wavefront's next turn

  const int u_begin = split * p.units_per_split;
  const int u_end = min(p.units, u_begin + p.units_per_split);
  const int gplane = p.OH * p.OW, xplane = p.IH * p.IW;

  // ---- this lane's elements of the tile image: slot i covers LDS unit i*NT + tid -------------------------------------------
  // g image [MBW * MPW][R][wce][4 channel blocks], x image [NBW*NPW][XR][xw][4 channel blocks]; element -> (cb, r, c).
  // lc = byte offset inside the image for unit origin (0, 0), rc = r << 16 | c (all ones: never valid)
  int glc[WGB_GSLOTS], xlc[WGB_XSLOTS];
  unsigned grc[WGB_GSLOTS], xrc[WGB_XSLOTS];
  const int gtot = MBW * MPW * 4 * R * p.wce, xtot = NBW * NPW * p.xcb * XR * p.xw;
  // a piece (this wavefront's 64 units of a slot) that lies wholly behind the image is not issued: the images can then be packed
  // piece-tight (xoff), and the counted waits below go by what THIS wavefront issues per unit
  int my_pieces = 0;
#pragma unroll
  for (int i = 0; i < WGB_GSLOTS; ++i) my_pieces += (i < p.ngs && i * NT + wave * 64 < gtot) ? 1 : 0;
#pragma unroll
  for (int i = 0; i < WGB_XSLOTS; ++i) my_pieces += (i < p.nxs && i * NT + wave * 64 < xtot) ? 1 : 0;
  {
#pragma unroll
    for (int i = 0; i < WGB_GSLOTS; ++i) {
      const int e = tid + i * NT;
      const int cb3 = e & 3, pix = e >> 2;
      const int rr = pix / p.wce, c = pix - rr * p.wce;
      const int cbg = rr / R, r = rr - cbg * R;
      const int cb = m0b + cbg * 4 + cb3;
      const bool ok = e < gtot && cb < p.Mbk;
      glc[i] = (cb * gplane + r * p.OW + c) * 16;
      grc[i] = ok ? (unsigned)(r << 16 | c) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < WGB_XSLOTS; ++i) {
      const int e = tid + i * NT;
      const int pix = p.xcb == 4 ? e >> 2 : e >> 1, cb3 = e - pix * p.xcb;
      const int rr = pix / p.xw, c = pix - rr * p.xw;
      const int cbg = rr / XR, r = rr - cbg * XR;
      const int cb = n0b + cbg * 4 + cb3;
      const bool ok = e < xtot && cb < p.Nb;
      xlc[i] = (cb * xplane + r * p.IW + c) * 16;
      xrc[i] = ok ? (unsigned)(r << 16 | c) : 0xFFFFFFFFu;
    }
  }

  // the unit whose DMA is issued next: (image, row group, column chunk), stepped without divisions
  int iu_cw = u_begin % p.nchunk_w, iu_rg = (u_begin / p.nchunk_w) % p.nrowg, iu_b = u_begin / (p.nchunk_w * p.nrowg);
// geometry of the unit whose DMA goes out next (scalars), then one slot at a time: the lean loop spreads the slots over its
// steps -- issued in one block behind the barrier, the DMA of 12 wavefronts (address checks + 96 kilobyte-pieces through the
// CU's one address unit) kept the matrix pipe idle for a quarter of every unit (stamps: 1.5 k of 9 k ticks, plus the skew
// it leaves between the wavefronts at the next barrier)
#define WB_ISSUE_PREP(BUF)                                                                                            \
    const int b_ = iu_b, oy0_ = iu_rg * R, ox0_ = iu_cw * p.base_w + min(iu_cw, p.rem_w);                             \
    const int wc_ = p.base_w + (iu_cw < p.rem_w ? 1 : 0);                                                             \
    if (++iu_cw == p.nchunk_w) {                                                                                      \
      iu_cw = 0;                                                                                                      \
      if (++iu_rg == p.nrowg) { iu_rg = 0; ++iu_b; }                                                                  \
    }                                                                                                                 \
    const i32x4 rs_g = make_rsrc(p.g + (size_t)b_ * p.Mbk * gplane, p.Mbk * gplane * 16);                             \
    const i32x4 rs_x = make_rsrc(p.x + (size_t)b_ * p.Nb * xplane, p.Nb * xplane * 16);                               \
    const unsigned lb_ = (unsigned)(((BUF) * p.bufu + wave * 64) * 16);  /* the dynamic LDS block starts at LDS address 0 */ \
    const int gorg_ = (oy0_ * p.OW + ox0_) * 16, rmax_ = p.OH - oy0_;                                                 \
    const int iy0_ = oy0_ * S - p.pad, ix0_ = ox0_ * S - p.pad;                                                       \
    const int xorg_ = (iy0_ * p.IW + ix0_) * 16;                                                                      \
    /* input columns this chunk's VALID gradient columns touch: the rest of the staged width is zero-filled, not fetched */ \
    const int xneed_ = (wc_ - 1) * S + (T == 1 ? 1 : 3);
#define WB_ISSUE_G(i)                                                                                                 \
      if ((i) < p.ngs && (i) * NT + wave * 64 < gtot && !WB_DIAG(4)) {                                                \
        const int c_ = (int)(grc[i] & 0xFFFFu), r_ = (int)(grc[i] >> 16);                                             \
        const bool ok_ = (c_ < wc_) && (r_ < rmax_) && !WB_DIAG(8) && !(WB_DIAG(16) && blockIdx.y == 1);              \
        dma16(rs_g, lb_ + (i) * NT * 16, ok_ ? glc[i] + gorg_ : (int)OOB);                                            \
      }
#define WB_ISSUE_X(i)                                                                                                 \
      if ((i) < p.nxs && (i) * NT + wave * 64 < xtot && !WB_DIAG(4)) {                                                \
        const int c_ = (int)(xrc[i] & 0xFFFFu), r_ = (int)(xrc[i] >> 16);                                             \
        const bool ok_ = ((unsigned)(iy0_ + r_) < (unsigned)p.IH) && ((unsigned)(ix0_ + c_) < (unsigned)p.IW) && (c_ < xneed_) && !WB_DIAG(8) && !(WB_DIAG(32) && r_ >= R); \
        dma16(rs_x, lb_ + (p.xoff + (i) * NT) * 16, ok_ ? xlc[i] + xorg_ : (int)OOB);                                 \
      }
#define WB_ISSUE_SLOTS                                                                                                \
    _Pragma("unroll") for (int i = 0; i < WGB_GSLOTS; ++i) { WB_ISSUE_G(i) }                                          \
    _Pragma("unroll") for (int i = 0; i < WGB_XSLOTS; ++i) { WB_ISSUE_X(i) }
#define WB_ISSUE(BUF)                                                                                                 \
  {                                                                                                                   \
    WB_ISSUE_PREP(BUF)                                                                                                \
    WB_ISSUE_SLOTS                                                                                                    \
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
  const int xpb = p.xcb * 16;                             // bytes per pixel of the x image (64, or 32 for 16-channel inputs)
  const int xblk = XR * p.xw * xpb;                       // bytes of one channel-block group of the x image
  const int gbase = mb * MPW * (R * p.wce * 64) + lane_ch_off;
  // 16-channel image: the lanes of channels 16-31 re-read channels 0-15 (those output columns lie beyond N and are dropped)
  const int xbase = p.xoff * 16 + nb * NPW * xblk + (lane_ch_off & (xpb - 1)) + (PACK2 ? (gi & 1) * xpb : 0);
  // (BLK4: k = 8 (gi >> 1) + 4 j + q -> block row 2 (gi >> 1) + j (j = the first / second transposed read), block column q)
  [[maybe_unused]] const int lean_a0 = BLK4 ? gbase + ((gi >> 1) * 2 * p.wce + q4) * 64 : gbase + lane_px * 64;                                     // (xpb = 64 on this path)
  [[maybe_unused]] const int lean_b0 = BLK4 ? xbase + (((gi >> 1) * 2 + tg) * p.xw + q4) * 64 : xbase + ((T == 1 ? 0 : tg) * p.xw + lane_px * WB_SC) * 64;

#define WB_LOAD(AV, BV, I)                                                                                      \
  {                                                                                                             \
    const int st_ = ks + min((I), cnt - 1) * KS;                                                                \
    const int r_ = st_ / ksteps_row;                                                                            \
    const int px_ = (st_ - r_ * ksteps_row) * 16 + lane_px;                                                     \
    const int ga_ = gbase + (r_ * p.wce + px_) * 64;                                                            \
    _Pragma("unroll") for (int m = 0; m < MPW; ++m)                                                             \
      AV[m] = lds_tr8(buf, ga_ + m * (R * p.wce * 64), ga_ + m * (R * p.wce * 64) + 4 * 64);                    \
    _Pragma("unroll") for (int q = 0; q < NPW; ++q)                                                             \
    _Pragma("unroll") for (int t = 0; t < TM; ++t) {                                                            \
      const int xa_ = xbase + q * xblk + (((r_ * S + (T == 1 ? 0 : tg)) * p.xw) + px_ * S + (PACK2 ? 2 * t : t)) * xpb; \
      BV[q][t] = lds_tr8(buf, xa_, xa_ + 4 * S * xpb);                                                          \
    }                                                                                                           \
  }
// bias gradient: the A operand already holds 8 pixels of "this lane's" output channel (padding pixels are zeros)
#define WB_BIAS(AV)                                                                  \
  if (do_bias) {                                                                     \
    _Pragma("unroll") for (int m = 0; m < MPW; ++m)                                  \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) bsum[m] += (float)AV[m][j];        \
  }
#define WB_MFMA(AV, BV)                                           \
  _Pragma("unroll") for (int m = 0; m < MPW; ++m)                 \
  _Pragma("unroll") for (int q = 0; q < NPW; ++q)                 \
  _Pragma("unroll") for (int t = 0; t < TM; ++t)                  \
    acc[m][q][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AV[m], BV[q][t], acc[m][q][t], 0, 0, 0);

  // ring of p.depth LDS buffers: unit u + depth - 1 streams in while unit u is multiplied.  Ordering (LDS-DMA is invisible to
  // the barrier): every wave waits for ITS loads of unit u with a counted vmcnt, then the barrier makes all of them visible and
  // proves that nobody still reads the buffer the next issue overwrites.
  for (int d = 0; d + 1 < p.depth; ++d)
    if (u_begin + d < u_end) WB_ISSUE(d)
  int ib = 0;
  [[maybe_unused]] unsigned long long tw_ = 0, ti_ = 0, tc_ = 0, t0_ = WB_T(), tk_ = t0_;
  // (LEAN: one copy of the unit loop per number of 16-pixel steps in a row, so that the steps of a unit unroll completely)
  // LEAN2: the pixel-split tilings (KS > 1: the 16/32/64-channel layers) with whole ROWS dealt to the splits (row r belongs to
  // split r mod KS) instead of single steps: a wavefront's steps then have affine addresses -- a per-wavefront base plus
  // immediates -- and unroll like the lean loop; the generic loop divides and multiplies its way to every operand address
  // (and, with the DMA switched off entirely, still takes 0.65 of the layer-1 kernel's 0.76 ms: it is bound by instruction
  // issue, not by memory).  Needs R % KS == 0 and the pixel size of the staged input image known at compile time.
  constexpr bool LEAN2_OK = KS > 1 && !ROT && (R % KS == 0) && MPW == 1 && NPW == 1;
  auto unit_loop = [&](auto kr_tag, auto mode_tag) {
  [[maybe_unused]] constexpr int KR = decltype(kr_tag)::value;
  [[maybe_unused]] constexpr bool LEAN2 = decltype(mode_tag)::value == 2;
  for (int u = u_begin; u < u_end; ++u) {
    wait_dma(max(0, min(p.depth - 2, u_end - 1 - u)) * my_pieces);   // (the units behind u that are already in flight)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WB_STAMP(tw_)
    const int ahead = p.depth - 1;
    const bool issue_ = u + ahead < u_end;
    int nb_ = ib + ahead;
    nb_ = nb_ >= p.depth ? nb_ - p.depth : nb_;
    if constexpr (!LEAN && !LEAN2) {
      if (issue_) WB_ISSUE(nb_)
    }
    WB_STAMP(ti_)
    const unsigned char* buf = smem_b + ib * p.bufu * 16;
    ib = ib + 1 == p.depth ? 0 : ib + 1;
    if constexpr (BLK4) {
      WB_ISSUE_PREP(nb_)
      // this unit's chunk: 33 columns take a ninth step
      const int cwi_ = u % p.nchunk_w;
      constexpr int NSB = KR;                                            // steps of an ordinary chunk (8: chunks of <= 36 columns, 4: <= 20)
      constexpr int GS = NSB == 8 ? WGB_GSLOTS : 2;                      // steps that carry a gradient slot of the next unit (the planner keeps ngs <= GS)
      const bool wide = p.base_w + (cwi_ < p.rem_w ? 1 : 0) > 4 * NSB;   // uniform
      const int rowa = p.wce * 64, rowx = p.xw * 64;
      const unsigned char* pa0 = buf + lean_a0;
      const unsigned char* pa0j = pa0 + rowa;
      const unsigned char* pa1 = pa0 + R * p.wce * 64;
      const unsigned char* pa1j = pa1 + rowa;
      const unsigned char* pb0 = buf + lean_b0;
      const unsigned char* pb0j = pb0 + rowx;
      typedef bf16x4 __attribute__((address_space(3))) * lds_v4b;
      auto tr2 = [&](const unsigned char* b0, const unsigned char* b1, int off) __attribute__((always_inline)) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4b)(b0 + off));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4b)(b1 + off));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      };
      bf16x8 av[2][MPW], bvv[2][NPW][TM];
      auto fetch = [&](auto e_tag) __attribute__((always_inline)) {
        constexpr int e = decltype(e_tag)::value, set = e & 1;
        av[set][0] = tr2(pa0, pa0j, e * 256);
#pragma unroll
        for (int t = 0; t < TM; ++t) bvv[set][0][t] = tr2(pb0, pb0j, e * 256 + t * 64);
        if constexpr (MPW == 2) av[set][1] = tr2(pa1, pa1j, e * 256);
      };
      if (!WB_DIAG(2)) {
        fetch(std::integral_constant<int, 0>{});
        auto step = [&](auto e_tag) __attribute__((always_inline)) {
          constexpr int e = decltype(e_tag)::value, set = e & 1;
          if constexpr (e + 1 < NSB) fetch(std::integral_constant<int, e + 1>{});
          else if constexpr (e + 1 == NSB) { if (wide) fetch(std::integral_constant<int, NSB>{}); }
          if (!WB_DIAG(1)) { WB_MFMA(av[set], bvv[set]); }
          if (issue_ && WB_DIAG(64)) {
            if constexpr (e == 0) { WB_ISSUE_SLOTS }
          } else
          if (issue_) {   // the next unit's DMA: one slot behind the MFMAs of each of the first steps
            if constexpr (e < GS) { WB_ISSUE_G(e) }
            else if constexpr (e < NSB && e - GS < WGB_XSLOTS) { WB_ISSUE_X(e - GS) }
            if constexpr (e == NSB - 1) {
#pragma unroll
              for (int sl = NSB - GS; sl < WGB_XSLOTS; ++sl) { WB_ISSUE_X(sl) }
            }
          }
          if (do_bias) {   // the NBW * TG wavefronts that hold the same gradient operand take turns summing it
            if (--bias_turn < 0) {
              bias_turn = NBW * TG - 1;
#pragma unroll
              for (int m = 0; m < MPW; ++m)
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum[m] += (float)av[set][m][j];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        };
        wb_static_for(step, std::make_integer_sequence<int, NSB>{});
        if (wide) step(std::integral_constant<int, NSB>{});
      } else if (issue_) {
        WB_ISSUE_SLOTS   // (diagnostic: no operand reads -- the DMA still goes out; PREP above has advanced the iterators)
      }
    } else
    if constexpr (LEAN) {
      // the next unit's DMA: its slots ride behind the MFMAs of the first step of every row (KSL slots per row)
      constexpr int KSL = (WGB_GSLOTS + WGB_XSLOTS + R - 1) / R;
      WB_ISSUE_PREP(nb_)
      // ---- lean step loop (KS = 1, 64-byte pixels): a wavefront beside two MFMA-busy partners issues an instruction only every
      //      5-10 cycles, so the loop is kept to the ds_reads and the MFMAs themselves -- the R x KR steps of a unit are unrolled,
      //      every step-dependent part of an operand address (step * 1 KB, tap * 64 B, second pixel half + 256 B) is an immediate
      //      of the ds_read, and the operands of step e + 1 are requested before the MFMAs of step e (two register sets): with
      //      one set the 10 transposed reads of a step and its 6 MFMAs alternated, and the three wavefronts of a SIMD did not
      //      cover each other (stamps: reads alone 4.5 k, MFMAs alone 4.4 k, together 8.2 k ticks per unit).
      constexpr int NS = R * KR;
      const unsigned char* pa0 = buf + lean_a0;
      const unsigned char* pa1 = buf + lean_a0 + R * p.wce * 64;
      const unsigned char* pb0 = buf + lean_b0;
      const int rowb = S * p.xw * 64;
      bf16x8 av[2][MPW], bvv[2][NPW][TM];
      auto fetch = [&](auto e_tag) {
        constexpr int e = decltype(e_tag)::value, set = e & 1, r = e / KR, kx = e % KR;
        av[set][0] = lds_tr8(pa0, e * 1024, e * 1024 + 256);
        const unsigned char* pb = pb0 + r * rowb;
#pragma unroll
        for (int q = 0; q < NPW; ++q)
#pragma unroll
          for (int t = 0; t < TM; ++t)
            bvv[set][q][t] = lds_tr8(pb + q * xblk, kx * 1024 * WB_SC + t * 64, kx * 1024 * WB_SC + t * 64 + 256 * WB_SC);
        if constexpr (MPW == 2) av[set][1] = lds_tr8(pa1, e * 1024, e * 1024 + 256);
      };
      if (!WB_DIAG(2)) {
      fetch(std::integral_constant<int, 0>{});
      auto step = [&](auto e_tag) {
        constexpr int e = decltype(e_tag)::value, set = e & 1, r = e / KR, kx = e % KR;
        if constexpr (e + 1 < NS) fetch(std::integral_constant<int, e + 1>{});
        if (!WB_DIAG(1)) { WB_MFMA(av[set], bvv[set]); }
        if (kx == 0 && issue_) {
#pragma unroll
          for (int j = 0; j < KSL; ++j) {
            constexpr int dummy = 0; (void)dummy;
            const int sl = r * KSL + j;   // (a compile-time constant after unrolling)
            if (sl < WGB_GSLOTS) { WB_ISSUE_G(sl) }
            else if (sl < WGB_GSLOTS + WGB_XSLOTS) { WB_ISSUE_X(sl - WGB_GSLOTS) }
          }
        }
        if (do_bias) {   // the NBW * TG wavefronts that hold the same gradient operand take turns summing it
          if (--bias_turn < 0) {
            bias_turn = NBW * TG - 1;
#pragma unroll
            for (int m = 0; m < MPW; ++m)
#pragma unroll
              for (int j = 0; j < 8; ++j) bsum[m] += (float)av[set][m][j];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      wb_static_for(step, std::make_integer_sequence<int, NS>{});
      }
    } else if constexpr (LEAN2) {
      constexpr int RJ = (R / KS) > 0 ? R / KS : 1, NS2 = RJ * KR, XPB = PACK2 ? 32 : 64;  // (RJ >= 1 wherever LEAN2_OK holds)
      constexpr int KSL = (WGB_GSLOTS + WGB_XSLOTS + RJ - 1) / RJ;
      WB_ISSUE_PREP(nb_)
      const int rowb = S * p.xw * XPB;
      const unsigned char* pa = buf + gbase + lane_px * 64 + ks * (KR * 1024);
      const unsigned char* pb0 = buf + xbase + ((T == 1 ? 0 : tg) * p.xw + lane_px * WB_SC) * XPB + ks * rowb;
      bf16x8 av[2][MPW], bvv[2][NPW][TM];
      auto fetch = [&](auto e_tag) {
        constexpr int e = decltype(e_tag)::value, set = e & 1, j = e / KR, kx = e % KR;
        av[set][0] = lds_tr8(pa, (j * KS * KR + kx) * 1024, (j * KS * KR + kx) * 1024 + 256);
        const unsigned char* pb = pb0 + (j * KS) * rowb;
#pragma unroll
        for (int t = 0; t < TM; ++t)
          bvv[set][0][t] = lds_tr8(pb, (kx * 16 * WB_SC + (PACK2 ? 2 * t : t)) * XPB, (kx * 16 * WB_SC + (PACK2 ? 2 * t : t)) * XPB + 4 * WB_SC * XPB);
      };
      if (!WB_DIAG(2)) {
        fetch(std::integral_constant<int, 0>{});
        auto step = [&](auto e_tag) {
          constexpr int e = decltype(e_tag)::value, set = e & 1, j = e / KR, kx = e % KR;
          if constexpr (e + 1 < NS2) fetch(std::integral_constant<int, e + 1>{});
          if (!WB_DIAG(1)) { WB_MFMA(av[set], bvv[set]); }
          if (kx == 0 && issue_) {
#pragma unroll
            for (int jj = 0; jj < KSL; ++jj) {
              const int sl = j * KSL + jj;   // (a compile-time constant after unrolling)
              if (sl < WGB_GSLOTS) { WB_ISSUE_G(sl) }
              else if (sl < WGB_GSLOTS + WGB_XSLOTS) { WB_ISSUE_X(sl - WGB_GSLOTS) }
            }
          }
          WB_BIAS(av[set])
          __builtin_amdgcn_sched_barrier(0);
        };
        wb_static_for(step, std::make_integer_sequence<int, NS2>{});
      } else if (issue_) {
        WB_ISSUE_SLOTS   // (diagnostic: no operand reads -- the DMA still has to go out; the iterators were advanced by PREP above)
      }
    } else
    if (cnt > 0) {
      if constexpr (NPW == 1 && MPW == 1) {  // operands of step i+1 are fetched before the MFMAs of step i
        bf16x8 a0[MPW], a1[MPW], b0[NPW][TM], b1[NPW][TM];
        WB_LOAD(a0, b0, 0);
        int i = 0;
        for (; i + 1 < cnt; i += 2) {
          WB_LOAD(a1, b1, i + 1);
          __builtin_amdgcn_sched_barrier(0);
          WB_MFMA(a0, b0);
          WB_BIAS(a0)
          __builtin_amdgcn_sched_barrier(0);
          WB_LOAD(a0, b0, i + 2);
          __builtin_amdgcn_sched_barrier(0);
          WB_MFMA(a1, b1);
          WB_BIAS(a1)
          __builtin_amdgcn_sched_barrier(0);
        }
        if (i < cnt) {
          WB_MFMA(a0, b0);
          WB_BIAS(a0)
        }
      } else {  // 6 accumulator tiles: one operand set (the three wavefronts of a SIMD hide each other's LDS latency)
        for (int i = 0; i < cnt; ++i) {
          bf16x8 a0[MPW], b0[NPW][TM];
          WB_LOAD(a0, b0, i);
          WB_MFMA(a0, b0);
          WB_BIAS(a0)
        }
      }
    }
    WB_STAMP(tc_)
  }
  };
  if constexpr (BLK4) {
    if (p.nsb == 4) unit_loop(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});
    else unit_loop(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{});
  } else if constexpr (LEAN) {
    switch (ksteps_row) {
      case 1: unit_loop(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); break;
      case 2: unit_loop(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}); break;
      case 3: unit_loop(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}); break;
      default: unit_loop(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{}); break;
    }
  } else if constexpr (LEAN2_OK) {
    if (p.xcb == (PACK2 ? 2 : 4) && !WB_DIAG(32)) {
      switch (ksteps_row) {
        case 1: unit_loop(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}); break;
        case 2: unit_loop(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}); break;
        case 3: unit_loop(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{}); break;
        default: unit_loop(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{}); break;
      }
    } else {
      unit_loop(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    }
  } else {
    unit_loop(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  }
#ifdef YOGO_DIAG
  if (p.stamps && tid == 0) {
    unsigned long long* d = p.stamps + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4;
    d[0] = tw_; d[1] = ti_; d[2] = tc_; d[3] = __builtin_amdgcn_s_memtime() - t0_;
  }
#endif
#undef WB_ISSUE
#undef WB_ISSUE_SLOTS
#undef WB_ISSUE_PREP
#undef WB_ISSUE_G
#undef WB_ISSUE_X
#undef WB_LOAD
#undef WB_MFMA
#undef WB_BIAS

  // ---- write the slab: [split*KS + ks][t][m][n] ----------------------------------------------------------------------
  float* sl = p.slab + (size_t)(split * KS + ks) * T * p.Mpad * p.Npad;
  const int m0 = m0b * 8, n0 = n0b * 8;
#pragma unroll
  for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
    for (int q = 0; q < NPW; ++q)
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        // PACK2: column l31 of tile t = input channel l31 & 15 of tap 2t + (l31 >> 4)
        const int tap = PACK2 ? 2 * t + (l31 >> 4) : t;
        const int n = n0 + (nb * NPW + q) * 32 + (PACK2 ? (l31 & 15) : l31);
        if (tap < TT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + (mb * MPW + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            sl[((size_t)(tg * TT + tap) * p.Mpad + m) * p.Npad + n] = acc[mi][q][t][r];
          }
        }
      }
  if (do_bias) {  // lane l31 and lane l31 + 32 hold the two pixel halves of channel (mb*MPW + mi)*32 + l31
    const int brow = LEAN ? (split * KS + ks) * (NBW * TG) + nb * TG + tg : split * KS + ks;
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) {
      const float tot = bsum[mi] + __shfl_xor(bsum[mi], 32, 64);
      if (half == 0) p.bias_part[(size_t)brow * p.Mpad + m0 + (mb * MPW + mi) * 32 + l31] = tot;
    }
  }
}

// defined in wgrad_f32.hip (the fixed-order slab reduction is shared)
extern "C" int yogo_internal_wgrad_reduce_q(void* queue, const float* slab, int nslab, int T, int M, int N, int Mpad, int Npad, float clip, float* dw,
                                            const float* bias_part, int nbias, float* db, hipStream_t stream);
extern "C" int yogo_internal_wgrad_reduce(const float* slab, int nslab, int T, int M, int N, int Mpad, int Npad, float clip, float* dw,
                                          const float* bias_part, int nbias, float* db, hipStream_t stream);

namespace {

thread_local char g_wb_plan_txt[256] = "";  // planner parameters of the launch in flight (launch log)

struct WbPlan {
  bool blk4;
  int MBW, NBW, NPW, MPW, KS, R, Mpad, Npad, nchunk_w, base_w, rem_w, wce, nrowg, xw, units, nsplit, units_per_split, ngs, nxs, bufu,
      depth, lds_bytes, xcb, xoff, nsb;
  dim3 grid;
};

bool wb_plan(int B, int N, int M, int IH, int IW, int ks, int stride, WbPlan* pl) {
  const int pad = ks == 3 ? 1 : 0;
  const int OH = (IH + 2 * pad - ks) / stride + 1, OW = (IW + 2 * pad - ks) / stride + 1;
  const int mblocks = cdiv(M, 32), nblocks = cdiv(N, 32);
  int MBW = min(4, mblocks);
  if (MBW == 3) MBW = 2;
  int NBW = min(4 / MBW, nblocks);
  if (NBW == 3) NBW = 2;
  // 128 output channels x >= 64 input channels: two ci-blocks per wavefront, so the gradient tile is staged once per 64 (not
  // 32) input channels
  int NPW = (MBW == 4 && NBW == 1 && nblocks >= 2 && ks == 3) ? 2 : 1;
  // ... or, the same 128 x 64 workgroup tile as 2 x 2 wavefronts of two co-blocks x one ci-block each: a third less LDS read
  // traffic per MFMA
  int MPW = 1;
  if (NPW == 2) { MBW = 2; NBW = 2; NPW = 1; MPW = 2; }
  const int KS = 4 / (MBW * NBW);
  // rows per unit: few channels -> little MFMA work per row, so take more rows per barrier
  const bool tall = MBW == 1 && NBW == 1 && ks == 3 && stride == 1 && OH >= 64;
  int R = (NPW == 2 || MPW == 2) ? (stride == 1 ? 3 : 2) : (stride == 1 ? (tall ? 8 : 4) : 2);
  bool blk4 = MPW == 2 && stride == 1 && ks == 3 && WGB_BLK4;   // 4 x 4-pixel k-steps (wgrad_bf16_kernel, BLK4)
  const int R_row = R;                                          // rows per unit of the row-step loop (when the BLK4 staging does not fit)
  pl->MBW = MBW; pl->NBW = NBW; pl->NPW = NPW; pl->MPW = MPW; pl->KS = KS;
  pl->Mpad = round_up(M, 32 * MBW * MPW);
  pl->Npad = round_up(N, 32 * NBW * NPW);
  if (blk4) R = 4;
  const int TG = ks == 3 ? 3 : 1;
  const int NT = 64 * MBW * NBW * KS * TG;
  const int XR = ks == 3 ? (R - 1) * stride + 3 : R;
  const int xcb = (NBW * NPW == 1 && N <= 16 && KS != 1) ? 2 : 4;  // 16-channel inputs: 32-byte pixels in the staged image (not on the lean KS = 1 path)
  pl->xcb = xcb;
  // column chunks: staged width a multiple of 16, at most 64; the widest that fits the slots and two LDS buffers, then the
  // count with the least zero padding
  bool found = false;
  int best_waste = 1 << 30;
  pl->depth = 0;
  if (blk4) {
    // chunks of at most w columns = nsb four-column steps + a last one for a wide chunk; the two images are packed piece-tight (the
    // kernel does not issue a piece that lies wholly behind its image), the ring is as deep as the LDS allows (at most four).  Among
    // the chunk counts the one with the fewest steps per row: 129 columns = 17 + 7 x 16 (33 steps), 258 = 11 x 20 + 2 x 19 (65).
    const int wlist[2] = {WGB_BLK4_W, 36};
    for (int wi = 0; wi < 2 && !found; ++wi) {
      const int w = wlist[wi], xw = w + 2, nsb = w / 4 - 1;
      if (nsb != 8 && nsb != 4) continue;
      const int gtot = MBW * MPW * 4 * R * w, xtot = NBW * NPW * xcb * XR * xw;
      const int ngs = cdiv(gtot, NT), nxs = cdiv(xtot, NT);
      const int xoff = round_up(gtot, 64), bufu = xoff + round_up(xtot, 64);
      if (ngs > (nsb == 8 ? WGB_GSLOTS : 2) || nxs > WGB_XSLOTS || 2 * bufu * 16 > WGB_LDS_MAX) continue;
      int best_steps = 1 << 30, best_nc = 0;
      for (int nc = cdiv(OW, w); nc <= cdiv(OW, w) + 3 && nc <= OW; ++nc) {
        const int base = OW / nc, rem = OW - base * nc;
        if (base + (rem ? 1 : 0) > w) continue;
        const int steps = rem * (base + 1 > 4 * nsb ? nsb + 1 : nsb) + (nc - rem) * (base > 4 * nsb ? nsb + 1 : nsb);
        if (steps < best_steps) { best_steps = steps; best_nc = nc; }
      }
      if (best_nc == 0) continue;
      pl->nchunk_w = best_nc; pl->wce = w; pl->xw = xw; pl->ngs = ngs; pl->nxs = nxs; pl->bufu = bufu; pl->xoff = xoff; pl->nsb = nsb;
      pl->depth = min(4, WGB_LDS_MAX / (bufu * 16));
      found = true;
    }
    if (!found) { blk4 = false; R = R_row; }   // (unreachable with today's constants: the row-step loop then takes the launch)
  }
  pl->blk4 = blk4;
  pl->R = R;
  const int XR_row = ks == 3 ? (R - 1) * stride + 3 : R;
  for (int wmax = 64; wmax >= 16 && !found; wmax -= 16) {
    for (int nc = cdiv(OW, wmax); nc <= cdiv(OW, wmax) + 4 && nc <= OW; ++nc) {
      const int w = round_up(cdiv(OW, nc), 16);
      if (w > wmax) continue;
      const int xw = (w - 1) * stride + (ks == 3 ? 3 : 1);
      const int ngs = cdiv(MBW * MPW * 4 * R * w, NT), nxs = cdiv(NBW * NPW * xcb * XR_row * xw, NT);
      const int bufu = (ngs + nxs) * NT;
      if (ngs > WGB_GSLOTS || nxs > WGB_XSLOTS || 2 * bufu * 16 > WGB_LDS_MAX) continue;
      const int waste = nc * w - OW;
      if (waste < best_waste) {
        best_waste = waste;
        pl->nchunk_w = nc; pl->wce = w; pl->xw = xw; pl->ngs = ngs; pl->nxs = nxs; pl->bufu = bufu; pl->xoff = ngs * NT; pl->nsb = 0;
        found = true;
      }
    }
  }
  if (!found) return false;
  pl->base_w = OW / pl->nchunk_w;
  pl->rem_w = OW - pl->base_w * pl->nchunk_w;
  if (pl->depth == 0) pl->depth = 3 * pl->bufu * 16 <= WGB_LDS_MAX ? 3 : 2;
  pl->lds_bytes = pl->depth * pl->bufu * 16;
  pl->nrowg = cdiv(OH, R);
  pl->units = B * pl->nrowg * pl->nchunk_w;
  const int gy = pl->Npad / (32 * NBW * NPW), gz = pl->Mpad / (32 * MBW * MPW);
  int nsplit = max(1, min(pl->units, 256 / max(1, gy * gz)));
  pl->units_per_split = cdiv(pl->units, nsplit);
  pl->nsplit = cdiv(pl->units, pl->units_per_split);
  pl->grid = dim3(pl->nsplit, gy, gz);
  return true;
}

// rows of the bias partial-sum region behind the slabs: one per slab, times -- on the lean KS = 1 path -- the NBW * TG wavefronts
// that share a gradient operand and each write their own row (the kernel's `brow`).  ONE definition for the workspace query,
// the launcher's pointer arithmetic and the reduction's row count, so they cannot drift apart.
int wb_bias_rows(const WbPlan& pl, int T) {
  const bool lean = pl.KS == 1;   // (= the kernel's LEAN: KS = 1 tilings never pack two taps)
  return pl.nsplit * pl.KS * (lean ? pl.NBW * (T == 1 ? 1 : 3) : 1);
}

template <int MBW, int NBW, int NPW, int KS, int T, int S, int R, int MPW = 1, bool ROT = false, bool PACK2 = false, bool BLK4 = false>
int wb_launch_one(const WgradBf16Params& p, const WbPlan& pl, hipStream_t stream) {
  // per device, once per instantiation: the dynamic-LDS limit (a second device in the process, or a first call from two host threads,
  // must not see another device's state)
  static std::mutex mu;
  static bool done[64] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    yogo_set_error("wgrad_bf16: hipGetDevice failed");
    return YOGO_ERR_HIP;
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!done[dev]) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<MBW, NBW, NPW, KS, T, S, R, MPW, ROT, PACK2, BLK4>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, WGB_LDS_MAX);
      if (e != hipSuccess) {
        yogo_set_error("wgrad_bf16: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", WGB_LDS_MAX, hipGetErrorString(e));
        return YOGO_ERR_HIP;
      }
      done[dev] = true;
    }
  }
  hipLaunchKernelGGL((wgrad_bf16_kernel<MBW, NBW, NPW, KS, T, S, R, MPW, ROT, PACK2, BLK4>), pl.grid, dim3(64 * MBW * NBW * KS * (T == 1 ? 1 : 3)),
                     pl.lds_bytes, stream, p);
  yogo_launch_log("wgrad_bf16_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %s, %s, %s> | %s", MBW, NBW, NPW, KS, T, S, R, MPW, ROT ? "true" : "false",
                  PACK2 ? "true" : "false", BLK4 ? "true" : "false", g_wb_plan_txt);
  return YOGO_OK;
}

template <int MBW, int NBW, int KS>
int wb_launch(const WgradBf16Params& p, const WbPlan& pl, int T, int stride, hipStream_t stream) {
  if (T == 1) return wb_launch_one<MBW, NBW, 1, KS, 1, 1, 4>(p, pl, stream);
  if (stride == 1) return wb_launch_one<MBW, NBW, 1, KS, 9, 1, 4>(p, pl, stream);
  return wb_launch_one<MBW, NBW, 1, KS, 9, 2, 2>(p, pl, stream);
}

}  // namespace

extern "C" int yogo_conv2d_wgrad_bf16_workspace_bytes(int B, int Cin, int Cout, int IH, int IW, int ks, int stride, size_t* bytes) {
  YOGO_CHECK_ARG(bytes && B > 0 && Cin > 0 && Cout > 0 && IH > 0 && IW > 0 && (ks == 1 || ks == 3) && (stride == 1 || stride == 2),
                 "wgrad_bf16_workspace_bytes: bad arguments");
  WbPlan pl;
  YOGO_CHECK_ARG(wb_plan(B, Cin, Cout, IH, IW, ks, stride, &pl), "wgrad_bf16: no LDS plan");
  // slabs + bias partial rows
  *bytes = ((size_t)pl.nsplit * pl.KS * ks * ks * pl.Mpad * pl.Npad + (size_t)wb_bias_rows(pl, ks * ks) * pl.Mpad) * sizeof(float);
  return YOGO_OK;
}

// fp32 dw (OIHW) / db from bf16 NCHW8c x and g on the bf16 matrix cores (fp32 accumulation); clamped to +-clip when clip > 0
static int conv2d_wgrad_bf16_impl(const void* x, const void* g, float* dw, float* db, void* workspace, int B, int Cin, int Cout, int IH, int IW,
                                 int ks, int stride, float clip, void* queue, hipStream_t stream);
extern "C" int yogo_conv2d_wgrad_bf16(const void* x, const void* g, float* dw, float* db, void* workspace, int B, int Cin, int Cout,
                                      int IH, int IW, int ks, int stride, float clip, hipStream_t stream) {
  return conv2d_wgrad_bf16_impl(x, g, dw, db, workspace, B, Cin, Cout, IH, IW, ks, stride, clip, nullptr, stream);
}
// the same with the split-K reduction DEFERRED: it is recorded in `queue` (yogo_wgrad_reduce_queue_create) and runs, together with
// every other recorded one, when yogo_wgrad_reduce_flush(queue, stream) is called -- one launch for the weight gradients of a
// whole backward pass instead of one per layer.  dw / db / workspace must stay valid until then; same bits as the immediate form.
extern "C" int yogo_conv2d_wgrad_bf16_deferred(const void* x, const void* g, float* dw, float* db, void* workspace, int B, int Cin, int Cout,
                                               int IH, int IW, int ks, int stride, float clip, void* queue, hipStream_t stream) {
  YOGO_CHECK_ARG(queue != nullptr, "conv2d_wgrad_bf16_deferred: null queue");
  return conv2d_wgrad_bf16_impl(x, g, dw, db, workspace, B, Cin, Cout, IH, IW, ks, stride, clip, queue, stream);
}
static int conv2d_wgrad_bf16_impl(const void* x, const void* g, float* dw, float* db, void* workspace, int B, int Cin, int Cout, int IH, int IW,
                                 int ks, int stride, float clip, void* queue, hipStream_t stream) {
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
  p.xw = pl.xw;
#ifdef YOGO_DIAG
  p.diag = g_wb_diag;
  {
    const size_t nwg = (size_t)pl.grid.x * pl.grid.y * pl.grid.z;
    p.stamps = (g_wb_stamps && nwg * 32 <= g_wb_stamps_bytes) ? g_wb_stamps : nullptr;
    if (p.stamps) (void)hipMemsetAsync(p.stamps, 0, nwg * 32, stream);
  }
#endif
  p.units = pl.units; p.units_per_split = pl.units_per_split; p.ngs = pl.ngs; p.nxs = pl.nxs; p.bufu = pl.bufu; p.depth = pl.depth; p.xcb = pl.xcb; p.xoff = pl.xoff; p.nsb = pl.nsb;
  if (yogo_launch_log_enabled())
    snprintf(g_wb_plan_txt, sizeof(g_wb_plan_txt), "N=%d M=%d in=%dx%d s=%d T=%d B=%d R=%d wce=%d nchunk_w=%d xw=%d nsb=%d slots=%d+%d depth=%d lds=%d units=%d units_per_split=%d grid=%ux%ux%u",
             Cin, Cout, IH, IW, stride, T, B, pl.R, pl.wce, pl.nchunk_w, pl.xw, pl.nsb, pl.ngs, pl.nxs, pl.depth, pl.lds_bytes, pl.units, pl.units_per_split,
             pl.grid.x, pl.grid.y, pl.grid.z);
  int lrc = YOGO_OK;
  if (pl.MPW == 2) {
    if (stride == 1 && pl.blk4) lrc = wb_launch_one<2, 2, 1, 1, 9, 1, 4, 2, false, false, true>(p, pl, stream);
    else if (stride == 1) lrc = wb_launch_one<2, 2, 1, 1, 9, 1, 3, 2>(p, pl, stream);
    else lrc = wb_launch_one<2, 2, 1, 1, 9, 2, 2, 2>(p, pl, stream);
  } else {
    const int cfg = pl.MBW * 100 + pl.NBW * 10 + pl.KS;
    if (pl.R == 8 && pl.xcb == 2) {
      lrc = wb_launch_one<1, 1, 1, 4, 9, 1, 8, 1, false, true>(p, pl, stream);
    } else if (pl.R == 8) {
      lrc = wb_launch_one<1, 1, 1, 4, 9, 1, 8>(p, pl, stream);
    } else
    switch (cfg) {
      case 411: lrc = wb_launch<4, 1, 1>(p, pl, T, stride, stream); break;
      case 221: lrc = wb_launch<2, 2, 1>(p, pl, T, stride, stream); break;
      case 212: lrc = wb_launch<2, 1, 2>(p, pl, T, stride, stream); break;
      case 141: lrc = wb_launch<1, 4, 1>(p, pl, T, stride, stream); break;
      case 122: lrc = wb_launch<1, 2, 2>(p, pl, T, stride, stream); break;
      case 114: lrc = wb_launch<1, 1, 4>(p, pl, T, stride, stream); break;
      default:
        yogo_set_error("wgrad_bf16: unsupported wave layout %d", cfg);
        return YOGO_ERR_ARG;
    }
  }
  if (lrc != YOGO_OK) return lrc;
  YOGO_CHECK_LAUNCH("conv2d_wgrad_bf16");
  const int nbias = wb_bias_rows(pl, T);
  return yogo_internal_wgrad_reduce_q(queue, p.slab, pl.nsplit * pl.KS, T, Cout, Cin, pl.Mpad, pl.Npad, clip, dw, p.bias_part, nbias, db, stream);
}
