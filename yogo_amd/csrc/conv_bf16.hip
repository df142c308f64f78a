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
#include "conv_bf16_ws.h"
#include "conv_bf16_ws16.h"
#include "conv_bf16_ws3.h"
#include <type_traits>
#include <utility>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// native vector types (HIP's u32x4/u32x2 are structs and get spilled to scratch when selected/stored through pointers)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

#define BF_MAX_TAPS 9
#define BF_LDS_BUDGET (80 * 1024)      // two 4-wavefront workgroups per CU
#define BF_LDS_MAX (160 * 1024)        // one workgroup per CU (all of the LDS)

struct ConvBf16Params {
  const u32x4* in;    // [B][Kb][IH][IW] units
  const u32x4* wp;    // [T][Kb][Mpad] units (8 input channels of one output channel each)
  const float* bias;  // [M] fp32 or null
  u32x2* out;         // bf16 8c viewed as 8-byte halves: [B][Mb][OH][OW][2]
  float* out_f32;     // when set: fp32 NCHW [B][M][OH][OW] instead of `out`
  u32x2* out_pre;     // optional second bf16 output: conv + bias BEFORE the activation (SiLU blocks keep it for backward)
  const u32x2* act_ref;     // training dgrad: multiply by act'(ref); ref = bf16 tensor shaped like `out` (8-byte halves)
  // sign map of a LeakyReLU block's output, [B][2][OH][OW][Mpad/16] bytes in the epilogue's lane order: byte (h, pixel, q),
  // bit i + 4e = (channel 4h + i of channel block 2q + e > 0), h, e in {0, 1}, i < 4.  The forward pass writes it (REF = 3), the data gradient reads it in place of act_ref
  // (REF = 2): 1/16 of the bytes of the bf16 reference.
  unsigned char* signs;
  const float* chan_scale;  // optional [B][M] Dropout2d channel mask (already scaled)
  float* stats_part;        // optional BatchNorm partial sums [B*gridDim.x][Mpad][2] of the fp32 pre-activation
  int ref_act;
  int B, Kb, M, Mpad, Mb;
  int IH, IW, OH, OW, a, T;
  int dy_min, dx_min, span_y, span_x;
  int ncb, TW, tiles_per_band;
  // ceil(2^32 / d) for the divisors of the prologue (exact for n * d < 2^32): grid x, grid x*y, grid y, tiles per band, band
  // width (full / last band), staged row length (full / last band)
  unsigned m_gx, m_gxy, m_gy, m_tpb, m_bw, m_bwl, m_lw, m_lwl;
  int CKb, ckb_shift, nchunk, ldsw_off, lds_dummy;
  int act;
  // dma = 1: a chunk fits the kernel's PF slots and two LDS buffers -> chunk c+1 streams into the other buffer by LDS-DMA
  // while the matrix cores work on chunk c.  ni_slots / n_slots: input / all slots of a chunk; bufu: units per buffer.
  int dma, ni_slots, n_slots, bufu;
  int bufs;  // LDS units between the buffers of consecutive chunks: bufu (two buffers) or 0 (dma = 2: one buffer, see below)
  int lean4; // 4-wavefront 3x3 tiles, 16-channel chunks, one LDS buffer: the unrolled step loop
  // rowdma (with lean4): the input tile is staged ROW by ROW -- one LDS-DMA piece = 64 consecutive units of one tile row of one
  // channel block, rows at a fixed pitch of lwp (64 or 128) units.  Row index, row validity, the row's byte offset and the LDS
  // address of a piece are SCALARS; the per-lane part (column offset, column validity) is the same for every piece and is
  // formed once.  The per-lane slot decode this replaces was ~16 vector instructions per slot and a quarter of the vector
  // instructions of a 16/32-channel tile, whose throughput is bound by vector-instruction issue (profiles/r03_issue_util.txt).
  int rowdma, lwp;
  int ring;  // stride-2 data gradient of the 128-channel tile: 16-channel chunks in a ring of 4 LDS buffers (fixed 2 + 3 slot layout)
#ifdef YOGO_DIAG
  // diagnostic build only (bash build.sh diag -> libyogo_hip_diag.so; tools/bench_conv_bf16.py): ablation bits and phase stamps.
  // The production library contains none of this code.
  int dbg;  // 1 = no output stores, 2 = no MFMA loop, 4 = no DMA, 8 / 16 = no input / weight DMA
  unsigned long long* stamps;  // [workgroup][16]: s_memtime at start / loop / epilogue / end; [4..7] / [8..11]: ping-pong phase sums of wave 0 / 4 (fetch, barrier, MFMA, barrier)
#endif
};

#ifdef YOGO_DIAG
#define BF_DBG(BIT) (p.dbg & (BIT))
#define BF_STAMP() (p.stamps ? __builtin_amdgcn_s_memtime() : 0ull)
#else
#define BF_DBG(BIT) 0
#define BF_STAMP() 0ull
#endif

// sum over the 32 lanes of each half-wave with DPP adds (no LDS traffic); the result is valid in lanes 16-31 / 48-63
__device__ __forceinline__ float half_wave_sum(float v) {
#define DPP_ADD(CTRL, ROWMASK)                                                                                         \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
  DPP_ADD(0xB1, 0xf)   // quad_perm [1,0,3,2]
  DPP_ADD(0x4E, 0xf)   // quad_perm [2,3,0,1]
  DPP_ADD(0x141, 0xf)  // row_half_mirror
  DPP_ADD(0x140, 0xf)  // row_mirror: every lane holds its row's (16 lanes) total
  DPP_ADD(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
#undef DPP_ADD
  return v;
}

// n / d through the precomputed magic number m = ceil(2^32 / d) (d = 1: m does not fit, n is returned)
__device__ __forceinline__ int udivm(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }

typedef int i32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor of `bytes` bytes at `ptr`
__device__ __forceinline__ i32x4 bf_make_rsrc(const void* ptr, int bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), bytes, 0x00020000};
}
// One LDS-DMA piece from inline asm (invisible to hipcc's waitcnt bookkeeping; retired by explicit s_waitcnt vmcnt): 64 lanes x
// 16 bytes from (descriptor, per-lane byte offset + scalar offset) to LDS bytes [lds_addr, lds_addr + 1024).  The dynamic LDS
// block of this kernel starts at LDS address 0 (no static __shared__ objects).
__device__ __forceinline__ void bf_dma16(i32x4 rsrc, unsigned lds_addr, int voff, int soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(lds_addr), "s"(rsrc), "s"(soff)
               : "memory");
}

// Five LDS-DMA pieces of one descriptor in ONE statement (ping-pong fetch phase: a wavefront beside an MFMA-saturating partner
// issues an instruction only every ~10 cycles, so the request sequence is kept to 3 instructions per piece): LDS destinations
// lds_addr + k * STRIDE (M0 is stepped in place), per-lane source offsets v0..v4, common scalar offset.
template <int STRIDE>
__device__ __forceinline__ void bf_dma16x5(i32x4 rsrc, unsigned lds_addr, int soff, int v0, int v1, int v2, int v3, int v4) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %7\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %1, %6, %8 offen lds\n\t"
      "s_add_u32 m0, m0, %9\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %6, %8 offen lds\n\t"
      "s_add_u32 m0, m0, %9\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %3, %6, %8 offen lds\n\t"
      "s_add_u32 m0, m0, %9\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %4, %6, %8 offen lds\n\t"
      "s_add_u32 m0, m0, %9\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %5, %6, %8 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "s"(rsrc), "s"(lds_addr), "s"(soff), "i"(STRIDE)
      : "memory", "scc");
}

// Five LDS-DMA pieces in one statement, the first two through descriptor ra (scalar offset sa), the other three through rb (sb):
// the 2 + 3 slot layout of the stride-2 data gradient's chunk ring (gradient tile, weight slices).
template <int STRIDE>
__device__ __forceinline__ void bf_dma16_2p3(i32x4 ra, i32x4 rb, unsigned lds_addr, int sa, int sb, int v0, int v1, int v2, int v3, int v4) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %8\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %1, %6, %9 offen lds\n\t"
      "s_add_u32 m0, m0, %11\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %6, %9 offen lds\n\t"
      "s_add_u32 m0, m0, %11\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %3, %7, %10 offen lds\n\t"
      "s_add_u32 m0, m0, %11\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %4, %7, %10 offen lds\n\t"
      "s_add_u32 m0, m0, %11\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %5, %7, %10 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "s"(ra), "s"(rb), "s"(lds_addr), "s"(sa), "s"(sb), "i"(STRIDE)
      : "memory", "scc");
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a compile-time unrolled loop whose index can feed constexpr
template <class F, int... I>
__device__ __forceinline__ void bf_static_for(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

// NWV wavefronts per workgroup, each owning NW 32-pixel groups x all MW channel blocks: NWV = 8 shares one staged weight
// slice between 512 output pixels (the weight slice is the larger part of the LDS traffic at 128 channels).
//
// S2D: data gradient of a stride-2 3x3 convolution, decomposed by output parity.  dx[2a+py][2b+px] only receives the taps
// with ky = py+1 (mod 2), kx = px+1 (mod 2): 1 + 2 + 2 + 4 = 9 tap-GEMMs per 2x2 output quad instead of the 36 a
// zero-upsampled formulation spends.  A workgroup owns one row parity py (blockIdx.y & 1) and computes both column parities
// (two accumulator sets) from ONE staged, un-upsampled dy tile with a one-unit halo; the weight slices arrive in class order
// (pack mode 2) so a workgroup stages only the 3 (py = 0) or 6 (py = 1) taps it needs.
//
// PF: slots (16-byte elements per lane and chunk) of the LDS-DMA pipeline.  When a chunk fits (p.dma) the kernel runs
//   barrier -> issue DMA(c+1 -> buffer (c+1)&1) -> MFMA(c from buffer c&1): one barrier per chunk, no staging registers, no
//   ds_write, no vector-ALU address work inside the loop (every lane's source offsets are decoded once per workgroup).
//
// PP on a 4-wavefront tile selects the UNROLLED single-buffer step loop of the 3x3 / 16-channel-chunk case instead (a separate
//   instantiation: merged with the generic loop it costs the registers that keep four workgroups on a CU).
// PP ("ping-pong", 8-wavefront workgroups = two wavefronts per SIMD, stride-1 / stride-2 forward-type 3x3 tiles): the two
//   wavefronts of a SIMD alternate roles phase by phase -- one issues a cluster of 3 * MW * NW back-to-back MFMAs from
//   operands it already holds while the other fetches its next 3 steps' operands from LDS and issues its share of the next
//   chunk's LDS-DMA -- separated by raw s_barriers (waves 4-7 run one phase behind waves 0-3).  The matrix pipe of every SIMD
//   then always has one wavefront whose MFMAs need nothing but registers; in the interleaved loop (PP = false) both
//   wavefronts wait for LDS / issue DMA at the same moments and the pipe idles (47 % busy measured, profiles/r02_*).
//   The LDS-DMA is issued from inline asm with a counted / explicit vmcnt: hipcc would otherwise drain it (vmcnt(0)) in front
//   of every later LDS read.
// LEPI (4-wavefront tiles with the unrolled step loop): the epilogue's LEAN order of operations only -- the launcher selects
//   it when no BatchNorm sums, SiLU, second output or non-LeakyReLU reference is asked for (every launch of the base_model
//   training step).  Merged with the general order behind a run-time test (the default of the 4-wavefront tiles) the unrolled
//   group loop carries both paths' live values through every join: with the stores switched off, the epilogue ARITHMETIC of the
//   16 -> 32 channel forward was 0.23 ms of its 0.77 ms (tools/stamps_conv_bf16.py, ablation bits 1 / 256).
template <int MW, int NW, int NWV, bool S2D, int PF, bool OUT_F32, int REF, bool PP = false, bool LEPI = false>
__global__ __launch_bounds__(64 * NWV, 2) void conv_bf16_kernel(const ConvBf16Params p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
  constexpr int BM = 32 * MW;
  constexpr int BM_SHIFT = MW == 4 ? 7 : (MW == 2 ? 6 : 5);
  constexpr int NT = 64 * NWV;
  constexpr int PT = NWV * NW * 32;
  constexpr int NC = S2D ? 2 : 1;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  [[maybe_unused]] const unsigned long long t_start = BF_STAMP();
  // XCD-aware bijective remap: contiguous runs of (image, tile) per XCD so halo rows hit the same L2
  const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
  const unsigned widx = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
  const int gx = gridDim.x, gy = gridDim.y;
  const int b = udivm((int)widx, gx * gy, p.m_gxy);
  const int rem_ = (int)widx - b * gx * gy;
  // stride-2 data gradient with <= 64 rows: the two row parities of a tile are NEIGHBOURS in the walk -- they stage the same gradient
  // tile, and 195 tiles apart (parity-major) the second fetch no longer finds it in the XCD's L2 (FETCH 2.2x the tensor).  Layer 2's data
  // gradient -6 % in the same-box A/B (gpurun_out/r4_abpair.log); the 128-row tile (layer 4, not bound by memory) loses 4 % and keeps
  // the parity-major order.
  constexpr bool PAIR = S2D && MW < 4;
  const int mb_ = udivm(PAIR ? (rem_ >> 1) : rem_, gx, p.m_gx);
  const int by = PAIR ? 2 * mb_ + (rem_ & 1) : mb_;
  const int bx = (PAIR ? (rem_ >> 1) : rem_) - mb_ * gx;
  const int py = S2D ? (by & 1) : 0;
  const int m0 = (S2D ? (by >> 1) : by) * BM;
  // S2D tiles the quad grid of this row parity; everything else tiles the output
  const int OHt = S2D ? ((p.OH - py + 1) >> 1) : p.OH;
  const int OWt = S2D ? ((p.OW + 1) >> 1) : p.OW;
  const int span_y = S2D ? 1 + py : p.span_y;
  const int tbase = S2D ? 3 * py : 0;       // first weight slice
  const int ntap = S2D ? 3 + 3 * py : p.T;  // weight slices this workgroup uses
  const int n0tap = S2D ? 1 + py : ntap;    // ... of which belong to column parity 0
  const int cb = udivm(bx, p.tiles_per_band, p.m_tpb);
  const int tb = bx - cb * p.tiles_per_band;
  const int j0 = cb * p.TW;
  const int bw = min(p.TW, OWt - j0);
  const int NPb = OHt * bw;
  const int p0 = tb * PT;
  if (p0 >= NPb) {
    if (!S2D && p.stats_part != nullptr && tid < BM) {
      float* dst = p.stats_part + (((size_t)b * gridDim.x + bx) * p.Mpad + m0 + tid) * 2;
      dst[0] = 0.f;
      dst[1] = 0.f;
    }
    return;
  }
  // bias / channel scale of this workgroup's channels: fetched now, parked in LDS for the epilogue
  float bias_reg = 0.f, scale_reg = 0.f;
  if (tid < BM && m0 + tid < p.M) {
    bias_reg = p.bias != nullptr ? p.bias[m0 + tid] : 0.f;
    scale_reg = p.chan_scale != nullptr ? p.chan_scale[(size_t)b * p.M + m0 + tid] : 1.f;
  }
  const int p1 = min(p0 + PT, NPb);
  const bool lastband = cb == p.ncb - 1;
  const unsigned m_bw = lastband ? p.m_bwl : p.m_bw;
  const int i_lo = udivm(p0, bw, m_bw), i_hi = udivm(p1 - 1, bw, m_bw);
  const int rows_in = (i_hi - i_lo) * p.a + span_y;
  const int iy0 = i_lo * p.a + p.dy_min;
  const int ix0 = j0 * p.a + p.dx_min;
  const int lw_need = (bw - 1) * p.a + p.span_x;   // units of a tile row the MFMAs read
  const int lw = (NWV == 4 && !S2D && p.rowdma) ? p.lwp : lw_need;   // row pitch of the LDS image

  int boff[NW], opix[NW];
  bool pvalid[NW], pvalid1[NW];
#pragma unroll
  for (int n = 0; n < NW; ++n) {
    const int pp = p0 + (wave * NW + n) * 32 + l31;
    pvalid[n] = pp < p1;
    const int pc = pvalid[n] ? pp : (p1 - 1);
    const int i = udivm(pc, bw, m_bw), j = pc - i * bw;
    boff[n] = ((i - i_lo) * p.a) * lw + j * p.a;
    if constexpr (S2D) {
      opix[n] = (2 * i + py) * p.OW + 2 * (j0 + j);
      pvalid1[n] = pvalid[n] && (2 * (j0 + j) + 1 < p.OW);
    } else {
      opix[n] = i * p.OW + j0 + j;
      pvalid1[n] = false;
    }
  }

  f32x16 acc[NC][MW][NW];
  // per-lane unit offsets of the operand reads: weights (channel pair half, output channel), input (pixel, channel pair half)
  const int a_vu = half * BM + l31;
  int b_vu[NW];
#pragma unroll
  for (int n = 0; n < NW; ++n) b_vu[n] = boff[n] + half * ((i_hi - i_lo) * p.a + span_y) * lw;
// zeroed after the first chunk's DMA is on its way (the 128 moves then cost nothing)
#define ACC_ZERO()                                                                  \
  _Pragma("unroll") for (int c = 0; c < NC; ++c)                                    \
  _Pragma("unroll") for (int mb = 0; mb < MW; ++mb)                                 \
  _Pragma("unroll") for (int n = 0; n < NW; ++n)                                    \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[c][mb][n][r] = 0.f;

  const u32x4* inb = p.in + (size_t)b * p.Kb * p.IH * p.IW;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // LDS image of a chunk: the input tile [CKb][rows_in][lw] units at its natural pitch (element e of the flattened staging
  // order sits at unit e -- the image LDS-DMA writes), the weight slices [ntap][CKb][BM] from unit p.ldsw_off on.
  // Tap offsets inside the tile, one per lane (read back with v_readlane):
  int toff_lane;
  {
    const int ti = min(tbase + lane, BF_MAX_TAPS - 1);
    if constexpr (S2D) toff_lane = ((0x190 >> ti) & 1) * lw + ((0x144 >> ti) & 1);  // dy (row, col) of the class-ordered slices
    else toff_lane = p.span_x == 3 ? (ti / 3) * lw + (ti % 3) : 0;
  }
  // ceil(2^32 / d) magic numbers for the flattened tile indexing; d = 1 uses 2^32 - 1 (umulhi gives e - 1 for e > 0) plus
  // a branch-free correction
  const int per_kb = rows_in * lw;
  const int one_lw = lw == 1, one_perkb = per_kb == 1;
  const unsigned inv_lw = one_lw ? 0xFFFFFFFFu : (lastband ? p.m_lwl : p.m_lw);
  // 0xFFFFFFFF / d + 1 = ceil(2^32 / d) for every d > 1 (a 32-bit division; d | 2^32 included)
  const unsigned inv_perkb = one_perkb ? 0xFFFFFFFFu : 0xFFFFFFFFu / (unsigned)per_kb + 1u;
  const int hk_shift = p.ckb_shift - 1;  // k-steps (16 channels) per tap and chunk = CKb / 2
  const int hk = 1 << hk_shift;
  // one chunk = CKb channel blocks of the input tile (rows_in x lw units, zero padded) followed by the weight slices
  // [ntap][CKb][BM]; both are staged through ONE flattened element index so every lane carries the same number of loads
  const int itotal = p.CKb * per_kb;
  const int total = itotal + ntap * p.CKb * BM;

#define ST_DECODE(E)                                                                               \
  const int e_ = min((E), total - 1);                                                              \
  const bool isw_ = e_ >= itotal;                                                                  \
  const int ei_ = isw_ ? 0 : e_;                                                                   \
  const int kc_ = (int)__umulhi((unsigned)ei_, inv_perkb) + (one_perkb & (ei_ != 0));              \
  const int rm_ = ei_ - kc_ * per_kb;                                                              \
  const int r_ = (int)__umulhi((unsigned)rm_, inv_lw) + (one_lw & (rm_ != 0));                     \
  const int x_ = rm_ - r_ * lw;                                                                    \
  const int w_ = e_ - itotal;
#define ST_LOAD(V, E, KB0)                                                                         \
  {                                                                                                \
    ST_DECODE(E)                                                                                   \
    const int iy_ = iy0 + r_, ix_ = ix0 + x_;                                                      \
    const bool ok_ = isw_ || ((iy_ >= 0) && (iy_ < p.IH) && (ix_ >= 0) && (ix_ < p.IW));           \
    const int R_ = w_ >> BM_SHIFT;                                                                 \
    const int wi_ = ((tbase + (R_ >> p.ckb_shift)) * p.Kb + (KB0) + (R_ & (p.CKb - 1))) * p.Mpad + \
                    m0 + (w_ & (BM - 1));                                                          \
    const int ii_ = ok_ ? (((KB0) + kc_) * p.IH + iy_) * p.IW + ix_ : 0;                           \
    const u32x4* src_ = isw_ ? p.wp + wi_ : inb + ii_;                                             \
    V = *src_;                                                                                     \
  }
// the decode is redone at commit time (from a laundered thread index) so nothing but the data stays live across the MFMAs
#define ST_STORE(V, E)                                                                             \
  {                                                                                                \
    ST_DECODE(E)                                                                                   \
    const int iy_ = iy0 + r_, ix_ = ix0 + x_;                                                      \
    const bool ok_ = isw_ || ((iy_ >= 0) && (iy_ < p.IH) && (ix_ >= 0) && (ix_ < p.IW));           \
    const int dst_ = (E) < total ? (isw_ ? p.ldsw_off + w_ : ei_) : p.lds_dummy;                   \
    smem4[dst_] = ok_ ? V : zero4;                                                                 \
  }
#define LAUNDER_TID(T) int T = tid; asm volatile("" : "+v"(T));

// operand fetch of step S: everything that depends on the step is scalar (tap, channel pair), the per-lane parts (a_vu, b_vu)
// are fixed for the whole kernel -- one vector add per ds_read_b128 address
#define BF_LOAD(AV, BV, S, SEND)                                                                  \
  {                                                                                               \
    const int s_ = min((S), (SEND)-1);                                                            \
    const int t_ = s_ >> hk_shift;                                                                \
    const int k2_ = 2 * (s_ & (hk - 1));                                                          \
    const u32x4* wI_ = ldsI + (k2_ * per_kb + __builtin_amdgcn_readlane(toff_lane, t_));          \
    const u32x4* wW_ = ldsW + ((t_ << p.ckb_shift) + k2_) * BM;                                   \
    if (!BF_DBG(64)) { _Pragma("unroll") for (int mb = 0; mb < MW; ++mb) AV[mb] = wW_[a_vu + mb * 32]; } \
    if (!BF_DBG(32)) { _Pragma("unroll") for (int n = 0; n < NW; ++n) BV[n] = wI_[b_vu[n]]; }     \
  }
#define BF_MFMA(CI, AV, BV)                                                                       \
  _Pragma("unroll") for (int mb = 0; mb < MW; ++mb)                                               \
  _Pragma("unroll") for (int n = 0; n < NW; ++n)                                                  \
    acc[CI][mb][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AV[mb]),  \
                                                             __builtin_bit_cast(bf16x8, BV[n]), acc[CI][mb][n], 0, 0, 0);
#define BF_HOOK()
// MFMA over steps [SBEG, SEND) = (tap, 16-channel step); operands of step s+1 are read before the MFMAs of step s.  The MFMA
// cluster runs at raised priority so that the two wavefronts of a SIMD fall out of phase (one fetches while the other multiplies)
// interleave directive for one (operand fetch of the next step, MFMA cluster of this step) pair: one ds_read_b128 behind each
// of the first MW + NW MFMAs, so a wavefront's own stream keeps the matrix pipe fed while its operands arrive
#define BF_INTERLEAVE()                                                                           \
  _Pragma("unroll") for (int q_ = 0; q_ < MW + NW; ++q_) {                                        \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                            \
  }                                                                                               \
  __builtin_amdgcn_sched_group_barrier(0x008, MW * NW - (MW + NW) > 0 ? MW * NW - (MW + NW) : 0, 0);
#define BF_RUN(CI, SBEG, SEND)                                                                    \
  {                                                                                               \
    u32x4 a0[MW] = {}, b0[NW] = {}, a1[MW] = {}, b1[NW] = {};                                     \
    int s = (SBEG);                                                                               \
    if constexpr (MW == 1) {  /* 32-row tiles are memory-bound: one operand set, fewer registers, more workgroups per CU */ \
      for (; s < (SEND); ++s) {                                                                   \
        BF_LOAD(a0, b0, s, SEND);                                                                 \
        BF_MFMA(CI, a0, b0);                                                                      \
        BF_HOOK()                                                                                 \
      }                                                                                           \
    } else {                                                                                      \
    BF_LOAD(a0, b0, s, SEND);                                                                     \
    for (; s + 1 < (SEND); s += 2) {                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      BF_LOAD(a1, b1, s + 1, SEND);                                                               \
      BF_MFMA(CI, a0, b0);                                                                        \
      if (MW * NW >= MW + NW) { BF_INTERLEAVE() }                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      BF_HOOK()                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      BF_LOAD(a0, b0, s + 2, SEND);                                                               \
      BF_MFMA(CI, a1, b1);                                                                        \
      if (MW * NW >= MW + NW) { BF_INTERLEAVE() }                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      BF_HOOK()                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    if (s < (SEND)) {                                                                             \
      BF_MFMA(CI, a0, b0);                                                                        \
      BF_HOOK()                                                                                   \
    }                                                                                             \
    }                                                                                             \
  }
#define BF_COMPUTE()                                                                              \
  {                                                                                               \
    BF_RUN(0, 0, n0tap * hk)                                                                      \
    if constexpr (S2D) BF_RUN(NC - 1, n0tap * hk, ntap * hk)                                      \
  }

  [[maybe_unused]] const unsigned long long t_loop = BF_STAMP();
  if (p.dma) {
    // Every lane owns PF fixed elements of a chunk: slots [0, ni) cover the input tile, [ni, ns) the weight slices.  The
    // source byte offset of each is decoded ONCE; image borders, tile tails and unused slices become out-of-range offsets,
    // which the buffer unit turns into zeros written to LDS.  A chunk then costs one `buffer_load_dwordx4 ... lds` per slot
    // with the chunk's base in the scalar offset.
    constexpr unsigned OOB = 0x80000000u;
    const int ni = p.ni_slots, ns = p.n_slots;
    const int wtotal = total - itotal;
    i32x16 voffv = {};  // the slot offsets: a register vector, so that a slot can also be picked at run time (relative indexing)
    [[maybe_unused]] int vcol0 = (int)OOB, vcol1 = (int)OOB;   // rowdma: this lane's column byte offset in segment 0 / 1 of a tile row
    const bool rowdma = NWV == 4 && !S2D && p.rowdma != 0;      // (uniform)
    if (rowdma) {
      const int c0 = lane, c1 = 64 + lane;
      const int x0 = ix0 + c0, x1 = ix0 + c1;
      vcol0 = (c0 < lw_need && x0 >= 0 && x0 < p.IW) ? x0 * 16 : (int)OOB;
      vcol1 = (c1 < lw_need && x1 >= 0 && x1 < p.IW) ? x1 * 16 : (int)OOB;
      // weight pieces of this wavefront: piece wave + NWV * j = units [64 * piece, 64 * piece + 64) of the chunk's slices
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int w_ = (wave + NWV * j) * 64 + lane;
        const bool wok = w_ < wtotal;
        const int R_ = w_ >> BM_SHIFT;
        const int wi_ = ((tbase + (R_ >> p.ckb_shift)) * p.Kb + (R_ & (p.CKb - 1))) * p.Mpad + m0 + (w_ & (BM - 1));
        voffv[j] = wok ? wi_ * 16 : (int)OOB;
      }
    } else
    {
      // input element tid + i*NT -> (channel block kc, tile row r, tile column x): slot 0 by division, every further slot by
      // stepping (NT = skc*per_kb + sr*lw + sx) with two carries -- adds and compares instead of quarter-rate multiplies
      const int skc = (int)__umulhi((unsigned)NT, inv_perkb) + one_perkb;  // NT != 0
      const int srm = NT - skc * per_kb;
      const int sr = (int)__umulhi((unsigned)srm, inv_lw) + (one_lw & (srm != 0));
      const int sx = srm - sr * lw;
      int kc_ = (int)__umulhi((unsigned)tid, inv_perkb) + (one_perkb & (tid != 0));
      const int rm0 = tid - kc_ * per_kb;
      int r_ = (int)__umulhi((unsigned)rm0, inv_lw) + (one_lw & (rm0 != 0));
      int x_ = rm0 - r_ * lw;
      const int rowb = p.IW * 16, kcb = p.IH * p.IW * 16;
      if constexpr (PP && NWV == 8) {
        // ping-pong tiles (fixed 5 + 5 layout): the weight slices of chunk 0 are requested BEFORE the input slots are decoded --
        // their offsets cost a dozen instructions each, and the first chunk's round trip then runs under the rest of the decode
        // (the first chunk is 7-9 k of a tile's 47-68 k ticks)
#pragma unroll
        for (int i = 5; i < 10; ++i) {
          const int w_ = tid + (i - 5) * NT;
          const bool wok = w_ < wtotal;
          const int R_ = w_ >> BM_SHIFT;
          const int wi_ = ((tbase + (R_ >> p.ckb_shift)) * p.Kb + (R_ & (p.CKb - 1))) * p.Mpad + m0 + (w_ & (BM - 1));
          voffv[i] = wok ? wi_ * 16 : (int)OOB;
        }
        if (!BF_DBG(4))
          bf_dma16x5<NT * 16>(bf_make_rsrc(p.wp, p.T * p.Kb * p.Mpad * 16), (unsigned)(wave * 64 * 16) + (unsigned)(5 * NT * 16), 0, voffv[5], voffv[6],
                              voffv[7], voffv[8], voffv[9]);
      }
#pragma unroll
      for (int i = 0; i < ((PP && NWV == 8) ? 5 : PF); ++i) {
        const int e = tid + i * NT;
        if (i < ni) {  // uniform (ping-pong layout: ni = PF / 2 slots are reserved for the input tile, unused ones fetch nothing)
          const int iy_ = iy0 + r_, ix_ = ix0 + x_;
          const bool iok = (e < itotal) && (iy_ >= 0) && (iy_ < p.IH) && (ix_ >= 0) && (ix_ < p.IW);
          voffv[i] = iok ? kc_ * kcb + iy_ * rowb + ix_ * 16 : (int)OOB;
          x_ += sx;
          r_ += sr;
          kc_ += skc;
          if (x_ >= lw) { x_ -= lw; ++r_; }
          if (r_ >= rows_in) { r_ -= rows_in; ++kc_; }
        } else {
          const int w_ = e - ni * NT;
          const bool wok = w_ < wtotal;
          const int R_ = w_ >> BM_SHIFT;
          const int wi_ = ((tbase + (R_ >> p.ckb_shift)) * p.Kb + (R_ & (p.CKb - 1))) * p.Mpad + m0 + (w_ & (BM - 1));
          voffv[i] = wok ? wi_ * 16 : (int)OOB;
        }
      }
    }
    const int ibytes = p.Kb * p.IH * p.IW * 16, wbytes = p.T * p.Kb * p.Mpad * 16;
    const int so_i = p.CKb * p.IH * p.IW * 16, so_w = p.CKb * p.Mpad * 16;
    const auto rs_i = __builtin_amdgcn_make_buffer_rsrc((void*)inb, (short)0, ibytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, (short)0, wbytes, 0x00020000);
#if defined(__HIP_DEVICE_COMPILE__)  // LDS address space and the DMA builtin exist in the device pass only
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define DMA_ISSUE(C)                                                                                                   \
  {                                                                                                                    \
    u32x4* lb_ = smem4 + ((C) & 1) * p.bufs + wave * 64;                                                               \
    _Pragma("unroll") for (int i = 0; i < PF; ++i) {                                                                   \
      if (i < ni) { if (!BF_DBG(8)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_i, (lds_ptr_t)(lb_ + i * NT), 16, voffv[i], (C) * so_i, 0, 0); } \
      else if (i < ns) { if (!BF_DBG(16)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(lb_ + i * NT), 16, voffv[i], (C) * so_w, 0, 0); } \
    }                                                                                                                  \
  }
// rowdma: the rows of chunk C's input tile (CKb channel blocks x the rows_in rows this tile reads x lwp / 64 segments) dealt
// round-robin to the wavefronts, then this wavefront's weight pieces.  Rows outside the image are zero-filled (all lanes out of range).
#define DMA_ISSUE_ROWS(C)                                                                                              \
  {                                                                                                                    \
    const int nseg_ = p.lwp >> 6, npr_ = rows_in * nseg_, rowb_ = p.IW * 16;                                           \
    u32x4* lbase_ = smem4 + ((C) & 1) * p.bufs;                                                                        \
    for (int kb_ = 0; kb_ < p.CKb; ++kb_) {                                                                            \
      for (int rs_ = wave; rs_ < npr_; rs_ += NWV) {                                                                   \
        const int r_ = nseg_ == 2 ? rs_ >> 1 : rs_, sg_ = nseg_ == 2 ? rs_ & 1 : 0;                                    \
        const int iy_ = iy0 + r_;                                                                                      \
        const bool rok_ = iy_ >= 0 && iy_ < p.IH;                                                                      \
        const int so_ = rok_ ? (C) * so_i + (kb_ * p.IH + iy_) * rowb_ : 0;                                            \
        const int vo_ = rok_ ? (sg_ ? vcol1 : vcol0) : (int)OOB;                                                       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_i, (lds_ptr_t)(lbase_ + (kb_ * rows_in + r_) * p.lwp + sg_ * 64), 16, vo_, so_, 0, 0); \
      }                                                                                                                \
    }                                                                                                                  \
    const int npw_ = wtotal >> 6;                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                    \
      if (wave + NWV * j < npw_)                                                                                       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(lbase_ + p.ldsw_off + (wave + NWV * j) * 64), 16, voffv[j], (C) * so_w, 0, 0); \
    }                                                                                                                  \
  }
// one slot of chunk C, chosen at run time: issued between the MFMA clusters of the previous chunk, so the DMA instructions
// never hold up the matrix cores.  The slot's offset comes out of the register array by relative indexing (voffv[I]), the
// descriptor / scalar offset / LDS address are scalar selects -- no branches.
#define DMA_ONE(I, C)                                                                                                  \
  {                                                                                                                    \
    const int i_ = (I);                                                                                                \
    const bool isw_ = i_ >= ni;                                                                                        \
    const auto rs_ = __builtin_amdgcn_make_buffer_rsrc(isw_ ? (void*)p.wp : (void*)inb, (short)0, isw_ ? wbytes : ibytes, 0x00020000); \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(smem4 + ((C) & 1) * p.bufs + wave * 64 + i_ * NT), 16, voffv[i_],   \
                                             (C) * (isw_ ? so_w : so_i), 0, 0);                                        \
  }
#else
#define DMA_ISSUE(C) (void)ns, (void)so_i, (void)so_w, (void)rs_i, (void)rs_w, (void)voffv;
#define DMA_ISSUE_ROWS(C) (void)vcol0, (void)vcol1;
#define DMA_ONE(I, C) (void)(I), (void)voffv;
#endif
    if constexpr (PP && NWV == 8) {
      // ---- ping-pong pipeline (see the kernel comment).  The launcher selects it only for 3x3 taps with 16-channel chunks
      //      (CKb = 2: one step per tap, 9 steps = 3 phases of 3 steps per chunk) through the double-buffered DMA pipeline.
      //      A wavefront issues one instruction every ~4-5 cycles whatever its kind, so the fetch phase has to be SHORT in
      //      instructions to fit beside the partner's 24 MFMAs (768 cycles): every step-dependent part of an operand address is a
      //      compile-time immediate of the ds_read (weights: tap * 4 KB + channel block * 512 B; input: kernel column * 16 B off
      //      a per-kernel-row base), the bases move once per chunk, and the DMA slots of a phase are fixed (first half of the
      //      slots in phase 0, second half in phase 1, none in phase 2 -- their data is needed one phase later).
      static_assert(NWV == 8 && !S2D, "ping-pong needs two wavefronts per SIMD");
      constexpr int PS = 3;
      const int grp = wave >> 2;                 // 0: waves 0-3 (lead), 1: waves 4-7 (one phase behind) -- scalar
      const i32x4 rsi = bf_make_rsrc(inb, ibytes), rsw = bf_make_rsrc(p.wp, wbytes);
      const unsigned bufbytes = (unsigned)p.bufs * 16u;
      const unsigned a_b0 = (unsigned)(p.ldsw_off + a_vu) * 16u;
      unsigned b_b0[NW][3];
#pragma unroll
      for (int n = 0; n < NW; ++n)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) b_b0[n][ky] = (unsigned)(b_vu[n] + ky * lw) * 16u;
      static_assert(PF == 10, "the ping-pong slot layout is 5 input + 5 weight slots");
      const unsigned la0 = (unsigned)(wave * 64 * 16);
      const unsigned char* lds_b = reinterpret_cast<const unsigned char*>(smem4);
// slots 0-4: input tile (descriptor of this image), slots 5-9: weight slices; slots that hold nothing carry out-of-range offsets
#define PP_DMA_IN(LA, SOI) if (!BF_DBG(4)) bf_dma16x5<NT * 16>(rsi, (LA), (SOI), voffv[0], voffv[1], voffv[2], voffv[3], voffv[4]);
#define PP_DMA_W(LA, SOW) if (!BF_DBG(4)) bf_dma16x5<NT * 16>(rsw, (LA) + (unsigned)(5 * NT * 16), (SOW), voffv[5], voffv[6], voffv[7], voffv[8], voffv[9]);
#define PP_BARRIER()                              \
  __builtin_amdgcn_sched_barrier(0);              \
  __builtin_amdgcn_s_barrier();                   \
  __builtin_amdgcn_sched_barrier(0);
      [[maybe_unused]] const unsigned long long tp_dec = BF_STAMP();   // (diagnostic: slots decoded, per-lane bases formed)
      PP_DMA_IN(la0, 0)   // (the weight slices of chunk 0 went out in front of the slot decode above)
      ACC_ZERO()
      [[maybe_unused]] const unsigned long long tp_iss = BF_STAMP();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      [[maybe_unused]] const unsigned long long tp_land = BF_STAMP();
      PP_BARRIER()                 // chunk 0 has landed for everybody
#ifdef YOGO_DIAG
      if (p.stamps && tid == 0) {   // [12..15] of a stamp row: decoded, requested + zeroed, landed, barrier passed
        unsigned long long* d = p.stamps + (size_t)widx * 16;
        d[12] = tp_dec; d[13] = tp_iss; d[14] = tp_land; d[15] = __builtin_amdgcn_s_memtime();
      }
#endif
      if (grp) { PP_BARRIER() }    // the trailing half starts one phase later
      u32x4 av[PS][MW], bv[PS][NW];
#ifdef YOGO_DIAG_PHASES
      unsigned long long ph_sum[4] = {0ull, 0ull, 0ull, 0ull}, ph_t = __builtin_amdgcn_s_memtime();
#define PP_STAMP(K)                                                         \
  if (p.stamps) {                                                           \
    __builtin_amdgcn_sched_barrier(0);                                      \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();             \
    ph_sum[K] += t_ - ph_t;                                                 \
    ph_t = t_;                                                              \
    __builtin_amdgcn_sched_barrier(0);                                      \
  }
#else
#define PP_STAMP(K)
#endif
      for (int c = 0; c < p.nchunk; ++c) {
        const unsigned par = (c & 1) ? bufbytes : 0u;
        const unsigned char* pa = lds_b + (a_b0 + par);
        const unsigned char* pb[NW][3];
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) pb[n][ky] = lds_b + (b_b0[n][ky] + par);
        const bool more = c + 1 < p.nchunk;                           // (nothing is fetched behind the last chunk)
        const unsigned lan = la0 + ((c & 1) ? 0u : bufbytes);         // this wavefront's LDS base in the OTHER buffer
        const int soi = (c + 1) * so_i, sow = (c + 1) * so_w;
#pragma unroll
        for (int q = 0; q < 3; ++q) {   // phase q = kernel row q: steps (q, 0..2)
          // ---- fetch phase.  Buffer (c + 1) & 1 is free: its last reader (the trailing half's fetch of chunk c - 1) finished
          //      before the barrier that opened this chunk.
#pragma unroll
          for (int j = 0; j < PS; ++j) {
            if (!BF_DBG(64)) {
#pragma unroll
              for (int mb = 0; mb < MW; ++mb)
                av[j][mb] = *reinterpret_cast<const u32x4*>(pa + ((q * 3 + j) * 2 * BM + mb * 32) * 16);
            }
            if (!BF_DBG(32)) {
#pragma unroll
              for (int n = 0; n < NW; ++n) bv[j][n] = *reinterpret_cast<const u32x4*>(pb[n][q] + j * 16);
            }
          }
          if (q < 2 && more) {
            __builtin_amdgcn_sched_barrier(0);
            if (q == 0) { PP_DMA_IN(lan, soi) } else { PP_DMA_W(lan, sow) }
          }
          if (q == 2 && grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (trailing half) my part of chunk c + 1 has landed
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          PP_STAMP(0)
          PP_BARRIER()
          PP_STAMP(1)
          // ---- MFMA phase: registers only
#pragma unroll
          for (int j = 0; j < PS; ++j) BF_MFMA(0, av[j], bv[j])
          if (q == 2 && !grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (leading half) ... before the barrier in front of its first fetch of chunk c + 1
          PP_STAMP(2)
          PP_BARRIER()
          PP_STAMP(3)
        }
      }
      if (!grp) { PP_BARRIER() }   // same number of barriers for both halves
#ifdef YOGO_DIAG_PHASES
      if (p.stamps && (tid == 0 || tid == 256)) {
        unsigned long long* d = p.stamps + (size_t)widx * 16 + 4 + (tid >> 8) * 4;
        d[0] = ph_sum[0]; d[1] = ph_sum[1]; d[2] = ph_sum[2]; d[3] = ph_sum[3];
      }
#endif
#undef PP_STAMP
#undef PP_DMA_IN
#undef PP_DMA_W
#undef PP_BARRIER
    } else {
    bool ring_done = false;
    if constexpr (S2D && NWV == 8 && MW == 4 && NW == 1) {
      // ---- stride-2 data gradient of the 128-channel tile: 16-channel chunks (one step per tap) in a RING of four LDS buffers,
      //      three chunks in flight.  Measured on the two-buffer loop below: ~55 instructions per step around 4 MFMAs make it
      //      issue-bound (28 k ticks per tile, the same with the DMA switched off), and once the steps are lean a chunk's
      //      LDS-DMA round trip (~5 k ticks) is what a chunk costs -- with one chunk in flight.  Here: the step loop is unrolled
      //      for the 3 (py = 0) or 6 (py = 1) taps of the workgroup's row parity, every step-dependent address part is an
      //      immediate (weights: tap * 4 KB + channel block * 512 B; gradient tile: column * 16 B off one base per tile row), a
      //      chunk's five DMA pieces (2 + 3 fixed slots) are one asm statement, and chunk c + 3 is requested when chunk c - 1
      //      has been consumed: counted waits (vector memory operations retire in order), one barrier per chunk.
      if (p.ring) {
        ring_done = true;
        constexpr int NB = 4, D = 3;  // buffers, chunks in flight (5 DMA pieces per chunk and wavefront: the counted waits below)
        const i32x4 rsi = bf_make_rsrc(inb, ibytes), rsw = bf_make_rsrc(p.wp, wbytes);
        const unsigned bufbytes = (unsigned)p.bufs * 16u;
        const unsigned la0 = (unsigned)(wave * 64 * 16);
        const unsigned char* lds_b = reinterpret_cast<const unsigned char*>(smem4);
        const unsigned a_b0 = (unsigned)(p.ldsw_off + a_vu) * 16u;
        const unsigned b_b0[2] = {(unsigned)b_vu[0] * 16u, (unsigned)(lw + b_vu[0]) * 16u};  // tile row of the tap
#define RING_DMA(C) if (!BF_DBG(4)) bf_dma16_2p3<NT * 16>(rsi, rsw, la0 + (unsigned)((C) & (NB - 1)) * bufbytes, (C) * so_i, (C) * so_w, \
                                                         voffv[0], voffv[1], voffv[2], voffv[3], voffv[4]);
#define RING_BARRIER()                            \
  __builtin_amdgcn_sched_barrier(0);              \
  __builtin_amdgcn_s_barrier();                   \
  __builtin_amdgcn_sched_barrier(0);
        const int nck = p.nchunk;
        RING_DMA(0)
        if (nck > 1) { RING_DMA(1) }
        if (nck > 2) { RING_DMA(2) }
        ACC_ZERO()
        // (one copy of the chunk loop per row parity: a branch inside it would merge 128 accumulator registers per chunk)
        auto chunks = [&](auto py_tag) {
          constexpr int PY = decltype(py_tag)::value;
          constexpr int NTAP = 3 + 3 * PY, N0 = 1 + PY;
          for (int c = 0; c < nck; ++c) {
            // chunk c has landed when at most the requests behind it are outstanding: min(D - 1, nck - 1 - c) chunks of 5 pieces
            const int behind = min(D - 1, nck - 1 - c);
            if (behind >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (behind == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            RING_BARRIER()   // ... for everybody; and everybody has consumed chunk c - 1, whose buffer chunk c + 3 may now take
            if (c + D < nck) { RING_DMA(c + D) }
            const unsigned par = (unsigned)(c & (NB - 1)) * bufbytes;
            const unsigned char* pa = lds_b + (a_b0 + par);
            const unsigned char* pb[2] = {lds_b + (b_b0[0] + par), lds_b + (b_b0[1] + par)};
            u32x4 av0[MW], av1[MW], bv0, bv1;
            auto fetch = [&](auto t_tag, u32x4 (&av)[MW], u32x4& bv) {
              constexpr int t_ = decltype(t_tag)::value, ti_ = 3 * PY + t_;
              constexpr int row_ = (0x190 >> ti_) & 1, col_ = (0x144 >> ti_) & 1;
#pragma unroll
              for (int mb = 0; mb < MW; ++mb) av[mb] = *reinterpret_cast<const u32x4*>(pa + (t_ * 2 * BM + mb * 32) * 16);
              bv = *reinterpret_cast<const u32x4*>(pb[row_] + col_ * 16);
            };
            fetch(std::integral_constant<int, 0>{}, av0, bv0);
            auto step = [&](auto t_tag) {
              constexpr int t_ = decltype(t_tag)::value;
              constexpr int cls = t_ < N0 ? 0 : 1;
              if constexpr (t_ + 1 < NTAP) {
                if constexpr (t_ & 1) fetch(std::integral_constant<int, t_ + 1>{}, av0, bv0);
                else fetch(std::integral_constant<int, t_ + 1>{}, av1, bv1);
              }
              if (!BF_DBG(2)) {
#pragma unroll
                for (int mb = 0; mb < MW; ++mb)
                  acc[cls][mb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (t_ & 1) ? av1[mb] : av0[mb]),
                                                                           __builtin_bit_cast(bf16x8, (t_ & 1) ? bv1 : bv0), acc[cls][mb][0], 0, 0, 0);
              }
              // one operand read of the next step behind each MFMA (two behind the first)
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_barrier(0);
            };
            __builtin_amdgcn_sched_barrier(0);
            bf_static_for(step, std::make_integer_sequence<int, NTAP>{});
          }
        };
        if (py) chunks(std::integral_constant<int, 1>{});
        else chunks(std::integral_constant<int, 0>{});
        RING_BARRIER()   // (the epilogue's scratch overlays the buffers)
#undef RING_DMA
#undef RING_BARRIER
      }
    }
    if (!ring_done) {
    int dnext = 0, dend = 0, dchunk = 0;
#undef BF_HOOK
#define BF_HOOK()                   \
  if (dnext < dend) {               \
    DMA_ONE(dnext, dchunk)          \
    ++dnext;                        \
    if (dnext < dend) {             \
      DMA_ONE(dnext, dchunk)        \
      ++dnext;                      \
    }                               \
  }
    [[maybe_unused]] const unsigned long long t_dec = BF_STAMP();   // (diagnostic: the slot offsets are decoded)
    if (rowdma) DMA_ISSUE_ROWS(0) else DMA_ISSUE(0)
    ACC_ZERO()
    [[maybe_unused]] const unsigned long long t_iss = BF_STAMP();   // (... chunk 0 is requested, the accumulators are zero)
    [[maybe_unused]] unsigned long long t_land = 0, t_c1 = 0;
    constexpr bool LEAN4 = PP && NWV == 4;   // (the kernel's PP flag selects the unrolled step loop for the 4-wavefront tiles)
    for (int c = 0; c < p.nchunk; ++c) {
      // dma = 2 (at most two chunks): ONE buffer, the next chunk is fetched after the MFMAs -- half the LDS, so more
      // workgroups per CU cover each other's latencies, which matters more than overlap inside a workgroup that short
      if (p.dma == 2 && c > 0) {
        __syncthreads();
        if (rowdma) DMA_ISSUE_ROWS(c) else DMA_ISSUE(c)
      }
      __syncthreads();  // chunk c has landed (every wave drained its DMA) and nobody reads the other buffer any more
      if (c == 0) t_land = BF_STAMP();
      if (c == 1) t_c1 = BF_STAMP();
      // two buffers: chunk c + 1 is requested slot by slot between the MFMA clusters of chunk c
      dnext = 0;
      dchunk = c + 1;
      dend = (p.dma == 1 && c + 1 < p.nchunk && !BF_DBG(4)) ? ns : 0;
      const u32x4* ldsI = smem4 + (c & 1) * p.bufs;
      const u32x4* ldsW = ldsI + p.ldsw_off;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (LEAN4) {
        // ---- single-buffer 3x3 tiles with 16-channel chunks (the 16/32-channel layers: one or two chunks, four workgroups per
        //      CU): the 9 steps UNROLLED, operand addresses = one base per kernel row + immediates.  The generic step loop
        //      spends ~55 instructions per step (tap / channel arithmetic, v_readlane tap offsets, DMA slot hooks); with 16
        //      wavefronts per CU these kernels issue more than one instruction per SIMD cycle (profiles/r02_mfma_util.txt:
        //      "active" 0.29-0.36 per wavefront x 4) -- they are bound by instruction issue, not by HBM.
        {
          if (!BF_DBG(2)) {
            const unsigned char* pa_ = reinterpret_cast<const unsigned char*>(ldsW + a_vu);
            const unsigned char* pi_ = reinterpret_cast<const unsigned char*>(ldsI);
            // (the row pitch is laundered per chunk: as a loop invariant the 3 x NW per-row bases would be hoisted into registers
            //  that the occupancy of these tiles does not have; a scalar add per base at its use is free)
            int lw_ = lw;
            asm volatile("" : "+s"(lw_));
            if constexpr (MW == 1) {  // (one operand set; the other workgroups of the CU cover the LDS latency)
#pragma unroll
              for (int ky = 0; ky < 3; ++ky) {
                const unsigned char* pb_[NW];
#pragma unroll
                for (int n = 0; n < NW; ++n) pb_[n] = pi_ + ky * lw_ * 16 + b_vu[n] * 16;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                  u32x4 a1_[MW], b1_[NW];
#pragma unroll
                  for (int mb = 0; mb < MW; ++mb) a1_[mb] = *reinterpret_cast<const u32x4*>(pa_ + (((ky * 3 + kx) * 2) * BM + mb * 32) * 16);
#pragma unroll
                  for (int n = 0; n < NW; ++n) b1_[n] = *reinterpret_cast<const u32x4*>(pb_[n] + kx * 16);
                  BF_MFMA(0, a1_, b1_)
                  __builtin_amdgcn_sched_barrier(0);  // (or every step's reads are hoisted to the top: 180 registers)
                }
              }
            } else {  // two operand sets: the reads of step s + 1 go out before the MFMAs of step s
              u32x4 a2_[2][MW], b2_[2][NW];
#pragma unroll
              for (int mb = 0; mb < MW; ++mb) a2_[0][mb] = *reinterpret_cast<const u32x4*>(pa_ + (mb * 32) * 16);
#pragma unroll
              for (int n = 0; n < NW; ++n) b2_[0][n] = *reinterpret_cast<const u32x4*>(pi_ + b_vu[n] * 16);
#pragma unroll
              for (int s9 = 0; s9 < 9; ++s9) {
                __builtin_amdgcn_sched_barrier(0);
                if (s9 + 1 < 9) {
                  const int ky = (s9 + 1) / 3, kx = (s9 + 1) % 3;
#pragma unroll
                  for (int mb = 0; mb < MW; ++mb) a2_[(s9 + 1) & 1][mb] = *reinterpret_cast<const u32x4*>(pa_ + (((s9 + 1) * 2) * BM + mb * 32) * 16);
#pragma unroll
                  for (int n = 0; n < NW; ++n) b2_[(s9 + 1) & 1][n] = *reinterpret_cast<const u32x4*>(pi_ + ky * lw_ * 16 + b_vu[n] * 16 + kx * 16);
                }
                BF_MFMA(0, a2_[s9 & 1], b2_[s9 & 1])
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
        }
      } else {
        if (!BF_DBG(2)) BF_COMPUTE()
      }
      while (dnext < dend) {  // fewer MFMA clusters than slots: the rest goes out now
        DMA_ONE(dnext, dchunk)
        ++dnext;
      }
    }
#ifdef YOGO_DIAG
    if (p.stamps && tid == 0) {   // [4..7] of a stamp row, single- / two-buffer loop: offsets decoded, chunk 0 requested, landed, chunk 1 landed
      unsigned long long* d = p.stamps + (size_t)widx * 16;
      d[4] = t_dec; d[5] = t_iss; d[6] = t_land; d[7] = t_c1;
    }
#endif
    }  // two-buffer loop
    }  // !PP
#undef DMA_ISSUE
#undef DMA_ONE
#undef BF_HOOK
#define BF_HOOK()
  } else {
    ACC_ZERO()
    for (int c = 0; c < p.nchunk; ++c) {
      const int kb0 = c * p.CKb;
      const u32x4* ldsI = smem4;
      const u32x4* ldsW = smem4 + p.ldsw_off;
      __syncthreads();
      for (int e0 = tid; e0 < total; e0 += NT * 4) {
        u32x4 l0, l1, l2, l3;
        ST_LOAD(l0, e0, kb0) ST_LOAD(l1, e0 + NT, kb0) ST_LOAD(l2, e0 + 2 * NT, kb0) ST_LOAD(l3, e0 + 3 * NT, kb0)
        __builtin_amdgcn_sched_barrier(0);
        ST_STORE(l0, e0) ST_STORE(l1, e0 + NT) ST_STORE(l2, e0 + 2 * NT) ST_STORE(l3, e0 + 3 * NT)
      }
      __syncthreads();
      BF_COMPUTE()
    }
  }
#undef ST_DECODE
#undef ST_LOAD
#undef ST_STORE
#undef LAUNDER_TID
#undef BF_LOAD
#undef BF_MFMA
#undef BF_RUN
#undef BF_INTERLEAVE
#undef BF_COMPUTE
#undef ACC_ZERO
#undef BF_HOOK

  [[maybe_unused]] const unsigned long long t_epi = BF_STAMP();
  // ---- epilogue: bias (+ BatchNorm partial sums of the fp32 pre-activation) + activation [or act'(ref)] + channel mask,
  //      then bf16 NCHW8c or fp32 NCHW.  The per-channel bias / scale of the BM channels go through LDS (fetched before the
  //      main loop); bf16 output: the two half-waves exchange one 8-byte group (v_permlane32_swap) so that every lane stores
  //      a whole 16-byte unit -- lanes 0-31 channel block cb, lanes 32-63 block cb + 1 -- through a buffer descriptor whose
  //      range check drops the tail pixels (no per-store address arithmetic, no branches).
  const size_t plane = (size_t)p.OH * p.OW;
  const bool do_stats = !S2D && p.stats_part != nullptr;
  float* eb = reinterpret_cast<float*>(smem4);  // [BM] bias
  float* es = eb + BM;                          // [BM] channel scale (0 for padding channels)
  float* red = es + BM;                         // [NWV waves][BM][2]
  // sign-map reference: this lane's bytes of a pixel (one per 16 channels of the tile) are contiguous -- one load per pixel,
  // requested before the LDS hand-over so that the barriers hide its latency
  constexpr int SW = MW == 4 ? 2 : 1;  // dwords holding the 2 * MW sign bytes of a pixel
  const int sq = p.Mpad >> 4;          // sign bytes per (pixel, half-wave)
  // 8-wavefront tiles take the request to write a sign map at run time (their registers do not decide the occupancy), so the
  // forward kernels of the 128-channel layers stay ONE kernel; the 4-wavefront tiles have a variant (REF = 3) for it
  constexpr bool SIGN_OUT = REF == 3 || (REF == 0 && NWV == 8 && !OUT_F32);
  const bool write_signs = (REF == 3 || (SIGN_OUT && p.signs != nullptr)) && !BF_DBG(512);  // uniform  (diagnostic bit 512: no sign map)
  unsigned sg[(REF == 2 || SIGN_OUT) ? NC : 1][NW][SW];
  if constexpr (REF == 2) {
    const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)(p.signs + (size_t)b * plane * 2 * sq), (short)0, (int)plane * 2 * sq, 0x00020000);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int n = 0; n < NW; ++n) {
        const bool valid = c == 0 ? pvalid[n] : pvalid1[n];
        const int vs = valid ? (half * (int)plane + opix[n] + c) * sq + (m0 >> 4) : (int)0x80000000u;
        if constexpr (MW == 4) {
          const u32x2 t = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_s, vs, 0, 0));
          sg[c][n][0] = t.x;
          sg[c][n][SW - 1] = t.y;
        } else if constexpr (MW == 2) {
          sg[c][n][0] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_s, vs, 0, 0);
        } else {
          sg[c][n][0] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs_s, vs, 0, 0);
        }
      }
  } else if constexpr (SIGN_OUT) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int n = 0; n < NW; ++n)
#pragma unroll
        for (int k = 0; k < SW; ++k) sg[c][n][k] = 0u;
  }
  __syncthreads();  // the staged tiles are dead
  if (tid < BM) {
    eb[tid] = bias_reg;
    es[tid] = scale_reg;
  }
  __syncthreads();
  if constexpr (OUT_F32) {
#pragma unroll
    for (int mb = 0; mb < MW; ++mb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cl = mb * 32 + 8 * g + 4 * half;
        const int cbase = m0 + cl;
        const float4 bs = *reinterpret_cast<const float4*>(eb + cl), cs = *reinterpret_cast<const float4*>(es + cl);
        const float bsa[4] = {bs.x, bs.y, bs.z, bs.w}, csa[4] = {cs.x, cs.y, cs.z, cs.w};
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int n = 0; n < NW; ++n) {
            const bool valid = (c == 0 ? pvalid[n] : pvalid1[n]) && !BF_DBG(1);
            if (valid) {
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (cbase + i < p.M)
                  p.out_f32[((size_t)b * p.M + cbase + i) * plane + opix[n] + c] = act_fwd(acc[c][mb][n][4 * g + i] + bsa[i], p.act) * csa[i];
            }
          }
      }
    }
  } else {
    const bool full_tile = p1 - p0 == PT;  // uniform
    constexpr bool has_ref = REF == 1 || REF == 2;  // act'(ref) epilogue (training data gradient into a block without BatchNorm)
    constexpr bool sign_ref = REF == 2;  // ... with the LeakyReLU sign map in place of the bf16 reference
    const int plane16 = (int)plane * 16;
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)b * p.Mb * plane * 2), (short)0, p.Mb * plane16, 0x00020000);
    // cache policy of the bf16 output stores: non-temporal for the sign-writing forward of the 4-wavefront tiles (layer 1 forward: 1.8 GB out per
    // 0.8 GB in, -5.3 % in the same-box A/B gpurun_out/r5_nt_ab3.log); everywhere else it made no difference or lost (head data gradient +43 %)
#ifndef BF_NT_L1
#define BF_NT_L1 1   // (A/B variant builds: 0 = cached stores everywhere)
#endif
    constexpr int BF_ST_AUX = (BF_NT_L1 && REF == 3 && NWV == 4 && !S2D) ? 2 : 0;
#ifndef BF_S2D_HALF_DENSE
#define BF_S2D_HALF_DENSE 0   // (A/B variant builds: 1 = the stride-2 data gradient's units go out as they are, one half-filled store per column parity)
#endif
    // stride-2 data gradient: byte offset of the pixel this lane stores after the second exchange (lanes 0-31 the even, 32-63 the odd column)
    [[maybe_unused]] int vd[NW];
    if constexpr (S2D) {
#pragma unroll
      for (int n = 0; n < NW; ++n) vd[n] = ((half == 0 ? pvalid[n] : pvalid1[n]) && !BF_DBG(1)) ? (opix[n] + half) * 16 : (int)0x80000000u;
    }
    int vo[NC][NW];  // byte offset of this lane's unit inside its image: lanes 32-63 write the next channel block
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int n = 0; n < NW; ++n) {
        const bool valid = (c == 0 ? pvalid[n] : pvalid1[n]) && !BF_DBG(1);
        vo[c][n] = valid ? (opix[n] + c) * 16 + half * plane16 : (int)0x80000000u;
      }
    // training dgrad into a block without BatchNorm: act'(ref).  All of this lane's reference values (8 bytes per group) are
    // requested up front so that their latency is paid once, behind the LDS hand-over above.
    // (32-row tiles fetch them per channel-block pair instead: half the registers keep four workgroups on a CU)
    constexpr bool RF_PER_GP = MW == 1;
    u32x2 rf[REF == 1 ? MW : 1][RF_PER_GP ? 1 : 2][NC][NW][2];
    if constexpr (REF == 1 && !RF_PER_GP) {
      const auto rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.act_ref + (size_t)b * p.Mb * plane * 2), (short)0, p.Mb * plane16, 0x00020000);
#pragma unroll
      for (int mb = 0; mb < MW; ++mb)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int cb = (m0 >> 3) + mb * 4 + 2 * gp;
          if (cb >= p.Mb) continue;
#pragma unroll
          for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int n = 0; n < NW; ++n) {
              const bool valid = c == 0 ? pvalid[n] : pvalid1[n];
              const int vr = valid ? (opix[n] + c) * 16 + half * 8 + cb * plane16 : (int)0x80000000u;
              rf[REF == 1 ? mb : 0][RF_PER_GP ? 0 : gp][c][n][0] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_r, vr, 0, 0));
              rf[REF == 1 ? mb : 0][RF_PER_GP ? 0 : gp][c][n][1] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_r, vr + plane16, 0, 0));
            }
        }
    }
    // The group loop exists twice: the GENERAL order of operations (BatchNorm partial sums, SiLU, a second pre-activation output,
    // a non-LeakyReLU reference) and the lean one every launch of the training step takes.  One uniform test up here instead of
    // one per unrolled block: merged, the blocks carry each other's live values (16 statistics registers zeroed per group,
    // moves at every join) and the epilogue -- two wavefronts per SIMD, bound by instruction issue -- is a fifth of a tile's time.
    [[maybe_unused]] const bool leaky = p.act == ACT_LEAKY;
    [[maybe_unused]] const bool general = do_stats || p.act == ACT_SILU || p.out_pre != nullptr || (has_ref && p.ref_act != ACT_LEAKY);
    // (the 4-wavefront tiles and the bf16-reference variant keep ONE merged loop with the test inside: split, they need more
    //  registers than their occupancy allows)
    constexpr bool SPLIT = NWV == 8 && REF != 1;
    if (BF_DBG(256)) {   // (diagnostic: no epilogue arithmetic and no stores -- what the group loop costs altogether)
    } else
    if constexpr (LEPI) {
      constexpr int MODE = 0;
#include "conv_bf16_epi_groups.inc"
    } else
    if constexpr (SPLIT) {
      auto epi_groups = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
#include "conv_bf16_epi_groups.inc"
      };
      if (general) epi_groups(std::integral_constant<int, 1>{});
      else epi_groups(std::integral_constant<int, 0>{});
    } else {
      constexpr int MODE = 2;
#include "conv_bf16_epi_groups.inc"
    }
    if constexpr (SIGN_OUT) if (write_signs) {  // the sign bytes of a pixel go out together
      const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)(p.signs + (size_t)b * plane * 2 * sq), (short)0, (int)plane * 2 * sq, 0x00020000);
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int n = 0; n < NW; ++n) {
          const int vs = vo[c][n] < 0 ? (int)0x80000000u : (vo[c][n] >> 4) * sq + (m0 >> 4);
          if constexpr (MW == 4) {
            const u32x2 t = {sg[c][n][0], sg[c][n][SW - 1]};
            __builtin_amdgcn_raw_buffer_store_b64(t, rs_s, vs, 0, 0);
          } else if constexpr (MW == 2) {
            __builtin_amdgcn_raw_buffer_store_b32(sg[c][n][0], rs_s, vs, 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)sg[c][n][0], rs_s, vs, 0, 0);
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
#ifdef YOGO_DIAG
  if (p.stamps && tid == 0) {
    unsigned long long* d = p.stamps + (size_t)widx * 16;
    d[0] = t_start; d[1] = t_loop; d[2] = t_epi; d[3] = __builtin_amdgcn_s_memtime();
  }
#endif
}

// ---- weight packing: OIHW fp32 (x optional per-output-channel scale = folded BatchNorm) -> [T][Kb][Mpad] units ----------
// dgrad = 1: GEMM roles swapped (k = co, m = ci) and the kernel flipped, so the data gradient is a plain stride-1 conv
// dgrad = 2: as 1, slices stored in the parity-class order of the stride-2 data gradient (see conv_bf16_kernel, S2D):
//            row parity 0 -> taps (1,1) | (1,0) (1,2); row parity 1 -> taps (0,1) (2,1) | (0,0) (0,2) (2,0) (2,2)
__global__ void conv_bf16_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, u32x4* __restrict__ wp,
                                      int Cin, int Cout, int ks, int Kb, int Mpad, int dgrad) {
  const int T = ks * ks;
  const int total = T * Kb * Mpad;
  const int Kc = dgrad ? Cout : Cin, Mc = dgrad ? Cin : Cout;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int m = e % Mpad;
    const int kb = (e / Mpad) % Kb;
    const int t = e / (Mpad * Kb);
    const int ts = dgrad == 2 ? (int)((0x862071534ull >> (4 * t)) & 15ull) : t;
    const int tt = dgrad ? (T - 1 - ts) : ts;  // flipped tap
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

// all the packings of a training step in ONE launch: table[n][8] int64 = {w, scale, packed, Cin, Cout, ks, mode, first
// block}; a workgroup finds its entry by its block index and packs 256 units of it (the body of conv_bf16_pack_kernel)
__global__ void conv_bf16_pack_multi_kernel(const long long* __restrict__ table, int n) {
  int e = 0;
  for (int k = 1; k < n; ++k)
    if ((int)table[k * 8 + 7] <= (int)blockIdx.x) e = k;
  const long long* t = table + e * 8;
  const float* w = reinterpret_cast<const float*>(t[0]);
  const float* scale = reinterpret_cast<const float*>(t[1]);
  u32x4* wp = reinterpret_cast<u32x4*>(t[2]);
  const int Cin = (int)t[3], Cout = (int)t[4], ks = (int)t[5], dgrad = (int)t[6];
  const int Kc = dgrad ? Cout : Cin, Mc = dgrad ? Cin : Cout;
  const int Kb = ((Kc + 15) / 16) * 2, mw = Mc <= 32 ? 1 : (Mc <= 64 ? 2 : 4), Mpad = ((Mc + 32 * mw - 1) / (32 * mw)) * (32 * mw);
  const int T = ks * ks, total = T * Kb * Mpad;
  const int el = ((int)blockIdx.x - (int)t[7]) * 256 + (int)threadIdx.x;
  if (el >= total) return;
  const int m = el % Mpad, kb = (el / Mpad) % Kb, tp = el / (Mpad * Kb);
  const int ts = dgrad == 2 ? (int)((0x862071534ull >> (4 * tp)) & 15ull) : tp;
  const int tt = dgrad ? (T - 1 - ts) : ts;
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
  wp[el] = __builtin_bit_cast(u32x4, o);
}

namespace {

int bf_pick_mw(int M) { return M <= 32 ? 1 : (M <= 64 ? 2 : 4); }
int bf_pick_nw(int mw) { return mw == 1 ? 4 : 2; }
int bf_kb_of(int K) { return round_up(K, 16) / 8; }
int bf_mpad_of(int M) { return round_up(M, 32 * bf_pick_mw(M)); }

struct BfTiling {
  int ncb, TW, tiles_per_band, CKb, rows_max, LW, ldsw_off, lds_dummy, lds_bytes, dma, ni_slots, n_slots, bufu;
};

// (OH, OW): the grid the workgroups tile (output pixels; quads for the stride-2 data gradient), T: weight slices staged at
// most, PF: DMA slots of the kernel variant.  LDS image of a chunk: input tile from unit 0, weight slices from ldsw_off.
bool bf_plan(int OH, int OW, int a, int T, int span, int Kb, int MW, int NW, int NWV, int PF, int budget, BfTiling* out,
             bool force_single = false, int force_ckb = 0, int force_nbuf = 0, int force_ni = 0) {
  const int BM = 32 * MW, PT = 32 * NWV * NW, NT = 64 * NWV;
  BfTiling best{};
  long long best_score = -1;
  for (int ncb = 1; ncb <= 24 && ncb <= OW; ++ncb) {
    const int TW = cdiv(OW, ncb);
    const int bw_min = OW - (cdiv(OW, TW) - 1) * TW;
    if (cdiv(OW, TW) != ncb || bw_min <= 0) continue;
    const int LW = (TW - 1) * a + span;
    const int nrow_lat = min(OH, 1 + cdiv(PT - 1, bw_min));
    const int rows_max = (nrow_lat - 1) * a + span;
    const int chs = rows_max * LW;
    for (int CKb : {8, 4, 2}) {
      if (Kb % CKb || (force_ckb && CKb != force_ckb)) continue;
      const int ni_need = cdiv(CKb * chs, NT), nw = cdiv(T * CKb * BM, NT);
      if (force_ni && ni_need > force_ni) continue;
      const int ni = force_ni ? force_ni : ni_need;  // (a fixed slot layout: unused slots carry out-of-range offsets)
      const int ldsw_off = ni * NT;  // slot-aligned, so a DMA slot is all input or all weights
      const int bufu = (ni + nw) * NT;
      const int nbuf = force_nbuf ? force_nbuf : ((Kb / CKb <= 2 || force_single) ? 1 : 2);  // short contractions: one buffer (more workgroups per CU)
      const int dma = (ni + nw <= PF && nbuf * bufu * 16 <= budget) ? (nbuf == 1 ? 2 : 1) : 0;
      if (force_nbuf && !dma) continue;
      const int dummy = ldsw_off + T * CKb * BM;
      const int bytes = dma ? nbuf * bufu * 16 : (dummy + 1) * 16;
      if (bytes > budget) continue;
      // pipelined chunks first, then deep chunks, then the least staged input over the whole image (halo overhead)
      const long long staged = (long long)ncb * cdiv(OH * TW, PT) * rows_max * LW;  // units per channel block and image
      const long long score = (dma ? 1 : 0) * 100000000000000LL + (long long)CKb * 100000000000LL - staged * 100 - ncb;
      if (best_score < 0 || score > best_score) {
        best_score = score;
        best = BfTiling{ncb, TW, cdiv(OH * TW, PT), CKb, rows_max, LW, ldsw_off, dummy, bytes, dma, ni, ni + nw, bufu};
      }
    }
  }
  if (best_score < 0) return false;
  *out = best;
  return true;
}

// Row-staged tiling of the single-buffer 4-wavefront kernels (ConvBf16Params::rowdma): rows at a pitch of 64 or 128 units, so
// the band may be as wide as the pitch allows at no extra LDS -- fewer bands, less halo, fewer pieces.  16-channel chunks only.
bool bf_plan_rows(int OH, int OW, int a, int T, int span, int Kb, int MW, int NW, int NWV, int budget, BfTiling* out, int* lwp_out) {
  const int BM = 32 * MW, PT = 32 * NWV * NW, CKb = 2;
  if (Kb % CKb || Kb / CKb > 2 || T * CKb * BM / 64 > 8 * NWV) return false;
  long long best_score = -1;
  BfTiling best{};
  int best_lwp = 0;
  for (int ncb = 1; ncb <= 24 && ncb <= OW; ++ncb) {
    const int TW = cdiv(OW, ncb);
    const int bw_min = OW - (cdiv(OW, TW) - 1) * TW;
    if (cdiv(OW, TW) != ncb || bw_min <= 0) continue;
    const int LW = (TW - 1) * a + span;
    const int lwp = LW <= 64 ? 64 : ((LW <= 128 && a == 1) ? 128 : 0);   // (stride 2: one piece per row, or the piece count doubles)
    if (!lwp) continue;
    const int nrow_lat = min(OH, 1 + cdiv(PT - 1, bw_min));
    const int rows_max = (nrow_lat - 1) * a + span;
    const int units_in = CKb * rows_max * lwp, wunits = T * CKb * BM;
    const int bytes = (units_in + wunits) * 16;
    if (bytes > budget) continue;
    const long long staged = (long long)ncb * cdiv(OH * TW, PT) * rows_max * lwp;  // staged units per channel block and image
    if (best_score < 0 || staged < best_score) {
      best_score = staged;
      best = BfTiling{ncb, TW, cdiv(OH * TW, PT), CKb, rows_max, LW, units_in, units_in + wunits, bytes, 2, 0, 0, units_in + wunits};
      best_lwp = lwp;
    }
  }
  if (best_score < 0) return false;
  *out = best;
  *lwp_out = best_lwp;
  return true;
}

}  // namespace

// =========================================================================================================
// C ABI
// =========================================================================================================
// mode 0: forward (k = ci, m = co); mode 1: dgrad (k = co, m = ci, flipped taps); mode 2: dgrad of a stride-2 3x3 conv
// (as 1, slices in parity-class order)
extern "C" int yogo_conv_bf16_packed_bytes(int Cin, int Cout, int ks, int mode, size_t* bytes) {
  YOGO_CHECK_ARG(bytes && Cin > 0 && Cout > 0 && (ks == 1 || ks == 3), "conv_bf16_packed_bytes: bad arguments");
  const int K = mode ? Cout : Cin, M = mode ? Cin : Cout;
  *bytes = (size_t)ks * ks * bf_kb_of(K) * bf_mpad_of(M) * 16;
  return YOGO_OK;
}

// scale (optional, [Cout]): per-output-channel factor folded into the weights (eval-mode BatchNorm: gamma / sqrt(var + eps))
extern "C" int yogo_conv_bf16_pack(const float* w_oihw, const float* scale, void* packed, int Cin, int Cout, int ks, int mode,
                                   hipStream_t stream) {
  YOGO_CHECK_ARG(w_oihw && packed && Cin > 0 && Cout > 0 && (ks == 1 || ks == 3) && mode >= 0 && mode <= 2 && !(mode == 2 && ks != 3),
                 "conv_bf16_pack: bad arguments");
  const int K = mode ? Cout : Cin, M = mode ? Cin : Cout;
  const int Kb = bf_kb_of(K), Mpad = bf_mpad_of(M);
  const int total = ks * ks * Kb * Mpad;
  hipLaunchKernelGGL(conv_bf16_pack_kernel, dim3(min(1024, cdiv(total, 256))), dim3(256), 0, stream, w_oihw, scale,
                     reinterpret_cast<u32x4*>(packed), Cin, Cout, ks, Kb, Mpad, mode);
  YOGO_CHECK_LAUNCH("conv_bf16_pack");
  return YOGO_OK;
}

// yogo_conv_bf16_pack for a list of tensors in one launch.  table: device int64 [n][8] = {w pointer, scale pointer or 0,
// packed pointer, Cin, Cout, ks, mode, first block}; entry k owns blocks [first block k, first block k + 1) with
// yogo_conv_bf16_pack_blocks blocks each; total_blocks = the sum.
extern "C" int yogo_conv_bf16_pack_blocks(int Cin, int Cout, int ks, int mode, int* blocks) {
  YOGO_CHECK_ARG(blocks && Cin > 0 && Cout > 0 && (ks == 1 || ks == 3) && mode >= 0 && mode <= 2, "conv_bf16_pack_blocks: bad arguments");
  const int K = mode ? Cout : Cin, M = mode ? Cin : Cout;
  *blocks = cdiv(ks * ks * bf_kb_of(K) * bf_mpad_of(M), 256);
  return YOGO_OK;
}
extern "C" int yogo_conv_bf16_pack_multi(const void* table, int n, int total_blocks, hipStream_t stream) {
  YOGO_CHECK_ARG(table && n > 0 && total_blocks > 0, "conv_bf16_pack_multi: bad arguments");
  hipLaunchKernelGGL(conv_bf16_pack_multi_kernel, dim3(total_blocks), dim3(256), 0, stream, reinterpret_cast<const long long*>(table), n);
  YOGO_CHECK_LAUNCH("conv_bf16_pack_multi");
  return YOGO_OK;
}

// channel blocks of a bf16 NCHW8c tensor with C channels AS THE NEXT LAYER READS IT (padded to 16 channels = 2 blocks)
extern "C" int yogo_bf16_channel_blocks(int C) { return bf_kb_of(C); }

#ifdef BF_NO_ROWDMA   // (A/B variant builds: bash build.sh variant slot conv_bf16 -DBF_NO_ROWDMA)
static bool g_bf_rowdma = false;
#else
static bool g_bf_rowdma = true;
#endif
// (diagnostic build: yogo_diag_conv_bf16_rowdma(0) keeps the per-lane slot staging of the lean 4-wavefront tiles)
static bool g_bf_lean4 = true; // (diagnostic build: yogo_diag_conv_bf16_lean4(0) selects the generic step loop of the 4-wavefront tiles)
static bool g_bf_ring = true; // (diagnostic build: yogo_diag_conv_bf16_ring(0) selects the two-buffer loop of the stride-2 data gradient)
// the persistent wavefront-specialised kernels (conv_bf16_ws.hip, conv_bf16_ws3.hip) take the launches they are eligible for.  The
// product has NO run-time plan switch (no mutable global state in the library): the switch below exists in the test-hooks build
// (build.sh: libyogo_hip_hooks.so, -DYOGO_TEST_HOOKS; loaded by tests/ and tools/ only) and in the diagnostic build.
#if defined(YOGO_TEST_HOOKS) || defined(YOGO_DIAG)
static bool g_bf_ws = true;
// 0 = every launch goes to the tiled conv_bf16_kernel (A/B runs and the bit-identity tests of the two kernel families)
extern "C" int yogo_hook_conv_bf16_persistent(int on) { g_bf_ws = on != 0; return YOGO_OK; }
// conv_bf16_ws16_kernel (16x16x32 MFMAs) for the plain-epilogue launches: 0 = never (conv_bf16_ws_kernel<0> takes them, as until round 5),
// 1 = the product's rule (the launches WITHOUT a bias: the data gradients), 2 = every eligible launch (bias too: its tests and A/B runs)
static int g_bf_ws16 = 1;
extern "C" int yogo_hook_conv_bf16_ws16(int mode) { g_bf_ws16 = mode; return YOGO_OK; }
// the direct (weights-resident, no staging) stride-2 data gradients (conv_bf16_direct.hip)
static bool g_bf_direct = true;
extern "C" int yogo_hook_conv_bf16_direct(int on) { g_bf_direct = on != 0; return YOGO_OK; }
// the independent-wavefront kernel of the thin forward-type layers (conv_bf16_staged.hip)
static bool g_bf_staged = true;
extern "C" int yogo_hook_conv_bf16_staged(int on) { g_bf_staged = on != 0; return YOGO_OK; }
// the 1x1 head's forward with the weights in registers (conv_bf16_head.hip)
static bool g_bf_head = true;
extern "C" int yogo_hook_conv_bf16_head(int on) { g_bf_head = on != 0; return YOGO_OK; }
#else
static constexpr bool g_bf_head = true;
static constexpr bool g_bf_staged = true;
static constexpr bool g_bf_ws = true;
#ifdef BF_WS16_MODE   // (A/B variant builds: bash build.sh variant nows16 conv_bf16 -DBF_WS16_MODE=0)
static constexpr int g_bf_ws16 = BF_WS16_MODE;
#else
static constexpr int g_bf_ws16 = 1;
#endif
static constexpr bool g_bf_direct = true;
#endif
bool conv_bf16_s2d_direct_eligible(int K, int M, int OH, int OW, int B);
int launch_conv_bf16_s2d_direct(const void* in, const void* packed, void* out, const void* signs, const float* chan_scale, int B, int K, int M, int IH, int IW,
                                int OH, int OW, hipStream_t stream);
bool conv_bf16_staged_eligible(int K, int M, int stride, int IH, int IW, int OH, int OW, int B);
int launch_conv_bf16_staged(const void* in, const void* packed, const float* bias, void* out, void* signs, const float* chan_scale, int B, int K, int M, int IH,
                            int IW, int OH, int OW, int stride, int act, hipStream_t stream);
bool conv_bf16_head_fwd_eligible(int K, int M, int plane, int B);
int launch_conv_bf16_head_fwd(const void* in, const void* packed, const float* bias, float* out_f32, int B, int K, int M, int plane, hipStream_t stream);
static bool g_bf_pp = true;   // (diagnostic build: yogo_diag_conv_bf16_pp(0) selects the interleaved main loop for A/B runs)
#ifdef YOGO_DIAG
extern "C" int yogo_diag_conv_bf16_pp(int on) { g_bf_pp = on != 0; return YOGO_OK; }
extern "C" int yogo_diag_conv_bf16_ring(int on) { g_bf_ring = on != 0; return YOGO_OK; }
extern "C" int yogo_diag_conv_bf16_lean4(int on) { g_bf_lean4 = on != 0; return YOGO_OK; }
extern "C" int yogo_diag_conv_bf16_rowdma(int on) { g_bf_rowdma = on != 0; return YOGO_OK; }
// diagnostic build only: ablation bits, synchronous staging, and a caller-owned stamp buffer ([workgroups][4] u64)
static int g_diag_dbg = 0, g_diag_nodma = 0;
static unsigned long long* g_diag_stamps = nullptr;
static size_t g_diag_stamps_bytes = 0;
// stamps: [workgroups][16] u64 (see ConvBf16Params::stamps)
extern "C" int yogo_diag_conv_bf16(int dbg_bits, int no_dma, void* stamps, size_t stamps_bytes) {
  g_diag_dbg = dbg_bits; g_diag_nodma = no_dma;
  g_diag_stamps = reinterpret_cast<unsigned long long*>(stamps); g_diag_stamps_bytes = stamps_bytes;
  return YOGO_OK;
}
#endif

namespace {

template <int MW, int NW, int NWV, bool S2D, int PF, bool F32, int REF, bool PP, bool LEPI = false>
void bf_launch_one(dim3 grid, int lds_bytes, hipStream_t stream, const ConvBf16Params& p, const char* plan_txt) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<MW, NW, NWV, S2D, PF, F32, REF, PP, LEPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, BF_LDS_MAX);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_bf16_kernel<MW, NW, NWV, S2D, PF, F32, REF, PP, LEPI>), grid, dim3(64 * NWV), lds_bytes, stream, p);
  yogo_launch_log("conv_bf16_kernel<%d, %d, %d, %s, %d, %s, %d, %s, %s> | %s", MW, NW, NWV, S2D ? "true" : "false", PF, F32 ? "true" : "false", REF,
                  PP ? "true" : "false", LEPI ? "true" : "false", plan_txt);
}
// the ping-pong main loop exists for the 8-wavefront tiles with bf16 output (two wavefronts per SIMD)
template <int MW, int NW, int NWV, bool S2D, int PF, bool F32, int REF>
void bf_launch_t(dim3 grid, int lds_bytes, hipStream_t stream, const ConvBf16Params& p, bool use_pp, const char* plan_txt) {
  if constexpr (NWV == 8 && !S2D && !F32) {
    if (use_pp) {
      bf_launch_one<MW, NW, NWV, S2D, PF, F32, REF, true>(grid, lds_bytes, stream, p, plan_txt);
      return;
    }
  }
  if constexpr (NWV == 4 && !S2D && !F32) {
    if (p.lean4) {
      // the epilogue's general order of operations (kernel: `general`) is only compiled into the LEPI = false instantiation
      const bool general = p.stats_part != nullptr || p.act == ACT_SILU || p.out_pre != nullptr || ((REF == 1 || REF == 2) && p.ref_act != ACT_LEAKY);
#ifndef BF_NO_LEPI   // (A/B variant builds: bash build.sh variant nolepi conv_bf16 -DBF_NO_LEPI)
      if constexpr (REF != 1) {
        if (!general) {
          bf_launch_one<MW, NW, NWV, S2D, PF, F32, REF, true, true>(grid, lds_bytes, stream, p, plan_txt);
          return;
        }
      }
#else
      (void)general;
#endif
      bf_launch_one<MW, NW, NWV, S2D, PF, F32, REF, true>(grid, lds_bytes, stream, p, plan_txt);
      return;
    }
  }
  // the 4-wavefront stride-2 data gradient (layer 2): the lean-epilogue instantiation too (the same arithmetic in the same order, -1 % in
  // the same-box A/B, gpurun_out/r5_abs2dlepi.log)
  if constexpr (NWV == 4 && S2D && !F32 && REF != 1) {
    const bool general = p.stats_part != nullptr || p.act == ACT_SILU || p.out_pre != nullptr || (REF == 2 && p.ref_act != ACT_LEAKY);
    if (!general) {
      bf_launch_one<MW, NW, NWV, S2D, PF, F32, REF, false, true>(grid, lds_bytes, stream, p, plan_txt);
      return;
    }
  }
  bf_launch_one<MW, NW, NWV, S2D, PF, F32, REF, false>(grid, lds_bytes, stream, p, plan_txt);
}

// One launcher for forward and data-gradient.  (K, M) are the GEMM contraction / output channel counts, (IH, IW) the
// physical input dims, (OH, OW) the output dims, `a` the input step per output pixel; s2d = 1: parity-decomposed data
// gradient of a stride-2 3x3 convolution (input = dy, output = dx, weights packed with mode 2).
int launch_conv_bf16(const void* in, const void* packed, const float* bias, void* out, float* out_f32, void* out_pre, const void* act_ref,
                     int ref_act, void* signs, bool signs_read, const float* chan_scale, float* stats_part, int B, int K, int M, int IH, int IW, int OH,
                     int OW, int ks, int a, int s2d, int act, hipStream_t stream, int* stats_rows, int* stats_mpad) {
  const int T = ks * ks, pad = ks == 3 ? 1 : 0;
  // stride-1 3x3 convolutions with 128 output channels and the lean epilogue: the persistent wavefront-specialised kernel
  if (in != nullptr && g_bf_ws && !s2d && a == 1 && ks == 3 && out_f32 == nullptr && out_pre == nullptr && act_ref == nullptr && !signs_read &&
      stats_part == nullptr && (act == ACT_NONE || act == ACT_LEAKY) && !(signs != nullptr && act != ACT_LEAKY) &&
      conv_bf16_ws_eligible(K, M, IH, IW, B)) {
    ConvWsParams q{};
    q.in = in; q.wp = packed; q.bias = bias; q.out = out; q.signs = reinterpret_cast<unsigned char*>(signs); q.chan_scale = chan_scale;
    q.B = B; q.Kb = bf_kb_of(K); q.IH = IH; q.IW = IW; q.act = act;
#ifdef YOGO_DIAG
    q.dbg = g_diag_dbg;
    q.stamps = (g_diag_stamps != nullptr && g_diag_stamps_bytes >= 512 * 128) ? g_diag_stamps : nullptr;
    if (q.stamps) (void)hipMemsetAsync(g_diag_stamps, 0, 512 * 128, stream);
#endif
    if (conv_bf16_ws_plan(&q)) {
      // the plain epilogue without a bias (the data gradients of the 128 -> 128 layers): the 16x16x32 member of the family.  With a bias
      // (layer 5's forward) it measured +7 % inside the training step (profiles/r06_kernel_stats.txt history) and stays
      // on conv_bf16_ws_kernel<0>
      if ((g_bf_ws16 == 2 || (g_bf_ws16 == 1 && bias == nullptr)) && act == ACT_NONE && signs == nullptr && chan_scale == nullptr &&
          conv_bf16_ws16_eligible(K, M, IH, IW, B))
        return launch_conv_bf16_ws16(q, stream);
      return launch_conv_bf16_ws(q, stream);
    }
  }
  // stride-2 3x3 forward with 128 output channels and the lean epilogue: its persistent wavefront-specialised member (conv_bf16_ws3.hip)
  if (in != nullptr && g_bf_ws && !s2d && a == 2 && ks == 3 && out_f32 == nullptr && out_pre == nullptr && act_ref == nullptr && !signs_read &&
      stats_part == nullptr && (act == ACT_NONE || act == ACT_LEAKY) && !(signs != nullptr && act != ACT_LEAKY) && conv_bf16_ws3_eligible(K, M, IH, IW, B)) {
    ConvWs3Params q{};
    q.in = in; q.wp = packed; q.bias = bias; q.out = out; q.signs = reinterpret_cast<unsigned char*>(signs); q.chan_scale = chan_scale;
    q.B = B; q.Kb = bf_kb_of(K); q.IH = IH; q.IW = IW; q.OH = OH; q.OW = OW; q.act = act;
#ifdef YOGO_DIAG
    q.dbg = g_diag_dbg;
    q.stamps = (g_diag_stamps != nullptr && g_diag_stamps_bytes >= 512 * 128) ? g_diag_stamps : nullptr;
    if (q.stamps) (void)hipMemsetAsync(g_diag_stamps, 0, 512 * 128, stream);
#endif
    if (conv_bf16_ws3_plan(&q)) return launch_conv_bf16_ws3(q, stream);
  }
  // the 1x1 head's fp32-output forward (bias only) with the weights in registers
  if (in != nullptr && g_bf_head && ks == 1 && a == 1 && !s2d && out_pre == nullptr && act_ref == nullptr && stats_part == nullptr && act == ACT_NONE &&
      out_f32 != nullptr && out == nullptr && signs == nullptr && chan_scale == nullptr && conv_bf16_head_fwd_eligible(K, M, OH * OW, B))
    return launch_conv_bf16_head_fwd(in, packed, bias, out_f32, B, K, M, OH * OW, stream);
  // 3x3 convolutions out of 16 / 32 channels into <= 64 with the lean epilogue: independent wavefronts, weights resident, private LDS staging
  if (in != nullptr && g_bf_staged && !s2d && ks == 3 && out_f32 == nullptr && out_pre == nullptr && act_ref == nullptr && !signs_read && stats_part == nullptr &&
      (act == ACT_NONE || act == ACT_LEAKY) && !(signs != nullptr && act != ACT_LEAKY) && conv_bf16_staged_eligible(K, M, a, IH, IW, OH, OW, B))
    return launch_conv_bf16_staged(in, packed, bias, out, signs, chan_scale, B, K, M, IH, IW, OH, OW, a, act, stream);
  // stride-2 3x3 data gradient into <= 32 or 65 - 128 channels (scale / LeakyReLU-sign-map epilogue): weights resident in LDS, operands straight from memory
  if (in != nullptr && g_bf_direct && s2d && ks == 3 && out_f32 == nullptr && out_pre == nullptr && act_ref == nullptr && bias == nullptr &&
      stats_part == nullptr && act == ACT_NONE && (!signs_read || (signs != nullptr && ref_act == ACT_LEAKY)) && (signs_read || signs == nullptr) &&
      conv_bf16_s2d_direct_eligible(K, M, OH, OW, B))
    return launch_conv_bf16_s2d_direct(in, packed, out, signs_read ? signs : nullptr, chan_scale, B, K, M, IH, IW, OH, OW, stream);
  const int MW = bf_pick_mw(M);
  const bool small_n = s2d || a == 2;  // two accumulator sets / four-fold input tile: half the pixel groups per wavefront
  // 64 GEMM rows at stride 1 with a long contraction: 8 wavefronts x 4 pixel groups (64 rows x 1024 px), the same staged
  // bytes per MFMA as the 128-row tile; everything else with <= 64 rows stays on 4-wavefront workgroups
  const bool wide64 = MW == 2 && !small_n && K >= 64 && OH * OW >= 4096;
  const int NW = MW == 1 ? (small_n ? 2 : 4) : (small_n ? 1 : (wide64 ? 4 : 2));
  const int NWV = (MW == 4 || wide64) ? 8 : 4;  // 128-channel tiles: 8 wavefronts share the staged weight slice
  const int PF = NWV == 8 ? 10 : 16;  // = 160 KB / 128 KB of LDS for the two buffers at most
  const int Kb = bf_kb_of(K), Mpad = bf_mpad_of(M);
  // the grid the workgroups tile: output pixels, or 2x2 output quads of one row parity
  const int OHt = s2d ? (OH + 1) / 2 : OH, OWt = s2d ? (OW + 1) / 2 : OW;
  BfTiling tl;
  // 4-wavefront workgroups: as many per CU as a pipelined tiling allows (4, 3, 2)
  bool planned = false;
  if (NWV == 4) {
    const int ladder[3] = {40 * 1024, 53 * 1024, BF_LDS_BUDGET};
    for (int i = 0; i < 3 && !planned; ++i)
      planned = bf_plan(OHt, OWt, s2d ? 1 : a, s2d ? 6 : T, s2d ? 2 : ks, Kb, MW, NW, NWV, PF, ladder[i], &tl) && tl.dma;
  }
  // stride-2 data gradient of the 128-channel tile: 16-channel chunks in a ring of four buffers with the fixed 2 + 3 slot layout
  bool ring = false;
  if (s2d && MW == 4 && Kb >= 8 && g_bf_ring) {
    ring = bf_plan(OHt, OWt, 1, 6, 2, Kb, MW, NW, NWV, 5, BF_LDS_MAX, &tl, false, 2, 4, 2);
    planned = ring;
  }
  if (!planned) planned = bf_plan(OHt, OWt, s2d ? 1 : a, s2d ? 6 : T, s2d ? 2 : ks, Kb, MW, NW, NWV, PF, BF_LDS_MAX, &tl);
  if (!planned) {
    yogo_set_error("conv_bf16: no LDS tiling fits (K=%d M=%d OW=%d a=%d)", K, M, OW, a);
    return YOGO_ERR_ARG;
  }
  // the lean single-buffer 4-wavefront tiles (3x3, 16-channel chunks): stage the input row by row when that keeps the
  // workgroups-per-CU level of the slot plan
  int rowdma = 0, lwp = 0;
  // (measured: -2.5 ... -5 % on the 16 <-> 32 channel layer in both directions; the stride-2 forward -2 ... -3.5 % with ONE piece
  //  per row -- bands of at most 31 output columns -- and +1 % with 87-unit rows in two pieces, so its pitch stays 64)
  if (!s2d && NWV == 4 && T == 9 && tl.CKb == 2 && tl.dma == 2 && out_f32 == nullptr && g_bf_lean4 && g_bf_rowdma) {
    const int ladder[3] = {40 * 1024, 53 * 1024, BF_LDS_BUDGET};
    int level = 0;
    while (level < 2 && tl.lds_bytes > ladder[level]) ++level;
    BfTiling tr;
    if (tl.lds_bytes <= ladder[level] && bf_plan_rows(OHt, OWt, a, T, ks, Kb, MW, NW, NWV, ladder[level], &tr, &lwp)) {
      tl = tr;
      rowdma = 1;
    }
  }
  dim3 grid(tl.ncb * tl.tiles_per_band, (Mpad / (32 * MW)) * (s2d ? 2 : 1), B);
  if (stats_rows) *stats_rows = B * (int)grid.x;
  if (stats_mpad) *stats_mpad = Mpad;
  if (in == nullptr) return YOGO_OK;  // shape query only
  ConvBf16Params p{};
  p.in = reinterpret_cast<const u32x4*>(in); p.wp = reinterpret_cast<const u32x4*>(packed); p.bias = bias;
  p.out = reinterpret_cast<u32x2*>(out); p.out_f32 = out_f32; p.out_pre = reinterpret_cast<u32x2*>(out_pre);
  p.act_ref = reinterpret_cast<const u32x2*>(act_ref); p.ref_act = ref_act; p.signs = reinterpret_cast<unsigned char*>(signs); p.chan_scale = chan_scale; p.stats_part = stats_part;
  p.B = B; p.Kb = Kb; p.M = M; p.Mpad = Mpad; p.Mb = bf_kb_of(M);
  p.IH = IH; p.IW = IW; p.OH = OH; p.OW = OW; p.T = T;
  if (s2d) {
    p.a = 1; p.dy_min = 0; p.dx_min = 0; p.span_y = 2; p.span_x = 2;
  } else {
    p.a = a; p.dy_min = -pad; p.dx_min = -pad; p.span_y = ks; p.span_x = ks;
  }
  p.ncb = tl.ncb; p.TW = tl.TW; p.tiles_per_band = tl.tiles_per_band;
  {
    auto magic = [](int d) -> unsigned { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(((1ull << 32) + (unsigned)d - 1ull) / (unsigned)d); };
    const int a_ = s2d ? 1 : a, span_ = s2d ? 2 : ks;
    const int bw_last = OWt - (tl.ncb - 1) * tl.TW;
    p.m_gx = magic((int)grid.x); p.m_gy = magic((int)grid.y); p.m_gxy = magic((int)(grid.x * grid.y)); p.m_tpb = magic(tl.tiles_per_band);
    p.m_bw = magic(tl.TW); p.m_bwl = magic(bw_last);
    p.m_lw = magic((tl.TW - 1) * a_ + span_); p.m_lwl = magic((bw_last - 1) * a_ + span_);
  }
  p.CKb = tl.CKb; p.ckb_shift = tl.CKb == 8 ? 3 : (tl.CKb == 4 ? 2 : 1); p.nchunk = Kb / tl.CKb;
  p.ldsw_off = tl.ldsw_off; p.lds_dummy = tl.lds_dummy; p.act = act;
  p.dma = tl.dma; p.ni_slots = tl.ni_slots; p.n_slots = tl.n_slots; p.bufu = tl.bufu; p.bufs = tl.dma == 1 ? tl.bufu : 0;
  p.ring = ring ? 1 : 0;
  p.lean4 = (!s2d && NWV == 4 && T == 9 && tl.CKb == 2 && tl.dma == 2 && out_f32 == nullptr && g_bf_lean4) ? 1 : 0;
  p.rowdma = rowdma; p.lwp = lwp;
#ifdef YOGO_DIAG
  p.dbg = g_diag_dbg;
  if (g_diag_nodma && (tl.lds_dummy + 1) * 16 <= BF_LDS_MAX) p.dma = 0;  // synchronous staging through registers
  {
    const size_t nwg = (size_t)grid.x * grid.y * grid.z;
    if (g_diag_stamps != nullptr && nwg * 128 <= g_diag_stamps_bytes) {
      (void)hipMemsetAsync(g_diag_stamps, 0, nwg * 128, stream);
      p.stamps = g_diag_stamps;
    }
  }
#endif
  if ((act_ref != nullptr || signs_read) && bias != nullptr) {
    yogo_set_error("conv_bf16: an activation reference goes with a data gradient (no bias)");
    return YOGO_ERR_ARG;
  }
  if (B == 0) return YOGO_OK;
  int lds_bytes = max(p.dma ? tl.lds_bytes : (tl.lds_dummy + 1) * 16, (2 + 2 * NWV) * 32 * MW * 4);
  // ping-pong main loop: 3x3 taps, 16-channel chunks (one step per tap) through the double-buffered LDS-DMA pipeline, with its
  // fixed slot layout: 5 slots for the input tile, 5 for the weight slices, two buffers of 10 slots = all of the LDS
  const bool use_pp = NWV == 8 && !s2d && T == 9 && p.dma == 1 && tl.CKb == 2 && tl.ni_slots <= 5 && tl.n_slots - tl.ni_slots <= 5 && g_bf_pp;
  if (use_pp) {
    p.ni_slots = 5; p.n_slots = 10; p.ldsw_off = 5 * 64 * NWV; p.bufu = 10 * 64 * NWV; p.bufs = p.bufu;
    lds_bytes = 2 * p.bufu * 16;
  }
  char plan_txt[256] = "";
  if (yogo_launch_log_enabled())
    snprintf(plan_txt, sizeof(plan_txt), "K=%d M=%d in=%dx%d out=%dx%d a=%d s2d=%d T=%d ncb=%d TW=%d tiles_per_band=%d CKb=%d nchunk=%d rows=%d LW=%d lds=%d dma=%d slots=%d+%d rowpitch=%d grid=%ux%ux%u",
             K, M, IH, IW, OH, OW, a, s2d, T, tl.ncb, tl.TW, tl.tiles_per_band, tl.CKb, Kb / tl.CKb, tl.rows_max, tl.LW, lds_bytes, p.dma, tl.ni_slots,
             tl.n_slots - tl.ni_slots, rowdma ? lwp : 0, grid.x, grid.y, grid.z);
#define BFLAUNCH__(MW_, NW_, NWV_, S2D_, PF_, F32_, REF_)                                                               \
  bf_launch_t<MW_, NW_, NWV_, S2D_, PF_, F32_, REF_>(grid, lds_bytes, stream, p, use_pp, plan_txt)
#define BFLAUNCH_(MW_, NW_, NWV_, S2D_, PF_, F32_)                             \
  do {                                                                         \
    if (act_ref != nullptr) BFLAUNCH__(MW_, NW_, NWV_, S2D_, PF_, false, 1);   \
    else if (signs_read) BFLAUNCH__(MW_, NW_, NWV_, S2D_, PF_, false, 2);      \
    else if (signs != nullptr && NWV_ != 8) BFLAUNCH__(MW_, NW_, NWV_, S2D_, PF_, false, 3); \
    else BFLAUNCH__(MW_, NW_, NWV_, S2D_, PF_, F32_, 0);                       \
  } while (0)
#define BFLAUNCH(MW_, NW_, NWV_, S2D_, PF_)                                    \
  do {                                                                         \
    if (out_f32 != nullptr) BFLAUNCH_(MW_, NW_, NWV_, S2D_, PF_, true);        \
    else BFLAUNCH_(MW_, NW_, NWV_, S2D_, PF_, false);                          \
  } while (0)
  if (s2d) {
    if (MW == 4) BFLAUNCH_(4, 1, 8, true, 10, false);
    else if (MW == 2) BFLAUNCH_(2, 1, 4, true, 16, false);
    else BFLAUNCH_(1, 2, 4, true, 16, false);
  } else if (a == 2) {
    if (MW == 4) BFLAUNCH(4, 1, 8, false, 10);
    else if (MW == 2) BFLAUNCH(2, 1, 4, false, 16);
    else BFLAUNCH(1, 2, 4, false, 16);
  } else {
    if (MW == 4) BFLAUNCH(4, 2, 8, false, 10);
    else if (MW == 2 && wide64) BFLAUNCH(2, 4, 8, false, 10);
    else if (MW == 2) BFLAUNCH(2, 2, 4, false, 16);
    else BFLAUNCH(1, 4, 4, false, 16);
  }
#undef BFLAUNCH
#undef BFLAUNCH_
#undef BFLAUNCH__
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
  return launch_conv_bf16(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, false, nullptr, nullptr, B, Cin, Cout, IH, IW,
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
  return launch_conv_bf16(in, packed, bias, out, out_f32, nullptr, nullptr, 0, nullptr, false, chan_scale, stats_part, B, Cin, Cout, IH, IW,
                          (IH + 2 * pad - ks) / stride + 1, (IW + 2 * pad - ks) / stride + 1, ks, stride, 0, act, stream,
                          nullptr, nullptr);
}

// the same with a second bf16 output `out_pre` = conv + bias before the activation (what the backward pass of a SiLU block
// without BatchNorm needs: silu'(z) is a function of the pre-activation); bf16 NCHW8c outputs only
extern "C" int yogo_conv2d_fwd_bf16_pre(const void* in, const void* packed, const float* bias, void* out, void* out_pre,
                                        const float* chan_scale, int B, int Cin, int Cout, int IH, int IW, int ks, int stride,
                                        int act, hipStream_t stream) {
  YOGO_CHECK_ARG(in && packed && out && out_pre, "conv2d_fwd_bf16_pre: null pointer");
  if (int e = check_bf16_conv(B, Cin, Cout, IH, IW, ks, stride)) return e;
  const int pad = ks == 3 ? 1 : 0;
  return launch_conv_bf16(in, packed, bias, out, nullptr, out_pre, nullptr, 0, nullptr, false, chan_scale, nullptr, B, Cin, Cout, IH, IW,
                          (IH + 2 * pad - ks) / stride + 1, (IW + 2 * pad - ks) / stride + 1, ks, stride, 0, act, stream,
                          nullptr, nullptr);
}

// bytes of the LeakyReLU sign map of a bf16 NCHW8c tensor with C channels: [B][2][H][W][Cpad/16] bytes, Cpad = C rounded up
// to the channel tile (32 / 64 / 128 for C <= 32 / <= 64 / more); byte (h, pixel, q), bit i + 4e = (channel 4h + i of channel
// block 2q + e > 0) -- the lane order of the conv epilogue, so a lane's signs of one pixel are one contiguous store
extern "C" int yogo_bf16_signs_bytes(int B, int C, int H, int W, size_t* bytes) {
  YOGO_CHECK_ARG(bytes && B >= 0 && C > 0 && H > 0 && W > 0, "bf16_signs_bytes: bad arguments");
  *bytes = (size_t)B * H * W * (bf_mpad_of(C) / 8);
  return YOGO_OK;
}

// yogo_conv2d_fwd_bf16 (bf16 output) that also writes the sign map of its output: what the data gradient of the NEXT layer
// needs of a LeakyReLU block without BatchNorm (yogo_conv2d_dgrad_bf16_signs), at 1/16 of the bytes of the output itself
extern "C" int yogo_conv2d_fwd_bf16_signs(const void* in, const void* packed, const float* bias, void* out, void* signs,
                                          const float* chan_scale, int B, int Cin, int Cout, int IH, int IW, int ks, int stride,
                                          int act, hipStream_t stream) {
  YOGO_CHECK_ARG(in && packed && out && signs, "conv2d_fwd_bf16_signs: null pointer");
  if (int e = check_bf16_conv(B, Cin, Cout, IH, IW, ks, stride)) return e;
  const int pad = ks == 3 ? 1 : 0;
  return launch_conv_bf16(in, packed, bias, out, nullptr, nullptr, nullptr, 0, signs, false, chan_scale, nullptr, B, Cin, Cout, IH, IW,
                          (IH + 2 * pad - ks) / stride + 1, (IW + 2 * pad - ks) / stride + 1, ks, stride, 0, act, stream,
                          nullptr, nullptr);
}

// yogo_conv2d_dgrad_bf16 with act = LeakyReLU and the sign map of the reference in place of the reference:
// dx = conv_transpose(dy) * (sign bit ? 1 : 0.01) * chan_scale
extern "C" int yogo_conv2d_dgrad_bf16_signs(const void* dy, const void* packed_dgrad, void* dx, const void* signs,
                                            const float* chan_scale, int B, int Cin, int Cout, int IH, int IW, int ks, int stride,
                                            hipStream_t stream) {
  YOGO_CHECK_ARG(dy && packed_dgrad && dx && signs, "conv2d_dgrad_bf16_signs: null pointer");
  if (int e = check_bf16_conv(B, Cin, Cout, IH, IW, ks, stride)) return e;
  const int pad = ks == 3 ? 1 : 0;
  const int OHf = (IH + 2 * pad - ks) / stride + 1, OWf = (IW + 2 * pad - ks) / stride + 1;
  return launch_conv_bf16(dy, packed_dgrad, nullptr, dx, nullptr, nullptr, nullptr, ACT_LEAKY, const_cast<void*>(signs), true, chan_scale,
                          nullptr, B, Cout, Cin, OHf, OWf, IH, IW, ks, 1, (stride == 2 && ks == 3) ? 1 : 0, ACT_NONE, stream, nullptr,
                          nullptr);
}

// dx = conv_transpose(dy) * act'(act_ref) * chan_scale, all bf16 NCHW8c; (IH, IW) = the forward conv's INPUT dims.
// Stride 1: weights packed with mode 1.  Stride 2 (3x3): weights packed with mode 2; the gradient is computed per output
// parity class from the un-upsampled dy tile (9 tap-GEMMs per 2x2 output quad).
extern "C" int yogo_conv2d_dgrad_bf16(const void* dy, const void* packed_dgrad, void* dx, const void* act_ref, int ref_act,
                                      const float* chan_scale, int B, int Cin, int Cout, int IH, int IW, int ks, int stride,
                                      hipStream_t stream) {
  YOGO_CHECK_ARG(dy && packed_dgrad && dx, "conv2d_dgrad_bf16: null pointer");
  if (int e = check_bf16_conv(B, Cin, Cout, IH, IW, ks, stride)) return e;
  const int pad = ks == 3 ? 1 : 0;
  const int OHf = (IH + 2 * pad - ks) / stride + 1, OWf = (IW + 2 * pad - ks) / stride + 1;
  return launch_conv_bf16(dy, packed_dgrad, nullptr, dx, nullptr, nullptr, act_ref, ref_act, nullptr, false, chan_scale, nullptr, B, Cout, Cin, OHf, OWf,
                          IH, IW, ks, 1, (stride == 2 && ks == 3) ? 1 : 0, ACT_NONE, stream, nullptr, nullptr);
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
