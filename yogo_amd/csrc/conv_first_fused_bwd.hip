// Layer 1's data gradient, layer 0's backward sums and (WG) layer 1's weight gradient in ONE sweep over the gradient w.r.t. layer 1's output
// (base_model: yogo/model_defns.py:34-41 under autograd; C ABI: yogo_conv2d_dgrad[_wgrad]_bf16_first_bwd).
//
// Unfused, the gradient w.r.t. layer 0's output (16 channels at 386 x 516 x 128 images = 816 MB of bf16) is written by layer 1's data
// gradient and read back once by conv_first_bn_wgrad_pk2_kernel, whose only products are SUMS over the pixels: A1[c][j] = sum gb[c] * patch_j
// and S1[c] = sum gb[c], gb = dy * LeakyReLU'(sign map of layer 0's output), patch = the 3x3 stride-2 window of the uint8 image.  Layer 0
// has no data gradient of its own, so that tensor need not exist: this kernel computes layer 1's data gradient tile by tile (independent
// wavefronts, LDS-DMA staged input, conv_bf16_staged.hip's skeleton on v_mfma_f32_16x16x32_bf16 -- all 32 gradient channels in one MFMA, the
// nine weight operands resident in registers for the whole launch) and contracts every 32 output pixels with their image windows on the
// matrix cores as well, instead of storing them (see the kernel).  HBM traffic: g (1.63 GB) + image + sign map instead of g + dy written +
// dy read + image + sign map.
//
// Arithmetic: dy is rounded to bf16 (what the data gradient would have stored); dy * pixel products are exact in fp32 and are summed by the
// MFMA's fp32 accumulation; LeakyReLU' multiplies the sum of the negative-side products by 0.01 once instead of every element.  The unfused
// pair rounds 0.01 dy per element and sums in another order, so the two agree to fp32 rounding of sums over the whole batch, not bit for
// bit (tests/test_gpu_first_fused_bwd.py: both against a CPU fp64 reference).
#include "common.h"
#include <mutex>
#include <type_traits>
#include <utility>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct DgFirstParams {
  const u32x4* g;                // bf16 NCHW8c [B][4][H][W] units: gradient w.r.t. layer 1's conv output
  const u32x4* wp;               // layer 1's data-gradient packing (yogo_conv_bf16_pack mode 1): [9][4][Mpad] units
  const unsigned char* img;      // uint8 [B][2H][2W]: layer 0's input
  const unsigned short* signs;   // [B][H*W]: layer 0's sign map (yogo_conv_first_mfma_signs), or null when act is ACT_NONE
  float* part;                   // [gridDim.x * wavefronts][ncol]
  // with layer 1's weight gradient (WG): its input = layer 0's output, and the per-workgroup partial results
  const u32x4* x;                // bf16 NCHW8c [B][2][H][W] units
  float* slab;                   // [gridDim.x][9][32][16]: dW[tap][co][ci]
  float* bias_part;              // [gridDim.x][32]
  int B, H, W, Mpad, act, ncol;
  // a wavefront walks SEGMENTS: seg consecutive tiles of one tile column, top to bottom, so that the two halo rows a tile shares with the one
  // above it are as recent as they can be (-8 % against tile-linear order; most still come from memory: 4.6 MB pass through an XCD's 4 MB L2
  // per round of tiles, FETCH = 1.41 x the algorithmic bytes); consecutive segments are horizontal neighbours
  int tiles_per_row, tile_rows, seg, segs_per_img, nsegs;
  unsigned m_tpr, m_spi;         // ceil(2^32 / d)
};

namespace {
constexpr int DF_NJ = 9, DF_PER = 2 * DF_NJ + 2, DF_COUT = 16;   // the partial rows of conv_first_bn_wgrad_*: [Cout][A1 9 | A2 9 | S1 | S2] + P[9] + G[9][9]
__device__ __forceinline__ int df_udivm1(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }
__device__ __forceinline__ unsigned df_u(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ i32x4 df_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  return i32x4{(int)df_u((unsigned)a), (int)df_u((unsigned)(a >> 32) & 0xFFFFu), (int)df_u(bytes), 0x00020000};
}
// LDS-DMA pieces (conv_bf16_staged.hip): 64 lanes x 16 bytes -> LDS [lds_addr, + 1024); 64 lanes x 4 bytes -> [lds_addr, + 256)
__device__ __forceinline__ void df_dma16(i32x4 rs, unsigned lds_addr, int voff) {
  unsigned keep;
  lds_addr = df_u(lds_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(rs) : "memory");
}
__device__ __forceinline__ void df_dma4(i32x4 rs, unsigned lds_addr, int voff) {
  unsigned keep;
  lds_addr = df_u(lds_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dword %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(rs) : "memory");
}
__device__ __forceinline__ void df_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// eight pixels of this lane's channel: two transposed reads of four pixels each, `second` bytes apart (wgrad_bf16.hip: lds_tr8)
__device__ __forceinline__ bf16x8 df_tr8(const unsigned char* a, int second) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(a));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(a + second));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
}  // namespace

// R: output rows of a wavefront's tile (R x 32 pixels); NWV: wavefronts per workgroup.
//
// Two contractions per 32 pixels, both on the matrix cores.  (1) The data gradient, TRANSPOSED: D'[pixel][ci] = sum over (tap, gradient channel)
// of g * w -- the MFMA's A operand is the staged gradient (row = pixel of a 16-pixel block), its B operand the resident weights (column =
// ci), so lane (c16, g4) ends up with channel ci = c16 of pixels 4 g4 .. 4 g4 + 3 of the block.  (2) The sums over pixels:
// D2[j][ci] += sum over the 32 pixels of X[j][pixel] * GB[pixel][ci] with K = pixel: the lane's own eight values (two blocks x four pixels),
// rounded to bf16 -- what the data gradient would have stored -- ARE its B operand (column ci, K group g4), no lane exchange; the A operand
// is row j of the image window (j < 9: tap j of the lane's eight pixels, bytes picked out of the staged image; j = 9: ones, which makes row 9
// the plain sum S1; j > 9: zeros).  uint8 values and bf16 gradients multiply exactly; LeakyReLU' = 1 or 0.01 is applied by SPLITTING the
// gradient by its sign bit into two operands with two accumulators, A1 = D2pos + 0.01 D2neg at the end.  (A first form kept D in the usual
// orientation and folded every value into 40 per-lane fp32 accumulators with packed FMAs: 624 vector instructions per 4 x 32-pixel tile
// beside 72 MFMAs, VALU-bound at 502 us; profiles/r06_first_fused_bwd.log.)
#ifndef DF_WAVES_PER_EU
#define DF_WAVES_PER_EU 2
#endif
#ifndef DF_NPF
#define DF_NPF 6   // operand reads in flight ahead of their MFMA
#endif
#ifndef DF_NPF_WG
#define DF_NPF_WG 3   // ... with the weight gradient's 80 accumulator registers beside them
#endif
namespace {
template <class F, int... I>
__device__ __forceinline__ void df_static_for(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
}  // namespace
//
// WG: layer 1's WEIGHT gradient in the same sweep -- it reads the same gradient tile (1.63 GB that wgrad_bf16_kernel would read again) and
// layer 0's output x (the tile's own pixels, no halo).  dW[tap][co][ci] = sum over pixels p of g[co][p - tap + 1] x[ci][p], K = the 32 pixels of a
// tile row: both operands come out of the staged channel-fastest units through ds_read_b64_tr_b16 (16 lanes read a 4 pixel x 16 channel block,
// each lane gets 4 pixels of its channel; wgrad_bf16.hip), the tap shifts the address of the gradient's halo tile; 2 x 9 accumulator tiles
// of 16 output x 16 input channels per wavefront for the whole launch, and a ones operand for the bias gradient.
template <int R, int NWV, bool WG>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(DF_WAVES_PER_EU))) void conv_bf16_dgrad_first_bwd_kernel(const DgFirstParams p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds_u[];   // (the dynamic block starts at LDS address 0: LDS-DMA takes addresses)
  constexpr int OOB = (int)0x80000000u;
  constexpr int KB = 4, NR = R + 2, ROWU = 34, TU = KB * NR * ROWU, NDMA = (TU + 63) / 64;
  constexpr int IR = 2 * R + 1, IROWB = 72, IDW = IROWB / 4, IU = IR * IDW, NIDMA = (IU + 63) / 64;   // staged image: IR rows of 18 dwords = columns 2 ox0 - 4 .. 2 ox0 + 67
  constexpr int SU = R * 16, NSDMA = (SU + 63) / 64;   // staged sign words: R rows of 16 dwords (32 pixels x 2 bytes)
  static_assert(!WG || (R % 2) == 0, "a piece of the staged x tile is two whole rows");
  constexpr int NXDMA = WG ? R : 0;   // staged x tile: [2 channel blocks][R rows][32 columns] units
  constexpr int WAVE_BYTES = NDMA * 1024 + NIDMA * 256 + NSDMA * 256 + NXDMA * 1024;
  constexpr int CONST_BYTES = ((IR * IROWB + 15) / 16) * 16;   // an "image" of ones and one of zeros behind the wavefronts' pieces (rows 9 and 10..15 of X)
  const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W, IH0 = 2 * H, IW0 = 2 * W;
  const int kcb = H * W * 16;
  const bool leaky = p.act == ACT_LEAKY;   // uniform
  const unsigned my_addr = (unsigned)(wave * WAVE_BYTES), my_img_addr = my_addr + NDMA * 1024, my_sg_addr = my_img_addr + NIDMA * 256;
  [[maybe_unused]] const unsigned my_x_addr = my_sg_addr + NSDMA * 256;
  const unsigned ones_addr = (unsigned)(NWV * WAVE_BYTES), zeros_addr = ones_addr + CONST_BYTES;
  const u32x4* my_tile = lds_u + wave * (WAVE_BYTES / 16);
  const unsigned char* lds8 = reinterpret_cast<const unsigned char*>(lds_u);
  {
    unsigned* cw = reinterpret_cast<unsigned*>(lds_u) + ones_addr / 4;
    for (int e = tid; e < CONST_BYTES / 4; e += 64 * NWV) {
      cw[e] = 0x01010101u;
      cw[e + CONST_BYTES / 4] = 0u;
    }
  }
  __syncthreads();

  // the nine weight operands of this lane: [n = input channel c16][k = gradient channels 8 g4 .. 8 g4 + 7] per tap, resident
  bf16x8 wv[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) wv[tp] = __builtin_bit_cast(bf16x8, p.wp[(tp * KB + g4) * p.Mpad + c16]);

  // this lane's elements of the staged gradient tile (piece i = units 64 i + lane -> channel block, row, column), of the staged image
  // (dword e -> row, dword of the row) and of the staged sign words
  int rel[NDMA];
  unsigned rc[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int u = i * 64 + lane;
    const int kb = u / (NR * ROWU), rem = u - kb * (NR * ROWU);
    const int r = rem / ROWU, cc = rem - r * ROWU;
    rel[i] = u < TU ? kb * kcb + (r * W + cc) * 16 : OOB;
    rc[i] = u < TU ? (unsigned)(r << 16 | cc) : 0xFFFFFFFFu;
  }
  unsigned irc[NIDMA];
#pragma unroll
  for (int i = 0; i < NIDMA; ++i) {
    const int e = i * 64 + lane;
    const int r = e / IDW, d = e - r * IDW;
    irc[i] = e < IU ? (unsigned)(r << 16 | d) : 0xFFFFFFFFu;
  }
  // row j = c16 of X: tap (kh, kw) of the lane's pixels 4 g4 + i: bytes A, A + 2, A + 4, A + 6 of a staged image row, A = 3 + 8 g4 + kw (+ 32 per
  // pixel block, + 144 per tile row, + 72 kh): three aligned dwords from x_addr on, shifted by x_sh bytes
  const int xj = c16 < 9 ? c16 : 0, kh = xj / 3, kw = xj - 3 * kh;
  const unsigned x_addr = c16 < 9 ? my_img_addr + (unsigned)(kh * IROWB + ((3 + 8 * g4 + kw) & ~3)) : (c16 == 9 ? ones_addr : zeros_addr) + (unsigned)(8 * g4);
  const unsigned x_sh = c16 < 9 ? (unsigned)((3 + kw) & 3) : 0u;
  const unsigned sg_addr = my_sg_addr + (unsigned)(8 * g4);   // this lane's four pixels of pixel block 0, tile row 0
  const unsigned sg_pos = (unsigned)(c16 < 4 ? c16 : c16 < 8 ? c16 + 4 : c16 < 12 ? c16 - 4 : c16);   // channel c16 in a pixel's sign word

  f32x4 d2p = {0.f, 0.f, 0.f, 0.f}, d2n = {0.f, 0.f, 0.f, 0.f};   // rows 4 g4 + i of D2, column c16
  // WG: rows co = 16 cbk + 4 g4 + i, column ci = c16 of dW[tap]; the bias gradient in every column of bacc
  f32x4 wacc[WG ? 2 : 1][WG ? 9 : 1], bacc[WG ? 2 : 1];
#pragma unroll
  for (int c = 0; c < (WG ? 2 : 1); ++c) {
    bacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < (WG ? 9 : 1); ++t) wacc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // transposed reads: lane (q = (c16 >> 2), pc = c16 & 3) of a 16-lane group supplies channels 4 pc .. 4 pc + 3 of pixel 8 g4 + q (+ 4: the second read)
  [[maybe_unused]] const unsigned tr_g = my_addr + (unsigned)((((c16 & 3) >> 1) * NR * ROWU + 8 * g4 + (c16 >> 2)) * 16 + (c16 & 1) * 8);
  [[maybe_unused]] const unsigned tr_x = my_x_addr + (unsigned)((((c16 & 3) >> 1) * R * 32 + 8 * g4 + (c16 >> 2)) * 16 + (c16 & 1) * 8);
  const u32x4 ones_op = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

  // workgroup -> XCD -> a contiguous eighth of the segments (conv_bf16_staged.hip)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int s8 = (p.nsegs + 7) >> 3, s_end = min(p.nsegs, (xcd + 1) * s8);
  for (int sgm = xcd * s8 + slot * NWV + wave; sgm < s_end; sgm += nslot * NWV) {
    const int b = df_udivm1(sgm, p.segs_per_img, p.m_spi);
    const int t = sgm - b * p.segs_per_img;
    const int sy = df_udivm1(t, p.tiles_per_row, p.m_tpr), ox0 = (t - sy * p.tiles_per_row) * 32;
    const int ty_end = min(p.tile_rows, (sy + 1) * p.seg);
   for (int ty = sy * p.seg; ty < ty_end; ++ty) {
    const int oy0 = ty * R;
    // ---- stage the gradient tile (rows oy0 - 1 .. oy0 + R, columns ox0 - 1 .. ox0 + 32; zeros beyond the image), the image window and the sign words
    {
      const i32x4 rs_g = df_rsrc(p.g + (size_t)b * KB * H * W, (unsigned)(KB * kcb));
      const int iy0 = oy0 - 1, ix0 = ox0 - 1;
      const int base = (iy0 * W + ix0) * 16;
      if (iy0 >= 0 && iy0 + NR <= H && ix0 >= 0 && ix0 + ROWU <= W) {   // (uniform) an interior tile: no border tests
#pragma unroll
        for (int i = 0; i < NDMA; ++i) df_dma16(rs_g, my_addr + (unsigned)i * 1024u, (i + 1) * 64 <= TU ? rel[i] + base : (rel[i] == OOB ? OOB : rel[i] + base));
      } else {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
          const int r = (int)(rc[i] >> 16), c = (int)(rc[i] & 0xFFFFu);
          const bool ok = ((unsigned)(iy0 + r) < (unsigned)H) && ((unsigned)(ix0 + c) < (unsigned)W);   // (all ones: r = 65535 fails the row test)
          df_dma16(rs_g, my_addr + (unsigned)i * 1024u, ok ? rel[i] + base : OOB);
        }
      }
      const i32x4 rs_i = df_rsrc(p.img + (size_t)b * IH0 * IW0, (unsigned)(IH0 * IW0));
      const int jy0 = 2 * oy0 - 1, jx0 = 2 * ox0 - 4;
#pragma unroll
      for (int i = 0; i < NIDMA; ++i) {
        const int r = (int)(irc[i] >> 16), d = (int)(irc[i] & 0xFFFFu);
        const int iy = jy0 + r, ix = jx0 + 4 * d;   // (a dword is inside the row or outside it as a whole: 2 W is a multiple of 4)
        const bool ok = ((unsigned)iy < (unsigned)IH0) && ((unsigned)ix < (unsigned)IW0);
        df_dma4(rs_i, my_img_addr + (unsigned)i * 256u, ok ? iy * IW0 + ix : OOB);
      }
      if (leaky) {
        const i32x4 rs_s = df_rsrc(p.signs + (size_t)b * H * W, (unsigned)(H * W * 2));
#pragma unroll
        for (int i = 0; i < NSDMA; ++i) {
          const int e = i * 64 + lane, r = e >> 4, d = e & 15;   // dword d of tile row r: pixels ox0 + 2 d, + 1 (W is even: inside the row or outside as a whole)
          const bool ok = e < SU && oy0 + r < H && ox0 + 2 * d < W;
          df_dma4(rs_s, my_sg_addr + (unsigned)i * 256u, ok ? ((oy0 + r) * W + ox0 + 2 * d) * 2 : OOB);
        }
      }
      if constexpr (WG) {   // piece i = rows 2 i, 2 i + 1 of the flattened [channel block][row]: lanes 0-31 / 32-63, column = lane & 31
        const i32x4 rs_x = df_rsrc(p.x + (size_t)b * 2 * H * W, (unsigned)(2 * kcb));
        const int hi = lane >> 5, col = lane & 31;
        const int xo = ((oy0 + hi) * W + ox0 + col) * 16;
        const bool cok = ox0 + col < W;
#pragma unroll
        for (int i = 0; i < NXDMA; ++i) {
          constexpr int dummy = 0; (void)dummy;
          const int kb = (2 * i) / R, row0 = (2 * i) % R;
          df_dma16(rs_x, my_x_addr + (unsigned)i * 1024u, (cok && oy0 + row0 + hi < H) ? xo + kb * kcb + row0 * W * 16 : OOB);
        }
      }
    }
    const bool right = ox0 + 32 > W;   // (uniform) pixel columns beyond the image: their "gradient" is the halo's, not zero
    df_wait_dma();
    const u32x4* bt = my_tile + g4 * NR * ROWU + c16;   // this lane's pixel of tile row 0 / pixel block 0, channel block g4, tap (0, 0)
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      if (oy0 + rr >= H) break;   // (uniform)
      // A row is ONE stream of (operand read, MFMA) steps: 2 x 9 of the data gradient (pixel block, tap), then [WG] 2 x 9 of the weight
      // gradient (output-channel block, tap).  Operands run DF_NPF steps ahead of their MFMA in a ring of registers, and a scheduling barrier
      // behind every step keeps that order: left to itself hipcc sinks every LDS read next to its use (it has 256 registers for 130
      // accumulator / weight registers) and each 16-cycle MFMA then waits a full LDS round trip (the first form of this loop: matrix pipes 0.35 busy).
      // The sums over pixels of the row (X and GB operands, two MFMAs) ride in the gaps of the weight-gradient steps.
      constexpr int NST = WG ? 36 : 18, NPF = WG ? DF_NPF_WG : DF_NPF;
      f32x4 acc[2];
      bf16x8 ring[NPF];
      [[maybe_unused]] bf16x8 xt;
      auto load = [&](auto s_tag) {
        constexpr int S = decltype(s_tag)::value;
        if constexpr (S < 18) {
          constexpr int pb = S / 9, tp = S % 9, ky = tp / 3, kx = tp % 3;
          ring[S % NPF] = __builtin_bit_cast(bf16x8, bt[(rr + ky) * ROWU + 16 * pb + kx]);
        } else {
          constexpr int cbk = (S - 18) / 9, tp = (S - 18) % 9, ky = tp / 3, kx = tp % 3;
          ring[S % NPF] = df_tr8(lds8 + tr_g + ((cbk * 2 * NR + rr + 2 - ky) * ROWU + 2 - kx) * 16, 64);
        }
      };
      auto use = [&](auto s_tag) {
        constexpr int S = decltype(s_tag)::value;
        if constexpr (S < 18) {
          constexpr int pb = S / 9, tp = S % 9;
          acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[S % NPF], wv[tp], tp == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[pb], 0, 0, 0);
        } else if constexpr (WG) {
          constexpr int cbk = (S - 18) / 9, tp = (S - 18) % 9;
          wacc[cbk][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[S % NPF], xt, wacc[cbk][tp], 0, 0, 0);
          if constexpr (tp == 4) bacc[cbk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[S % NPF], __builtin_bit_cast(bf16x8, ones_op), bacc[cbk], 0, 0, 0);
        }
      };
      // the sums over pixels in slices: 0 = the LDS reads, 1 / 2 = the X operand of pixel block 0 / 1, 3 / 4 = GB of pixel block 0 / 1, 5 / 6 = the MFMAs
      unsigned xr[2][3];
      unsigned sw[2][2] = {{0xFFFFFFFFu, 0xFFFFFFFFu}, {0xFFFFFFFFu, 0xFFFFFFFFu}};
      u32x4 xop, gpos, gneg;
      auto p3 = [&](auto k_tag) {
        constexpr int K = decltype(k_tag)::value;
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        if constexpr (K == 0) {
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            const unsigned char* xa = lds8 + x_addr + rr * (2 * IROWB) + 32 * pb;
            xr[pb][0] = *reinterpret_cast<const unsigned*>(xa);
            xr[pb][1] = *reinterpret_cast<const unsigned*>(xa + 4);
            xr[pb][2] = *reinterpret_cast<const unsigned*>(xa + 8);
            if (leaky) {
              const unsigned long long s2 = *reinterpret_cast<const unsigned long long*>(lds8 + sg_addr + rr * 64 + 32 * pb);
              sw[pb][0] = (unsigned)s2;
              sw[pb][1] = (unsigned)(s2 >> 32);
            }
          }
        } else if constexpr (K == 1 || K == 2) {   // X: row j of the image window for this lane's 4 pixels of the block, as bf16
          constexpr int pb = K - 1;
          const unsigned lo4 = __builtin_amdgcn_alignbyte(xr[pb][1], xr[pb][0], x_sh), hi4 = __builtin_amdgcn_alignbyte(xr[pb][2], xr[pb][1], x_sh);
          const unsigned xs = __builtin_amdgcn_perm(hi4, lo4, 0x06040200u);   // bytes A, A + 2, A + 4, A + 6
          const bf16x2 q0 = {(__bf16)(float)(xs & 0xFFu), (__bf16)(float)((xs >> 8) & 0xFFu)};
          const bf16x2 q1 = {(__bf16)(float)((xs >> 16) & 0xFFu), (__bf16)(float)(xs >> 24)};
          xop[2 * pb] = __builtin_bit_cast(unsigned, q0);
          xop[2 * pb + 1] = __builtin_bit_cast(unsigned, q1);
        } else if constexpr (K == 3 || K == 4) {   // GB: the lane's four gradients of the block as bf16, split by the sign bit of layer 0's output
          constexpr int pb = K - 3;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const bf16x2 dq = {(__bf16)acc[pb][2 * q], (__bf16)acc[pb][2 * q + 1]};
            unsigned dw = __builtin_bit_cast(unsigned, dq);
            if (right && ox0 + 16 * pb + 4 * g4 + 2 * q >= W) dw = 0u;
            const unsigned m = leaky ? __umul24((sw[pb][q] >> sg_pos) & 0x00010001u, 0xFFFFu) : 0xFFFFFFFFu;
            const unsigned pos = dw & m;
            gpos[2 * pb + q] = pos;
            gneg[2 * pb + q] = dw ^ pos;
          }
        } else if constexpr (K == 5) {
          d2p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xop), __builtin_bit_cast(bf16x8, gpos), d2p, 0, 0, 0);
        } else {
          d2n = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xop), __builtin_bit_cast(bf16x8, gneg), d2n, 0, 0, 0);
        }
      };
      df_static_for([&](auto s_tag) { load(s_tag); }, std::make_integer_sequence<int, NPF>{});
      __builtin_amdgcn_sched_barrier(0);
      df_static_for(
          [&](auto s_tag) {
            constexpr int S = decltype(s_tag)::value;
            use(s_tag);
            if constexpr (S + NPF < NST) load(std::integral_constant<int, S + NPF>{});
            if constexpr (WG) {
              if constexpr (S == 10) xt = df_tr8(lds8 + tr_x + rr * 512, 64);
              if constexpr (S == 18) p3(std::integral_constant<int, 0>{});
              if constexpr (S == 23) p3(std::integral_constant<int, 1>{});
              if constexpr (S == 25) p3(std::integral_constant<int, 2>{});
              if constexpr (S == 27) p3(std::integral_constant<int, 3>{});
              if constexpr (S == 29) p3(std::integral_constant<int, 4>{});
              if constexpr (S == 31) p3(std::integral_constant<int, 5>{});
              if constexpr (S == 32) p3(std::integral_constant<int, 6>{});
            } else {
              if constexpr (S == 12) p3(std::integral_constant<int, 0>{});   // (the reads; the rest needs the finished data gradient)
              if constexpr (S == 15) p3(std::integral_constant<int, 1>{});
              if constexpr (S == 16) p3(std::integral_constant<int, 2>{});
            }
            __builtin_amdgcn_sched_barrier(0);
          },
          std::make_integer_sequence<int, NST>{});
      if constexpr (!WG) {
        p3(std::integral_constant<int, 3>{});
        p3(std::integral_constant<int, 4>{});
        p3(std::integral_constant<int, 5>{});
        p3(std::integral_constant<int, 6>{});
      }
    }
   }
  }
  // ---- the wavefront's partial row: lane (c16, g4) holds rows j = 4 g4 + i of column ci = c16: A1[ci][j] for j < 9, S1[ci] for j = 9
  float* prow = p.part + (size_t)(blockIdx.x * NWV + wave) * p.ncol;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = 4 * g4 + i;
    const float v = d2p[i] + LEAKY_SLOPE * d2n[i];
    if (j < DF_NJ) prow[c16 * DF_PER + j] = v;
    else if (j == DF_NJ) prow[c16 * DF_PER + 2 * DF_NJ] = v;
  }
  if (g4 == 0) prow[c16 * DF_PER + 2 * DF_NJ + 1] = 0.f;   // (S2: derived by the finalize kernel)
  // (the A2 columns come from the Gram matrix, P / G from the caller's: zeros here so that the reduced row is finite)
  for (int e = lane; e < DF_COUT * DF_NJ; e += 64) prow[(e / DF_NJ) * DF_PER + DF_NJ + e % DF_NJ] = 0.f;
  for (int e = lane; e < DF_NJ + DF_NJ * DF_NJ; e += 64) prow[DF_COUT * DF_PER + e] = 0.f;
  if constexpr (WG) {   // one partial result per WORKGROUP: the four wavefronts' accumulators meet in LDS (the tile buffers are free now)
    constexpr int NE = 9 * 32 * 16, STRIDE = NE + 32;
    static_assert(NWV * STRIDE * 4 <= NWV * WAVE_BYTES + 2 * CONST_BYTES, "the reduction uses the kernel's LDS allocation");
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds_u) + wave * STRIDE;
#pragma unroll
    for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = 16 * cbk + 4 * g4 + i;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) red[(tp * 32 + co) * 16 + c16] = wacc[cbk][tp][i];
        if (c16 == 0) red[NE + co] = bacc[cbk][i];
      }
    __syncthreads();
    const float* r0 = reinterpret_cast<const float*>(lds_u);
    float* sl = p.slab + (size_t)blockIdx.x * NE;
    float* bp = p.bias_part + (size_t)blockIdx.x * 32;
    for (int e = tid; e < STRIDE; e += 64 * NWV) {
      float v = r0[e];
#pragma unroll
      for (int w = 1; w < NWV; ++w) v += r0[w * STRIDE + e];
      if (e < NE) sl[e] = v;
      else bp[e - NE] = v;
    }
  }
}

// =========================================================================================================
// host side
// =========================================================================================================
// tile height / wavefronts per workgroup (variant builds: build.sh variant TAG conv_first_fused_bwd -DDF_TILE_ROWS=8 ...)
#ifndef DF_TILE_ROWS
#define DF_TILE_ROWS 6
#endif
#ifndef DF_TILE_ROWS_WG
#define DF_TILE_ROWS_WG 4
#endif
#ifndef DF_WAVES
#define DF_WAVES 4
#endif
namespace {
constexpr int DF_NWV = DF_WAVES;
constexpr int df_rows(bool wg) { return wg ? DF_TILE_ROWS_WG : DF_TILE_ROWS; }
constexpr int df_lds_bytes(int R, int NWV, bool wg) {   // the wavefronts' pieces (gradient tile, image window, sign words[, x tile]) + the two constant images
  return NWV * (((4 * (R + 2) * 34 + 63) / 64) * 1024 + (((2 * R + 1) * 18 + 63) / 64) * 256 + ((R * 16 + 63) / 64) * 256 + (wg ? R * 1024 : 0)) +
         2 * ((((2 * R + 1) * 72 + 15) / 16) * 16);
}
int df_n_cu() {
  static std::mutex mu;
  static int n_cu_of[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  std::lock_guard<std::mutex> lk(mu);
  if (n_cu_of[dev] == 0) {
    hipDeviceProp_t prop;
    n_cu_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n_cu_of[dev];
}
// the launch: workgroups, and the segment length -- short enough that the wavefronts of an XCD get equal shares, long enough that few tiles
// re-read their upper halo from memory (cost = the busiest wavefront's tiles x (1 + the halo rows a segment's first tile fetches))
struct DfPlan { int grid, seg; };
DfPlan df_plan(int B, int H, int W, int n_cu, bool wg) {
  const int R = df_rows(wg);
  const int nty = cdiv(H, R), ntx = cdiv(W, 32);
  const int lds = df_lds_bytes(R, DF_NWV, wg);
  const int per_cu = max(1, min(16 / DF_NWV, (160 * 1024) / lds));
#ifdef DF_SEG
  const int seg_lo = DF_SEG, seg_hi = DF_SEG;
#else
  const int seg_lo = 1, seg_hi = 16;
#endif
  DfPlan best{8, 1};
  double best_cost = -1.0;
  for (int seg = seg_lo; seg <= seg_hi && (seg <= nty || seg == seg_lo); ++seg) {
    const long long nsegs = (long long)B * cdiv(nty, seg) * ntx;
    const int s8 = (int)((nsegs + 7) / 8);
    const int grid = 8 * max(1, min(cdiv(s8, DF_NWV), per_cu * n_cu / 8));
    const int waves_per_xcd = grid / 8 * DF_NWV;
    const double cost = (double)cdiv(s8, waves_per_xcd) * min(seg, nty) * (1.0 + (2.0 / R) / min(seg, nty));
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = DfPlan{grid, seg}; }
  }
  return best;
}
bool df_shape_ok(int Cmid, int Cout1, int H, int W, int B, int act0, bool wg) {
  if (Cmid != 16 || Cout1 != 32 || H < 1 || W < 2 || (W & 1) || B < 1 || (act0 != ACT_NONE && act0 != ACT_LEAKY)) return false;
  const int R = df_rows(wg);
  if ((long long)4 * H * W * 16 >= (1ll << 31) || (long long)4 * H * W >= (1ll << 31) || (long long)B * cdiv(H, R) * cdiv(W, 32) >= (1ll << 31)) return false;
  // the kernel's divisions by multiplication (segment -> image, segment of the image -> segment row), whatever segment length the plan picks
  const long long tpr = cdiv(W, 32);
  for (int seg = 1; seg <= 16; ++seg) {
    const long long spi = (long long)cdiv(cdiv(H, R), seg) * tpr;
    if (!magic_div_exact((long long)B * spi - 1, (int)spi) || !magic_div_exact(spi, (int)tpr)) return false;
  }
  return true;
}

template <bool WG>
int df_launch(DgFirstParams p, int B, int H, int W, int act0, const char* what, hipStream_t stream) {
  const int n_cu = df_n_cu();
  if (n_cu < 0) {
    yogo_set_error("%s: hipGetDevice failed", what);
    return YOGO_ERR_HIP;
  }
  constexpr int R = df_rows(WG);
  p.B = B; p.H = H; p.W = W; p.Mpad = 32; p.act = act0; p.ncol = DF_COUT * DF_PER + DF_NJ + DF_NJ * DF_NJ;
  const DfPlan pl = df_plan(B, H, W, n_cu, WG);
  p.tiles_per_row = cdiv(W, 32); p.tile_rows = cdiv(H, R); p.seg = pl.seg;
  p.segs_per_img = cdiv(p.tile_rows, pl.seg) * p.tiles_per_row; p.nsegs = B * p.segs_per_img;
  auto magic = [](int d) -> unsigned { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(((1ull << 32) + (unsigned)d - 1ull) / (unsigned)d); };
  p.m_tpr = magic(p.tiles_per_row); p.m_spi = magic(p.segs_per_img);
  const int grid = pl.grid;
  constexpr int lds = df_lds_bytes(R, DF_NWV, WG);
  static_assert(lds <= 160 * 1024, "LDS");
  if (lds > 64 * 1024) {   // more than 64 KB of dynamic LDS has to be asked for: once per device and instantiation
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < 64 && !done[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_dgrad_first_bwd_kernel<R, DF_NWV, WG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) {
        yogo_set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
        return YOGO_ERR_HIP;
      }
      done[dev] = true;
    }
  }
  hipLaunchKernelGGL((conv_bf16_dgrad_first_bwd_kernel<R, DF_NWV, WG>), dim3(grid), dim3(64 * DF_NWV), lds, stream, p);
  if (yogo_launch_log_enabled())
    yogo_launch_log("conv_bf16_dgrad_first_bwd_kernel<%d, %d, %s> | K=32 M=16 at %dx%d segments=%d of %d tiles grid=%d lds=%d act0=%d", R, DF_NWV, WG ? "true" : "false", H, W,
                    p.nsegs, pl.seg, grid, lds, act0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    yogo_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return YOGO_ERR_HIP;
  }
  return YOGO_OK;
}
}  // namespace

// split-K reduction shared with the weight-gradient kernels (wgrad_f32.hip)
extern "C" int yogo_internal_wgrad_reduce_q(void* queue, const float* slab, int nslab, int T, int M, int N, int Mpad, int Npad, float clip, float* dw,
                                            const float* bias_part, int nbias, float* db, hipStream_t stream);

// 1 when the fused sweep takes (layer 1: 3x3, stride 1, Cmid -> Cout1 channels at H x W; layer 0: one uint8 channel, stride 2, Cmid outputs)
extern "C" int yogo_conv2d_dgrad_first_bwd_supported(int Cmid, int Cout1, int H, int W, int B, int act0) {
  return (df_shape_ok(Cmid, Cout1, H, W, B, act0, false) && df_shape_ok(Cmid, Cout1, H, W, B, act0, true)) ? 1 : 0;
}

// rows of the partial buffer (cols: yogo_conv_first_bn_wgrad_cols(1, Cmid)) on the current device; with_wgrad: of the _wgrad form below
extern "C" int yogo_conv2d_dgrad_first_bwd_rows(int B, int H, int W, int with_wgrad, int* rows) {
  YOGO_CHECK_ARG(rows && B > 0 && H > 0 && W > 0, "conv2d_dgrad_first_bwd_rows: bad arguments");
  const int n_cu = df_n_cu();
  if (n_cu < 0) {
    yogo_set_error("conv2d_dgrad_first_bwd_rows: hipGetDevice failed");
    return YOGO_ERR_HIP;
  }
  *rows = df_plan(B, H, W, n_cu, with_wgrad != 0).grid * DF_NWV;
  return YOGO_OK;
}

// g: gradient w.r.t. layer 1's conv output (bf16 NCHW8c [B][Cout1 / 8][H][W]); packed: layer 1's mode-1 packing; image: uint8 [B][2H][2W];
// signs: layer 0's sign map [B][H*W][2] bytes (NULL with ACT_NONE); part: rows x cols floats in the layout of
// yogo_conv_first_bn_wgrad_bf16_xs -- finish with yogo_partials_reduce and yogo_conv_first_bn_wgrad_finalize_xs
extern "C" int yogo_conv2d_dgrad_bf16_first_bwd(const void* g, const void* packed, const void* image, const void* signs, float* part, int B, int Cmid,
                                                int Cout1, int H, int W, int act0, hipStream_t stream) {
  YOGO_CHECK_ARG(g && packed && image && part, "conv2d_dgrad_bf16_first_bwd: null pointer");
  YOGO_CHECK_ARG(yogo_conv2d_dgrad_first_bwd_supported(Cmid, Cout1, H, W, B, act0), "conv2d_dgrad_bf16_first_bwd: unsupported shape");
  YOGO_CHECK_ARG(signs != nullptr || act0 == ACT_NONE, "conv2d_dgrad_bf16_first_bwd: LeakyReLU needs the sign map");
  DgFirstParams p{};
  p.g = reinterpret_cast<const u32x4*>(g); p.wp = reinterpret_cast<const u32x4*>(packed); p.img = reinterpret_cast<const unsigned char*>(image);
  p.signs = reinterpret_cast<const unsigned short*>(signs); p.part = part;
  return df_launch<false>(p, B, H, W, act0, "conv2d_dgrad_bf16_first_bwd", stream);
}

// The same sweep with layer 1's WEIGHT gradient: x = layer 1's input (= layer 0's output, bf16 NCHW8c [B][Cmid / 8][H][W]); dw (OIHW fp32
// [Cout1][Cmid][3][3]) and db ([Cout1], may be NULL), clamped to +-clip when clip > 0, exactly as yogo_conv2d_wgrad_bf16[_deferred] delivers
// them; workspace: yogo_conv2d_dgrad_wgrad_first_bwd_workspace_bytes; queue: NULL (reduce now) or a yogo_wgrad_reduce_queue (deferred)
extern "C" int yogo_conv2d_dgrad_wgrad_first_bwd_workspace_bytes(int B, int H, int W, size_t* bytes) {
  YOGO_CHECK_ARG(bytes && B > 0 && H > 0 && W > 0, "conv2d_dgrad_wgrad_first_bwd_workspace_bytes: bad arguments");
  const int n_cu = df_n_cu();
  if (n_cu < 0) {
    yogo_set_error("conv2d_dgrad_wgrad_first_bwd_workspace_bytes: hipGetDevice failed");
    return YOGO_ERR_HIP;
  }
  *bytes = (size_t)df_plan(B, H, W, n_cu, true).grid * (9 * 32 * 16 + 32) * sizeof(float);
  return YOGO_OK;
}
extern "C" int yogo_conv2d_dgrad_wgrad_bf16_first_bwd(const void* g, const void* packed, const void* x, const void* image, const void* signs, float* part,
                                                      float* dw, float* db, void* workspace, int B, int Cmid, int Cout1, int H, int W, int act0, float clip,
                                                      void* queue, hipStream_t stream) {
  YOGO_CHECK_ARG(g && packed && x && image && part && dw && workspace, "conv2d_dgrad_wgrad_bf16_first_bwd: null pointer");
  YOGO_CHECK_ARG(yogo_conv2d_dgrad_first_bwd_supported(Cmid, Cout1, H, W, B, act0), "conv2d_dgrad_wgrad_bf16_first_bwd: unsupported shape");
  YOGO_CHECK_ARG(signs != nullptr || act0 == ACT_NONE, "conv2d_dgrad_wgrad_bf16_first_bwd: LeakyReLU needs the sign map");
  const int n_cu = df_n_cu();
  if (n_cu < 0) {
    yogo_set_error("conv2d_dgrad_wgrad_bf16_first_bwd: hipGetDevice failed");
    return YOGO_ERR_HIP;
  }
  const int rows = df_plan(B, H, W, n_cu, true).grid;   // of the weight-gradient slab: one per workgroup
  DgFirstParams p{};
  p.g = reinterpret_cast<const u32x4*>(g); p.wp = reinterpret_cast<const u32x4*>(packed); p.img = reinterpret_cast<const unsigned char*>(image);
  p.signs = reinterpret_cast<const unsigned short*>(signs); p.part = part;
  p.x = reinterpret_cast<const u32x4*>(x); p.slab = reinterpret_cast<float*>(workspace); p.bias_part = p.slab + (size_t)rows * 9 * 32 * 16;
  if (int e = df_launch<true>(p, B, H, W, act0, "conv2d_dgrad_wgrad_bf16_first_bwd", stream)) return e;
  return yogo_internal_wgrad_reduce_q(queue, p.slab, rows, 9, Cout1, Cmid, 32, 16, clip, dw, p.bias_part, rows, db, stream);
}
