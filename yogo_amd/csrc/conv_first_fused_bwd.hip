// Layer 1's data gradient and layer 0's backward sums in ONE sweep (base_model: yogo/model_defns.py:34-41 under autograd).
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
  int B, H, W, Mpad, act, ncol;
  int tiles_per_row, tiles_per_img, ntiles;
  unsigned m_tpr, m_tpi;         // ceil(2^32 / d)
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
template <int R, int NWV>
__global__ __launch_bounds__(64 * NWV) void conv_bf16_dgrad_first_bwd_kernel(const DgFirstParams p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds_u[];   // (the dynamic block starts at LDS address 0: LDS-DMA takes addresses)
  constexpr int OOB = (int)0x80000000u;
  constexpr int KB = 4, NR = R + 2, ROWU = 34, TU = KB * NR * ROWU, NDMA = (TU + 63) / 64;
  constexpr int IR = 2 * R + 1, IROWB = 72, IDW = IROWB / 4, IU = IR * IDW, NIDMA = (IU + 63) / 64;   // staged image: IR rows of 18 dwords = columns 2 ox0 - 4 .. 2 ox0 + 67
  constexpr int SU = R * 16, NSDMA = (SU + 63) / 64;   // staged sign words: R rows of 16 dwords (32 pixels x 2 bytes)
  constexpr int WAVE_BYTES = NDMA * 1024 + NIDMA * 256 + NSDMA * 256;
  constexpr int CONST_BYTES = ((IR * IROWB + 15) / 16) * 16;   // an "image" of ones and one of zeros behind the wavefronts' pieces (rows 9 and 10..15 of X)
  const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W, IH0 = 2 * H, IW0 = 2 * W;
  const int kcb = H * W * 16;
  const bool leaky = p.act == ACT_LEAKY;   // uniform
  const unsigned my_addr = (unsigned)(wave * WAVE_BYTES), my_img_addr = my_addr + NDMA * 1024, my_sg_addr = my_img_addr + NIDMA * 256;
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

  // workgroup -> XCD -> a contiguous eighth of the tiles (conv_bf16_staged.hip)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int t8 = (p.ntiles + 7) >> 3, t_end = min(p.ntiles, (xcd + 1) * t8);
  for (int tile = xcd * t8 + slot * NWV + wave; tile < t_end; tile += nslot * NWV) {
    const int b = df_udivm1(tile, p.tiles_per_img, p.m_tpi);
    const int t = tile - b * p.tiles_per_img;
    const int ty = df_udivm1(t, p.tiles_per_row, p.m_tpr), ox0 = (t - ty * p.tiles_per_row) * 32;
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
    }
    const bool right = ox0 + 32 > W;   // (uniform) pixel columns beyond the image: their "gradient" is the halo's, not zero
    df_wait_dma();
    const u32x4* bt = my_tile + g4 * NR * ROWU + c16;   // this lane's pixel of tile row 0 / pixel block 0, channel block g4, tap (0, 0)
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      if (oy0 + rr >= H) break;   // (uniform)
      f32x4 acc[2];
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        acc[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const int ky = tp / 3, kx = tp % 3;
          const bf16x8 bv = __builtin_bit_cast(bf16x8, bt[(rr + ky) * ROWU + 16 * pb + kx]);
          acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, wv[tp], acc[pb], 0, 0, 0);
        }
      }
      // X: row j of the image window for this lane's 2 x 4 pixels, as bf16
      u32x4 xop;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const unsigned char* xa = lds8 + x_addr + rr * (2 * IROWB) + 32 * pb;
        const unsigned d0 = *reinterpret_cast<const unsigned*>(xa), d1 = *reinterpret_cast<const unsigned*>(xa + 4), d2 = *reinterpret_cast<const unsigned*>(xa + 8);
        const unsigned lo4 = __builtin_amdgcn_alignbyte(d1, d0, x_sh), hi4 = __builtin_amdgcn_alignbyte(d2, d1, x_sh);
        const unsigned xs = __builtin_amdgcn_perm(hi4, lo4, 0x06040200u);   // bytes A, A + 2, A + 4, A + 6
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 q0 = {(__bf16)(float)(xs & 0xFFu), (__bf16)(float)((xs >> 8) & 0xFFu)};
        const bf16x2 q1 = {(__bf16)(float)((xs >> 16) & 0xFFu), (__bf16)(float)(xs >> 24)};
        xop[2 * pb] = __builtin_bit_cast(unsigned, q0);
        xop[2 * pb + 1] = __builtin_bit_cast(unsigned, q1);
      }
      // GB: the lane's eight gradients as bf16, split by the sign bit of layer 0's output
      u32x4 gpos, gneg;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        unsigned sw[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
        if (leaky) {
          const unsigned long long s2 = *reinterpret_cast<const unsigned long long*>(lds8 + sg_addr + rr * 64 + 32 * pb);
          sw[0] = (unsigned)s2;
          sw[1] = (unsigned)(s2 >> 32);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const bf16x2 dq = {(__bf16)acc[pb][2 * q], (__bf16)acc[pb][2 * q + 1]};
          unsigned dw = __builtin_bit_cast(unsigned, dq);
          if (right && ox0 + 16 * pb + 4 * g4 + 2 * q >= W) dw = 0u;
          const unsigned m = leaky ? __umul24((sw[q] >> sg_pos) & 0x00010001u, 0xFFFFu) : 0xFFFFFFFFu;
          const unsigned pos = dw & m;
          gpos[2 * pb + q] = pos;
          gneg[2 * pb + q] = dw ^ pos;
        }
      }
      const bf16x8 xo = __builtin_bit_cast(bf16x8, xop);
      d2p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xo, __builtin_bit_cast(bf16x8, gpos), d2p, 0, 0, 0);
      d2n = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xo, __builtin_bit_cast(bf16x8, gneg), d2n, 0, 0, 0);
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
}

// =========================================================================================================
// host side
// =========================================================================================================
// tile height / wavefronts per workgroup (variant builds: build.sh variant TAG conv_first_fused_bwd -DDF_TILE_ROWS=8 ...)
#ifndef DF_TILE_ROWS
#define DF_TILE_ROWS 4
#endif
#ifndef DF_WAVES
#define DF_WAVES 4
#endif
namespace {
constexpr int DF_R = DF_TILE_ROWS, DF_NWV = DF_WAVES;
constexpr int df_lds_bytes(int R, int NWV) {   // the wavefronts' pieces (gradient tile, image window, sign words) + the two constant images
  return NWV * (((4 * (R + 2) * 34 + 63) / 64) * 1024 + (((2 * R + 1) * 18 + 63) / 64) * 256 + ((R * 16 + 63) / 64) * 256) + 2 * ((((2 * R + 1) * 72 + 15) / 16) * 16);
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
int df_grid(int B, int H, int W, int n_cu) {
  const int ntiles = B * cdiv(H, DF_R) * cdiv(W, 32);
  const int lds = df_lds_bytes(DF_R, DF_NWV);
  const int per_cu = max(1, min(16 / DF_NWV, (160 * 1024) / lds));
  const int t8 = cdiv(ntiles, 8);
  return 8 * max(1, min(cdiv(t8, DF_NWV), per_cu * n_cu / 8));
}
}  // namespace

// 1 when the fused sweep takes (layer 1: 3x3, stride 1, Cmid -> Cout1 channels at H x W; layer 0: one uint8 channel, stride 2, Cmid outputs)
extern "C" int yogo_conv2d_dgrad_first_bwd_supported(int Cmid, int Cout1, int H, int W, int B, int act0) {
  if (Cmid != 16 || Cout1 != 32 || H < 1 || W < 2 || (W & 1) || B < 1 || (act0 != ACT_NONE && act0 != ACT_LEAKY)) return 0;
  if ((long long)4 * H * W * 16 >= (1ll << 31) || (long long)4 * H * W >= (1ll << 31) || (long long)B * cdiv(H, DF_R) * cdiv(W, 32) >= (1ll << 31)) return 0;
  const long long tpr = cdiv(W, 32), tpi = (long long)cdiv(H, DF_R) * tpr;
  if (!magic_div_exact((long long)B * tpi - 1, (int)tpi) || !magic_div_exact(tpi, (int)tpr)) return 0;
  return 1;
}

// rows of the partial buffer (cols: yogo_conv_first_bn_wgrad_cols(1, Cmid)) on the current device
extern "C" int yogo_conv2d_dgrad_first_bwd_rows(int B, int H, int W, int* rows) {
  YOGO_CHECK_ARG(rows && B > 0 && H > 0 && W > 0, "conv2d_dgrad_first_bwd_rows: bad arguments");
  const int n_cu = df_n_cu();
  if (n_cu < 0) {
    yogo_set_error("conv2d_dgrad_first_bwd_rows: hipGetDevice failed");
    return YOGO_ERR_HIP;
  }
  *rows = df_grid(B, H, W, n_cu) * DF_NWV;
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
  const int n_cu = df_n_cu();
  if (n_cu < 0) {
    yogo_set_error("conv2d_dgrad_bf16_first_bwd: hipGetDevice failed");
    return YOGO_ERR_HIP;
  }
  DgFirstParams p{};
  p.g = reinterpret_cast<const u32x4*>(g); p.wp = reinterpret_cast<const u32x4*>(packed); p.img = reinterpret_cast<const unsigned char*>(image);
  p.signs = reinterpret_cast<const unsigned short*>(signs); p.part = part;
  p.B = B; p.H = H; p.W = W; p.Mpad = 32; p.act = act0; p.ncol = DF_COUT * DF_PER + DF_NJ + DF_NJ * DF_NJ;
  p.tiles_per_row = cdiv(W, 32); p.tiles_per_img = cdiv(H, DF_R) * p.tiles_per_row; p.ntiles = B * p.tiles_per_img;
  auto magic = [](int d) -> unsigned { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(((1ull << 32) + (unsigned)d - 1ull) / (unsigned)d); };
  p.m_tpr = magic(p.tiles_per_row); p.m_tpi = magic(p.tiles_per_img);
  const int grid = df_grid(B, H, W, n_cu);
  constexpr int lds = df_lds_bytes(DF_R, DF_NWV);
  static_assert(lds <= 160 * 1024, "LDS");
  if (lds > 64 * 1024) {   // more than 64 KB of dynamic LDS has to be asked for: once per device
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < 64 && !done[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_dgrad_first_bwd_kernel<DF_R, DF_NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) {
        yogo_set_error("conv2d_dgrad_bf16_first_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return YOGO_ERR_HIP;
      }
      done[dev] = true;
    }
  }
  hipLaunchKernelGGL((conv_bf16_dgrad_first_bwd_kernel<DF_R, DF_NWV>), dim3(grid), dim3(64 * DF_NWV), lds, stream, p);
  if (yogo_launch_log_enabled())
    yogo_launch_log("conv_bf16_dgrad_first_bwd_kernel<%d, %d> | K=%d M=%d at %dx%d tiles=%d grid=%d lds=%d act0=%d", DF_R, DF_NWV, Cout1, Cmid, H, W, p.ntiles, grid,
                    lds, act0);
  YOGO_CHECK_LAUNCH("conv2d_dgrad_bf16_first_bwd");
  return YOGO_OK;
}
