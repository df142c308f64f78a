// Layer 0 of the bf16 training path on the matrix cores: Conv2d(1 -> <=16, 3x3, stride 2, pad 1) on uint8 images
// (yogo/model_defns.py:34 + the uint8 -> float cast of yogo/model.py:272), with the BatchNorm that follows it
// (model_defns.py:35) either as partial sums (pass A) or applied together with the activation (pass B).
//
// K = 9 taps is short, but the direct VALU kernel (conv_first.hip) spends ~330 lane-instructions per output pixel on 144
// FMAs, nine bounds-checked byte loads and the BatchNorm sums; here a wavefront turns 32 output pixels into ONE
// v_mfma_f32_32x32x16_bf16: K = 16 slots = (row ky, column kx | pad) x 4 rows (the 4th is padding), M = 32 rows of which the
// Cout <= 16 real channels are used, N = 32 pixels.  uint8 inputs are exact in bf16; the weights are rounded to bf16 (what
// autocast does to the reference's conv, yogo/train.py:315-318).  Per pixel the lane work is the operand build (two 16-bit
// loads + three byte->float conversions + two packs per image row) and the epilogue -- about 100 lane-instructions -- so
// the kernel runs at the rate of its stores.
//
// Training forward of a BatchNorm block = two sweeps over the IMAGES instead of one over the images and two over the
// activations: pass A forms the batch statistics without writing anything, pass B recomputes the convolution and writes
// z (saved for backward) and y = act(BN(z)) (the next layer's input) -- 0.1 + 0.1 GB read and 1.6 GB written per 128 images
// where conv + separate BatchNorm apply read 0.9 GB and wrote 1.6 GB.
#include "common.h"
#ifndef CFM_ST_AUX
#define CFM_ST_AUX 2   // non-temporal y / z stores of the pair kernel: 0.243 -> 0.211 ms in the same-box A/B (gpurun_out/r5_nt_ab5.log; A/B variant builds: 0 = cached)
#endif

typedef float cfm_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 cfm_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int cfm_u32x4 __attribute__((ext_vector_type(4)));

struct ConvFirstMfmaParams {
  const unsigned char* in;  // [B][IH][IW] uint8
  const float* w;           // [Cout][3][3] fp32 (rounded to bf16 here)
  const float* bias;        // optional [Cout]
  cfm_u32x4* z;             // optional: conv output, bf16 NCHW8c [B][2][OH*OW] units
  cfm_u32x4* y;             // optional: act(BatchNorm(z)), same layout
  unsigned char* signs;     // optional (with y): [B][OH*OW][2] bytes, bit = (BatchNorm output > 0) -- see yogo_conv_first_mfma_signs
  const float* mean;        // for y: [Cout] each
  const float* invstd;
  const float* gamma;
  const float* beta;
  float* stats_part;        // optional: [workgroups of the grid][16][2] partial (sum, sum of squares) of conv + bias in fp32
  int B, Cout, IH, IW, OH, OW, act;
  int gpi;                  // 32-pixel groups per image
  int total;                // B * gpi
  unsigned m_ow;            // ceil(2^32 / OW)
};

namespace {

// sum over the 32 lanes of each half-wave with DPP adds; the result is valid in lanes 16-31 / 48-63
__device__ __forceinline__ float cfm_half_wave_sum(float v) {
#define CFM_DPP_ADD(CTRL, ROWMASK) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
  CFM_DPP_ADD(0xB1, 0xf)   // quad_perm [1,0,3,2]
  CFM_DPP_ADD(0x4E, 0xf)   // quad_perm [2,3,0,1]
  CFM_DPP_ADD(0x141, 0xf)  // row_half_mirror
  CFM_DPP_ADD(0x140, 0xf)  // row_mirror
  CFM_DPP_ADD(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
#undef CFM_DPP_ADD
  return v;
}

// three bytes (1, 2, 3) of a word as bf16 values: x = [b1 | b2 << 16], y = [b3 | 0]; small integers are exact in bf16, so
// the bf16 pattern is the upper half of the float's
__device__ __forceinline__ void cfm_bytes_to_bf16(unsigned word, unsigned& x, unsigned& y) {
  const unsigned f1 = __builtin_bit_cast(unsigned, (float)((word >> 8) & 0xFFu));
  const unsigned f2 = __builtin_bit_cast(unsigned, (float)((word >> 16) & 0xFFu));
  const unsigned f3 = __builtin_bit_cast(unsigned, (float)(word >> 24));
  x = (f1 >> 16) | (f2 & 0xFFFF0000u);
  y = f3 >> 16;
}

__global__ __launch_bounds__(256) void conv_first_mfma_kernel(const ConvFirstMfmaParams p) {
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave_g = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = gridDim.x * 4;  // (scalar: descriptors stay in SGPRs)
  const int npix = p.OH * p.OW;
  constexpr unsigned OOB = 0x80000000u;

  // A operand: row m = l31 (output channel), K slots 8 * half + j = (ky = slot / 4, kx = slot % 4); kx = 3 and ky = 3 are padding
  cfm_bf16x8 wa;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int slot = 8 * half + j, ky = slot >> 2, kx = slot & 3;
    const bool ok = ky < 3 && kx < 3 && l31 < p.Cout;
    wa[j] = (__bf16)(ok ? p.w[l31 * 9 + min(ky, 2) * 3 + min(kx, 2)] : 0.f);
  }
  // this lane's output channels: 4 * half + i (i < 4) and 8 + 4 * half + i -- the accumulator rows of a 32x32 tile
  float bs[8], sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = (i < 4 ? 0 : 8) + 4 * half + (i & 3);
    const bool ok = ch < p.Cout;
    bs[i] = (ok && p.bias != nullptr) ? p.bias[ch] : 0.f;
    sc[i] = sh[i] = 0.f;  // padding channels: y = act(0) = 0
    if (ok && p.y != nullptr) {  // y = z * sc + sh with sh = beta - mean * sc (the form of bn_apply_act_8c_kernel)
      sc[i] = p.invstd[ch] * p.gamma[ch];
      sh[i] = fmaf(-p.mean[ch], sc[i], p.beta[ch]);
    }
  }
  float s8[8], q8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s8[i] = q8[i] = 0.f;

  // (image, group in the image) of this wavefront's next group, stepped without divisions; the raw image bytes of group
  // i + 1 are requested before group i is multiplied
  int nb = wave_g / p.gpi, ngi = wave_g - nb * p.gpi;
  unsigned n_lo0 = 0, n_hi0 = 0, n_lo1 = 0, n_hi1 = 0;
  int n_pix = 0;
#define CFM_FETCH()                                                                                                    \
  {                                                                                                                    \
    n_pix = ngi * 32 + l31;                                                                                            \
    const bool valid_ = n_pix < npix && nb < p.B;                                                                      \
    const int pc_ = valid_ ? n_pix : npix - 1;                                                                         \
    const int oy_ = (int)__umulhi((unsigned)pc_, p.m_ow), ox_ = pc_ - oy_ * p.OW;                                      \
    /* image rows of this lane's K slots: half 0 -> ky = 0, 1; half 1 -> ky = 2 (and the padding row) */              \
    const int r0_ = 2 * oy_ - 1 + 2 * half;                                                                            \
    const auto rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)min(nb, p.B - 1) * p.IH * p.IW), (short)0, \
                                                       p.IH * p.IW, 0x00020000);                                       \
    const int o0_ = r0_ * p.IW + 2 * ox_ - 2; /* bytes (2ox - 2 .. 2ox + 1) of the row: two aligned 16-bit loads */    \
    const bool ok0_ = valid_ && r0_ >= 0, ok1_ = valid_ && half == 0; /* rows <= IH - 1: even sizes only */            \
    n_lo0 = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs_, (ok0_ && ox_ > 0) ? o0_ : (int)OOB, 0, 0);             \
    n_hi0 = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs_, ok0_ ? o0_ + 2 : (int)OOB, 0, 0);                      \
    n_lo1 = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs_, (ok1_ && ox_ > 0) ? o0_ + p.IW : (int)OOB, 0, 0);      \
    n_hi1 = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs_, ok1_ ? o0_ + p.IW + 2 : (int)OOB, 0, 0);               \
  }
  CFM_FETCH()
  for (int g = wave_g; g < p.total; g += nwaves) {  // g is uniform over the wavefront
    const int b = nb, pix = n_pix;
    const bool valid = pix < npix;
    const unsigned lo0 = n_lo0, hi0 = n_hi0, lo1 = n_lo1, hi1 = n_hi1;
    ngi += nwaves;
    while (ngi >= p.gpi) {
      ngi -= p.gpi;
      ++nb;
    }
    CFM_FETCH()
    unsigned bx, by, bz, bw_;
    cfm_bytes_to_bf16((lo0 & 0xFFFFu) | (hi0 << 16), bx, by);
    cfm_bytes_to_bf16((lo1 & 0xFFFFu) | (hi1 << 16), bz, bw_);
    const cfm_u32x4 bw = {bx, by, bz, bw_};
    cfm_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, __builtin_bit_cast(cfm_bf16x8, bw), acc, 0, 0, 0);

    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = acc[i];
    if (p.bias != nullptr) {  // (uniform)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += bs[i];
    }
    if (p.stats_part != nullptr) {
      const float m = valid ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float vm = v[i] * m;
        s8[i] += vm;
        q8[i] = fmaf(vm, v[i], q8[i]);
      }
    }
    if (p.y == nullptr && p.act != ACT_NONE) {  // inference (BatchNorm folded into w / bias): z = act(conv + bias)
      if (p.act == ACT_LEAKY) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], LEAKY_SLOPE * v[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = act_fwd(v[i], ACT_SILU);
      }
    }
    if (p.z != nullptr || p.y != nullptr) {
      cfm_bf16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
      // lanes 0-31 end up with channel block 0 (channels 0-7) of their pixel, lanes 32-63 with block 1: one 16-byte unit each
      const int vo = valid ? (half * npix + pix) * 16 : (int)OOB;
      if (p.z != nullptr) {
        const cfm_u32x4 w4 = __builtin_bit_cast(cfm_u32x4, o);
        const auto r0s = __builtin_amdgcn_permlane32_swap(w4.x, w4.z, false, false);
        const auto r1s = __builtin_amdgcn_permlane32_swap(w4.y, w4.w, false, false);
        const cfm_u32x4 st = {r0s[0], r1s[0], r0s[1], r1s[1]};
        const auto rs_z = __builtin_amdgcn_make_buffer_rsrc((void*)(p.z + (size_t)b * 2 * npix), (short)0, 2 * npix * 16, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(st, rs_z, vo, 0, 0);
      }
      if (p.y != nullptr) {  // BatchNorm + activation of the ROUNDED z, as yogo_bn_apply_act_bf16 computes it from the stored tensor
        float r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = fmaf((float)o[i], sc[i], sh[i]);
        if (p.signs != nullptr) {  // what the backward pass needs of the pre-activation: one byte per lane (its 8 channels of the pixel)
          unsigned m = 0;   // bit i = (r[i] > 0): compare into vcc, add-with-carry shifts it in (values 7 down to 0) -- 2 instructions per value
#define CFM_SGN(I) "v_cmp_lt_f32_e32 vcc, 0, %" #I "\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
          asm(CFM_SGN(8) CFM_SGN(7) CFM_SGN(6) CFM_SGN(5) CFM_SGN(4) CFM_SGN(3) CFM_SGN(2) "v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
              : "+v"(m)
              : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7])
              : "vcc");
#undef CFM_SGN
          const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)(p.signs + (size_t)b * 2 * npix), (short)0, 2 * npix, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b8((unsigned char)m, rs_s, valid ? 2 * pix + half : (int)OOB, 0, 0);
        }
        if (p.act == ACT_LEAKY) {  // (uniform branches around whole blocks)
#pragma unroll
          for (int i = 0; i < 8; ++i) r[i] = fmaxf(r[i], LEAKY_SLOPE * r[i]);
        } else if (p.act == ACT_SILU) {
#pragma unroll
          for (int i = 0; i < 8; ++i) r[i] = act_fwd(r[i], ACT_SILU);
        }
        cfm_bf16x8 yo;
#pragma unroll
        for (int i = 0; i < 8; ++i) yo[i] = (__bf16)r[i];
        const cfm_u32x4 w4 = __builtin_bit_cast(cfm_u32x4, yo);
        const auto r0s = __builtin_amdgcn_permlane32_swap(w4.x, w4.z, false, false);
        const auto r1s = __builtin_amdgcn_permlane32_swap(w4.y, w4.w, false, false);
        const cfm_u32x4 st = {r0s[0], r1s[0], r0s[1], r1s[1]};
        const auto rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (size_t)b * 2 * npix), (short)0, 2 * npix * 16, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(st, rs_y, vo, 0, 0);
      }
    }
  }
#undef CFM_FETCH
  if (p.stats_part != nullptr) {  // one row of partial sums per workgroup (the four wavefronts meet in LDS)
    __shared__ float red[4][16][2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float s = cfm_half_wave_sum(s8[i]), q = cfm_half_wave_sum(q8[i]);
      if (l31 == 31) {
        const int ch = (i < 4 ? 0 : 8) + 4 * half + (i & 3);
        red[tid >> 6][ch][0] = s;
        red[tid >> 6][ch][1] = q;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int ch = tid >> 1, k = tid & 1;
      p.stats_part[((size_t)blockIdx.x * 16 + ch) * 2 + k] = (red[0][ch][k] + red[1][ch][k]) + (red[2][ch][k] + red[3][ch][k]);
    }
  }
}

// The training sweep (y, optionally z and the sign map; no statistics) with TWO horizontally adjacent output pixels per lane and step
// (output width even): a wavefront takes 64 consecutive pixels as two MFMAs -- the even and the odd ones.  The kernel above issues 4
// 16-bit image loads, a 16-byte store and a byte store per lane and 32 pixels and waits on the CU's vector-memory path; a pixel pair
// (2m, 2m + 1) reads image columns 4m - 1 .. 4m + 3: two ALIGNED dwords per row, and its sign bytes leave as one dword per pair.
// Same arithmetic per element: bit-identical outputs.
__global__ __launch_bounds__(256) void conv_first_mfma2_kernel(const ConvFirstMfmaParams p) {
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave_g = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = gridDim.x * 4;
  const int npix = p.OH * p.OW;
  constexpr unsigned OOB = 0x80000000u;
  cfm_bf16x8 wa;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int slot = 8 * half + j, ky = slot >> 2, kx = slot & 3;
    const bool ok = ky < 3 && kx < 3 && l31 < p.Cout;
    wa[j] = (__bf16)(ok ? p.w[l31 * 9 + min(ky, 2) * 3 + min(kx, 2)] : 0.f);
  }
  float bs[8], sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = (i < 4 ? 0 : 8) + 4 * half + (i & 3);
    const bool ok = ch < p.Cout;
    bs[i] = (ok && p.bias != nullptr) ? p.bias[ch] : 0.f;
    sc[i] = sh[i] = 0.f;
    if (ok && p.y != nullptr) {
      sc[i] = p.invstd[ch] * p.gamma[ch];
      sh[i] = fmaf(-p.mean[ch], sc[i], p.beta[ch]);
    }
  }
  const int gpi2 = (npix + 63) >> 6, total2 = p.B * gpi2;
  int nb = wave_g / gpi2, ngi = wave_g - nb * gpi2;
  unsigned n_a0 = 0, n_b0 = 0, n_a1 = 0, n_b1 = 0;
  int n_pix = 0;
#define CFM2_FETCH()                                                                                                   \
  {                                                                                                                    \
    n_pix = ngi * 64 + 2 * l31;                                                                                        \
    const bool valid_ = n_pix < npix && nb < p.B;                                                                      \
    const int pc_ = valid_ ? n_pix : npix - 2;                                                                         \
    const int oy_ = (int)__umulhi((unsigned)pc_, p.m_ow), ox_ = pc_ - oy_ * p.OW;                                      \
    const int r0_ = 2 * oy_ - 1 + 2 * half;                                                                            \
    const auto rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)min(nb, p.B - 1) * p.IH * p.IW), (short)0, \
                                                       p.IH * p.IW, 0x00020000);                                       \
    const int o0_ = r0_ * p.IW + 2 * ox_; /* columns 4m .. 4m + 3 of the row; 4m - 4 .. 4m - 1 for the left neighbour */ \
    const bool ok0_ = valid_ && r0_ >= 0, ok1_ = valid_ && half == 0;                                                  \
    n_a0 = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_, (ok0_ && ox_ > 0) ? o0_ - 4 : (int)OOB, 0, 0);          \
    n_b0 = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_, ok0_ ? o0_ : (int)OOB, 0, 0);                           \
    n_a1 = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_, (ok1_ && ox_ > 0) ? o0_ + p.IW - 4 : (int)OOB, 0, 0);   \
    n_b1 = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_, ok1_ ? o0_ + p.IW : (int)OOB, 0, 0);                    \
  }
  CFM2_FETCH()
  for (int g = wave_g; g < total2; g += nwaves) {  // g is uniform over the wavefront
    const int b = nb, pix = n_pix;
    const bool valid = pix < npix;
    const unsigned a0 = n_a0, b0 = n_b0, a1 = n_a1, b1 = n_b1;
    ngi += nwaves;
    while (ngi >= gpi2) {
      ngi -= gpi2;
      ++nb;
    }
    CFM2_FETCH()
    unsigned sgn2 = 0;   // this lane's sign bytes of the pair: pixel 2m in bits 0..7, pixel 2m + 1 in bits 16..23
    cfm_u32x4 sty[2];    // [e]: y of pixel 2m + e: lanes 0-31 channel block 0, lanes 32-63 block 1 (after the half-wave exchange)
    cfm_u32x4 stz[2];    // ... and z (the inference forward stores z = act(conv + bias) only)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      // bytes 1, 2, 3 of a word = kernel columns 0, 1, 2: pixel 2m reads (4m - 1, 4m, 4m + 1), pixel 2m + 1 (4m + 1, 4m + 2, 4m + 3)
      const unsigned w0 = e ? b0 : (((a0 >> 24) << 8) | (b0 << 16));
      const unsigned w1 = e ? b1 : (((a1 >> 24) << 8) | (b1 << 16));
      unsigned bx, by, bz, bw_;
      cfm_bytes_to_bf16(w0, bx, by);
      cfm_bytes_to_bf16(w1, bz, bw_);
      const cfm_u32x4 bw = {bx, by, bz, bw_};
      cfm_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, __builtin_bit_cast(cfm_bf16x8, bw), acc, 0, 0, 0);
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = acc[i];
      if (p.bias != nullptr) {  // (uniform)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += bs[i];
      }
      if (p.y == nullptr && p.act != ACT_NONE) {  // inference (BatchNorm folded into w / bias): z = act(conv + bias)
        if (p.act == ACT_LEAKY) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], LEAKY_SLOPE * v[i]);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = act_fwd(v[i], ACT_SILU);
        }
      }
      cfm_bf16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
      if (p.z != nullptr) {
        const cfm_u32x4 w4 = __builtin_bit_cast(cfm_u32x4, o);
        const auto r0s = __builtin_amdgcn_permlane32_swap(w4.x, w4.z, false, false);
        const auto r1s = __builtin_amdgcn_permlane32_swap(w4.y, w4.w, false, false);
        stz[e] = cfm_u32x4{r0s[0], r1s[0], r0s[1], r1s[1]};
      }
      if (p.y == nullptr) continue;   // (uniform)
      float r[8];   // BatchNorm + activation of the ROUNDED z, as yogo_bn_apply_act_bf16 computes it from the stored tensor
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = fmaf((float)o[i], sc[i], sh[i]);
      if (p.signs != nullptr) {
        unsigned m = 0;
#define CFM_SGN(I) "v_cmp_lt_f32_e32 vcc, 0, %" #I "\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
        asm(CFM_SGN(8) CFM_SGN(7) CFM_SGN(6) CFM_SGN(5) CFM_SGN(4) CFM_SGN(3) CFM_SGN(2) "v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
            : "+v"(m)
            : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7])
            : "vcc");
#undef CFM_SGN
        sgn2 |= m << (16 * e);
      }
      if (p.act == ACT_LEAKY) {  // (uniform branches around whole blocks)
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = fmaxf(r[i], LEAKY_SLOPE * r[i]);
      } else if (p.act == ACT_SILU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = act_fwd(r[i], ACT_SILU);
      }
      cfm_bf16x8 yo;
#pragma unroll
      for (int i = 0; i < 8; ++i) yo[i] = (__bf16)r[i];
      const cfm_u32x4 w4 = __builtin_bit_cast(cfm_u32x4, yo);
      const auto r0s = __builtin_amdgcn_permlane32_swap(w4.x, w4.z, false, false);
      const auto r1s = __builtin_amdgcn_permlane32_swap(w4.y, w4.w, false, false);
      sty[e] = cfm_u32x4{r0s[0], r1s[0], r0s[1], r1s[1]};
    }
    if (p.z != nullptr) {   // (uniform) the pair's z units as two contiguous kilobytes, non-temporal: the same exchange as for y below
      const auto x0 = __builtin_amdgcn_permlane32_swap(stz[0].x, stz[1].x, false, false);
      const auto x1 = __builtin_amdgcn_permlane32_swap(stz[0].y, stz[1].y, false, false);
      const auto x2 = __builtin_amdgcn_permlane32_swap(stz[0].z, stz[1].z, false, false);
      const auto x3 = __builtin_amdgcn_permlane32_swap(stz[0].w, stz[1].w, false, false);
      const cfm_u32x4 se0 = {x0[0], x1[0], x2[0], x3[0]}, se1 = {x0[1], x1[1], x2[1], x3[1]};
      const auto rs_z = __builtin_amdgcn_make_buffer_rsrc((void*)(p.z + (size_t)b * 2 * npix), (short)0, 2 * npix * 16, 0x00020000);
      const int vdz = valid ? (pix + half) * 16 : (int)OOB;
      __builtin_amdgcn_raw_buffer_store_b128(se0, rs_z, vdz, 0, CFM_ST_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(se1, rs_z, valid ? vdz + npix * 16 : (int)OOB, 0, CFM_ST_AUX);
    }
#ifndef CFM_DENSE
#define CFM_DENSE 1   // (A/B variant builds: 0 = one half-filled store per pixel parity)
#endif
    if (p.y != nullptr && !CFM_DENSE) {
      const auto rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (size_t)b * 2 * npix), (short)0, 2 * npix * 16, 0x00020000);
#pragma unroll
      for (int e = 0; e < 2; ++e) __builtin_amdgcn_raw_buffer_store_b128(sty[e], rs_y, valid ? (half * npix + pix + e) * 16 : (int)OOB, 0, 0);
    } else
    if (p.y != nullptr) {   // (uniform)
      // A lane's two pixels are neighbours, so the store of pixel parity e alone fills HALF of every 64-byte segment it touches (16 bytes
      // every 32).  A second half-wave exchange (parity 0's block-1 half trades places with parity 1's block-0 half) turns the two
      // half-filled stores into two CONTIGUOUS kilobytes: lanes 0-31 pixel 2m, lanes 32-63 pixel 2m + 1 of one channel block
      // (measured first on the stride-2 data gradients, conv_bf16_direct.hip: -12 %).
      const auto x0 = __builtin_amdgcn_permlane32_swap(sty[0].x, sty[1].x, false, false);
      const auto x1 = __builtin_amdgcn_permlane32_swap(sty[0].y, sty[1].y, false, false);
      const auto x2 = __builtin_amdgcn_permlane32_swap(sty[0].z, sty[1].z, false, false);
      const auto x3 = __builtin_amdgcn_permlane32_swap(sty[0].w, sty[1].w, false, false);
      const cfm_u32x4 se0 = {x0[0], x1[0], x2[0], x3[0]}, se1 = {x0[1], x1[1], x2[1], x3[1]};
      const auto rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (size_t)b * 2 * npix), (short)0, 2 * npix * 16, 0x00020000);
      const int vd = valid ? (pix + half) * 16 : (int)OOB;   // (npix is even: a pair is valid or not as a whole)
      __builtin_amdgcn_raw_buffer_store_b128(se0, rs_y, vd, 0, CFM_ST_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(se1, rs_y, valid ? vd + npix * 16 : (int)OOB, 0, CFM_ST_AUX);   // (vector offset: conv_bf16_epi_groups.inc on scalar store offsets)
    }
    if (p.signs != nullptr) {
      // a pixel's two bytes are (half 0, half 1): the partner lane's pair comes over the crossbar, the lower half-wave stores the pair's
      // four bytes as one dword
      const unsigned other = (unsigned)__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, (int)sgn2);
      const unsigned word = (sgn2 & 0xFFu) | ((other & 0xFFu) << 8) | ((sgn2 & 0xFF0000u)) | ((other & 0xFF0000u) << 8);
      const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)(p.signs + (size_t)b * 2 * npix), (short)0, 2 * npix, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b32(word, rs_s, (valid && half == 0) ? 2 * pix : (int)OOB, 0, 0);
    }
  }
#undef CFM2_FETCH
}

// ---- batch statistics without the convolution: P = sum patch (9), G = sum patch (x) patch (9 x 9) --------------------------
// z_c = w_c . patch + b_c is linear in the patch, so  sum z_c = w_c . P + N b_c  and  sum (z_c - b_c)^2 = w_c^T G w_c:  the
// statistics of ALL channels follow from 54 channel-independent sums over the uint8 image.  They are accumulated as INTEGERS
// (32 pixels per lane: <= 2.1e6 per accumulator; a workgroup: <= 5.4e8), so G and P are exact and the variance comes out of
// a double-precision quadratic form -- more accurate than summing rounded fp32 outputs.  The layer-0 backward pass needs the
// same G (conv_first_bn_wgrad_kernel) and takes it from here instead of sweeping the image again.
#define CFG_PPT 32
__device__ __forceinline__ unsigned cfm_wave_sum_u32(unsigned v) {
#define CFM_DPP_ADDU(CTRL, ROWMASK) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);
  CFM_DPP_ADDU(0xB1, 0xf)
  CFM_DPP_ADDU(0x4E, 0xf)
  CFM_DPP_ADDU(0x141, 0xf)
  CFM_DPP_ADDU(0x140, 0xf)
  CFM_DPP_ADDU(0x142, 0xa)
  CFM_DPP_ADDU(0x143, 0xc)
#undef CFM_DPP_ADDU
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// part: [gridDim.y * gridDim.x][54] uint32 = P[9], then the upper triangle of G row by row
__global__ __launch_bounds__(256) void conv_first_gram_kernel(const unsigned char* __restrict__ in, unsigned* __restrict__ part, int IH,
                                                              int IW, int OH, int OW) {
  __shared__ unsigned red[4][54];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, npix = OH * OW;
  const unsigned char* ib = in + (size_t)b * IH * IW;
  unsigned acc[54];
#pragma unroll
  for (int q = 0; q < 54; ++q) acc[q] = 0u;
  const int pbase = blockIdx.x * (256 * CFG_PPT);
  for (int k = 0; k < CFG_PPT; ++k) {
    const int pix = pbase + k * 256 + tid;
    const bool ok = pix < npix;
    const int pc = ok ? pix : 0;
    const int oy = pc / OW, ox = pc - oy * OW;
    unsigned x[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = 2 * oy + kh - 1;  // <= IH - 1 (even sizes)
      const bool rok = ok && iy >= 0;
      const int o = rok ? iy * IW + 2 * ox : 0;  // bytes o - 2 .. o + 1
      const unsigned hi = *reinterpret_cast<const unsigned short*>(ib + o);
      const unsigned lo = *reinterpret_cast<const unsigned short*>(ib + ((rok && ox > 0) ? o - 2 : o));
      x[kh * 3 + 0] = (rok && ox > 0) ? (lo >> 8) : 0u;
      x[kh * 3 + 1] = rok ? (hi & 0xFFu) : 0u;
      x[kh * 3 + 2] = rok ? (hi >> 8) : 0u;
    }
    int q = 9;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      acc[j] += x[j];
#pragma unroll
      for (int j2 = j; j2 < 9; ++j2) {
        acc[q] = __umul24(x[j], x[j2]) + acc[q];
        ++q;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 54; ++q) {
    const unsigned v = cfm_wave_sum_u32(acc[q]);
    if (lane == 0) red[wave][q] = v;
  }
  __syncthreads();
  if (tid < 54) part[((size_t)b * gridDim.x + blockIdx.x) * 54 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// one workgroup per partial column: exact 64-bit sum over the rows -> gram (double) and gram_f32, both [90] = P[9], G[9][9]
// The same sums with FOUR pixels per integer instruction (output width a multiple of 4, image rows 4-byte aligned): a lane takes
// groups of 4 adjacent output pixels, builds for every tap ONE word holding that tap's byte of the 4 pixels (three aligned dword
// loads per image row, v_perm_b32 picks the even / odd bytes, v_alignbit_b32 shifts the left neighbour in) and accumulates
// P[j] += dot4(X_j, 1), G[j][j2] += dot4(X_j, X_j2) with v_dot4_u32_u8 -- 54 dot products per 4 pixels where the kernel above
// spends 4 x 54 multiply-adds.  That kernel is bound by vector-instruction issue (0.88 of its SIMD cycles,
// profiles/r03_issue_util.txt).  Exact integers either way: same partial rows, same fold.
__global__ __launch_bounds__(256) void conv_first_gram4_kernel(const unsigned char* __restrict__ in, unsigned* __restrict__ part, int IH,
                                                               int IW, int OH, int OW) {
  __shared__ unsigned red[4][54];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, owg = OW >> 2, ngroups = OH * owg;
  const unsigned char* ib = in + (size_t)b * IH * IW;
  unsigned acc[54];
#pragma unroll
  for (int q = 0; q < 54; ++q) acc[q] = 0u;
  const int gbase = blockIdx.x * (256 * (CFG_PPT / 4));
  for (int k = 0; k < CFG_PPT / 4; ++k) {
    const int gi = gbase + k * 256 + tid;
    const bool ok = gi < ngroups;
    const int gc = ok ? gi : 0;
    const int oy = gc / owg, ox0 = 4 * (gc - oy * owg);
    unsigned x[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = 2 * oy + kh - 1;  // <= IH - 1 (even sizes)
      const bool rok = ok && iy >= 0;
      const unsigned* row = reinterpret_cast<const unsigned*>(ib + (size_t)(rok ? iy : 0) * IW + 2 * ox0);   // 4-byte aligned
      const unsigned d1 = rok ? row[0] : 0u, d2 = rok ? row[1] : 0u;            // input columns 2 ox0 .. 2 ox0 + 7
      const unsigned d0 = (rok && ox0 > 0) ? row[-1] : 0u;                      // ... 2 ox0 - 4 .. 2 ox0 - 1 (left border: zeros)
      const unsigned odd = __builtin_amdgcn_perm(d2, d1, 0x07050301u);          // columns +1 +3 +5 +7  = tap kx = 2 of the 4 pixels
      x[kh * 3 + 1] = __builtin_amdgcn_perm(d2, d1, 0x06040200u);               // columns +0 +2 +4 +6  = tap kx = 1
      x[kh * 3 + 2] = odd;
      x[kh * 3 + 0] = __builtin_amdgcn_alignbit(odd, d0, 24);                   // columns -1 +1 +3 +5  = tap kx = 0
    }
    int q = 9;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      acc[j] = __builtin_amdgcn_udot4(x[j], 0x01010101u, acc[j], false);
#pragma unroll
      for (int j2 = j; j2 < 9; ++j2) {
        acc[q] = __builtin_amdgcn_udot4(x[j], x[j2], acc[q], false);
        ++q;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 54; ++q) {
    const unsigned v = cfm_wave_sum_u32(acc[q]);
    if (lane == 0) red[wave][q] = v;
  }
  __syncthreads();
  if (tid < 54) part[((size_t)b * gridDim.x + blockIdx.x) * 54 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

__global__ __launch_bounds__(256) void conv_first_gram_fold_kernel(const unsigned* __restrict__ part, int rows, double* __restrict__ gram,
                                                                   float* __restrict__ gram_f32) {
  __shared__ unsigned long long sh[256];
  const int q = blockIdx.x;
  unsigned long long s = 0ull;
  for (int r = threadIdx.x; r < rows; r += 256) s += part[(size_t)r * 54 + q];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double v = (double)sh[0];
    if (q < 9) {
      gram[q] = v;
      gram_f32[q] = (float)v;
    } else {  // packed upper triangle -> both halves of the full matrix
      int t = q - 9, j = 0;
      while (t >= 9 - j) {
        t -= 9 - j;
        ++j;
      }
      const int j2 = j + t;
      gram[9 + j * 9 + j2] = v; gram[9 + j2 * 9 + j] = v;
      gram_f32[9 + j * 9 + j2] = (float)v; gram_f32[9 + j2 * 9 + j] = (float)v;
    }
  }
}

// mean / invstd / running statistics of conv(x, bf16(w)) + bias from P and G (the semantics of bn_finalize_kernel)
__global__ void bn_stats_from_gram_kernel(const double* __restrict__ gram, const float* __restrict__ w, const float* __restrict__ bias,
                                          int Cout, double count, float eps, float momentum, float* __restrict__ mean_out,
                                          float* __restrict__ invstd_out, float* __restrict__ running_mean,
                                          float* __restrict__ running_var, long long* __restrict__ num_batches_tracked) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cout) return;
  double wv[9];
  for (int j = 0; j < 9; ++j) wv[j] = (double)(float)(__bf16)w[c * 9 + j];
  double sp = 0.0, quad = 0.0;
  for (int j = 0; j < 9; ++j) {
    sp += wv[j] * gram[j];
    double row = 0.0;
    for (int k = 0; k < 9; ++k) row += wv[k] * gram[9 + j * 9 + k];
    quad += wv[j] * row;
  }
  const double m0 = sp / count;                      // mean of the bias-free convolution
  double var = quad / count - m0 * m0;
  if (var < 0.0) var = 0.0;
  const double mean = m0 + (bias != nullptr ? (double)bias[c] : 0.0);
  mean_out[c] = (float)mean;
  invstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean != nullptr) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
    running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
  }
  if (c == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;
}

int cfm_grid(int total) { return max(1, min(1024, cdiv(total, 4 * 8))); }  // >= 8 groups per wavefront when there is work
unsigned cfm_magic(int d) { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(((1ull << 32) + (unsigned)d - 1ull) / (unsigned)d); }

}  // namespace

// 1 when the matrix-core kernel takes the shape (else use yogo_conv_first_fwd_train_bf16 + yogo_bn_apply_act_bf16)
extern "C" int yogo_conv_first_mfma_supported(int in_dtype, int Cin, int Cout, int IH, int IW, int stride) {
  return in_dtype == 0 && Cin == 1 && Cout >= 1 && Cout <= 16 && stride == 2 && IH >= 4 && IW >= 4 && IH % 2 == 0 && IW % 2 == 0 &&
         (long long)IH * IW < (1ll << 30) && (IH / 2) * (IW / 2) > 1;
}

// rows of the BatchNorm partial-sum buffer ([rows][16][2]) a launch with stats_part fills; row stride (mpad) is 16
extern "C" int yogo_conv_first_mfma_stats_rows(int B, int IH, int IW, int* rows) {
  YOGO_CHECK_ARG(rows && B >= 0 && IH > 0 && IW > 0, "conv_first_mfma_stats_rows: bad arguments");
  *rows = cfm_grid(B * cdiv((IH / 2) * (IW / 2), 32));
  return YOGO_OK;
}

// in: uint8 [B][1][IH][IW]; w: fp32 [Cout][1][3][3] (rounded to bf16 inside); any of the outputs may be NULL:
//   stats_part: partial (sum, sumsq) of conv + bias (fp32, before rounding) -> yogo_bn_finalize(part, rows, 16, ...)
//   z: conv + bias in bf16 NCHW8c;  y: act((z - mean) * invstd * gamma + beta) in bf16 NCHW8c (needs mean/invstd/gamma/beta);
//   without y the activation goes onto z (inference with BatchNorm folded into w and bias): z = act(conv + bias)
// two pixels per lane where the output width is even; a switch only in the test-hooks / diagnostic builds (see conv_bf16.hip)
#if defined(YOGO_TEST_HOOKS) || defined(YOGO_DIAG)
static bool g_cfm_pairs = true;
extern "C" int yogo_hook_conv_first_mfma_pairs(int on) {
  g_cfm_pairs = on != 0;
  return YOGO_OK;
}
#else
static constexpr bool g_cfm_pairs = true;
#endif
static int conv_first_mfma_impl(const void* in, const float* w, const float* bias, void* z, void* y, void* signs, const float* mean,
                               const float* invstd, const float* gamma, const float* beta, float* stats_part, int B, int Cout,
                               int IH, int IW, int act, hipStream_t stream);
extern "C" int yogo_conv_first_mfma(const void* in, const float* w, const float* bias, void* z, void* y, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta, float* stats_part, int B, int Cout,
                                    int IH, int IW, int act, hipStream_t stream) {
  return conv_first_mfma_impl(in, w, bias, z, y, nullptr, mean, invstd, gamma, beta, stats_part, B, Cout, IH, IW, act, stream);
}
// the same with the SIGN MAP of the BatchNorm output beside y: signs = [B][OH*OW][2] bytes; byte h of a pixel, bit i = (pre-activation
// of channel 4h + i > 0), bit 4 + i = (... of channel 8 + 4h + i > 0) -- the lane order of the kernel, one byte store per lane.  It is
// all the layer-0 backward pass needs of z (yogo_conv_first_bn_wgrad_bf16_xs): with it z (16x the bytes) need not be written at all.
extern "C" int yogo_conv_first_mfma_signs(const void* in, const float* w, const float* bias, void* z, void* y, void* signs,
                                          const float* mean, const float* invstd, const float* gamma, const float* beta, int B, int Cout,
                                          int IH, int IW, int act, hipStream_t stream) {
  YOGO_CHECK_ARG(y && signs, "conv_first_mfma_signs: the sign map goes with y");
  return conv_first_mfma_impl(in, w, bias, z, y, signs, mean, invstd, gamma, beta, nullptr, B, Cout, IH, IW, act, stream);
}
static int conv_first_mfma_impl(const void* in, const float* w, const float* bias, void* z, void* y, void* signs, const float* mean,
                               const float* invstd, const float* gamma, const float* beta, float* stats_part, int B, int Cout,
                               int IH, int IW, int act, hipStream_t stream) {
  YOGO_CHECK_ARG(in && w && (z || y || stats_part), "conv_first_mfma: null pointer");
  YOGO_CHECK_ARG(yogo_conv_first_mfma_supported(0, 1, Cout, IH, IW, 2) && B >= 0, "conv_first_mfma: unsupported shape %dx%d Cout=%d", IH, IW, Cout);
  YOGO_CHECK_ARG(y == nullptr || (mean && invstd && gamma && beta), "conv_first_mfma: y needs mean / invstd / gamma / beta");
  if (B == 0) return YOGO_OK;
  ConvFirstMfmaParams p{};
  p.in = reinterpret_cast<const unsigned char*>(in); p.w = w; p.bias = bias;
  p.z = reinterpret_cast<cfm_u32x4*>(z); p.y = reinterpret_cast<cfm_u32x4*>(y); p.signs = reinterpret_cast<unsigned char*>(signs);
  p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta; p.stats_part = stats_part;
  p.B = B; p.Cout = Cout; p.IH = IH; p.IW = IW; p.OH = IH / 2; p.OW = IW / 2; p.act = act;
  p.gpi = cdiv(p.OH * p.OW, 32); p.total = B * p.gpi;
  p.m_ow = cfm_magic(p.OW);
  YOGO_CHECK_ARG((long long)p.total * 32 < (1ll << 31) && (long long)p.OH * p.OW * p.OW < (1ll << 32), "conv_first_mfma: batch / image too large");
  // the sweeps that write tensors (training: y [+ z] [+ sign map]; inference: z = act(conv + bias)) at an even output width: two pixels per lane
  if (g_cfm_pairs && (y != nullptr || z != nullptr) && stats_part == nullptr && p.OW % 2 == 0 && p.OW >= 4)
    hipLaunchKernelGGL(conv_first_mfma2_kernel, dim3(cfm_grid(B * cdiv(p.OH * p.OW, 64))), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(conv_first_mfma_kernel, dim3(cfm_grid(p.total)), dim3(256), 0, stream, p);
  YOGO_CHECK_LAUNCH("conv_first_mfma");
  return YOGO_OK;
}

// P = sum of the 3x3 stride-2 patches, G = sum of their outer products over a uint8 batch [B][1][IH][IW] (even sizes), exact:
// part = scratch of yogo_conv_first_gram_rows x 54 uint32; gram = double[90], gram_f32 = float[90] (P[9], then G[9][9])
extern "C" int yogo_conv_first_gram_rows(int B, int IH, int IW, int* rows) {
  YOGO_CHECK_ARG(rows && B >= 0 && IH > 0 && IW > 0, "conv_first_gram_rows: bad arguments");
  *rows = B * cdiv((IH / 2) * (IW / 2), 256 * CFG_PPT);
  return YOGO_OK;
}
extern "C" int yogo_conv_first_gram(const void* in, void* part, double* gram, float* gram_f32, int B, int IH, int IW, hipStream_t stream) {
  YOGO_CHECK_ARG(in && part && gram && gram_f32 && B > 0, "conv_first_gram: bad arguments");
  YOGO_CHECK_ARG(yogo_conv_first_mfma_supported(0, 1, 1, IH, IW, 2) && B <= 65535, "conv_first_gram: unsupported shape %dx%d", IH, IW);
  const int OH = IH / 2, OW = IW / 2, tiles = cdiv(OH * OW, 256 * CFG_PPT);
  // four pixels per integer instruction when a group of 4 output pixels is three aligned dwords of an image row
  if (OW % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 3) == 0) {
    hipLaunchKernelGGL(conv_first_gram4_kernel, dim3(tiles, B), dim3(256), 0, stream, reinterpret_cast<const unsigned char*>(in),
                       reinterpret_cast<unsigned*>(part), IH, IW, OH, OW);
  } else {
    hipLaunchKernelGGL(conv_first_gram_kernel, dim3(tiles, B), dim3(256), 0, stream, reinterpret_cast<const unsigned char*>(in),
                       reinterpret_cast<unsigned*>(part), IH, IW, OH, OW);
  }
  hipLaunchKernelGGL(conv_first_gram_fold_kernel, dim3(54), dim3(256), 0, stream, reinterpret_cast<const unsigned*>(part), tiles * B, gram,
                     gram_f32);
  YOGO_CHECK_LAUNCH("conv_first_gram");
  return YOGO_OK;
}
// BatchNorm batch statistics of conv(x, bf16(w)) + bias for all Cout channels from the Gram sums (count = B * OH * OW);
// running statistics and num_batches_tracked are updated like yogo_bn_finalize does
extern "C" int yogo_bn_stats_from_gram(const double* gram, const float* w, const float* bias, int Cout, long long count, float eps,
                                       float momentum, float* mean_out, float* invstd_out, float* running_mean, float* running_var,
                                       long long* num_batches_tracked, hipStream_t stream) {
  YOGO_CHECK_ARG(gram && w && mean_out && invstd_out && Cout > 0 && count > 0, "bn_stats_from_gram: bad arguments");
  hipLaunchKernelGGL(bn_stats_from_gram_kernel, dim3(cdiv(Cout, 64)), dim3(64), 0, stream, gram, w, bias, Cout, (double)count, eps, momentum,
                     mean_out, invstd_out, running_mean, running_var, num_batches_tracked);
  YOGO_CHECK_LAUNCH("bn_stats_from_gram");
  return YOGO_OK;
}
