// The forward of the 1x1 head of the network (yogo/model_defns.py:66, conv_block_8: 128 -> 5 + num_classes) as independent wavefronts with the
// weights in REGISTERS: a 1x1 convolution has one tap, so the pixel operands come straight from global memory with no amplification at all
// (one 16-byte unit per lane and 16-channel step) and the whole weight matrix of a wavefront is 8 operand quads.
//   conv_bf16_1x1_f32_kernel    bf16 NCHW8c [B][K <= 128][plane] -> fp32 NCHW [B][M <= 32][plane] (+ bias): what decode / loss / NMS read
// As tiles of conv_bf16_kernel this launch ran at 3.6 TB/s (8 chunk round trips per tile for 0.04 GFLOP per image); here every wavefront streams
// tiles of 32 pixels: 8 loads, 8 MFMAs, 12 stores.  Same products in the same order, the same epilogue formula (bit-identical); same-box A/B
// -18.6 % (0.118 -> 0.096 ms, gpurun_out/r5_hd_ab2.log).  The head's DATA GRADIENT in the same form (1 load, 4 MFMAs, 8 stores per tile) was
// 6 - 12 % slower than the tiled kernel (same log) and is not in the tree.
//   DEC = true (round 6): the box decode of yogo/model.py:277-313 applied to the pixel's 5 + C outputs where they sit in registers (SURVEY.md 8b,
//   head1x1_decode_fwd): sigmoid / exp / softmax and the grid offset with the expressions of decode_fwd_kernel (decode_loss.hip) in its order
//   of operations -- this file is built with -ffp-contract=off like that one -- so `model(x)` in eval mode gets the bits the two launches gave
//   and the raw head output never goes through memory.
#include "common.h"
#include <mutex>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ConvHeadParams {
  const u32x4* in;     // bf16 NCHW8c [B][Kb][plane] units
  const u32x4* wp;     // packed weights [1][Kb][Mpad] units
  const float* bias;   // [M] or null
  float* out_f32;      // [B][M][plane]
  int B, Kb, M, plane;
  int tiles_per_img, ntiles;
  unsigned m_tpi;
  // DEC: the decode's operands (yogo_decode_fwd)
  const float* cxs;    // [plane]
  const float* cys;    // [plane]
  float inv_sx, inv_sy, anchor_w, anchor_h, wmul, hmul;
  int inference;       // class channels: softmax (1) or raw logits (0)
};

namespace {
__device__ __forceinline__ int ch_udivm1(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }
unsigned ch_magic(int d) { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(((1ull << 32) + (unsigned)d - 1ull) / (unsigned)d); }
int ch_n_cu() {
  static std::mutex mu;
  static int n_cu_of[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    yogo_set_error("conv_bf16_head: hipGetDevice failed");
    return -1;
  }
  std::lock_guard<std::mutex> lk(mu);
  if (n_cu_of[dev] == 0) {
    hipDeviceProp_t prop;
    n_cu_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n_cu_of[dev];
}
}  // namespace

__device__ __forceinline__ float ch_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }   // (decode_loss.hip: sigmoidf_)

// NK: 16-channel steps of the contraction (K / 16, <= 8); M <= 32 (DEC: 6 <= M <= 16)
template <int NK, bool DEC = false>
__global__ __launch_bounds__(256) void conv_bf16_1x1_f32_kernel(const ConvHeadParams p) {
  constexpr int OOB = (int)0x80000000u;
  constexpr int KB = 2 * NK;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = p.plane, plane16 = plane * 16;
  u32x4 Aq[NK];   // weight unit [2 kc + half][l31] (Mpad = 32)
#pragma unroll
  for (int kc = 0; kc < NK; ++kc) Aq[kc] = p.wp[(2 * kc + half) * 32 + l31];
  float bs[8];    // bias of this lane's channels: 4 half + i and 8 + 4 half + i
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = 8 * (i >> 2) + 4 * half + (i & 3);
    bs[i] = (p.bias != nullptr && ch < p.M) ? p.bias[ch] : 0.f;
  }
  float bs2[8];   // channels 16 + ... (M > 16)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = 16 + 8 * (i >> 2) + 4 * half + (i & 3);
    bs2[i] = (p.bias != nullptr && ch < p.M) ? p.bias[ch] : 0.f;
  }
  const int nwaves = gridDim.x * 4;
  for (int tile = blockIdx.x * 4 + wave; tile < p.ntiles; tile += nwaves) {
    const int b = ch_udivm1(tile, p.tiles_per_img, p.m_tpi);
    const int px = (tile - b * p.tiles_per_img) * 32 + l31;
    const bool ov = px < plane;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)b * KB * plane), (short)0, KB * plane16, 0x00020000);
    const int v0 = ov ? (half * plane + px) * 16 : OOB;
    u32x4 Bq[NK];
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) Bq[kc] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v0, (2 * kc) * plane16, 0));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < NK; ++kc)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Aq[kc]), __builtin_bit_cast(bf16x8, Bq[kc]), acc, 0, 0, 0);
    // fp32 NCHW: a channel's 32 pixels are 128 contiguous bytes
    float* ob = p.out_f32 + (size_t)b * p.M * plane + px;
    if constexpr (DEC) {
      // this lane's channels of pixel px: half 0 -> 0..3 (the box) and 8..11, half 1 -> 4 (objectness), 5..7 and 12..15; class c = channel 5 + c
      const int M = p.M;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = acc[i] + bs[i];
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = v[i];
      const int pxc = ov ? px : 0;
      if (half == 0) {
        o[0] = p.inv_sx * ch_sigmoid(v[0]) + p.cxs[pxc];
        o[1] = p.inv_sy * ch_sigmoid(v[1]) + p.cys[pxc];
        o[2] = p.anchor_w * expf(fminf(v[2], 80.f)) * p.wmul;
        o[3] = p.anchor_h * expf(fminf(v[3], 80.f)) * p.hmul;
      } else {
        o[0] = ch_sigmoid(v[0]);
      }
      if (p.inference) {
        // softmax over the class channels in decode_fwd_kernel's order: max over all, then sum += expf(v - mx) channel by channel -- the
        // running sum crosses the half-wavefronts where the channel order does (5..7 | 8..11 | 12..15)
        const int i0 = half == 0 ? 4 : 1;   // this lane's first class slot (half 1: slots 1..3 = channels 5..7, slots 4..7 = 12..15)
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int ch = 8 * (i >> 2) + 4 * half + (i & 3);
          if (i >= i0 && ch < M) mx = fmaxf(mx, v[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = expf(v[i] - mx);
        float sum = 0.f;
        if (half == 1) {
#pragma unroll
          for (int i = 1; i < 4; ++i)
            if (4 + i < M) sum += e[i];                 // channels 5, 6, 7
        }
        const float fromA = __shfl_xor(sum, 32, 64);    // half 0 takes the sum over 5..7
        if (half == 0) {
          sum = fromA;
#pragma unroll
          for (int i = 4; i < 8; ++i)
            if (4 + i < M) sum += e[i];                 // channels 8..11
        }
        const float fromB = __shfl_xor(sum, 32, 64);    // half 1 takes the sum over 5..11
        if (half == 1) {
          sum = fromB;
#pragma unroll
          for (int i = 4; i < 8; ++i)
            if (8 + i < M) sum += e[i];                 // channels 12..15
        }
        const float tot = __shfl_xor(sum, 32, 64);      // the complete sum sits on half 1
        if (half == 0) sum = tot;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i >= i0) o[i] = e[i] / sum;
      }
      if (ov) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int ch = 8 * (i >> 2) + 4 * half + (i & 3);
          if (ch < M) ob[(size_t)ch * plane] = o[i];
        }
      }
      continue;
    }
    if (ov) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ch = 8 * (i >> 2) + 4 * half + (i & 3);
        if (ch < p.M) ob[(size_t)ch * plane] = acc[i] + bs[i];
      }
      if (p.M > 16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int ch = 16 + 8 * (i >> 2) + 4 * half + (i & 3);
          if (ch < p.M) ob[(size_t)ch * plane] = acc[8 + i] + bs2[i];
        }
      }
    }
  }
}

bool conv_bf16_head_fwd_eligible(int K, int M, int plane, int B) {
  return K >= 16 && K <= 128 && K % 16 == 0 && M >= 1 && M <= 32 && B > 0 && plane > 0 && (long long)(K / 8) * plane * 16 < (1ll << 31) &&
         (long long)B * ((plane + 31) / 32) < (1ll << 31) && magic_div_exact((long long)B * ((plane + 31) / 32) - 1, (plane + 31) / 32);   // (tile -> image by multiplication)
}
struct ConvHeadDecode {   // the decode's operands (null cxs: no decode)
  const float* cxs;
  const float* cys;
  float inv_sx, inv_sy, anchor_w, anchor_h, wmul, hmul;
  int inference;
};
static int launch_head(const void* in, const void* packed, const float* bias, float* out_f32, int B, int K, int M, int plane, const ConvHeadDecode* dec,
                       hipStream_t stream) {
  const int n_cu = ch_n_cu();
  if (n_cu < 0) return YOGO_ERR_HIP;
  ConvHeadParams p{};
  p.in = reinterpret_cast<const u32x4*>(in); p.wp = reinterpret_cast<const u32x4*>(packed); p.bias = bias; p.out_f32 = out_f32;
  p.B = B; p.Kb = K / 8; p.M = M; p.plane = plane;
  p.tiles_per_img = cdiv(plane, 32); p.ntiles = B * p.tiles_per_img; p.m_tpi = ch_magic(p.tiles_per_img);
  if (dec != nullptr) {
    p.cxs = dec->cxs; p.cys = dec->cys; p.inv_sx = dec->inv_sx; p.inv_sy = dec->inv_sy; p.anchor_w = dec->anchor_w; p.anchor_h = dec->anchor_h;
    p.wmul = dec->wmul; p.hmul = dec->hmul; p.inference = dec->inference;
  }
  if (p.ntiles <= 0) return YOGO_OK;
  const int grid = min(cdiv(p.ntiles, 4), 4 * n_cu);
  switch (K / 16) {
#define CH_CASE(NK)                                                                                                             \
  case NK:                                                                                                                      \
    if (dec != nullptr) hipLaunchKernelGGL((conv_bf16_1x1_f32_kernel<NK, true>), dim3(grid), dim3(256), 0, stream, p);           \
    else hipLaunchKernelGGL((conv_bf16_1x1_f32_kernel<NK, false>), dim3(grid), dim3(256), 0, stream, p);                         \
    break;
    CH_CASE(1) CH_CASE(2) CH_CASE(3) CH_CASE(4) CH_CASE(5) CH_CASE(6) CH_CASE(7) CH_CASE(8)
#undef CH_CASE
    default: yogo_set_error("conv_bf16_head_fwd: K = %d", K); return YOGO_ERR_ARG;
  }
  if (yogo_launch_log_enabled())
    yogo_launch_log("conv_bf16_1x1_f32_kernel<%d, %s> | K=%d M=%d plane=%d tiles=%d grid=%d bias=%d", K / 16, dec != nullptr ? "true" : "false", K, M, plane, p.ntiles,
                    grid, bias != nullptr);
  YOGO_CHECK_LAUNCH("conv_bf16_head_fwd");
  return YOGO_OK;
}
int launch_conv_bf16_head_fwd(const void* in, const void* packed, const float* bias, float* out_f32, int B, int K, int M, int plane, hipStream_t stream) {
  return launch_head(in, packed, bias, out_f32, B, K, M, plane, nullptr, stream);
}

// The 1x1 head + box decode of eval-mode `model(x)` in one launch (yogo/model.py:275-313: self.model(x) then the decode; SURVEY.md 8b
// head1x1_decode_fwd): x bf16 NCHW8c [B][Cin <= 128][Sy][Sx], packed = yogo_conv_bf16_pack(w [P][Cin][1][1], mode 0), out fp32 [B][P][Sy][Sx]
// = what yogo_conv2d_fwd_bf16 (fp32 output) followed by yogo_decode_fwd writes, bit for bit.  6 <= P <= 16, Cin a multiple of 16.
extern "C" int yogo_head1x1_decode_fwd_bf16(const void* x, const void* packed, const float* bias, float* out, const float* cxs, const float* cys, int B, int Cin,
                                           int P, int Sy, int Sx, float anchor_w, float anchor_h, float width_mult, float height_mult, int inference,
                                           hipStream_t stream) {
  YOGO_CHECK_ARG(x && packed && out && cxs && cys, "head1x1_decode_fwd_bf16: null pointer");
  YOGO_CHECK_ARG(B > 0 && Sy > 0 && Sx > 0 && P >= 6 && P <= 16 && conv_bf16_head_fwd_eligible(Cin, P, Sy * Sx, B),
                 "head1x1_decode_fwd_bf16: unsupported shape (6 <= P <= 16, Cin a multiple of 16 and <= 128)");
  ConvHeadDecode d{cxs, cys, (float)(1.0 / Sx), (float)(1.0 / Sy), anchor_w, anchor_h, width_mult, height_mult, inference};
  return launch_head(x, packed, bias, out, B, Cin, P, Sy * Sx, &d, stream);
}
