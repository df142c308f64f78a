// "Direct" bf16 convolutions for the thin, memory-bound layers: the packed weights are RESIDENT in LDS (9 - 36 KB, staged once per
// workgroup), every wavefront owns whole tiles of 32 pixels and fetches its MFMA pixel operands STRAIGHT FROM GLOBAL MEMORY into
// registers (buffer_load_dwordx4: a lane's 16-byte unit of its pixel, shifted per tap; neighbouring taps and rows hit the CU's L1), and
// after the one barrier behind the weight staging the wavefronts never synchronise again.
//
// Why (round 5): the thin layers move 2.5 GB per launch and ran at 3.2 - 4.7 TB/s as LDS-staged tiles; what they lose is
// not arithmetic (0.1 ms of MFMAs) but the coupling of a workgroup's wavefronts -- prologue, LDS-DMA round trip, barrier, epilogue and
// stores of a tile happen one after the other, and a load issued behind a tile's stores waits for them (the CU's vector-memory path is in
// order).  Sixteen INDEPENDENT wavefronts per CU, each a simple load -> MFMA -> store stream over its own tiles, is the shape of the
// kernels that do reach 5 - 6 TB/s here (the BatchNorm sweeps).
//
// conv_bf16_s2d_direct_kernel: the data gradient of a stride-2 3x3 convolution into <= 32 channels (layer 2 of base_model: autograd of
// yogo/model_defns.py:44-46), decomposed by output parity as conv_bf16_kernel<.., S2D> (dx[2a+py][2b+px] receives the taps with ky = py + 1,
// kx = px + 1 (mod 2): 1 + 2 + 2 + 4 tap-GEMMs per 2x2 output quad).  A tile = 32 consecutive quads (= dy pixels) of an image: 4 accumulator
// tiles (py, px), per 16-channel step four pixel operands (the quad's dy pixel and its right / lower / diagonal neighbours) serve the nine
// taps.  Epilogue as the tiled kernel: x channel scale [x LeakyReLU'(sign bit)], bf16, half-wave exchange, 16-byte stores.
#include "common.h"
// cache policy of the output stores: 2 = nt (non-temporal).  The 1.6 GB a launch writes are read again by a later kernel, long after they
// left the 4 MB L2: stored as ordinary lines they push the gradient rows the tiles of the next quad row (and the partner pass) are about
// to read out of it.  Same-box A/B (gpurun_out/r5_nt_ab1.log, r5_nt_ab2.log): L4 -8.5 ... -10 %, L2 -5 % (A/B variant builds: 0 = cached)
#ifndef YOGO_ST_AUX
#define YOGO_ST_AUX 2
#endif
#include <mutex>

#ifndef DD_ABL
#define DD_ABL 0   // (ablation variant builds: 1 = no stores, 2 = no MFMAs, 4 = no operand loads, 8 = no weight reads)
#endif
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ConvDirectS2dParams {
  const u32x4* in;    // dy: bf16 NCHW8c [B][Kb][IH][IW] units
  const u32x4* wp;    // packed weights, mode 2: [9 slices in parity-class order][Kb][Mpad] units
  u32x4* out;         // dx: bf16 NCHW8c [B][Mb][OH][OW] units
  const unsigned char* signs;   // optional LeakyReLU sign map of the block output dx flows into ([B][2][OH][OW][Mpad / 16] bytes), or null
  const float* chan_scale;      // optional [B][M]
  int B, Kb, M, Mb, Mpad, IH, IW, OH, OW;   // Mb: channel blocks of dx (even)
  int npass;                    // channel passes: Mpad / (32 NMB); the workgroups of a pass hold ITS weight slices
  int tiles_per_img, ntiles;
  unsigned m_iw, m_tpi;         // ceil(2^32 / d) magic numbers of IW and tiles_per_img
};

namespace {
__device__ __forceinline__ int dd_udivm1(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }
}  // namespace

// NK: 16-channel steps of the contraction (K / 16); NMB: 32-channel blocks of dx per wavefront tile (1: 4 wavefronts per workgroup, four
// workgroups per CU; 2: 8 wavefronts, ONE workgroup per CU whose 147 KB of LDS hold the nine slices of 64 of the 128 channels -- the other
// 64 belong to the partner workgroup of the same XCD, 8 workgroup ids further on, which walks the same tiles at the same time: the second
// read of a gradient tile comes out of the XCD's L2)
template <int NK, int NMB, bool SIGNS>
__global__ __launch_bounds__(256 * NMB) __attribute__((amdgpu_waves_per_eu(NMB == 1 ? 4 : 2, NMB == 1 ? 4 : 2))) void conv_bf16_s2d_direct_kernel(
    const ConvDirectS2dParams p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds_w[];   // [9][2 NK][32 NMB] units, then [wavefronts][32 NMB] fp32 channel scales
  constexpr int OOB = (int)0x80000000u;
  constexpr int KB = 2 * NK, MP = 32 * NMB, NT = 256 * NMB, NWV = 4 * NMB;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (XCD, pass, slot): consecutive workgroup ids go round the 8 XCDs; the passes of one slot sit 8 ids apart (same XCD).  An
  // XCD owns a CONTIGUOUS eighth of the tiles and its wavefronts walk it together: a gradient row is read by the tiles of two quad rows
  // (as the own row of one, as the lower neighbour of the other, IW / 32 tiles apart) -- dealt round-robin over the chip they met in
  // different L2s and every row came from HBM twice (FETCH 2.45x the gradient, profiles/r05_hbm_traffic.txt before this change)
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int pass = (bid >> 3) % p.npass;
  const int slot = bid / (8 * p.npass);
  const int nslot = gridDim.x / (8 * p.npass);
  const int t8 = (p.ntiles + 7) >> 3;                 // tiles of an XCD
  const int t_end = min(p.ntiles, (xcd + 1) * t8);
  for (int i = tid; i < 9 * KB * MP; i += NT) {
    const int m = i % MP, r = i / MP;   // r = slice * KB + kb
    lds_w[i] = p.wp[(size_t)r * p.Mpad + pass * MP + m];
  }
  float* my_scale = reinterpret_cast<float*>(lds_w + 9 * KB * MP) + wave * MP;
  __syncthreads();
  const int IH = p.IH, IW = p.IW, OH = p.OH, OW = p.OW;
  const int kcb = IH * IW * 16, plane = OH * OW, plane16 = plane * 16, nq = IH * IW;
  const int sq = p.Mpad >> 4;   // sign bytes per (pixel, half-wave)
  const int ch0 = pass * MP;    // first channel of this pass
  // a tile's geometry for this lane: its quad (a, c) of image b, the offsets of the four dy operands, which members of the quad exist
  struct Geo {
    int b, pix;          // image; output pixel (2a, 2c) of the quad
    int vsh[4];          // dy operand offsets: the quad's pixel and its right / lower / diagonal neighbours, channel block `half` of a 16-channel step
    bool qv, vx, vy;     // the quad exists; its odd column / row exists
  };
  auto geo_of = [&](int tile) __attribute__((always_inline)) {
    Geo g;
    g.b = dd_udivm1(tile, p.tiles_per_img, p.m_tpi);
    const int t = tile - g.b * p.tiles_per_img;
    const int q = t * 32 + l31;
    g.qv = q < nq;
    const int qc = g.qv ? q : nq - 1;
    const int a = dd_udivm1(qc, IW, p.m_iw), c = qc - a * IW;
    const bool cx = c + 1 < IW, cy = a + 1 < IH;   // the right / lower dy neighbour exists
    g.vx = 2 * c + 1 < OW; g.vy = 2 * a + 1 < OH;
    const int v00 = (a * IW + c) * 16 + half * kcb;
    g.vsh[0] = v00; g.vsh[1] = cx ? v00 + 16 : OOB; g.vsh[2] = cy ? v00 + IW * 16 : OOB; g.vsh[3] = (cx && cy) ? v00 + IW * 16 + 16 : OOB;
    g.pix = 2 * a * OW + 2 * c;
    return g;
  };
  u32x4 Bq[2][4];
  auto load_step = [&](const Geo& g, int kc, u32x4 (&dst)[4]) __attribute__((always_inline)) {
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)g.b * KB * IH * IW), (short)0, KB * kcb, 0x00020000);
#pragma unroll
    for (int s = 0; s < 4; ++s) dst[s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (DD_ABL & 4) ? OOB : g.vsh[s], (2 * kc) * kcb, 0));
  };
  static_assert(NK % 2 == 0, "the first operand set of the NEXT tile is requested while the last step still reads the second");
  const int tstep = nslot * NWV;
  int tile = xcd * t8 + slot * NWV + wave;
  Geo G = geo_of(min(tile, p.ntiles - 1));
  if (tile < t_end) load_step(G, 0, Bq[0]);
  for (; tile < t_end; tile += tstep) {
    const int b = G.b, pix = G.pix;
    const bool qv = G.qv, vx = G.vx, vy = G.vy;
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)b * p.Mb * plane), (short)0, p.Mb * plane16, 0x00020000);
    // the epilogue's small inputs first (they are older than the operand loads: landed when the MFMAs are through)
    float my_s = 0.f;   // lanes < MP: the scale of channel ch0 + lane (1 without a scale, 0 for the padding channels)
    if (lane < MP) my_s = ch0 + lane < p.M ? (p.chan_scale != nullptr ? p.chan_scale[(size_t)b * p.M + ch0 + lane] : 1.f) : 0.f;
    unsigned sg[2][2] = {{0u, 0u}, {0u, 0u}};   // byte j = the lane's sign bits of channel group 2 pass NMB + j
    if constexpr (SIGNS) {
      const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)(p.signs + (size_t)b * plane * 2 * sq), (short)0, plane * 2 * sq, 0x00020000);
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const bool ok = qv && (px == 0 || vx) && (py == 0 || vy);
          const int vs = ok ? (half * plane + pix + py * OW + px) * sq + pass * 2 * NMB : OOB;
          if constexpr (NMB == 2) sg[py][px] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_s, vs, 0, 0);
          else sg[py][px] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs_s, vs, 0, 0);
        }
    }
    f32x16 acc[NMB][2][2];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mb][py][px][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) {
      if (kc + 1 < NK) load_step(G, kc + 1, Bq[(kc + 1) & 1]);
      const u32x4(&Bc)[4] = Bq[kc & 1];
      const u32x4* wk = lds_w + (2 * kc + half) * MP + l31;   // slice s at + s * KB * MP, channel block mb at + 32 mb
      // slice -> (row parity, column parity, dy shift): 0 (0,0,0) | 1 (0,1,0) 2 (0,1,1) | 3 (1,0,0) 4 (1,0,2) | 5 (1,1,0) 6 (1,1,1) 7 (1,1,2) 8 (1,1,3)
#define DD_MF(PY, PX, S, SH)                                                                                                                     \
  _Pragma("unroll") for (int mb = 0; mb < NMB; ++mb) if (!(DD_ABL & 2)) acc[mb][PY][PX] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(               \
      __builtin_bit_cast(bf16x8, (DD_ABL & 8) ? Bc[(SH) ^ 1] : wk[(S) * KB * MP + 32 * mb]), __builtin_bit_cast(bf16x8, Bc[SH]), acc[mb][PY][PX], 0, 0, 0); \
  else acc[mb][PY][PX][0] += __builtin_bit_cast(float, Bc[SH].x);
      // (three taps at a time between scheduling barriers: left alone, hipcc hoists all the weight quads of a tile and spills 150 registers)
      __builtin_amdgcn_sched_barrier(0);
      DD_MF(0, 0, 0, 0) DD_MF(0, 1, 1, 0) DD_MF(1, 0, 3, 0)
      __builtin_amdgcn_sched_barrier(0);
      DD_MF(1, 1, 5, 0) DD_MF(0, 1, 2, 1) DD_MF(1, 1, 6, 1)
      __builtin_amdgcn_sched_barrier(0);
      DD_MF(1, 0, 4, 2) DD_MF(1, 1, 7, 2) DD_MF(1, 1, 8, 3)
      __builtin_amdgcn_sched_barrier(0);
#undef DD_MF
    }
    __builtin_amdgcn_sched_barrier(0);
    // the NEXT tile's first operands go out in front of this tile's stores (the CU's vector-memory path is in order; measured: no
    // difference either way, gpurun_out/r5_d128_ab2.log -- the other wavefronts of the SIMD cover the wait)
    if (tile + tstep < t_end) {
      G = geo_of(tile + tstep);
      load_step(G, 0, Bq[0]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the tile's channel scales through this wavefront's own LDS row (no barrier: a wavefront's LDS operations complete in order)
    if (lane < MP) my_scale[lane] = my_s;
    // ---- epilogue: (mb, gp, py) -> two 16-byte stores per lane
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const int cb = pass * 4 * NMB + 4 * mb + 2 * gp;   // channel block of the group's lower half (uniform)
      const int cl = 32 * mb + 16 * gp + 4 * half;       // this lane's channels: cl .. cl + 3 and cl + 8 .. cl + 11 of the pass
      const float4 sA = *reinterpret_cast<const float4*>(my_scale + cl), sB = *reinterpret_cast<const float4*>(my_scale + cl + 8);
      const float sa[8] = {sA.x, sA.y, sA.z, sA.w, sB.x, sB.y, sB.z, sB.w};
      float sl[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) sl[i] = LEAKY_SLOPE * sa[i];
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        u32x4 un[2];   // [px]: this lane's unit of channel block cb + half
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          float v[8];
          if constexpr (SIGNS) {
            const unsigned m = sg[py][px] >> (8 * (2 * mb + gp));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int tt = (int)(m << (31 - i)) >> 31;   // bit i spread over the word selects scale or 0.01 * scale
              unsigned f;
              asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(f) : "v"(tt), "v"(sa[i]), "v"(sl[i]));
              v[i] = acc[mb][py][px][8 * gp + i] * __builtin_bit_cast(float, f);
            }
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaf(acc[mb][py][px][8 * gp + i], sa[i], 0.f * sa[i]);
          }
          bf16x8 o;
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
          const u32x4 w = __builtin_bit_cast(u32x4, o);
          const auto r0 = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);
          un[px] = u32x4{r0[0], r1[0], r0[1], r1[1]};
        }
        // a second exchange makes every store a CONTIGUOUS kilobyte: lanes 0-31 the even pixel, lanes 32-63 the odd pixel of ONE channel
        // block (un[0]'s upper half-wave = (px 0, block cb + 1) trades places with un[1]'s lower half-wave = (px 1, block cb))
        u32x4 se[2];   // [e]: lanes 0-31 pixel (2c), lanes 32-63 pixel (2c + 1) of channel block cb + e
        {
          const auto x0 = __builtin_amdgcn_permlane32_swap(un[0].x, un[1].x, false, false);
          const auto x1 = __builtin_amdgcn_permlane32_swap(un[0].y, un[1].y, false, false);
          const auto x2 = __builtin_amdgcn_permlane32_swap(un[0].z, un[1].z, false, false);
          const auto x3 = __builtin_amdgcn_permlane32_swap(un[0].w, un[1].w, false, false);
          se[0] = u32x4{x0[0], x1[0], x2[0], x3[0]};
          se[1] = u32x4{x0[1], x1[1], x2[1], x3[1]};
        }
        // (this lane now stores pixel px = half of its quad)
        const bool okp = qv && (half == 0 || vx) && (py == 0 || vy);
#pragma unroll
        for (int e = 0; e < 2; ++e)
          __builtin_amdgcn_raw_buffer_store_b128(se[e], rs_o, (okp && cb + e < p.Mb && !(DD_ABL & 1)) ? (pix + py * OW + half) * 16 + (cb + e) * plane16 : OOB, 0, YOGO_ST_AUX);
      }
    }
  }
}

bool conv_bf16_s2d_direct_eligible(int K, int M, int OH, int OW, int B) {
  if (M < 1 || !((M <= 32 && (K == 32 || K == 64)) || (M > 64 && M <= 128 && K == 128))) return false;
  if (OH < 2 || OW < 2 || B <= 0) return false;
  const long long IH = (OH + 1) / 2, IW = (OW + 1) / 2, Mb = ((M + 15) / 16) * 2;
  if ((long long)(K / 8) * IH * IW * 16 >= (1ll << 31) || Mb * OH * OW * 16 >= (1ll << 31) || (long long)B * ((IH * IW + 31) / 32) >= (1ll << 31)) return false;
  // the kernel's divisions by multiplication: tile -> image (by tiles per image), pixel -> row (by IW)
  const long long tpi = (IH * IW + 31) / 32;
  if (!magic_div_exact(B * tpi - 1, (int)tpi) || !magic_div_exact(tpi * 32, (int)IW)) return false;
  return true;
}

int launch_conv_bf16_s2d_direct(const void* in, const void* packed, void* out, const void* signs, const float* chan_scale, int B, int K, int M, int IH, int IW,
                                int OH, int OW, hipStream_t stream) {
  static std::mutex mu;
  static int n_cu_of[64] = {0};
  static bool attr_set[64] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    yogo_set_error("conv_bf16_s2d_direct: hipGetDevice failed");
    return YOGO_ERR_HIP;
  }
  const int nmb = M <= 32 ? 1 : 2;
  const int lds = 9 * (K / 8) * 32 * nmb * 16 + 4 * nmb * 32 * nmb * 4;   // 18 / 36 KB (four workgroups per CU); 147 KB + 2 KB (one)
  int n_cu;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (n_cu_of[dev] == 0) {
      hipDeviceProp_t prop;
      n_cu_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    n_cu = n_cu_of[dev];
    if (nmb == 2 && !attr_set[dev]) {   // more than 64 KB of dynamic LDS has to be asked for
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_s2d_direct_kernel<8, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_s2d_direct_kernel<8, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) {
        yogo_set_error("conv_bf16_s2d_direct: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return YOGO_ERR_HIP;
      }
      attr_set[dev] = true;
    }
  }
  ConvDirectS2dParams p{};
  p.in = reinterpret_cast<const u32x4*>(in); p.wp = reinterpret_cast<const u32x4*>(packed); p.out = reinterpret_cast<u32x4*>(out);
  p.signs = reinterpret_cast<const unsigned char*>(signs); p.chan_scale = chan_scale;
  p.B = B; p.Kb = K / 8; p.M = M; p.Mb = ((M + 15) / 16) * 2; p.Mpad = nmb == 1 ? 32 : 128; p.IH = IH; p.IW = IW; p.OH = OH; p.OW = OW;
  p.npass = p.Mpad / (32 * nmb);
  p.tiles_per_img = cdiv(IH * IW, 32);
  p.ntiles = B * p.tiles_per_img;
  auto magic = [](int d) -> unsigned { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(((1ull << 32) + (unsigned)d - 1ull) / (unsigned)d); };
  p.m_iw = magic(IW); p.m_tpi = magic(p.tiles_per_img);
  if (p.ntiles <= 0) return YOGO_OK;
  // whole rounds of the 8 XCDs (x the passes): NMB = 1 up to four workgroups per CU, NMB = 2 one
  const int t8 = cdiv(p.ntiles, 8);
  const int grid = nmb == 1 ? 8 * max(1, min(cdiv(t8, 4), 4 * n_cu / 8)) : 16 * max(1, min(cdiv(t8, 8), n_cu / 16));
  const bool sg = signs != nullptr;
#define DD_LAUNCH(NK, NMB, S) hipLaunchKernelGGL((conv_bf16_s2d_direct_kernel<NK, NMB, S>), dim3(grid), dim3(256 * NMB), lds, stream, p)
  if (K == 128) { if (sg) DD_LAUNCH(8, 2, true); else DD_LAUNCH(8, 2, false); }
  else if (K == 64) { if (sg) DD_LAUNCH(4, 1, true); else DD_LAUNCH(4, 1, false); }
  else { if (sg) DD_LAUNCH(2, 1, true); else DD_LAUNCH(2, 1, false); }
#undef DD_LAUNCH
  if (yogo_launch_log_enabled())
    yogo_launch_log("conv_bf16_s2d_direct_kernel<%d, %d, %s> | K=%d M=%d dy=%dx%d dx=%dx%d tiles=%d grid=%d lds=%d signs=%d scale=%d", K / 16, nmb, sg ? "true" : "false",
                    K, M, IH, IW, OH, OW, p.ntiles, grid, lds, sg, chan_scale != nullptr);
  YOGO_CHECK_LAUNCH("conv_bf16_s2d_direct");
  return YOGO_OK;
}
