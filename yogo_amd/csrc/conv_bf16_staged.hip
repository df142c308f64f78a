// Thin 3x3 convolutions (stride 2 out of 16 / 32 channels into <= 64, stride 1 out of 16 into <= 32: layers 2 and 1 of base_model forward,
// yogo/model_defns.py:36-46) as INDEPENDENT wavefronts: the packed weights are resident in LDS (4.6 - 36 KB, staged once per workgroup), every wavefront owns tiles of 32 consecutive
// output pixels of one output row, stages ITS three input rows into ITS OWN piece of LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 4 - 13
// kilobyte pieces per tile), waits for them with its own vmcnt, multiplies and stores -- no barrier after the weight staging.
//
// Why (round 5): these layers move 2.4 - 2.5 GB per launch with 0.1 ms of MFMAs; as tiles of a 4-wavefront workgroup
// (conv_bf16_kernel) a tile's life is prologue -> offset decode -> DMA issue -> round trip -> barrier -> MFMAs -> second chunk -> epilogue, one
// after the other (tools/stamps_conv_bf16.py: 18 k ticks per workgroup of which the MFMAs are < 2 k), three or four such lives per CU, and every
// workgroup re-stages the weight slices (36 KB per 16 KB of output for layer 2).  The direct stride-2 data gradients (conv_bf16_direct.hip)
// showed what independent wavefronts with resident weights are worth; the forward-type layers cannot take their operands straight from
// memory (nine taps = nine loads per pixel block, measured +14 ... +45 %), so the input still goes through LDS -- once, privately.
//
// Staged image of a tile (16-byte units): [channel block kb][3 rows][ROWU columns]; stride 2 keeps the even and the odd input columns of a
// row apart ([33 even | 33 odd]) so that a tap's 32 pixels read 32 consecutive units (conv_bf16_ws3.h).  Products accumulate chunk-major /
// tap-minor as conv_bf16_kernel with 16-channel chunks does (bit-identical); the epilogue is its lean form: fma(acc, scale, bias * scale),
// LeakyReLU as max(v, 0.01 v), the sign map of the result, bf16, half-wave exchange, 512 contiguous bytes per half-wave and store.
#include "common.h"
#include <mutex>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CS_MAXDMA 16   // LDS-DMA pieces of a tile: ceil(2 NK x 3 x ROWU / 64)

struct ConvStagedParams {
  const u32x4* in;    // bf16 NCHW8c [B][Kb][IH][IW] units
  const u32x4* wp;    // packed weights (mode 0 / 1): [9][Kb][Mpad] units
  const float* bias;  // [M] or null
  u32x4* out;         // bf16 NCHW8c [B][Mb][OH][OW] units
  unsigned char* signs;         // optional LeakyReLU sign map of the output ([B][2][OH][OW][Mpad / 16] bytes), written
  const float* chan_scale;      // optional [B][M]
  int B, Kb, M, Mb, IH, IW, OH, OW, act;
  int tiles_per_row, tiles_per_img, ntiles;
  unsigned m_tpr, m_tpi;        // ceil(2^32 / d) magic numbers
};

namespace {
__device__ __forceinline__ int cs_udivm1(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }
__device__ __forceinline__ unsigned cs_u(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
// One LDS-DMA piece: 64 lanes x 16 bytes from (descriptor, per-lane byte offset) to LDS bytes [lds_addr, lds_addr + 1024).  From inline asm:
// hipcc would make every later LDS read wait for all of its own LDS-DMA loads AND stores (vmcnt(0)) wherever it liked; cs_wait_dma() below
// is the one place this kernel waits.
__device__ __forceinline__ void cs_dma16(i32x4 rs, unsigned lds_addr, int voff) {
  unsigned keep;
  rs = i32x4{(int)cs_u((unsigned)rs.x), (int)cs_u((unsigned)rs.y), (int)cs_u((unsigned)rs.z), (int)cs_u((unsigned)rs.w)};
  lds_addr = cs_u(lds_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(lds_addr), "s"(rs)
               : "memory");
}
__device__ __forceinline__ void cs_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ i32x4 cs_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
}  // namespace

// S: stride; NK: 16-channel steps of the contraction (1 or 2); NMB: 32-channel blocks of the output (1 or 2); NWV: wavefronts per workgroup;
// R: output rows of a tile (R x 32 pixels: (R - 1) S + 3 staged input rows -- taller tiles re-read fewer halo rows and reuse a weight operand R times);
// NTS: non-temporal output stores (where a launch writes more than it reads: layer 1 forward)
template <int S, int NK, int NMB, bool SIGN_OUT, int NWV, int R, bool NTS>
__global__ __launch_bounds__(64 * NWV) void conv_bf16_staged_kernel(const ConvStagedParams p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds_u[];   // (the dynamic block starts at LDS address 0: LDS-DMA takes addresses, not pointers)
  constexpr int OOB = (int)0x80000000u;
  constexpr int KB = 2 * NK, MP = 32 * NMB, WU = 9 * KB * MP, NT = 64 * NWV;
  constexpr int PW = 33;                       // stride 2: units of a column-parity plane of a staged row
  constexpr int ROWU = S == 2 ? 2 * PW : 34;   // units of a staged row
  constexpr int NR = (R - 1) * S + 3;          // staged input rows
  constexpr int TU = KB * NR * ROWU;           // units of a tile image
  constexpr int NDMA = (TU + 63) / 64;
  static_assert(NDMA <= CS_MAXDMA, "tile image too large");
  constexpr int TILE_U = NDMA * 64;            // units of a wavefront's private LDS piece
  // LDS: weights [9][KB][MP] | bias [MP] fp32 | channel scales [NWV][MP] fp32 | tile images [NWV][TILE_U]
  constexpr int OFF_B = WU, OFF_S = OFF_B + MP / 4, OFF_T = OFF_S + NWV * MP / 4;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* lds_b = reinterpret_cast<float*>(lds_u + OFF_B);
  float* my_scale = reinterpret_cast<float*>(lds_u + OFF_S) + wave * MP;
  for (int i = tid; i < WU; i += NT) lds_u[i] = p.wp[i];
  if (tid < MP) lds_b[tid] = (p.bias != nullptr && tid < p.M) ? p.bias[tid] : 0.f;
  __syncthreads();
  const int IH = p.IH, IW = p.IW, OW = p.OW;
  const int kcb = IH * IW * 16, plane = p.OH * OW, plane16 = plane * 16;
  const bool leaky = p.act == ACT_LEAKY;   // uniform
  const u32x4* my_tile = lds_u + OFF_T + wave * TILE_U;
  const unsigned my_tile_addr = (unsigned)((OFF_T + wave * TILE_U) * 16);
  // this lane's elements of the tile image: piece i covers units 64 i + lane -> (channel block, row, column) -> byte offset relative to the
  // tile's first input pixel (iy0, ix0), and (row << 16 | column) for the border tests; all ones = beyond the image of the tile
  int rel[NDMA];
  unsigned rc[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int u = i * 64 + lane;
    const int kb = u / (NR * ROWU), rem = u - kb * (NR * ROWU);
    const int r = rem / ROWU, cc = rem - r * ROWU;
    const int col = S == 2 ? 2 * (cc % PW) + cc / PW : cc;
    rel[i] = kb * kcb + (r * IW + col) * 16;
    rc[i] = u < TU ? (unsigned)(r << 16 | col) : 0xFFFFFFFFu;
  }
  // workgroup -> XCD -> a contiguous eighth of the tiles (conv_bf16_direct.hip): the rows two tiles share meet in one L2
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int t8 = (p.ntiles + 7) >> 3, t_end = min(p.ntiles, (xcd + 1) * t8);
  for (int tile = xcd * t8 + slot * NWV + wave; tile < t_end; tile += nslot * NWV) {
    const int b = cs_udivm1(tile, p.tiles_per_img, p.m_tpi);
    const int t = tile - b * p.tiles_per_img;
    const int ty = cs_udivm1(t, p.tiles_per_row, p.m_tpr), ox0 = (t - ty * p.tiles_per_row) * 32;
    const int oy0 = ty * R;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    // ---- stage the tile image (zeros where the padding or the image border is: out-of-range offsets)
    {
      const i32x4 rs_in = cs_rsrc(p.in + (size_t)b * KB * IH * IW, (unsigned)(KB * kcb));
      const int base = (iy0 * IW + ix0) * 16;
#pragma unroll
      for (int i = 0; i < NDMA; ++i) {
        const int r = (int)(rc[i] >> 16), c = (int)(rc[i] & 0xFFFFu);
        const bool ok = ((unsigned)(iy0 + r) < (unsigned)IH) && ((unsigned)(ix0 + c) < (unsigned)IW);   // (all ones: r = 65535 fails the row test)
        cs_dma16(rs_in, my_tile_addr + (unsigned)i * 1024u, ok ? rel[i] + base : OOB);
      }
    }
    // the tile's channel scales ride along (lanes < MP: the scale of channel `lane`; 1 without a scale, 0 for the padding channels)
    float my_s = 0.f;
    if (lane < MP) my_s = lane < p.M ? (p.chan_scale != nullptr ? p.chan_scale[(size_t)b * p.M + lane] : 1.f) : 0.f;
    f32x16 acc[R][NMB];
#pragma unroll
    for (int rr = 0; rr < R; ++rr)
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rr][mb][r] = 0.f;
    cs_wait_dma();
    if (lane < MP) my_scale[lane] = my_s;
    // ---- MFMAs: chunk-major, tap-minor
    const u32x4* bt = my_tile + half * NR * ROWU + l31;   // this lane's pixel of tile row 0, channel block `half` of a 16-channel step, tap (0, 0)
    const u32x4* wk = lds_u + half * MP + l31;            // weight unit [tap][2 kc + half][32 mb + l31]
#pragma unroll
    for (int kc = 0; kc < NK; ++kc)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        constexpr int dummy = 0; (void)dummy;
        const int ky = tp / 3, kx = tp % 3;
        const int boff = (2 * kc * NR + ky) * ROWU + (S == 2 ? (kx & 1) * PW + (kx >> 1) : kx);
        // (a few taps at a time between scheduling barriers: left alone, hipcc hoists every operand of the tile and spills)
        if (tp % (R * NMB >= 4 ? 1 : 3) == 0) __builtin_amdgcn_sched_barrier(0);
        bf16x8 av[NMB];
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) av[mb] = __builtin_bit_cast(bf16x8, wk[(tp * KB + 2 * kc) * MP + 32 * mb]);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
          const bf16x8 bv = __builtin_bit_cast(bf16x8, bt[boff + rr * S * ROWU]);
#pragma unroll
          for (int mb = 0; mb < NMB; ++mb) acc[rr][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mb], bv, acc[rr][mb], 0, 0, 0);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue: (mb, gp) -> one 16-byte store per lane: lanes 0-31 channel block 4 mb + 2 gp, lanes 32-63 the next one
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)b * p.Mb * plane), (short)0, p.Mb * plane16, 0x00020000);
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
    const bool ov = ox0 + l31 < OW && oy0 + rr < p.OH;
    const int o = (oy0 + rr) * OW + ox0 + l31;
    unsigned sg = 0;   // this lane's sign bytes: byte j = channel group 2 mb + gp
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const int cb = 4 * mb + 2 * gp;
        const int cl = 32 * mb + 16 * gp + 4 * half;   // this lane's channels: cl .. cl + 3 and cl + 8 .. cl + 11
        const float4 bA = *reinterpret_cast<const float4*>(lds_b + cl), bB = *reinterpret_cast<const float4*>(lds_b + cl + 8);
        const float4 sA = *reinterpret_cast<const float4*>(my_scale + cl), sB = *reinterpret_cast<const float4*>(my_scale + cl + 8);
        const float ba[8] = {bA.x, bA.y, bA.z, bA.w, bB.x, bB.y, bB.z, bB.w};
        const float sa[8] = {sA.x, sA.y, sA.z, sA.w, sB.x, sB.y, sB.z, sB.w};
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(acc[rr][mb][8 * gp + i], sa[i], ba[i] * sa[i]);
        if (leaky) {   // max(v, 0.01 v) as a bare v_max_f32 (conv_bf16_epi_groups.inc)
#pragma unroll
          for (int i = 0; i < 8; i += 2) {
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t sv = (f32x2_t){v[i], v[i + 1]} * (f32x2_t){LEAKY_SLOPE, LEAKY_SLOPE};
            asm("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(sv.x));
            asm("v_max_f32 %0, %1, %2" : "=v"(v[i + 1]) : "v"(v[i + 1]), "v"(sv.y));
          }
        }
        if constexpr (SIGN_OUT) {   // byte = sum of (v[i] > 0) << i (compare into vcc, add-with-carry shifts it in: values 7 down to 0)
          unsigned m = 0;
#pragma unroll
          for (int i = 7; i >= 0; --i) asm("v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(v[i]) : "vcc");
          sg |= m << (8 * (2 * mb + gp));
        }
        bf16x8 ob;
#pragma unroll
        for (int i = 0; i < 8; ++i) ob[i] = (__bf16)v[i];
        const u32x4 w = __builtin_bit_cast(u32x4, ob);
        const auto r0 = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);
        const u32x4 st = {r0[0], r1[0], r0[1], r1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(st, rs_o, (ov && cb + half < p.Mb) ? o * 16 + (cb + half) * plane16 : OOB, 0, NTS ? 2 : 0);
      }
    if constexpr (SIGN_OUT) {
      constexpr int sq = 2 * NMB;   // sign bytes per (pixel, half-wave)
      const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)(p.signs + (size_t)b * plane * 2 * sq), (short)0, plane * 2 * sq, 0x00020000);
      const int vs = ov ? (half * plane + o) * sq : OOB;
      if constexpr (NMB == 2) __builtin_amdgcn_raw_buffer_store_b32(sg, rs_s, vs, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b16((unsigned short)sg, rs_s, vs, 0, 0);
    }
    }
  }
}

bool conv_bf16_staged_eligible(int K, int M, int stride, int IH, int IW, int OH, int OW, int B) {
  if ((K != 16 && K != 32) || M < 1 || M > 64 || (stride != 2 && !(stride == 1 && K == 16 && M <= 32))) return false;   // (stride 1: see launch_conv_bf16_staged)
  if (IH < 1 || IW < 1 || OH < 1 || OW < 1 || B <= 0 || IH >= 32000 || IW >= 32768) return false;
  const long long Mb = ((M + 15) / 16) * 2;
  if ((long long)(K / 8) * IH * IW * 16 >= (1ll << 31) || Mb * OH * OW * 16 >= (1ll << 31) || (long long)B * OH * ((OW + 31) / 32) >= (1ll << 31)) return false;
  // the kernel's divisions by multiplication: tile -> image (by tiles per image, at most OH * tiles per row), tile in image -> row group
  const long long tpr = (OW + 31) / 32;
  const int R = stride == 1 ? 8 : 1;   // (the row-tile height launch_conv_bf16_staged instantiates for this stride)
  const long long tpi = ((OH + R - 1) / R) * tpr;
  if (!magic_div_exact((long long)B * tpi - 1, (int)tpi) || !magic_div_exact(tpi, (int)tpr)) return false;
  return true;
}

namespace {
unsigned cs_magic(int d) { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(((1ull << 32) + (unsigned)d - 1ull) / (unsigned)d); }
template <int S, int NK, int NMB, int NWV, int R, bool NT = false>
int cs_launch(ConvStagedParams p, int dev, int n_cu, bool sg, hipStream_t stream, int* grid_out, int* lds_out) {
  constexpr int KB = 2 * NK, MP = 32 * NMB, ROWU = S == 2 ? 66 : 34, NDMA = (KB * ((R - 1) * S + 3) * ROWU + 63) / 64;
  p.tiles_per_row = cdiv(p.OW, 32);
  p.tiles_per_img = cdiv(p.OH, R) * p.tiles_per_row;
  p.ntiles = p.B * p.tiles_per_img;
  p.m_tpr = cs_magic(p.tiles_per_row); p.m_tpi = cs_magic(p.tiles_per_img);
  if (p.ntiles <= 0) return YOGO_OK;
  constexpr int lds = (9 * KB * MP + MP / 4 + NWV * MP / 4 + NWV * NDMA * 64) * 16;
  static_assert(lds <= 160 * 1024, "LDS");
  const int per_cu = max(1, min(16 / NWV, (160 * 1024) / lds));   // workgroups a CU holds
  const int t8 = cdiv(p.ntiles, 8);
  const int grid = 8 * max(1, min(cdiv(t8, NWV), per_cu * n_cu / 8));
  if (lds > 64 * 1024) {   // more than 64 KB of dynamic LDS has to be asked for: once per device and instantiation
    static std::mutex mu;
    static bool done[64][2] = {};
    std::lock_guard<std::mutex> lk(mu);
    if (!done[dev][sg ? 1 : 0]) {
      hipError_t e = sg ? hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_staged_kernel<S, NK, NMB, true, NWV, R, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)
                        : hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_staged_kernel<S, NK, NMB, false, NWV, R, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) {
        yogo_set_error("conv_bf16_staged: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return YOGO_ERR_HIP;
      }
      done[dev][sg ? 1 : 0] = true;
    }
  }
  if (sg) hipLaunchKernelGGL((conv_bf16_staged_kernel<S, NK, NMB, true, NWV, R, NT>), dim3(grid), dim3(64 * NWV), lds, stream, p);
  else hipLaunchKernelGGL((conv_bf16_staged_kernel<S, NK, NMB, false, NWV, R, NT>), dim3(grid), dim3(64 * NWV), lds, stream, p);
  *grid_out = grid; *lds_out = lds;
  if (yogo_launch_log_enabled())
    yogo_launch_log("conv_bf16_staged_kernel<%d, %d, %d, %s, %d, %d, %s> | K=%d M=%d in=%dx%d out=%dx%d tiles=%d grid=%d lds=%d act=%d bias=%d scale=%d", S, NK, NMB,
                    sg ? "true" : "false", NWV, R, NT ? "true" : "false", 16 * NK, p.M, p.IH, p.IW, p.OH, p.OW, p.ntiles, grid, lds, p.act, p.bias != nullptr, p.chan_scale != nullptr);
  return YOGO_OK;
}
}  // namespace

int launch_conv_bf16_staged(const void* in, const void* packed, const float* bias, void* out, void* signs, const float* chan_scale, int B, int K, int M, int IH,
                            int IW, int OH, int OW, int stride, int act, hipStream_t stream) {
  static std::mutex mu;
  static int n_cu_of[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    yogo_set_error("conv_bf16_staged: hipGetDevice failed");
    return YOGO_ERR_HIP;
  }
  int n_cu;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (n_cu_of[dev] == 0) {
      hipDeviceProp_t prop;
      n_cu_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    n_cu = n_cu_of[dev];
  }
  ConvStagedParams p{};
  p.in = reinterpret_cast<const u32x4*>(in); p.wp = reinterpret_cast<const u32x4*>(packed); p.bias = bias; p.out = reinterpret_cast<u32x4*>(out);
  p.signs = reinterpret_cast<unsigned char*>(signs); p.chan_scale = chan_scale;
  p.B = B; p.Kb = K / 8; p.M = M; p.Mb = ((M + 15) / 16) * 2; p.IH = IH; p.IW = IW; p.OH = OH; p.OW = OW; p.act = act;
  if (B <= 0 || OH <= 0 || OW <= 0) return YOGO_OK;
  const int nmb = M <= 32 ? 1 : 2, nk = K / 16;
  const bool sg = signs != nullptr;
  int grid = 0, lds = 0, rc;
  // wavefronts per workgroup: 4 where four workgroups' weights + tile images fit a CU, 8 for the 36 KB of layer 2's weights (one workgroup per CU)
  // stride 1 out of 16 into <= 32 channels (layer 1 forward): tiles of 8 rows and non-temporal stores (it writes twice what it reads): 7 - 9 % ahead of
  // the tiled kernel with the same stores (gpurun_out/r5_s1_ab1.log, r5_cs_ab4.log); with cached stores the two were equal, and layer 1's data
  // gradient (32 -> 16 channels: reads twice what it writes) came out equal either way (-2 ... +1 %): it stays with the tiled kernel, not instantiated
  if (stride == 1) rc = cs_launch<1, 1, 1, 4, 8, true>(p, dev, n_cu, sg, stream, &grid, &lds);
  else
  if (nk == 1) rc = nmb == 1 ? cs_launch<2, 1, 1, 4, 1>(p, dev, n_cu, sg, stream, &grid, &lds) : cs_launch<2, 1, 2, 4, 1>(p, dev, n_cu, sg, stream, &grid, &lds);
  else rc = nmb == 1 ? cs_launch<2, 2, 1, 4, 1>(p, dev, n_cu, sg, stream, &grid, &lds) : cs_launch<2, 2, 2, 8, 1>(p, dev, n_cu, sg, stream, &grid, &lds);
  if (rc != YOGO_OK) return rc;
  YOGO_CHECK_LAUNCH("conv_bf16_staged");
  return YOGO_OK;
}
