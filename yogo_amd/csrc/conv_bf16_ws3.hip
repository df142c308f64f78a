// Persistent, wavefront-specialised FORWARD of a stride-2 3x3 convolution with 128 output channels (layer 4 of base_model:
// yogo/model_defns.py:54-56 -- the one launch of north_star's "3x3 conv GEMM" forward set that conv_bf16_ws_kernel does not take).
// Same MFMA sequence per accumulator (16-channel chunks, taps 0..8 inside a chunk) and the same epilogue formula as
// conv_bf16_kernel<4,1,8,...,PP>: outputs are bit-identical (tests/test_gpu_ws.py).
//
// The tiled kernel (profiles/r04_*: 619 us, 0.31 of the MFMA peak, 3.3 TB/s) runs one 8-wavefront workgroup per CU whose first
// chunk's HBM round trip, epilogue and stores overlap with nothing.  Here one persistent workgroup per CU walks tiles of 128 output
// pixels: wavefronts 4-7 keep a continuous LDS-DMA stream going across tile seams (weight slices one period ahead -- L2 hits --, the
// input tile two periods ahead -- HBM), wavefronts 0-3 compute 64 channels x 64 pixels each and store their own output.  What the
// stride costs is LDS: an output pixel's taps touch a 3x3 input window at stride 2, so a tile's input footprint is ~4x its
// output -- 128-pixel tiles (27 KB per 16-channel chunk, ring of three) where the stride-1 kernel has 256.  The staged rows hold
// their even and odd columns de-interleaved, so a tap's 32 consecutive output pixels read 32 consecutive units and every tap offset is
// an immediate of the ds_read.  Operands are requested two K steps ahead (conv_bf16_ws2.hip: an LDS read takes 200+ cycles beside
// the LDS-DMA stream, a step's 4 MFMAs 128).
#include "conv_bf16_ws3.h"
#include <mutex>
#include <type_traits>
#include <utility>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ int w3_udivm(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }   // n / d, m = ceil(2^32 / d), d > 1
__device__ __forceinline__ int w3_udivm1(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }
__device__ __forceinline__ int w3_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ i32x4 w3_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ unsigned w3_u(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ i32x4 w3_u4(i32x4 r) {
  return i32x4{__builtin_amdgcn_readfirstlane(r.x), __builtin_amdgcn_readfirstlane(r.y), __builtin_amdgcn_readfirstlane(r.z), __builtin_amdgcn_readfirstlane(r.w)};
}
// nine LDS-DMA pieces (64 lanes x 16 bytes each) of one descriptor with one per-lane offset: LDS + 4 KB each, scalar offset + step each
// (this wavefront's quarter of the nine taps' weight slices)
__device__ __forceinline__ void w3_dma9(i32x4 rs, unsigned lds, int voff, unsigned soff, unsigned step) {
  unsigned so;
  rs = w3_u4(rs); lds = w3_u(lds); soff = w3_u(soff); step = w3_u(step);
#define W3_P9 "s_add_u32 m0, m0, 4096\n\ts_add_u32 %0, %0, %5\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\t"
  asm volatile("s_mov_b32 m0, %3\n\ts_mov_b32 %0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\t" W3_P9 W3_P9 W3_P9 W3_P9 W3_P9 W3_P9 W3_P9 W3_P9
               : "=&s"(so)
               : "v"(voff), "s"(rs), "s"(lds), "s"(soff), "s"(step)
               : "memory", "scc");
#undef W3_P9
}
// N (6 or 7) pieces of one descriptor with one scalar offset and their own per-lane offsets: LDS + 4 KB each (the input slots of a chunk)
template <int N>
__device__ __forceinline__ void w3_dman(i32x4 rs, unsigned lds, const int (&v)[W3_NI], unsigned soff) {
  rs = w3_u4(rs); lds = w3_u(lds); soff = w3_u(soff);
#define W3_PI(K) "s_add_u32 m0, m0, 4096\n\ts_nop 0\n\tbuffer_load_dwordx4 %" #K ", %7, %9 offen lds\n\t"
  if constexpr (N == 7)
    asm volatile("s_mov_b32 m0, %8\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %7, %9 offen lds\n\t" W3_PI(1) W3_PI(2) W3_PI(3) W3_PI(4) W3_PI(5) W3_PI(6)
                 ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "s"(rs), "s"(lds), "s"(soff) : "memory", "scc");
  else
    asm volatile("s_mov_b32 m0, %8\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %7, %9 offen lds\n\t" W3_PI(1) W3_PI(2) W3_PI(3) W3_PI(4) W3_PI(5)
                 ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "s"(rs), "s"(lds), "s"(soff) : "memory", "scc");
#undef W3_PI
}
__device__ __forceinline__ void w3_dma_dword(i32x4 rs, unsigned lds, int voff, unsigned soff) {
  rs = w3_u4(rs); lds = w3_u(lds); soff = w3_u(soff);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dword %0, %2, %3 offen lds" ::"v"(voff), "s"(lds), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void w3_store16(u32x4 data, int voff, i32x4 rs, unsigned soff) {
  rs = w3_u4(rs); soff = w3_u(soff);
  // (s_nop in front: the descriptor may come from v_readfirstlane; behind: a 16-byte store's data registers must not be overwritten by
  //  the next vector instruction -- hipcc does not look inside asm statements)
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(data), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void w3_vmwait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void w3_barrier() { asm volatile("s_barrier" ::: "memory"); }

// ---- the asm statements of a compute wavefront's K steps.  The four 32x32 accumulator tiles (row block mb, pixel group n) -> tile 2 mb + n
// are a[0:63], OWNED BY THE ASM STATEMENTS (named literally, listed as clobbers; build.sh audits that no compiler-generated instruction
// touches an AGPR: conv_bf16_ws.hip)
#define W3_ACC_CLOBBER "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"
#ifndef W3_ABL
#define W3_ABL 0   // (compile-time ablations for timing runs, results wrong: bit 0 = no MFMAs, 1 = no operand reads, 2 = no weight DMA, 3 = no epilogue)
#endif
#if W3_ABL & 1
#define W3_MF(K, A, B) "s_nop 0\n\t"
#define W3_MF0(K, A, B) "s_nop 0\n\t"
#else
#define W3_MF(K, A, B) "v_mfma_f32_32x32x16_bf16 a[16*" #K ":16*" #K "+15], %[" #A "], %[" #B "], a[16*" #K ":16*" #K "+15]\n\t"
#define W3_MF0(K, A, B) "v_mfma_f32_32x32x16_bf16 a[16*" #K ":16*" #K "+15], %[" #A "], %[" #B "], 0\n\t"
#endif
#if W3_ABL & 2
#define W3_RD(D, P, O) "s_nop 0\n\t"
#else
#define W3_RD(D, P, O) "ds_read_b128 %[" #D "], %[" #P "] offset:%[" #O "]\n\t"
#endif
#define W3_OPS_IN [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1)
// the 4 MFMAs of a K step (one tap x 16 channels) + the reads of the step TWO ahead: its two weight quads (pa + AOFF, + 512) and two
// pixel quads (pb0 / pb1 + BIMM); ends when all but these four reads are done (the next step's operands have landed)
template <bool ZERO, int AOFF, int BIMM>
__device__ __forceinline__ void w3_k_rd(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, u32x4& an0, u32x4& an1, u32x4& bn0, u32x4& bn1,
                                        unsigned pa, unsigned pb0, unsigned pb1) {
  if constexpr (ZERO)
    asm volatile(W3_MF0(0, a0, b0) W3_RD(an0, pa, ao) W3_RD(an1, pa, ao1) W3_MF0(1, a0, b1) W3_RD(bn0, pb0, bo) W3_RD(bn1, pb1, bo) W3_MF0(2, a1, b0) W3_MF0(3, a1, b1)
                 "s_waitcnt lgkmcnt(4)"
                 : [an0] "=&v"(an0), [an1] "=&v"(an1), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1)
                 : W3_OPS_IN, [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1), [ao] "n"(AOFF), [ao1] "n"(AOFF + 512), [bo] "n"(BIMM)
                 : "memory", W3_ACC_CLOBBER);
  else
    asm volatile(W3_MF(0, a0, b0) W3_RD(an0, pa, ao) W3_RD(an1, pa, ao1) W3_MF(1, a0, b1) W3_RD(bn0, pb0, bo) W3_RD(bn1, pb1, bo) W3_MF(2, a1, b0) W3_MF(3, a1, b1)
                 "s_waitcnt lgkmcnt(4)"
                 : [an0] "=&v"(an0), [an1] "=&v"(an1), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1)
                 : W3_OPS_IN, [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1), [ao] "n"(AOFF), [ao1] "n"(AOFF + 512), [bo] "n"(BIMM)
                 : "memory", W3_ACC_CLOBBER);
}
// step 7: no reads (the next period's buffers are not ready before the barrier); every read of this period is in registers -> the period's barrier
__device__ __forceinline__ void w3_k_bar(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, u32x4& x0, u32x4& x1, u32x4& x2, u32x4& x3) {
  asm volatile(W3_MF(0, a0, b0) W3_MF(1, a0, b1) W3_MF(2, a1, b0) W3_MF(3, a1, b1) "s_waitcnt lgkmcnt(0)\n\ts_barrier"
               : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3)
               : W3_OPS_IN
               : "memory", W3_ACC_CLOBBER);
}
// step 8: the operands of the NEXT period's steps 0 and 1 (weights pa + 0 / + 4096, pixel quads pb + BIMM0 / + BIMM1); ends when step 0's have landed
template <int BIMM0, int BIMM1>
__device__ __forceinline__ void w3_k_next(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, u32x4& an0, u32x4& an1, u32x4& bn0, u32x4& bn1,
                                          u32x4& am0, u32x4& am1, u32x4& bm0, u32x4& bm1, unsigned pa, unsigned pb0, unsigned pb1) {
  asm volatile(W3_MF(0, a0, b0) W3_RD(an0, pa, z0) W3_RD(an1, pa, z512) W3_RD(bn0, pb0, bo0) W3_RD(bn1, pb1, bo0) W3_MF(1, a0, b1) W3_RD(am0, pa, z4096) W3_RD(am1, pa, z4608)
               W3_RD(bm0, pb0, bo1) W3_RD(bm1, pb1, bo1) W3_MF(2, a1, b0) W3_MF(3, a1, b1) "s_waitcnt lgkmcnt(4)"
               : [an0] "=&v"(an0), [an1] "=&v"(an1), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1), [am0] "=&v"(am0), [am1] "=&v"(am1), [bm0] "=&v"(bm0), [bm1] "=&v"(bm1)
               : W3_OPS_IN, [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1), [z0] "n"(0), [z512] "n"(512), [z4096] "n"(4096), [z4608] "n"(4608), [bo0] "n"(BIMM0),
                 [bo1] "n"(BIMM1)
               : "memory", W3_ACC_CLOBBER);
}
// step 8 of a tile's last period: the MFMAs and the wait states between an MFMA and a read of its result (hipcc does not look inside asm statements)
__device__ __forceinline__ void w3_k_end(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1) {
  asm volatile(W3_MF(0, a0, b0) W3_MF(1, a0, b1) W3_MF(2, a1, b0) W3_MF(3, a1, b1) "s_nop 15\n\ts_nop 15" : : W3_OPS_IN : "memory", W3_ACC_CLOBBER);
}
// the operands of a tile's steps 0 and 1
template <int BIMM0, int BIMM1>
__device__ __forceinline__ void w3_k_first(u32x4& an0, u32x4& an1, u32x4& bn0, u32x4& bn1, u32x4& am0, u32x4& am1, u32x4& bm0, u32x4& bm1, unsigned pa, unsigned pb0,
                                           unsigned pb1) {
  asm volatile(W3_RD(an0, pa, z0) W3_RD(an1, pa, z512) W3_RD(bn0, pb0, bo0) W3_RD(bn1, pb1, bo0) W3_RD(am0, pa, z4096) W3_RD(am1, pa, z4608) W3_RD(bm0, pb0, bo1)
               W3_RD(bm1, pb1, bo1) "s_waitcnt lgkmcnt(4)"
               : [an0] "=&v"(an0), [an1] "=&v"(an1), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1), [am0] "=&v"(am0), [am1] "=&v"(am1), [bm0] "=&v"(bm0), [bm1] "=&v"(bm1)
               : [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1), [z0] "n"(0), [z512] "n"(512), [z4096] "n"(4096), [z4608] "n"(4608), [bo0] "n"(BIMM0), [bo1] "n"(BIMM1)
               : "memory");
}
// eight consecutive accumulator registers -> VGPRs in ONE statement (it clobbers every accumulator register: conv_bf16_ws.hip, ws_acc_read8)
template <int R>
__device__ __forceinline__ void w3_acc_read8(float (&r)[8]) {
  asm volatile(
      "v_accvgpr_read_b32 %0, a[%8]\n\tv_accvgpr_read_b32 %1, a[%8+1]\n\tv_accvgpr_read_b32 %2, a[%8+2]\n\tv_accvgpr_read_b32 %3, a[%8+3]\n\t"
      "v_accvgpr_read_b32 %4, a[%8+4]\n\tv_accvgpr_read_b32 %5, a[%8+5]\n\tv_accvgpr_read_b32 %6, a[%8+6]\n\tv_accvgpr_read_b32 %7, a[%8+7]"
      : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7])
      : "n"(R)
      : W3_ACC_CLOBBER);
}
// byte offset of tap t's pixel quads inside the staged chunk: row 2 (i - i_lo) + ky, column plane (kx = 1: even columns, index j; kx = 0: odd
// columns, index j; kx = 2: odd columns, index j + 1)
__device__ constexpr int w3_bimm(int t) { return ((t / 3) * W3_LW + ((t % 3) == 1 ? 0 : W3_PL + ((t % 3) == 2 ? 1 : 0))) * 16; }

template <class F, int... I>
__device__ __forceinline__ void w3_static_for(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

#ifdef YOGO_DIAG
#define W3_DBG(BIT) (p.dbg & (BIT))
#define W3_STAMP() __builtin_amdgcn_s_memtime()
#else
#define W3_DBG(BIT) 0
#define W3_STAMP() 0ull
#endif

}  // namespace

// MODE: bit 0 = LeakyReLU, bit 1 = channel scale (Dropout2d mask), bit 2 = + the sign map of the output (with LeakyReLU)
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_bf16_ws3_kernel(const ConvWs3Params p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
  constexpr unsigned OOB = 0x80000000u;
  float* const ldsf = reinterpret_cast<float*>(smem4);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, tw = wave & 3;   // team 0 computes, team 1 loads; wavefronts tw and tw + 4 share a SIMD
  const int mh = tw >> 1, nh = tw & 1;         // compute wavefront tw: channels mh * 64 ..., pixels nh * 64 ... of the tile
  [[maybe_unused]] const unsigned long long t_start = W3_STAMP();
  const int OH = p.OH, OW = p.OW, IH = p.IH, IW = p.IW;
  const int plane = OH * OW, plane16 = plane * 16;
  const int nck = p.nchunk;
  constexpr bool leaky = (MODE & 1) != 0, SCALED = (MODE & 2) != 0, write_signs = (MODE & 4) != 0;

  // ---- tile walk (as conv_bf16_ws_kernel): virtual block lin = slot + k * G, an XCD's workgroups share a contiguous run of tiles
  const unsigned NV = (unsigned)p.ntiles, G = gridDim.x, slot = blockIdx.x;
  const unsigned xq = NV >> 3, xr = NV & 7;
  struct TileS { int b, j0, bw, p0, p1, lastband; };
  auto find_tile = [&](unsigned& k, TileS& t) __attribute__((always_inline)) -> bool {   // (uniform) next non-empty tile of this workgroup from ordinal k on
    for (;; ++k) {
      const unsigned lin = slot + k * G;
      if (lin >= NV) return false;
      const unsigned xcd = lin & 7;
      const int widx = (int)((xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3));
      const int b = w3_udivm1(widx, p.gx, p.m_gx);
      const int bx = widx - b * p.gx;
      const int cb = w3_udivm1(bx, p.tiles_per_band, p.m_tpb);
      const int tb = bx - cb * p.tiles_per_band;
      const int j0 = cb * p.TW;
      const int bw = min(p.TW, OW - j0);
      const int NPb = OH * bw;
      const int p0 = tb * p.PT;
      if (p0 >= NPb) continue;
      t.b = b; t.j0 = j0; t.bw = bw; t.p0 = p0; t.p1 = min(p0 + p.PT, NPb); t.lastband = cb == p.ncb - 1;
      return true;
    }
  };

  unsigned k_ord = 0;
  TileS T{};
  if (!find_tile(k_ord, T)) return;
  // bias (the same for every tile) and a unit channel scale when there is none
  if (tid < 128) {
    ldsf[W3_EB / 4 + tid] = p.bias != nullptr ? p.bias[tid] : 0.f;
    if (p.chan_scale == nullptr) {
      ldsf[W3_ES / 4 + tid] = 1.f;
      ldsf[W3_ES / 4 + 128 + tid] = 1.f;
    }
  }
  __syncthreads();

  if (team == 1) {
    // =====================================================================================================================
    // LOADERS: per period -- the weight slices of the next chunk (9 pieces per wavefront; needed at this period's barrier), the input
    // tile of the chunk after it (7 pieces, 6 for wavefront 3; it may stay in flight), wait, barrier.  The stream crosses tile seams.
    // =====================================================================================================================
    const int lane = w3_lane();
    const int ttid = tw * 64 + lane;
    const int rowb = IW * 16, kcb = IH * IW * 16;
    const unsigned ibytes = (unsigned)p.Kb * kcb, wbytes = 9u * p.Kb * 2048u;
    const unsigned so_i = 2u * kcb;                  // bytes between the 16-channel chunks of an image
    const unsigned wstep = (unsigned)p.Kb * 2048u;   // bytes between the taps of the packed weights
    const i32x4 rs_w = w3_rsrc(p.wp, wbytes);
    const int lane16 = W3_DBG(4) ? (int)OOB : lane * 16;
    const bool has_scale = p.chan_scale != nullptr;
    const i32x4 rs_sc = w3_rsrc(p.chan_scale, has_scale ? (unsigned)p.B * 512u : 0u);
    auto issue_scale = [&](int b, int par) __attribute__((always_inline)) {   // [128] channel scale of image b -> es[par] (loaders 0 and 1, 64 floats each)
      if (has_scale && tw < 2) w3_dma_dword(rs_sc, (unsigned)(W3_ES + par * 512 + tw * 256), lane * 4, (unsigned)((b * 128 + tw * 64) * 4));
    };
    // DMA source offsets of this lane's input slots: element e = ttid + i * 256 of the staged chunk [2 channel blocks][9 rows][even | odd columns]
    auto decode_slots = [&](const TileS& t, int (&voff)[W3_NI]) __attribute__((always_inline)) {
      const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
      const int bw = t.bw;
      const int i_lo = w3_udivm1(t.p0, bw, m_bw), i_hi = w3_udivm1(t.p1 - 1, bw, m_bw);
      const int rows_in = 2 * (i_hi - i_lo) + 3;
      const int iy0 = 2 * i_lo - 1;
#pragma unroll
      for (int i = 0; i < W3_NI; ++i) {
        const int e = ttid + i * 256;
        const int kbs = e >= W3_KBU ? 1 : 0;
        const int rem = e - kbs * W3_KBU;
        const int r = rem / W3_LW, x = rem - r * W3_LW;
        const int pl = x >= W3_PL ? 1 : 0, q = x - pl * W3_PL;
        const int iy = iy0 + r, ix = 2 * (t.j0 + q) - pl;
        const bool ok = e < 2 * W3_KBU && r < rows_in && q < bw + pl && iy >= 0 && iy < IH && ix >= 0 && ix < IW && !W3_DBG(4);
        voff[i] = ok ? kbs * kcb + iy * rowb + ix * 16 : (int)OOB;
      }
    };
    // weight slices of chunk cn (9 pieces of this wavefront: its quarter of every tap's [2 channel blocks][128 channels]) -> weight buffer wb
    auto req_w = [&](int cn, int wb) __attribute__((always_inline)) {
      if constexpr ((W3_ABL & 4) != 0) return;
      w3_dma9(rs_w, (unsigned)(wb * W3_WB + tw * 1024), lane16, (unsigned)((2 * cn) * 2048 + tw * 1024), wstep);
    };
    // input tile of chunk cn of the tile described by (rs, voff) -> input buffer ib
    auto req_i = [&](i32x4 rs, const int (&voff)[W3_NI], int cn, int ib) __attribute__((always_inline)) {
      const unsigned dst = (unsigned)(W3_I0 + ib * W3_IB + tw * 1024);
      if (tw == 3) w3_dman<6>(rs, dst, voff, (unsigned)cn * so_i);   // (wavefront 3's seventh piece would lie behind the buffer)
      else w3_dman<7>(rs, dst, voff, (unsigned)cn * so_i);
    };
    auto rs_in_of = [&](int b) __attribute__((always_inline)) { return w3_rsrc(reinterpret_cast<const unsigned char*>(p.in) + (size_t)b * ibytes, ibytes); };
    int voff[W3_NI], voff_n[W3_NI];
    decode_slots(T, voff);
#pragma unroll
    for (int i = 0; i < W3_NI; ++i) voff_n[i] = (int)OOB;
    i32x4 rs_in = rs_in_of(T.b), rs_in_n = rs_in;
    issue_scale(T.b, 0);
    req_i(rs_in, voff, 0, 0);
    req_i(rs_in, voff, 1, 1);   // (nck >= 4)
    req_w(0, 0);
    w3_vmwait<0>();
    w3_barrier();   // (#1) chunks 0 (and the input of chunk 1) of the first tile have landed
    int wpar = 0;    // weight buffer of the chunk being computed
    int ib2 = 2;     // ring slot of the next input request (the chunk two periods ahead)
    int tpar = 0;
    bool has_next = true;
    TileS Tn{};
    while (has_next) {
      for (int c = 0; c < nck; ++c) {
        // oldest first: the weight slices of the NEXT chunk (needed at this period's barrier) ...
        if (c + 1 < nck) req_w(c + 1, wpar ^ 1);
        else if (has_next) req_w(0, wpar ^ 1);
        // ... then the input tile of the chunk after it (needed one barrier later: it may stay in flight)
        bool req = false;   // (uniform)
        if (c + 2 < nck) {
          req_i(rs_in, voff, c + 2, ib2);
          req = true;
        } else {
          if (c + 2 == nck && has_next) {   // the request stream crosses into the next tile
#pragma unroll
            for (int i = 0; i < W3_NI; ++i) voff[i] = voff_n[i];
            rs_in = rs_in_n;
            issue_scale(Tn.b, tpar ^ 1);
          }
          if (has_next) {
            req_i(rs_in, voff, c + 2 - nck, ib2);
            req = true;
          }
        }
        ib2 = ib2 == 2 ? 0 : ib2 + 1;
        if (c == 0) {   // the NEXT tile is looked up and decoded here, behind this period's requests
          unsigned kn = k_ord + 1;
          has_next = find_tile(kn, Tn);
          k_ord = kn;
          if (has_next) {
            decode_slots(Tn, voff_n);
            rs_in_n = rs_in_of(Tn.b);
          }
        }
        // vector-memory operations retire in order: everything but this period's input request has to be done
        if (req) {
          if (tw == 3) w3_vmwait<6>();
          else w3_vmwait<7>();
        } else {
          w3_vmwait<0>();
        }
        w3_barrier();
        wpar ^= 1;
      }
      tpar ^= 1;
    }
    return;
  }

  // =======================================================================================================================
  // COMPUTE
  // =======================================================================================================================
  const int lane = w3_lane(), l31 = lane & 31, half = lane >> 5;
  const unsigned a_b0 = (unsigned)(half * 128 + mh * 64 + l31) * 16u;   // weight unit [channel block half][channel] of row block mb = 0
  const unsigned obytes = 16u * plane16;
  unsigned pbr[2];
  int vo[2];
  auto decode_pix = [&](const TileS& t) __attribute__((always_inline)) {
    const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
    const int bw = t.bw;
    const int i_lo = w3_udivm1(t.p0, bw, m_bw);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int pp = t.p0 + (nh * 2 + n) * 32 + l31;
      const bool pv = pp < t.p1;
      const int pc = pv ? pp : (t.p1 - 1);
      const int i = w3_udivm1(pc, bw, m_bw), j = pc - i * bw;
      pbr[n] = (unsigned)(2 * (i - i_lo) * W3_LW + j + half * W3_KBU) * 16u;
      vo[n] = (pv && !W3_DBG(1)) ? (i * OW + t.j0 + j) * 16 + half * plane16 : (int)OOB;
    }
  };
  decode_pix(T);
  u32x4 A[3][2], B[3][2];   // the operand sets of the K steps: step s in set s % 3
  int tpar = 0;
  // ---- the epilogue: 4 accumulator tiles -> 8 stores.  Channel group (mb, gp) of pixel group n: 8 values per lane -> bias / scale / LeakyReLU
  //      -> bf16; the two half-waves exchange one 8-byte group so that every lane stores a whole 16-byte unit (conv_bf16_kernel's lean epilogue:
  //      the same formula, the same bits)
  auto epilogue = [&](i32x4 rs_o) __attribute__((always_inline)) {
    if constexpr ((W3_ABL & 8) != 0) return;
    [[maybe_unused]] unsigned sg[2] = {0u, 0u};   // this lane's sign bytes of the tile: [pixel group], byte mb * 2 + gp
    w3_static_for([&](auto q_tag) __attribute__((always_inline)) {
      constexpr int Q = decltype(q_tag)::value, MB = Q >> 1, GP = Q & 1;
      const int cl = mh * 64 + MB * 32 + 16 * GP + 4 * half;   // channel of group A; group B = cl + 8
      const float* eb = ldsf + W3_EB / 4 + cl;
      const float4 bA = *reinterpret_cast<const float4*>(eb), bB = *reinterpret_cast<const float4*>(eb + 8);
      const float ba[8] = {bA.x, bA.y, bA.z, bA.w, bB.x, bB.y, bB.z, bB.w};
      float sa[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, bs[8];
      if constexpr (SCALED) {
        const float* es = ldsf + W3_ES / 4 + tpar * 128 + cl;
        const float4 sA = *reinterpret_cast<const float4*>(es), sB = *reinterpret_cast<const float4*>(es + 8);
        sa[0] = sA.x; sa[1] = sA.y; sa[2] = sA.z; sa[3] = sA.w; sa[4] = sB.x; sa[5] = sB.y; sa[6] = sB.z; sa[7] = sB.w;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) bs[i] = SCALED ? ba[i] * sa[i] : ba[i];
      const unsigned so = (unsigned)(mh * 8 + MB * 4 + GP * 2) * (unsigned)plane16;   // channel block the lower half-wave stores
      w3_static_for([&](auto n_tag) __attribute__((always_inline)) {
        constexpr int n = decltype(n_tag)::value;
        float v[8], r[8];
        w3_acc_read8<16 * (2 * MB + n) + 8 * GP>(r);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = SCALED ? fmaf(r[i], sa[i], bs[i]) : r[i] + bs[i];   // (fma(acc, 1, bias) = acc + bias: the same bits)
        if constexpr (leaky) {   // max(v, 0.01 v) as bare v_max_f32 (conv_bf16_epi_groups.inc's lean order)
          float sv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) sv[i] = v[i] * LEAKY_SLOPE;
#pragma unroll
          for (int i = 0; i < 8; ++i) asm("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(sv[i]));
        }
        if constexpr (write_signs) {   // byte = sum of (v[i] > 0) << i: compare into vcc, add-with-carry shifts it in (values 7 down to 0; conv_bf16_ws.hip)
          unsigned mA = 0;
#define W3_SGN(I) "v_cmp_lt_f32_e32 vcc, 0, %" #I "\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
          asm(W3_SGN(8) W3_SGN(7) W3_SGN(6) W3_SGN(5) W3_SGN(4) W3_SGN(3) W3_SGN(2) "v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
              : "+v"(mA)
              : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7])
              : "vcc");
#undef W3_SGN
          sg[n] |= mA << (8 * Q);
        }
        if (W3_DBG(2)) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = r[i];
        }
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
        const u32x4 w = __builtin_bit_cast(u32x4, o);   // (x, y) = this lane's 4 channels of block cb, (z, w) = of block cb + 1
        const auto r0 = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);
        const u32x4 st = {r0[0], r1[0], r0[1], r1[1]};
        w3_store16(st, vo[n], rs_o, so);
      }, std::make_integer_sequence<int, 2>{});
    }, std::make_integer_sequence<int, 4>{});
    if constexpr (write_signs) {   // this wavefront's 4 sign bytes of a pixel (its 64 channels' share of the half-wave's 8) go out together
      const i32x4 rs_s = w3_rsrc(p.signs + (size_t)T.b * plane * 16, (unsigned)plane * 16u);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int vs = vo[n] < 0 ? (int)OOB : (vo[n] >> 1) + mh * 4;   // ((half * plane + pixel) * 8 bytes: ConvBf16Params::signs)
        asm volatile("s_nop 4\n\tbuffer_store_dword %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(sg[n]), "v"(vs), "s"(w3_u4(rs_s)) : "memory");
      }
    }
  };
  // one period = the nine taps of a 16-channel chunk.  pa / pan: this lane's weight unit in this / the next period's buffer; ib / ibn: input buffers
  auto period = [&](auto first_tag, auto last_tag, unsigned pa, unsigned pan, unsigned ib, unsigned ibn) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value, LASTP = decltype(last_tag)::value;
    const unsigned pb0 = pbr[0] + ib, pb1 = pbr[1] + ib;
    w3_static_for([&](auto s_tag) __attribute__((always_inline)) {
      constexpr int S = decltype(s_tag)::value;
      u32x4(&Ac)[2] = A[S % 3];
      u32x4(&Bc)[2] = B[S % 3];
      if constexpr (S < 7) {
        constexpr int S2 = S + 2;
        w3_k_rd<FIRST && S == 0, S2 * 4096, w3_bimm(S2)>(Ac[0], Ac[1], Bc[0], Bc[1], A[S2 % 3][0], A[S2 % 3][1], B[S2 % 3][0], B[S2 % 3][1], pa, pb0, pb1);
      } else if constexpr (S == 7) {
        w3_k_bar(Ac[0], Ac[1], Bc[0], Bc[1], A[2][0], A[2][1], B[2][0], B[2][1]);   // (step 8's operands: set 8 % 3 = 2)
      } else if constexpr (!LASTP) {
        w3_k_next<w3_bimm(0), w3_bimm(1)>(Ac[0], Ac[1], Bc[0], Bc[1], A[0][0], A[0][1], B[0][0], B[0][1], A[1][0], A[1][1], B[1][0], B[1][1], pan, pbr[0] + ibn,
                                          pbr[1] + ibn);
      } else {
        w3_k_end(Ac[0], Ac[1], Bc[0], Bc[1]);
      }
    }, std::make_integer_sequence<int, 9>{});
  };
  using TT = std::true_type;
  using FT = std::false_type;

  w3_barrier();   // (#1)
  int wpar = 0;
  unsigned ibo = W3_I0;   // input buffer of the chunk being computed (ring of three)
  [[maybe_unused]] unsigned long long t_k = 0, t_e = 0;
  for (;;) {
    const i32x4 rs_o = w3_rsrc(reinterpret_cast<unsigned char*>(p.out) + (size_t)T.b * obytes, obytes);
    [[maybe_unused]] const unsigned long long s0 = W3_STAMP();
    w3_k_first<w3_bimm(0), w3_bimm(1)>(A[0][0], A[0][1], B[0][0], B[0][1], A[1][0], A[1][1], B[1][0], B[1][1], (unsigned)(wpar * W3_WB) + a_b0, pbr[0] + ibo, pbr[1] + ibo);
    for (int c = 0; c < nck; ++c) {
      const unsigned pa = (unsigned)(wpar * W3_WB) + a_b0, pan = (unsigned)((wpar ^ 1) * W3_WB) + a_b0;
      const unsigned ibn = ibo == W3_I0 + 2 * W3_IB ? (unsigned)W3_I0 : ibo + W3_IB;
      if (c == 0) period(TT{}, FT{}, pa, pan, ibo, ibn);          // (nck >= 4)
      else if (c == nck - 1) period(FT{}, TT{}, pa, pan, ibo, ibn);
      else period(FT{}, FT{}, pa, pan, ibo, ibn);
      wpar ^= 1;
      ibo = ibn;
    }
    [[maybe_unused]] const unsigned long long s1 = W3_STAMP();
    epilogue(rs_o);
    [[maybe_unused]] const unsigned long long s2 = W3_STAMP();
    t_k += s1 - s0; t_e += s2 - s1;
    unsigned kn = k_ord + 1;
    const bool has_next = find_tile(kn, T);
    k_ord = kn;
    if (!has_next) break;
    decode_pix(T);
    tpar ^= 1;
  }
#ifdef YOGO_DIAG
  if (p.stamps && tw == 0 && lane == 0) {
    unsigned long long* d = p.stamps + (size_t)blockIdx.x * 16;
    d[0] = t_start; d[1] = __builtin_amdgcn_s_memtime(); d[2] = t_k; d[3] = t_e; d[6] = k_ord;
  }
#endif
}

// =========================================================================================================
// host side: eligibility, tiling, launch
// =========================================================================================================
bool conv_bf16_ws3_eligible(int K, int M, int IH, int IW, int B) {
  const int Kb = round_up(K, 16) / 8;
  if (M != 128 || Kb < 8 || (Kb % 4) != 0) return false;   // nchunk = Kb / 2 even and >= 4
  if (IH < 3 || IW < 3 || B <= 0) return false;
  const long long OH = (IH - 1) / 2 + 1, OW = (IW - 1) / 2 + 1;
  if ((long long)Kb * IH * IW * 16 >= (1ll << 31) || 16ll * OH * OW * 16 >= (1ll << 31)) return false;   // per-image descriptors, bit 31 = "out of range"
  return true;
}

// column bands of TW <= 47 output columns, tiles of PT <= 128 consecutive output pixels of a band (row-major inside the band) that touch at
// most 4 output rows (9 staged input rows): PT = 128 where the narrowest band is >= 43 columns wide (or the image <= 4 rows high), fewer
// otherwise (a tile costs its 128 pixels' MFMAs either way); among the band counts the fewest tiles per image
bool conv_bf16_ws3_plan(ConvWs3Params* p) {
  const int OH = p->OH, OW = p->OW;
  long long best = -1;
  int best_ncb = 0, best_pt = 0;
  for (int ncb = 1; ncb <= 64 && ncb <= OW; ++ncb) {
    const int TW = cdiv(OW, ncb);
    if (cdiv(OW, TW) != ncb || TW > W3_PL - 1) continue;
    const int bw_min = OW - (ncb - 1) * TW;
    const int pt = OH <= 4 ? W3_PT : min(W3_PT, 3 * bw_min + 1);   // 1 + ceil((pt - 1) / bw) <= 4 rows for every band width >= bw_min
    const long long tiles = (long long)(ncb - 1) * cdiv(OH * TW, pt) + cdiv(OH * bw_min, pt);
    if (best < 0 || tiles < best) { best = tiles; best_ncb = ncb; best_pt = pt; }
  }
  if (best < 0) return false;
  p->ncb = best_ncb;
  p->TW = cdiv(OW, best_ncb);
  p->PT = best_pt;
  p->tiles_per_band = cdiv(OH * p->TW, p->PT);
  p->gx = p->ncb * p->tiles_per_band;
  p->ntiles = p->B * p->gx;
  auto magic = [](int d) -> unsigned { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(((1ull << 32) + (unsigned)d - 1ull) / (unsigned)d); };
  const int bw_last = OW - (p->ncb - 1) * p->TW;
  p->m_gx = magic(p->gx); p->m_tpb = magic(p->tiles_per_band);
  p->m_bw = magic(p->TW); p->m_bwl = magic(bw_last);
  p->nchunk = p->Kb / 2;
  // the kernel's divisions by multiplication: tile -> image / band / tile of the band, pixel -> row of its band
  if (!magic_div_exact((long long)p->ntiles - 1, p->gx) || !magic_div_exact(p->gx, p->tiles_per_band) || !magic_div_exact((long long)OH * p->TW, p->TW) ||
      !magic_div_exact((long long)OH * bw_last, bw_last))
    return false;
  return true;
}

int launch_conv_bf16_ws3(const ConvWs3Params& p, hipStream_t stream) {
  static std::mutex mu;
  static int n_cu_of[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    yogo_set_error("conv_bf16_ws3: hipGetDevice failed");
    return YOGO_ERR_HIP;
  }
  int n_cu;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (n_cu_of[dev] == 0) {
      hipError_t e = hipSuccess;
#define W3_ATTR(M) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_ws3_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS_BYTES);
      W3_ATTR(0) W3_ATTR(1) W3_ATTR(2) W3_ATTR(3) W3_ATTR(5) W3_ATTR(7)
#undef W3_ATTR
      if (e != hipSuccess) {
        yogo_set_error("conv_bf16_ws3: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", W3_LDS_BYTES, hipGetErrorString(e));
        return YOGO_ERR_HIP;
      }
      hipDeviceProp_t prop;
      n_cu_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    n_cu = n_cu_of[dev];
  }
  if (p.ntiles <= 0) return YOGO_OK;
  int grid = min(p.ntiles, n_cu);
  if (grid >= 8) grid &= ~7;
  const int mode = (p.act == ACT_LEAKY ? 1 : 0) | (p.chan_scale != nullptr ? 2 : 0) | (p.signs != nullptr ? 4 : 0);   // (a sign map goes with LeakyReLU: the dispatch checks)
#define W3_LAUNCH(M) case M: hipLaunchKernelGGL(conv_bf16_ws3_kernel<M>, dim3(grid), dim3(512), W3_LDS_BYTES, stream, p); break;
  switch (mode) { W3_LAUNCH(0) W3_LAUNCH(1) W3_LAUNCH(2) W3_LAUNCH(3) W3_LAUNCH(5) W3_LAUNCH(7) }
#undef W3_LAUNCH
  if (yogo_launch_log_enabled())
    yogo_launch_log("conv_bf16_ws3_kernel<%d> | Kb=%d in=%dx%d out=%dx%d ncb=%d TW=%d PT=%d tiles_per_band=%d nchunk=%d ntiles=%d grid=%d lds=%d act=%d scale=%d signs=%d", mode, p.Kb,
                    p.IH, p.IW, p.OH, p.OW, p.ncb, p.TW, p.PT, p.tiles_per_band, p.nchunk, p.ntiles, grid, W3_LDS_BYTES, p.act, p.chan_scale != nullptr, p.signs != nullptr);
  YOGO_CHECK_LAUNCH("conv_bf16_ws3");
  return YOGO_OK;
}
