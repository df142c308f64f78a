// Persistent, wavefront-specialised bf16 convolution for the stride-1 3x3 layers with 128 output channels (forward of layers
// 3/5/6 and the data gradients of layers 5/6 of base_model: yogo/model_defns.py:49-65 and their autograd) -- the dominant kernel
// of the training step.  Same arithmetic, accumulation order and epilogue formula as conv_bf16_kernel<4,2,8,...,PP>
// (conv_bf16.hip): outputs are bit-identical (tests/test_gpu_ws.py).
//
// Why this shape.  Measured on the first persistent form (4 wavefronts of 128 x 96 outputs with 192 AGPR accumulators, one per
// SIMD, everything issued from the MFMA wavefronts; round 4, gpurun_out/r4_p4_stamps1.log): parity-green and exactly as fast
// as the tiled kernel -- every vector-memory instruction (an LDS-DMA piece, a 16-byte output store) blocks the issuing
// wavefront for ~60 cycles (MI355X_MICROARCH.md "LDS-DMA piece issue cost"), ~40 of which no MFMA of that SIMD covers, and a
// tile needs 120 pieces + 24 stores per wavefront: 5 k of a tile's 33 k cycles, next to 4.5-8 k of epilogue arithmetic at
// the seam and 2 k of next-tile address decode.  So the vector-memory work moves to wavefronts of its own:
//   * wavefronts 0-3 COMPUTE: one per SIMD, 128 channels x 64 pixels each (8 accumulator tiles of 32x32 in a[0:127], owned by
//     the asm statements below: named literally, listed as clobbers; hipcc never allocates AGPRs here).  Their instruction
//     stream is MFMAs, the 6 ds_read_b128 per 8 MFMAs of the next K step in the MFMA gaps, one ds_write_b128 per step while
//     the previous tile's output is handed over, and one s_barrier per 16-channel chunk;
//   * wavefronts 4-7 LOAD (wavefront w + 4 shares its SIMD with w: tools/probes/wave_simd.hip): per chunk period they issue the
//     13 LDS-DMA pieces each of the NEXT chunk (across tile seams: the stream never stops), decode the next tile's addresses,
//     and move the previous tile's output from the LDS staging area to global memory (4 x 1 KB per period and loader);
//   * the output STORES of tile t overlap tile t + 1: at the seam the compute wavefronts read their accumulators out, apply
//     bias / channel scale / LeakyReLU / bf16 conversion / the half-wave exchange; the first 8 of a lane's 16 16-byte units go
//     straight to the LDS staging area (2 regions x 4 units x 1 KB per wavefront), the other 8 are parked in VGPRs and follow
//     in period 2 of the next tile (one ds_write_b128 per K step); the loaders store the staging area in periods 1 and 3.
//     One barrier per chunk orders everything: DMA landed (loaders wait vmcnt first), staging written (compute waits
//     lgkmcnt), staging read, buffers free.
#include "conv_bf16_ws.h"
#include <mutex>
#include <type_traits>
#include <utility>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// The eight 32x32 accumulator tiles of a compute wavefront are a[0:127], OWNED BY THE ASM STATEMENTS: named literally and listed
// as clobbers, so the kernel descriptor allocates them.  A clobber does not reserve a register, though: in the seam hipcc may park
// values of its own in "free" AGPRs (v_accvgpr_write) -- seen in round 4 on the 256-register instantiations, where the first
// accumulator rows came out wrong.  Pinning the tiles to "+a" C++ variables instead makes hipcc shuffle 16-register tuples
// between every statement (1 700 v_accvgpr moves, 190-400 spilled registers).  So: the seam is kept within the arch VGPRs
// (sched_barrier between its groups), and build.sh AUDITS the assembly of this file: no compiler v_accvgpr_* outside the asm
// statements, .vgpr_spill_count 0, no scratch (cdna_hip_programming.md 5.7 item 4).
// accumulator tile (mb, n) = a[16 * (2 mb + n) : +15]
#define WS_ACC_CLOBBER "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127"
// compile-time ablations (build.sh variant TAG conv_bf16_ws -DWS_ABL=bits; timings only, the results are wrong): see WS_ABLATE below;
// 256 = every 32x32x16 MFMA becomes two 16x16x32 MFMAs on eight of its sixteen accumulator registers (same operand registers, same
// FLOPs, same LDS reads: the in-situ test of the clock the other MFMA shape holds, MI355X_MICROARCH.md "DVFS give-back" item 7)
#ifndef WS_ABL
#define WS_ABL 0
#endif
// one MFMA on accumulator tile T (text of 16 * tile), operands A / B (text), T accumulates (WS_MF) or starts from zero (WS_MFZ)
#if (WS_ABL & 256)
#define WS_MF(T, A, B) "v_mfma_f32_16x16x32_bf16 a[" T ":" T "+3], " A ", " B ", a[" T ":" T "+3]\n\tv_mfma_f32_16x16x32_bf16 a[" T "+4:" T "+7], " A ", " B ", a[" T "+4:" T "+7]"
#define WS_MFZ(T, A, B) "v_mfma_f32_16x16x32_bf16 a[" T ":" T "+3], " A ", " B ", 0\n\tv_mfma_f32_16x16x32_bf16 a[" T "+4:" T "+7], " A ", " B ", 0"
#else
#define WS_MF(T, A, B) "v_mfma_f32_32x32x16_bf16 a[" T ":" T "+15], " A ", " B ", a[" T ":" T "+15]"
#define WS_MFZ(T, A, B) "v_mfma_f32_32x32x16_bf16 a[" T ":" T "+15], " A ", " B ", 0"
#endif
#define WS_MFMA(TILE, AOP, BOP) WS_MF("16*" #TILE, "%[" #AOP "]", "%[" #BOP "]") "\n\t"
#define WS_MFMA0(TILE, AOP, BOP) WS_MFZ("16*" #TILE, "%[" #AOP "]", "%[" #BOP "]") "\n\t"
// operand reads of tap T1 (kernel column KX1): weights at pa + T1 * 4 KB + mb * 512 B, input at the row base + KX1 * 16 B
// (T1 / KX1 name "n" operands of the statement, or are literal numbers)
#define WS_RDA(DST, MB, T1) "ds_read_b128 %[" #DST "], %[pa] offset:4096*" T1 "+512*" #MB "\n\t"
#define WS_RDB(DST, SRC, KX1) "ds_read_b128 %[" #DST "], %[" #SRC "] offset:16*" KX1 "\n\t"

namespace {

__device__ __forceinline__ int ws_udivm(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }   // n / d, m = ceil(2^32 / d), d > 1
__device__ __forceinline__ int ws_udivm1(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }
// lane index, re-derived at every use (volatile: never merged with an earlier copy that would have to stay live or be spilled)
__device__ __forceinline__ int ws_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ i32x4 ws_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
// one LDS-DMA piece (64 lanes x 16 bytes -> LDS bytes [lds, lds + 1024))
__device__ __forceinline__ void ws_dma1(i32x4 rs, unsigned lds, int voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" ::"v"(voff), "s"(lds), "s"(rs), "s"(soff) : "memory");
}
// four pieces of one descriptor with one scalar offset: LDS destinations lds + k * 4 KB (the input slots of a chunk)
__device__ __forceinline__ void ws_dma4(i32x4 rs, unsigned lds, int v0, int v1, int v2, int v3, unsigned soff) {
  asm volatile(
      "s_mov_b32 m0, %5\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %4, %6 offen lds\n\t"
      "s_add_u32 m0, m0, 4096\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %4, %6 offen lds\n\t"
      "s_add_u32 m0, m0, 4096\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, %6 offen lds\n\t"
      "s_add_u32 m0, m0, 4096\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, %6 offen lds"
      ::"v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(rs), "s"(lds), "s"(soff) : "memory", "scc");
}
// nine pieces of one descriptor with one per-lane offset: LDS destinations lds + j * 4 KB, scalar offsets soff + j * step (the
// weight slices of the nine taps): three scalar instructions per piece -- a loader beside an MFMA-saturating partner issues
// about one instruction per 10 cycles, whatever its kind
__device__ __forceinline__ void ws_dma9(i32x4 rs, unsigned lds, int voff, unsigned soff, unsigned step) {
  unsigned so;
#define WS_P9 "s_add_u32 m0, m0, 4096\n\ts_add_u32 %0, %0, %5\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\t"
  asm volatile("s_mov_b32 m0, %3\n\ts_mov_b32 %0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\t" WS_P9 WS_P9 WS_P9 WS_P9 WS_P9 WS_P9 WS_P9 WS_P9
               : "=&s"(so)
               : "v"(voff), "s"(rs), "s"(lds), "s"(soff), "s"(step)
               : "memory", "scc");
#undef WS_P9
}
// 64 lanes x 4 bytes -> LDS bytes [lds, lds + 256)
__device__ __forceinline__ void ws_dma_dword(i32x4 rs, unsigned lds, int voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dword %0, %2, %3 offen lds" ::"v"(voff), "s"(lds), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void ws_store16(u32x4 data, int voff, i32x4 rs, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(data), "v"(voff), "s"(rs), "s"(soff) : "memory");   // (s_nop: descriptor from v_readfirstlane)
}

template <class F, int... I>
__device__ __forceinline__ void ws_static_for(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

// ---- the asm statements of a K step (tap) of a compute wavefront ---------------------------------------------------------
// first half: accumulator rows mb = 0, 1 (4 MFMAs) with the 6 operand reads of the NEXT step in their gaps.  The reads are
// retired by the lgkmcnt(0) that ends the second half (ws_sb), which names their destinations "+v".
template <bool ZERO, int T1, int KX1>
__device__ __forceinline__ void ws_sa(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1,
                                      u32x4& an0, u32x4& an1, u32x4& an2, u32x4& an3, u32x4& bn0, u32x4& bn1, unsigned pa, unsigned pb0, unsigned pb1) {
#define WS_SA_BODY(M)                                                                                         \
  M(0, a0, b0) WS_RDA(an0, 0, "%[t1]") WS_RDB(bn0, pb0, "%[kx1]") M(1, a0, b1) WS_RDA(an1, 1, "%[t1]") WS_RDB(bn1, pb1, "%[kx1]") \
  M(2, a1, b0) WS_RDA(an2, 2, "%[t1]") M(3, a1, b1) WS_RDA(an3, 3, "%[t1]")
#define WS_SA_OPS                                                                                                        \
  : [an0] "=&v"(an0), [an1] "=&v"(an1), [an2] "=&v"(an2), [an3] "=&v"(an3), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1)            \
  : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1), [t1] "n"(T1), [kx1] "n"(KX1) \
  : "memory", WS_ACC_CLOBBER
  if constexpr (ZERO) asm volatile(WS_SA_BODY(WS_MFMA0) WS_SA_OPS);
  else asm volatile(WS_SA_BODY(WS_MFMA) WS_SA_OPS);
#undef WS_SA_BODY
#undef WS_SA_OPS
}
// second half: rows mb = 2, 3; optionally one parked output unit goes to the staging area (ds_write_b128 at stg + SOFF); ends
// by retiring the first half's operand reads (and the staging write)
template <bool ZERO, bool STAGE, int SOFF>
__device__ __forceinline__ void ws_sb(const u32x4& a2, const u32x4& a3, const u32x4& b0, const u32x4& b1,
                                      u32x4& an0, u32x4& an1, u32x4& an2, u32x4& an3, u32x4& bn0, u32x4& bn1, unsigned stg, const u32x2& sda,
                                      const u32x2& sdb) {
#define WS_SB_OPS                                                                                                        \
  : [an0] "+v"(an0), [an1] "+v"(an1), [an2] "+v"(an2), [an3] "+v"(an3), [bn0] "+v"(bn0), [bn1] "+v"(bn1)                  \
  : [a2] "v"(a2), [a3] "v"(a3), [b0] "v"(b0), [b1] "v"(b1), [stg] "v"(stg), [sda] "v"(sda), [sdb] "v"(sdb), [soff] "n"(SOFF) \
  : "memory", WS_ACC_CLOBBER
// (LDS operations retire in order: with the staging write behind the six reads, lgkmcnt(1) retires the reads and leaves the write --
//  ~150 cycles until it is acknowledged -- in flight; the chunk's barrier statement waits for it)
#define WS_SB_BODY(M, ST, W) M(4, a2, b0) ST M(5, a2, b1) M(6, a3, b0) M(7, a3, b1) "s_waitcnt lgkmcnt(" W ")"
// (a parked unit is un-swapped: this lane's 8 bytes of channel block cb go to the lower 512 B of the unit, those of cb + 1 above)
#define WS_STW "ds_write2st64_b64 %[stg], %[sda], %[sdb] offset0:%[soff]/512 offset1:%[soff]/512+1\n\t"
  if constexpr (ZERO) {
    if constexpr (STAGE) asm volatile(WS_SB_BODY(WS_MFMA0, WS_STW, "1") WS_SB_OPS);
    else asm volatile(WS_SB_BODY(WS_MFMA0, "", "0") WS_SB_OPS);
  } else {
    if constexpr (STAGE) asm volatile(WS_SB_BODY(WS_MFMA, WS_STW, "1") WS_SB_OPS);
    else asm volatile(WS_SB_BODY(WS_MFMA, "", "0") WS_SB_OPS);
  }
#undef WS_SB_BODY
#undef WS_SB_OPS
}
// step 8, first part: row mb = 0, then the chunk's barrier (the loaders arrive with the next chunk landed)
__device__ __forceinline__ void ws_x8(const u32x4& a0, const u32x4& b0, const u32x4& b1) {
  asm volatile(WS_MFMA(0, a0, b0) WS_MFMA(1, a0, b1) "s_waitcnt lgkmcnt(0)\n\ts_barrier" : : [a0] "v"(a0), [b0] "v"(b0), [b1] "v"(b1) : "memory", WS_ACC_CLOBBER);
}
// step 8, second part: rows mb = 1..3 with the operand reads of the next chunk's step 0 (the other buffer) up front
__device__ __forceinline__ void ws_y8(const u32x4& a1, const u32x4& a2,
                                      const u32x4& a3, const u32x4& b0, const u32x4& b1, u32x4& an0, u32x4& an1, u32x4& an2, u32x4& an3, u32x4& bn0,
                                      u32x4& bn1, unsigned pa, unsigned pb0, unsigned pb1) {
  asm volatile(WS_MFMA(2, a1, b0) WS_RDA(an0, 0, "0") WS_RDB(bn0, pb0, "0") WS_MFMA(3, a1, b1) WS_RDA(an1, 1, "0") WS_RDB(bn1, pb1, "0")
               WS_MFMA(4, a2, b0) WS_RDA(an2, 2, "0") WS_MFMA(5, a2, b1) WS_RDA(an3, 3, "0") WS_MFMA(6, a3, b0) WS_MFMA(7, a3, b1) "s_waitcnt lgkmcnt(0)"
               : [an0] "=&v"(an0), [an1] "=&v"(an1), [an2] "=&v"(an2), [an3] "=&v"(an3), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1)
               : [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [b0] "v"(b0), [b1] "v"(b1), [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1)
               : "memory", WS_ACC_CLOBBER);
}
// ... without reads (the tile's last chunk: the seam comes first, the next tile's operands after it)
// (ends with the wait states between the last MFMA and the first read of an accumulator register: 16 passes)
__device__ __forceinline__ void ws_y8_last(const u32x4& a1, const u32x4& a2, const u32x4& a3, const u32x4& b0, const u32x4& b1) {
  asm volatile(WS_MFMA(2, a1, b0) WS_MFMA(3, a1, b1) WS_MFMA(4, a2, b0) WS_MFMA(5, a2, b1) WS_MFMA(6, a3, b0) WS_MFMA(7, a3, b1) "s_nop 15\n\ts_nop 15"
               :
               : [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [b0] "v"(b0), [b1] "v"(b1)
               : "memory", WS_ACC_CLOBBER);
}
// eight consecutive accumulator registers -> VGPRs in ONE statement (hipcc pads every asm statement with a wait state).  The
// statement also CLOBBERS every accumulator register: hipcc then cannot keep a value of its own in an AGPR across any part of the
// seam (a clobbered register holds nothing live across the statement)
template <int R>
__device__ __forceinline__ void ws_acc_read8(float (&r)[8]) {
  asm volatile(
      "v_accvgpr_read_b32 %0, a[%8]\n\tv_accvgpr_read_b32 %1, a[%8+1]\n\tv_accvgpr_read_b32 %2, a[%8+2]\n\tv_accvgpr_read_b32 %3, a[%8+3]\n\t"
      "v_accvgpr_read_b32 %4, a[%8+4]\n\tv_accvgpr_read_b32 %5, a[%8+5]\n\tv_accvgpr_read_b32 %6, a[%8+6]\n\tv_accvgpr_read_b32 %7, a[%8+7]"
      : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7])
      : "n"(R)
      : WS_ACC_CLOBBER);
}

// ---- group-major K steps (the first and the last chunk of a tile): ONE MFMA per statement, so that C++ code placed between the
// statements -- the epilogue of the other half of the accumulators -- lands in the MFMA gaps.  Accumulator rows mb = 0, 1 are
// group 0 (tiles 0..3), rows 2, 3 group 1 (tiles 4..7); a group-major chunk runs the 9 taps of group 0, then those of group 1.
// MFMA + the reads of ONE weight quad (row MBA, tap T1) and ONE input quad (kernel column KX1) of the next step
template <int TILE, bool ZERO, int MBA, int T1, int KX1>
__device__ __forceinline__ void ws_g1(const u32x4& a, const u32x4& b, u32x4& an, u32x4& bn, unsigned pa, unsigned pb) {
#define WS_G1_OPS : [an] "=&v"(an), [bn] "=&v"(bn) : [a] "v"(a), [b] "v"(b), [pa] "v"(pa), [pb] "v"(pb), [tl] "n"(TILE), [mba] "n"(MBA), [t1] "n"(T1), [kx1] "n"(KX1) : "memory", WS_ACC_CLOBBER
#define WS_G1_RD "ds_read_b128 %[an], %[pa] offset:4096*%[t1]+512*%[mba]\n\tds_read_b128 %[bn], %[pb] offset:16*%[kx1]"
  if constexpr (ZERO) asm volatile(WS_MFZ("16*%[tl]", "%[a]", "%[b]") "\n\t" WS_G1_RD WS_G1_OPS);
  else asm volatile(WS_MF("16*%[tl]", "%[a]", "%[b]") "\n\t" WS_G1_RD WS_G1_OPS);
#undef WS_G1_OPS
#undef WS_G1_RD
}
// bare MFMA; TAIL: text behind it ("s_barrier", or nothing)
template <int TILE, bool ZERO, bool BARRIER = false>
__device__ __forceinline__ void ws_g0(const u32x4& a, const u32x4& b) {
#define WS_G0_OPS : : [a] "v"(a), [b] "v"(b), [tl] "n"(TILE) : "memory", WS_ACC_CLOBBER
  if constexpr (ZERO) asm volatile(WS_MFZ("16*%[tl]", "%[a]", "%[b]") "" WS_G0_OPS);
  else if constexpr (BARRIER) asm volatile(WS_MF("16*%[tl]", "%[a]", "%[b]") "\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" WS_G0_OPS);
  else asm volatile(WS_MF("16*%[tl]", "%[a]", "%[b]") "" WS_G0_OPS);
#undef WS_G0_OPS
}
// MFMA, then the step's four operand reads are retired
template <int TILE, bool ZERO>
__device__ __forceinline__ void ws_gw(const u32x4& a, const u32x4& b, u32x4& an0, u32x4& an1, u32x4& bn0, u32x4& bn1) {
#define WS_GW_OPS : [an0] "+v"(an0), [an1] "+v"(an1), [bn0] "+v"(bn0), [bn1] "+v"(bn1) : [a] "v"(a), [b] "v"(b), [tl] "n"(TILE) : "memory", WS_ACC_CLOBBER
  if constexpr (ZERO) asm volatile(WS_MFZ("16*%[tl]", "%[a]", "%[b]") "\n\ts_waitcnt lgkmcnt(0)" WS_GW_OPS);
  else asm volatile(WS_MF("16*%[tl]", "%[a]", "%[b]") "\n\ts_waitcnt lgkmcnt(0)" WS_GW_OPS);
#undef WS_GW_OPS
}
// last step of a tile's FIRST chunk, behind the barrier: MFMA + three of the six operand quads of the next (tap-major) chunk's
// step 0 -- weight rows MB0, MB0 + 1 and input quad pb; WAIT: retire all six (x0..x2 are the first statement's)
template <int TILE, int MB0, bool WAIT>
__device__ __forceinline__ void ws_g3(const u32x4& a, const u32x4& b, u32x4& an0, u32x4& an1, u32x4& bn, unsigned pa, unsigned pb, u32x4& x0, u32x4& x1,
                                      u32x4& x2) {
  if constexpr (WAIT)
    asm volatile(WS_MF("16*%[tl]", "%[a]", "%[b]") "\n\t"
                 "ds_read_b128 %[an0], %[pa] offset:512*%[mb0]\n\tds_read_b128 %[an1], %[pa] offset:512*%[mb0]+512\n\tds_read_b128 %[bn], %[pb]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [an0] "=&v"(an0), [an1] "=&v"(an1), [bn] "=&v"(bn), [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2)
                 : [a] "v"(a), [b] "v"(b), [pa] "v"(pa), [pb] "v"(pb), [tl] "n"(TILE), [mb0] "n"(MB0)
                 : "memory", WS_ACC_CLOBBER);
  else
    asm volatile(WS_MF("16*%[tl]", "%[a]", "%[b]") "\n\t"
                 "ds_read_b128 %[an0], %[pa] offset:512*%[mb0]\n\tds_read_b128 %[an1], %[pa] offset:512*%[mb0]+512\n\tds_read_b128 %[bn], %[pb]"
                 : [an0] "=&v"(an0), [an1] "=&v"(an1), [bn] "=&v"(bn)
                 : [a] "v"(a), [b] "v"(b), [pa] "v"(pa), [pb] "v"(pb), [tl] "n"(TILE), [mb0] "n"(MB0)
                 : "memory", WS_ACC_CLOBBER);
}
// operand reads of a group-major chunk's step 0 (weight rows 0, 1 and both input quads of tap 0) without MFMAs
__device__ __forceinline__ void ws_prefetch4(u32x4& an0, u32x4& an1, u32x4& bn0, u32x4& bn1, unsigned pa, unsigned pb0, unsigned pb1) {
  asm volatile(WS_RDA(an0, 0, "0") WS_RDA(an1, 1, "0") WS_RDB(bn0, pb0, "0") WS_RDB(bn1, pb1, "0") "s_waitcnt lgkmcnt(0)"
               : [an0] "=&v"(an0), [an1] "=&v"(an1), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1)
               : [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1)
               : "memory");
}
// four consecutive accumulator registers -> VGPRs (see ws_acc_read8: the statement clobbers every accumulator register)
template <int R>
__device__ __forceinline__ void ws_acc_read4(float& r0, float& r1, float& r2, float& r3) {
#if defined(WS_ABL) && (WS_ABL & 64)
  asm volatile("v_mov_b32 %0, 1.0\n\tv_mov_b32 %1, 1.0\n\tv_mov_b32 %2, 1.0\n\tv_mov_b32 %3, 1.0" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "n"(R) : WS_ACC_CLOBBER);
  return;
#endif
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%4+1]\n\tv_accvgpr_read_b32 %2, a[%4+2]\n\tv_accvgpr_read_b32 %3, a[%4+3]"
               : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
               : "n"(R)
               : WS_ACC_CLOBBER);
}
// operand reads of a chunk's step 0 without MFMAs
__device__ __forceinline__ void ws_prefetch(u32x4& an0, u32x4& an1, u32x4& an2, u32x4& an3, u32x4& bn0, u32x4& bn1, unsigned pa, unsigned pb0, unsigned pb1) {
  asm volatile(WS_RDA(an0, 0, "0") WS_RDA(an1, 1, "0") WS_RDA(an2, 2, "0") WS_RDA(an3, 3, "0") WS_RDB(bn0, pb0, "0") WS_RDB(bn1, pb1, "0")
               "s_waitcnt lgkmcnt(0)"
               : [an0] "=&v"(an0), [an1] "=&v"(an1), [an2] "=&v"(an2), [an3] "=&v"(an3), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1)
               : [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1)
               : "memory");
}

// compile-time ablations (build.sh variant TAG conv_bf16_ws -DWS_ABL=bits; timings only, the results are wrong): 8 = the loaders do
// not decode the next tile (its input then comes out of L2: the clock rises), 16 = no epilogue in the gaps of the
// group-major chunks, 32 = the slot decode twice, 64 = plain moves instead of accumulator reads, 128 = chunk 1 does not stage the parked units
// (0.3 k of layer 5's 22 k cycles per tile)
#define WS_ABLATE(BIT) ((WS_ABL & (BIT)) != 0)
#ifdef YOGO_DIAG
#define WS_DBG(BIT) (p.dbg & (BIT))
#define WS_STAMP() __builtin_amdgcn_s_memtime()
#else
#define WS_DBG(BIT) 0
#define WS_STAMP() 0ull
#endif

template <int N> using WsIC = std::integral_constant<int, N>;

}  // namespace

// MODE: the epilogue the kernel is compiled for, bit 0 = LeakyReLU, bit 1 = + sign map of the output, bit 2 = channel scale
// (Dropout2d mask).  0: conv + bias (layers 5 / 6 forward, data gradients without a channel mask); 1: eval-mode forward; 7: layer 3
// forward of the training step; 4 / 5 / 3: the other combinations a ModelDefn can ask for.  The epilogue is bound by
// vector-instruction issue, so what a launch does not need is compiled out -- and a variant that kept every run-time flag's
// operands live did not fit the 128 arch VGPRs.
#ifndef WS_PRIO_COMPUTE
#define WS_PRIO_COMPUTE 3
#endif
#ifndef WS_PRIO_LOADER
#define WS_PRIO_LOADER 0
#endif
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_bf16_ws_kernel(const ConvWsParams p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
  constexpr unsigned OOB = 0x80000000u;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, tw = wave & 3;   // team 0 computes, team 1 loads; wavefronts tw and tw + 4 share a SIMD
  [[maybe_unused]] const unsigned long long t_start = WS_STAMP();
  float* ldsf = reinterpret_cast<float*>(smem4);

  const int OH = p.IH, OW = p.IW;
  const int plane = OH * OW, plane16 = plane * 16;
  const int nck = p.nchunk;

  // ---- tile walk: virtual block lin = slot + k * G, remapped so that an XCD's workgroups share a contiguous run of tiles
  const unsigned NV = (unsigned)p.ntiles, G = gridDim.x, slot = blockIdx.x;
  const unsigned xq = NV >> 3, xr = NV & 7;
  struct TileS { int b, j0, bw, p0, p1, lastband; };
  auto find_tile = [&](unsigned& k, TileS& t) -> bool {   // (uniform) next non-empty tile of this workgroup from ordinal k on
    for (;; ++k) {
      const unsigned lin = slot + k * G;
      if (lin >= NV) return false;
      const unsigned xcd = lin & 7;
      const int widx = (int)((xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3));
      const int b = ws_udivm1(widx, p.gx, p.m_gx);
      const int bx = widx - b * p.gx;
      const int cb = ws_udivm1(bx, p.tiles_per_band, p.m_tpb);
      const int tb = bx - cb * p.tiles_per_band;
      const int j0 = cb * p.TW;
      const int bw = min(p.TW, OW - j0);
      const int NPb = OH * bw;
      const int p0 = tb * WS_PT;
      if (p0 >= NPb) continue;
      t.b = b; t.j0 = j0; t.bw = bw; t.p0 = p0; t.p1 = min(p0 + WS_PT, NPb); t.lastband = cb == p.ncb - 1;
      return true;
    }
  };
  // per-lane geometry of the two pixel groups of wavefront tw (compute: operand addresses; loader: output offsets of its partner)
  auto decode_pix = [&](const TileS& t, unsigned (&pbr)[2], int (&vo)[2], unsigned& lw16) {
    // (the lane's constants are re-derived from v_mbcnt: kept live across the tile loop -- or derived from threadIdx.x, which then
    //  has to stay live -- they cost registers the 256-register instantiations do not have and end up in scratch)
    const int ln_ = ws_lane();
    const int l31 = ln_ & 31, half = ln_ >> 5;
    const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
    const int bw = t.bw;   // >= 2 (planner)
    const int i_lo = ws_udivm(t.p0, m_bw), i_hi = ws_udivm(t.p1 - 1, m_bw);
    const int rows_in = i_hi - i_lo + 3;
    const int lw = bw + 2;
    const int per_kb = rows_in * lw;
    lw16 = (unsigned)lw * 16u;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int pp = t.p0 + (tw * 2 + n) * 32 + l31;
      const bool pv = pp < t.p1;
      const int pc = pv ? pp : (t.p1 - 1);
      const int i = ws_udivm(pc, m_bw), j = pc - i * bw;
      pbr[n] = (unsigned)((i - i_lo) * lw + j + half * per_kb) * 16u;
      vo[n] = pv ? (i * OW + t.j0 + j) * 16 + half * plane16 : (int)OOB;
    }
  };

  unsigned k_ord = 0;
  TileS T{};
  if (!find_tile(k_ord, T)) return;

  // bias (the same for every tile) and a unit channel scale when there is none
  if (tid < 128) {
    ldsf[WS_EB / 4 + tid] = p.bias != nullptr ? p.bias[tid] : 0.f;
    if (p.chan_scale == nullptr) {
      ldsf[WS_ES / 4 + tid] = 1.f;
      ldsf[WS_ES / 4 + 128 + tid] = 1.f;
    }
  }
  __syncthreads();

  if (team == 1) {
    // (lane indices are re-derived inside each role: a value of the prologue that both roles use stays live across the other
    //  role's code in hipcc's layout and is spilled there)
    const int lane = ws_lane();
    const int ttid = tw * 64 + lane;   // thread index inside the team
    if (WS_PRIO_LOADER) __builtin_amdgcn_s_setprio(WS_PRIO_LOADER);
    // =====================================================================================================================
    // LOADERS: per chunk period -- request the next chunk (13 LDS-DMA pieces per wavefront), move the staging region written
    // in the previous period to global memory, wait for the requests, barrier.
    // =====================================================================================================================
    const int rowb = p.IW * 16, kcb = p.IH * p.IW * 16;
    const unsigned ibytes = (unsigned)p.Kb * kcb, obytes = 16u * plane16, wbytes = 9u * p.Kb * 2048u;
    const unsigned so_i = 2u * kcb;                  // bytes between the 16-channel chunks of an image
    const unsigned wstep = (unsigned)p.Kb * 2048u;   // bytes between the taps of the packed weights
    const i32x4 rs_w = ws_rsrc(p.wp, wbytes);
    const int lane16 = WS_DBG(4) ? (int)OOB : lane * 16;
    const int kbw = tw >> 1, colh = tw & 1;          // this wavefront's weight pieces: channel block of the chunk, column half
    const bool has_scale = p.chan_scale != nullptr;
    const i32x4 rs_sc = ws_rsrc(p.chan_scale, has_scale ? (unsigned)p.B * 512u : 0u);
    auto issue_scale = [&](int b, int par) {   // [128] channel scale of image b -> es[par] (loaders 0 and 1, 64 floats each)
      if (has_scale && tw < 2)
        ws_dma_dword(rs_sc, (unsigned)__builtin_amdgcn_readfirstlane(WS_ES + par * 512 + tw * 256), lane * 4,
                     (unsigned)__builtin_amdgcn_readfirstlane((b * 128 + tw * 64) * 4));
    };
    // DMA source offsets of the 4 input slots: element ttid + i * 256 of the flattened [2][rows_in][lw] tile -> (channel
    // block, row, column): slot 0 by division, the others by stepping with two carries
    auto decode_slots = [&](const TileS& t, int (&voff)[WS_NI]) {
      const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
      const int bw = t.bw;
      const int i_lo = ws_udivm(t.p0, m_bw), i_hi = ws_udivm(t.p1 - 1, m_bw);
      const int rows_in = i_hi - i_lo + 3;
      const int iy0 = i_lo - 1, ix0 = t.j0 - 1;
      const int lw = bw + 2;
      const int per_kb = rows_in * lw;
      const unsigned inv_lw = t.lastband ? p.m_lwl : p.m_lw;
      const unsigned inv_perkb = 0xFFFFFFFFu / (unsigned)per_kb + 1u;
      const int skc = ws_udivm(WS_NT, inv_perkb);
      const int srm = WS_NT - skc * per_kb;
      const int sr = ws_udivm(srm, inv_lw);
      const int sx = srm - sr * lw;
      int kc_ = ws_udivm(ttid, inv_perkb);
      const int rm0 = ttid - kc_ * per_kb;
      int r_ = ws_udivm(rm0, inv_lw);
      int x_ = rm0 - r_ * lw;
#pragma unroll
      for (int i = 0; i < WS_NI; ++i) {
        const int iy_ = iy0 + r_, ix_ = ix0 + x_;
        const bool ok = (kc_ < 2) && (iy_ >= 0) && (iy_ < p.IH) && (ix_ >= 0) && (ix_ < p.IW) && !WS_DBG(4);
        voff[i] = ok ? kc_ * kcb + iy_ * rowb + ix_ * 16 : (int)OOB;
        x_ += sx; r_ += sr; kc_ += skc;
        if (x_ >= lw) { x_ -= lw; ++r_; }
        if (r_ >= rows_in) { r_ -= rows_in; ++kc_; }
      }
    };
    static_assert(WS_NI == 4, "ws_dma4 issues the four input slots");
    // weight slices of chunk cn (9 pieces of this wavefront) -> weight buffer cn & 1 (nchunk is even)
    auto req_w = [&](int cn) {
      const unsigned wb = (unsigned)(((2 * cn + kbw) * 128 + colh * 64) * 16);
      ws_dma9(rs_w, (unsigned)((cn & 1) * WS_WB + (kbw * 128 + colh * 64) * 16), lane16, wb, wstep);
    };
    // input tile of chunk cn of the tile described by (rs, voff) (4 pieces of this wavefront) -> input buffer ib
    auto req_i = [&](i32x4 rs, const int (&voff)[WS_NI], int cn, int ib) {
      ws_dma4(rs, (unsigned)(WS_I0 + ib * WS_IB + tw * 64 * 16), voff[0], voff[1], voff[2], voff[3], (unsigned)cn * so_i);
    };
    // Output hand-over.  The 8 staged units of compute wavefront tw (units U0 .. U0 + 7; unit = 2 q + n, q = 2 mb + gp: channel
    // block 2 q, pixel group n) are read into registers in periods 1 and 3 and stored FOUR PER PERIOD (periods 1, 2, 3 and 0 of
    // the next tile): 16 KB per CU and period.  All eight in one period is 32 KB = 3 k cycles of the chip's write rate (~10 B/clk
    // and CU with every CU storing, tools/probes/vmem_rate.hip), longer than the period -- the barrier then waits for the stores.
    const unsigned stg_rd = (unsigned)(WS_STG + tw * 4096 + lane * 16);
    u32x4 fifo[8];
    int f_vo[2] = {(int)OOB, (int)OOB};   // output offsets / descriptor / first unit of the tile the FIFO holds
    i32x4 f_rs = ws_rsrc(p.out, 0u);
    int f_u0 = 0;
    auto fifo_fill = [&](int U0, const int (&vop)[2], i32x4 rs_o) {
      const unsigned char* base = reinterpret_cast<const unsigned char*>(smem4) + stg_rd;
#pragma unroll
      for (int u = 0; u < 8; ++u) fifo[u] = *reinterpret_cast<const u32x4*>(base + (u >> 2) * 16384 + (u & 3) * 1024);
      f_vo[0] = vop[0]; f_vo[1] = vop[1]; f_rs = rs_o; f_u0 = U0;
    };
    auto fifo_store4 = [&](auto h_tag) {   // units 4 h .. 4 h + 3 of the FIFO
      constexpr int Hh = decltype(h_tag)::value;
      // (the descriptor and the unit base travel through assignments hipcc cannot prove uniform)
      const i32x4 rs = {__builtin_amdgcn_readfirstlane(f_rs.x), __builtin_amdgcn_readfirstlane(f_rs.y), __builtin_amdgcn_readfirstlane(f_rs.z),
                        __builtin_amdgcn_readfirstlane(f_rs.w)};
      const int u0 = __builtin_amdgcn_readfirstlane(f_u0);
#pragma unroll
      for (int u = 4 * Hh; u < 4 * Hh + 4; ++u) {
        const int q = (u0 + u) >> 1;
        ws_store16(fifo[u], (u & 1) ? f_vo[1] : f_vo[0], rs, (unsigned)(2 * q) * (unsigned)plane16);
      }
    };

    int voff[WS_NI];
    unsigned pbr_[2], lw16_;
    int vo[2], vo_prev[2] = {(int)OOB, (int)OOB};
    decode_slots(T, voff);
    decode_pix(T, pbr_, vo, lw16_);
    i32x4 rs_in = ws_rsrc(reinterpret_cast<const unsigned char*>(p.in) + (size_t)T.b * ibytes, ibytes);
    i32x4 rs_out = ws_rsrc(reinterpret_cast<unsigned char*>(p.out) + (size_t)T.b * obytes, obytes);
    i32x4 rs_out_prev = ws_rsrc(p.out, 0u);
    int tpar = 0;
    issue_scale(T.b, 0);
    req_i(rs_in, voff, 0, 0);
    req_i(rs_in, voff, 1, 1);
    req_w(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // (#1) chunks 0 (and the input of chunk 1) of the first tile have landed
    [[maybe_unused]] unsigned long long t_wait = 0, t_work = 0, t_wp[4] = {0, 0, 0, 0}, t_wv[4] = {0, 0, 0, 0};
    bool has_next = true;
    int ib2 = 2;   // ring slot of the next input request (the chunk two periods ahead)
    // the NEXT tile is looked up and decoded in period 0, behind that period's requests and stores: its ~1.5 k cycles of scalar
    // and vector work then run while the requests are in flight and the compute wavefronts are in their longest chunk (the
    // group-major first chunk with the previous tile's epilogue in its gaps).  (In period nchunk - 2, where the request stream
    // needs it, it delayed the barrier by that much once per tile.)
    TileS Tn{};
    int voff_n[WS_NI] = {(int)OOB, (int)OOB, (int)OOB, (int)OOB}, vo_n[2] = {(int)OOB, (int)OOB};
    i32x4 rs_in_n = rs_in, rs_out_n = rs_out;
    while (has_next) {
      for (int c = 0; c < nck; ++c) {
        [[maybe_unused]] const unsigned long long tw0 = WS_STAMP();
        // oldest first: the weight slices of the NEXT chunk (needed at this period's barrier) ...
        if (c + 1 < nck) req_w(c + 1);
        else if (has_next) req_w(0);
        // ... then the input tile of the chunk after it (needed one barrier later: it may stay in flight)
        bool req = false;   // (uniform)
        if (c + 2 < nck) {
          req_i(rs_in, voff, c + 2, ib2);
          req = true;
        } else {
          if (c + 2 == nck && has_next) {   // the request stream crosses into the next tile
#pragma unroll
            for (int i = 0; i < WS_NI; ++i) voff[i] = voff_n[i];
            rs_in = rs_in_n;
            issue_scale(Tn.b, tpar ^ 1);
          }
          if (has_next) {
            req_i(rs_in, voff, c + 2 - nck, ib2);
            req = true;
          }
        }
        ib2 = ib2 == 2 ? 0 : ib2 + 1;
        // period 0: the units the compute wavefronts wrote during the previous tile's last chunk; period 2: the parked units they
        // staged in period 1 -> the register FIFO; four stores in each of the periods 0..3
        if (c == 0) { fifo_fill(0, vo_prev, rs_out_prev); fifo_store4(WsIC<0>{}); }
        else if (c == 1) fifo_store4(WsIC<1>{});
        else if (c == 2) { fifo_fill(8, vo_prev, rs_out_prev); fifo_store4(WsIC<0>{}); }
        else if (c == 3) fifo_store4(WsIC<1>{});
        const bool dr = c < 4;   // (uniform) four stores were issued
        if (c == 0) {
          unsigned kn = k_ord + 1;
          has_next = find_tile(kn, Tn);
          k_ord = kn;
          if (has_next && WS_ABLATE(8)) {
#pragma unroll
            for (int i = 0; i < WS_NI; ++i) voff_n[i] = voff[i];
          } else if (has_next) {
            decode_slots(Tn, voff_n);
            if constexpr (WS_ABLATE(32)) {
              TileS T2 = Tn;
              asm volatile("" : "+s"(T2.p0), "+s"(T2.p1), "+s"(T2.bw));
              int v2[WS_NI];
              decode_slots(T2, v2);
#pragma unroll
              for (int i = 0; i < WS_NI; ++i) asm volatile("" : : "v"(v2[i]));
            }
            decode_pix(Tn, pbr_, vo_n, lw16_);
            rs_in_n = ws_rsrc(reinterpret_cast<const unsigned char*>(p.in) + (size_t)Tn.b * ibytes, ibytes);
            rs_out_n = ws_rsrc(reinterpret_cast<unsigned char*>(p.out) + (size_t)Tn.b * obytes, obytes);
          }
        }
        // the NEXT tile's geometry for the partner compute wavefront (same lanes, same pixels): written in period 1 -- the compute
        // wavefront read the previous message at its tile seam, in front of this tile's period 0 -- and read at the next seam
        if (c == 1) {
          unsigned char* mb = reinterpret_cast<unsigned char*>(smem4);
          *reinterpret_cast<u32x4*>(mb + WS_MB + (tw * 64 + lane) * 16) = u32x4{pbr_[0], pbr_[1], (unsigned)vo_n[0], (unsigned)vo_n[1]};
          if (tw == 0 && lane == 0) *reinterpret_cast<u32x4*>(mb + WS_MBS) = u32x4{has_next ? 1u : 0u, lw16_, (unsigned)Tn.b, 0u};
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (written before this period's barrier; read two or more barriers later)
        }
        [[maybe_unused]] const unsigned long long tw1 = WS_STAMP();
        // vector-memory operations retire in order: everything but this period's input request (4) and stores (4) has to be done
        if (dr && req) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (dr || req) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        [[maybe_unused]] const unsigned long long tw2 = WS_STAMP();
        __builtin_amdgcn_s_barrier();
        t_work += tw1 - tw0;
        t_wait += WS_STAMP() - tw1;
#ifdef YOGO_DIAG
        if (c < 4) { t_wp[c] += tw1 - tw0; t_wv[c] += tw2 - tw1; }
#endif
      }
      // the tile that was computed becomes the one whose output is handed over
#pragma unroll
      for (int n = 0; n < 2; ++n) vo_prev[n] = WS_DBG(1) ? (int)OOB : vo[n];
      rs_out_prev = rs_out;
      if (has_next) {
        vo[0] = vo_n[0]; vo[1] = vo_n[1];
        rs_out = rs_out_n;
        tpar ^= 1;
      }
    }
    // the last tile's output: the staged half (written before the last period's barrier), barrier, (compute: the parked half)
    // barrier, second half
    fifo_fill(0, vo_prev, rs_out_prev);
    fifo_store4(WsIC<0>{});
    fifo_store4(WsIC<1>{});
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    fifo_fill(8, vo_prev, rs_out_prev);
    fifo_store4(WsIC<0>{});
    fifo_store4(WsIC<1>{});
#ifdef YOGO_DIAG
    if (p.stamps && ttid == 0) {
      unsigned long long* d = p.stamps + (size_t)blockIdx.x * 16;
      d[8] = t_work; d[9] = t_wait;
      unsigned long long* e = p.stamps + (size_t)(gridDim.x + blockIdx.x) * 16;   // second table: the loaders' work / memory waits in periods 0..3
      e[0] = t_wp[0]; e[1] = t_wp[1]; e[2] = t_wp[2]; e[3] = t_wp[3];
      e[4] = t_wv[0]; e[5] = t_wv[1]; e[6] = t_wv[2]; e[7] = t_wv[3];
    }
#endif
    return;
  }

  // =======================================================================================================================
  // COMPUTE
  // =======================================================================================================================
  if (WS_PRIO_COMPUTE) __builtin_amdgcn_s_setprio(WS_PRIO_COMPUTE);
  const int lane = ws_lane(), l31 = lane & 31, half = lane >> 5;
  const unsigned a_b0 = (unsigned)(half * 128 + l31) * 16u;
  // staging address of this lane's 8 bytes of a unit's lower channel block (the upper block's are 512 B above)
  const unsigned stg_wr = (unsigned)(WS_STG + tw * 4096 + l31 * 16 + half * 8);
  unsigned pbr[2], lw16;
  int vo[2];
  decode_pix(T, pbr, vo, lw16);
  int tpar = 0;                       // channel-scale buffer of the tile being computed
  int epar = 1;                       // ... of the tile whose epilogue slices are running
  __builtin_amdgcn_s_barrier();   // (#1)
  u32x4 A0[4], B0[2], A1[4], B1[2];   // the two operand sets
  ws_prefetch4(A1[0], A1[1], B1[0], B1[1], a_b0, pbr[0] + WS_I0, pbr[1] + WS_I0);
  // parked output units 8..15 of the PREVIOUS tile (unit = 2 (2 mb + gp) + n: rows mb = 2, 3), un-swapped: hA = this lane's
  // 4 channels of the lower channel block, hB = of the upper one
  u32x2 hA[8], hB[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { hA[i] = u32x2{0u, 0u}; hB[i] = u32x2{0u, 0u}; }
  unsigned ibo = WS_I0;               // input buffer of the chunk being computed (ring of three)
  constexpr bool leaky = (MODE & 1) != 0, write_signs = (MODE & 2) != 0, SCALED = (MODE & 4) != 0;
  [[maybe_unused]] unsigned long long t_chunks = 0, t_seam = 0, t_x8 = 0, t_gmf = 0, t_gml = 0, t_c1 = 0, t_c2 = 0;

  // ---- the epilogue, cut into slices that run in the MFMA gaps of a group-major chunk.  Group GRP (0: accumulator rows 0, 1 =
  //      units 0..7, straight to the staging area; 1: rows 2, 3 = units 8..15, parked) has 8 units of 8 values per lane; unit UL
  //      takes slices 4 UL .. 4 UL + 3: [bias / scale of the channel group +] 4 accumulator reads | 4 reads | the arithmetic
  //      (fma(acc, scale, bias * scale) or acc + bias -- the same bits --, LeakyReLU, sign byte) | bf16 + hand-over.
  float e_r[8], e_v[8], e_ba[8], e_sa[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, e_bs[8];
  unsigned sg[2][2] = {{0u, 0u}, {0u, 0u}};   // sign bytes of the tile in the epilogue: [pixel group][dword]
  int vo_e[2] = {(int)OOB, (int)OOB};          // ... its output offsets / image (the sign map goes out when group 1 is through)
  int b_e = 0;
  // stage 0: the channel group's bias / scale from LDS (needed by a unit with N == 0 only) | 1, 2: four accumulator reads each | 3: the
  // arithmetic | 4: bf16 + hand-over
  auto epi = [&](auto grp_tag, auto ul_tag, auto stg_tag) {
    constexpr int GRP = decltype(grp_tag)::value, UL = decltype(ul_tag)::value, STG = decltype(stg_tag)::value;
    if constexpr (UL >= 0 && UL < 8) {
      constexpr int U = GRP * 8 + UL, Q = U >> 1, N = U & 1, MB = Q >> 1, GP = Q & 1, R = (MB * 2 + N) * 16 + 8 * GP;
      if constexpr (STG == 0) {
        if constexpr (N == 0) {
          const float* eb = ldsf + WS_EB / 4;
          const int cl = MB * 32 + 16 * GP + 4 * half;   // local channel of group A; group B = cl + 8
          const float4 bA = *reinterpret_cast<const float4*>(eb + cl), bB = *reinterpret_cast<const float4*>(eb + cl + 8);
          e_ba[0] = bA.x; e_ba[1] = bA.y; e_ba[2] = bA.z; e_ba[3] = bA.w; e_ba[4] = bB.x; e_ba[5] = bB.y; e_ba[6] = bB.z; e_ba[7] = bB.w;
          if constexpr (SCALED) {
            const float* es = ldsf + WS_ES / 4 + epar * 128;
            const float4 sA = *reinterpret_cast<const float4*>(es + cl), sB = *reinterpret_cast<const float4*>(es + cl + 8);
            e_sa[0] = sA.x; e_sa[1] = sA.y; e_sa[2] = sA.z; e_sa[3] = sA.w; e_sa[4] = sB.x; e_sa[5] = sB.y; e_sa[6] = sB.z; e_sa[7] = sB.w;
          }
        }
      } else if constexpr (STG == 1) {
        ws_acc_read4<R>(e_r[0], e_r[1], e_r[2], e_r[3]);
      } else if constexpr (STG == 2) {
        ws_acc_read4<R + 4>(e_r[4], e_r[5], e_r[6], e_r[7]);
      } else if constexpr (STG == 3) {
        if constexpr (N == 0) {   // (the first use of the LDS loads of stage 0)
#pragma unroll
          for (int i = 0; i < 8; ++i) e_bs[i] = SCALED ? e_ba[i] * e_sa[i] : e_ba[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) e_v[i] = SCALED ? fmaf(e_r[i], e_sa[i], e_bs[i]) : e_r[i] + e_bs[i];
        if constexpr (leaky) {   // max(v, 0.01 v) as bare v_max_f32 (the arithmetic of conv_bf16_epi_groups.inc's lean order), ONE statement
          // (single-issue multiplies, not v_pk_mul_f32: beside MFMAs a packed fp32 instruction costs ~11 cycles more than the two
          //  it replaces -- MI355X_MICROARCH.md "price of one filler beside MFMAs"; the file is built with -fno-slp-vectorize)
          float sv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) sv[i] = e_v[i] * LEAKY_SLOPE;
          asm("v_max_f32 %0, %0, %8\n\tv_max_f32 %1, %1, %9\n\tv_max_f32 %2, %2, %10\n\tv_max_f32 %3, %3, %11\n\t"
              "v_max_f32 %4, %4, %12\n\tv_max_f32 %5, %5, %13\n\tv_max_f32 %6, %6, %14\n\tv_max_f32 %7, %7, %15"
              : "+v"(e_v[0]), "+v"(e_v[1]), "+v"(e_v[2]), "+v"(e_v[3]), "+v"(e_v[4]), "+v"(e_v[5]), "+v"(e_v[6]), "+v"(e_v[7])
              : "v"(sv[0]), "v"(sv[1]), "v"(sv[2]), "v"(sv[3]), "v"(sv[4]), "v"(sv[5]), "v"(sv[6]), "v"(sv[7]));
        }
        if constexpr (write_signs) {   // byte = sum of (v[i] > 0) << i: compare into vcc, add-with-carry shifts it in (values 7 down to 0)
          unsigned mA = 0;
#define WS_SGN(I) "v_cmp_lt_f32_e32 vcc, 0, %" #I "\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
          asm(WS_SGN(8) WS_SGN(7) WS_SGN(6) WS_SGN(5) WS_SGN(4) WS_SGN(3) WS_SGN(2) "v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
              : "+v"(mA)
              : "v"(e_v[0]), "v"(e_v[1]), "v"(e_v[2]), "v"(e_v[3]), "v"(e_v[4]), "v"(e_v[5]), "v"(e_v[6]), "v"(e_v[7])
              : "vcc");
#undef WS_SGN
          sg[N][Q >> 2] |= mA << (8 * (Q & 3));
        }
      } else {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (__bf16)e_v[i];
        const u32x4 w = __builtin_bit_cast(u32x4, o);   // (x, y) = this lane's 4 channels of block cb, (z, w) = of block cb + 1
        const u32x2 wa = {w.x, w.y}, wb = {w.z, w.w};
        if constexpr (U < 8) {   // straight to the staging area: a unit is [2 channel blocks][32 pixels][16 B], 8 B per lane and block
          unsigned char* dst = reinterpret_cast<unsigned char*>(smem4) + stg_wr + (U >> 2) * 16384 + (U & 3) * 1024;
          *reinterpret_cast<u32x2*>(dst) = wa;
          *reinterpret_cast<u32x2*>(dst + 512) = wb;
        } else {
          hA[U - 8] = wa;
          hB[U - 8] = wb;
        }
      }
    }
  };
  auto store_signs = [&]() {   // the 8 sign bytes of a pixel (this lane's half of the 128 channels) go out together
    if constexpr (write_signs) {
      const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)(p.signs + (size_t)b_e * plane * 16), (short)0, plane * 16, 0x00020000);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int vs = vo_e[n] < 0 ? (int)OOB : (vo_e[n] >> 4) * 8;
        const u32x2 tsg = {sg[n][0], sg[n][1]};
        __builtin_amdgcn_raw_buffer_store_b64(tsg, rs_s, vs, 0, 0);
        sg[n][0] = 0u;
        sg[n][1] = 0u;
      }
    }
  };

  // one tap-major 16-channel chunk (every chunk of a tile but the first and the last): 9 K steps of 8 MFMAs.  P = parity of the chunk
  // (weight buffer AND operand set of step 0); STAGE: steps 0..7 hand the parked units over (the tile's second chunk)
  auto chunk = [&](auto stage_tag, auto p_tag) {
    constexpr bool STAGE = decltype(stage_tag)::value;
    constexpr int P = decltype(p_tag)::value;
    const unsigned pa = a_b0 + P * WS_WB;
    const unsigned ibn = ibo == WS_I0 + 2 * WS_IB ? (unsigned)WS_I0 : ibo + WS_IB;   // the next chunk's input buffer
    auto one = [&](auto s_tag) {
      constexpr int S = decltype(s_tag)::value;
      constexpr bool EVEN = ((P + S) & 1) == 0;   // operand set of this step: 0 when even
      u32x4(&Ac)[4] = EVEN ? A0 : A1;
      u32x4(&Bc)[2] = EVEN ? B0 : B1;
      u32x4(&An)[4] = EVEN ? A1 : A0;
      u32x4(&Bn)[2] = EVEN ? B1 : B0;
      if constexpr (S < 8) {
        constexpr int T1 = S + 1, KY1 = T1 / 3, KX1 = T1 % 3;
        const unsigned rowo = (unsigned)KY1 * lw16 + ibo;
        ws_sa<false, T1, KX1>(Ac[0], Ac[1], Bc[0], Bc[1], An[0], An[1], An[2], An[3], Bn[0], Bn[1], pa, pbr[0] + rowo, pbr[1] + rowo);
        constexpr int HI = STAGE ? S : 0;
        ws_sb<false, STAGE, (S >> 2) * 16384 + (S & 3) * 1024>(Ac[2], Ac[3], Bc[0], Bc[1], An[0], An[1], An[2], An[3], Bn[0], Bn[1], stg_wr, hA[HI], hB[HI]);
      } else {
        [[maybe_unused]] const unsigned long long tx0 = WS_STAMP();
        ws_x8(Ac[0], Bc[0], Bc[1]);
        t_x8 += WS_STAMP() - tx0;
        ws_y8(Ac[1], Ac[2], Ac[3], Bc[0], Bc[1], An[0], An[1], An[2], An[3], Bn[0], Bn[1], a_b0 + (1 - P) * WS_WB, pbr[0] + ibn, pbr[1] + ibn);
      }
    };
    ws_static_for(one, std::make_integer_sequence<int, 9>{});
    ibo = ibn;
  };
  // a GROUP-MAJOR chunk: the 9 taps of accumulator rows 0, 1 (group 0), then those of rows 2, 3 (group 1), one MFMA per statement
  // with a slice of the epilogue behind each.  FIRST = true: the tile's first chunk (weight buffer 0; both groups start from
  // zero; group 0's MFMAs carry the epilogue of the PREVIOUS tile's group 1, whose registers group 1's first MFMAs then
  // overwrite).  FIRST = false: the tile's last chunk (weight buffer 1; group 1's MFMAs carry the epilogue of THIS tile's group 0).
  // Operand set of step s (0..17): set 1 when s is even.
  auto gm_chunk = [&](auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int P = FIRST ? 0 : 1;
    const unsigned pa = a_b0 + P * WS_WB;
    const unsigned ibn = ibo == WS_I0 + 2 * WS_IB ? (unsigned)WS_I0 : ibo + WS_IB;
    auto one = [&](auto s_tag) {
      constexpr int S = decltype(s_tag)::value, G = S / 9, TAP = S % 9;
      constexpr bool ODD = (S & 1) != 0;   // operand set of this step: 1 when even
      u32x4(&Ac)[4] = ODD ? A0 : A1;
      u32x4(&Bc)[2] = ODD ? B0 : B1;
      u32x4(&An)[4] = ODD ? A1 : A0;
      u32x4(&Bn)[2] = ODD ? B1 : B0;
      constexpr bool ZERO = FIRST && TAP == 0;
      constexpr bool EPI = FIRST ? G == 0 : G == 1;   // this group's MFMAs carry the other group's epilogue
      constexpr int EG = FIRST ? 1 : 0;
      // The epilogue of unit TAP (0..7) of the other group, one stage per gap.  hipcc's own s_waitcnt for an LDS load of the
      // epilogue counts only ITS operations: behind this step's operand reads it would also wait for THEM (an LDS round trip
      // with the matrix pipe running dry).  So everything that touches LDS sits behind the step's LAST statement, whose
      // lgkmcnt(0) has just drained the queue: the first use of the bias / scale (stage 3), their loads for the next channel
      // group, and -- one step later, behind the first statement -- the hand-over write.
      auto slice = [&](auto m_tag) {
        constexpr int M = decltype(m_tag)::value;
        if constexpr (WS_ABLATE(16)) return;
        if constexpr (EPI) {
          if constexpr (M == 0) {
            if constexpr (TAP == 0) epi(WsIC<EG>{}, WsIC<0>{}, WsIC<0>{});
            else epi(WsIC<EG>{}, WsIC<TAP - 1>{}, WsIC<4>{});
          } else if constexpr (M == 1) {
            epi(WsIC<EG>{}, WsIC<TAP>{}, WsIC<1>{});
          } else if constexpr (M == 2) {
            epi(WsIC<EG>{}, WsIC<TAP>{}, WsIC<2>{});
          } else {
            epi(WsIC<EG>{}, WsIC<TAP>{}, WsIC<3>{});
            epi(WsIC<EG>{}, WsIC<TAP + 1>{}, WsIC<0>{});
          }
        }
      };
      if constexpr (S < 17) {
        constexpr int GN = (S + 1) / 9, T1 = (S + 1) % 9, KY1 = T1 / 3, KX1 = T1 % 3;   // the next step: group, tap
        const unsigned rowo = (unsigned)KY1 * lw16 + ibo;
        ws_g1<4 * G + 0, ZERO, 2 * GN, T1, KX1>(Ac[0], Bc[0], An[0], Bn[0], pa, pbr[0] + rowo);
        slice(WsIC<0>{});
        ws_g1<4 * G + 1, ZERO, 2 * GN + 1, T1, KX1>(Ac[0], Bc[1], An[1], Bn[1], pa, pbr[1] + rowo);
        slice(WsIC<1>{});
        ws_g0<4 * G + 2, ZERO>(Ac[1], Bc[0]);
        slice(WsIC<2>{});
        ws_gw<4 * G + 3, ZERO>(Ac[1], Bc[1], An[0], An[1], Bn[0], Bn[1]);
        slice(WsIC<3>{});
      } else {   // the chunk's last step: the last slice (31, in gap 32) behind its first MFMA, the barrier behind the second
        ws_g0<4, false>(Ac[0], Bc[0]);
        slice(WsIC<0>{});
        [[maybe_unused]] const unsigned long long tx0 = WS_STAMP();
        ws_g0<5, false, true>(Ac[0], Bc[1]);
        t_x8 += WS_STAMP() - tx0;
        if constexpr (FIRST) {   // ... then the six operand quads of the next (tap-major, weight buffer 1) chunk's step 0
          const unsigned pan = a_b0 + WS_WB;
          ws_g3<6, 0, false>(Ac[1], Bc[0], An[0], An[1], Bn[0], pan, pbr[0] + ibn, An[0], An[1], Bn[0]);
          ws_g3<7, 2, true>(Ac[1], Bc[1], An[2], An[3], Bn[1], pan, pbr[1] + ibn, An[0], An[1], Bn[0]);
        } else {
          ws_g0<6, false>(Ac[1], Bc[0]);
          ws_g0<7, false>(Ac[1], Bc[1]);
        }
      }
    };
    ws_static_for(one, std::make_integer_sequence<int, 18>{});
    ibo = ibn;
  };
  using TT = std::true_type;
  using FT = std::false_type;

  for (;;) {
    [[maybe_unused]] const unsigned long long tc0 = WS_STAMP();
    epar = tpar ^ 1;
    gm_chunk(TT{});                 // chunk 0; group 1 of the previous tile leaves its accumulators -> hA / hB
    store_signs();                  // (the previous tile's sign map is complete)
    t_gmf += WS_STAMP() - tc0;
    epar = tpar;                    // from here on the epilogue in progress is this tile's
    vo_e[0] = vo[0]; vo_e[1] = vo[1]; b_e = T.b;
    [[maybe_unused]] const unsigned long long tq0 = WS_STAMP();
    chunk(std::integral_constant<bool, !WS_ABLATE(128)>{}, WsIC<1>{});         // chunk 1: the parked units go to the staging area (ablation 128: they do not)
    [[maybe_unused]] const unsigned long long tq1 = WS_STAMP();
    chunk(FT{}, WsIC<0>{});
    t_c1 += tq1 - tq0;
    t_c2 += WS_STAMP() - tq1;
    for (int c = 3; c + 1 < nck; c += 2) {
      chunk(FT{}, WsIC<1>{});
      chunk(FT{}, WsIC<0>{});
    }
    [[maybe_unused]] const unsigned long long tl0 = WS_STAMP();
    gm_chunk(FT{});                 // chunk nck - 1; group 0 of this tile -> staging area
    [[maybe_unused]] const unsigned long long ts0 = WS_STAMP();
    t_gml += ts0 - tl0;
    t_chunks += ts0 - tc0;
    // the next tile: looked up and decoded by the partner loader wavefront (mailbox, written in period 1 of this tile)
    const u32x4 mbs = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(smem4) + WS_MBS);
    const u32x4 mbl = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(smem4) + WS_MB + (tw * 64 + lane) * 16);
    const bool has_next = __builtin_amdgcn_readfirstlane((int)mbs.x) != 0;
    ++k_ord;
    if (!has_next) break;
    lw16 = (unsigned)__builtin_amdgcn_readfirstlane((int)mbs.y);
    T.b = __builtin_amdgcn_readfirstlane((int)mbs.z);
    pbr[0] = mbl.x; pbr[1] = mbl.y; vo[0] = (int)mbl.z; vo[1] = (int)mbl.w;
    tpar ^= 1;
    ws_prefetch4(A1[0], A1[1], B1[0], B1[1], a_b0, pbr[0] + ibo, pbr[1] + ibo);   // (the last period's barrier: chunk 0 has landed)
    t_seam += WS_STAMP() - ts0;
  }
  // ---- the last tile: group 1's epilogue on its own, then the parked half's hand-over
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (the last MFMAs' results are in the accumulator file)
  ws_static_for([&](auto u_tag) {
    epi(WsIC<1>{}, u_tag, WsIC<0>{}); epi(WsIC<1>{}, u_tag, WsIC<1>{}); epi(WsIC<1>{}, u_tag, WsIC<2>{}); epi(WsIC<1>{}, u_tag, WsIC<3>{});
    epi(WsIC<1>{}, u_tag, WsIC<4>{});
  }, std::make_integer_sequence<int, 8>{});
  store_signs();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // (the loaders have read the staged half of the last tile)
  {
    // (the staging address is re-derived from v_mbcnt: live across the tile loop it costs a register the 256-register
    //  instantiations do not have)
    const int ln_ = ws_lane();
    const unsigned stg2 = (unsigned)(WS_STG + tw * 4096 + (ln_ & 31) * 16 + (ln_ >> 5) * 8);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      unsigned char* dst = reinterpret_cast<unsigned char*>(smem4) + stg2 + (u >> 2) * 16384 + (u & 3) * 1024;
      *reinterpret_cast<u32x2*>(dst) = hA[u];
      *reinterpret_cast<u32x2*>(dst + 512) = hB[u];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef YOGO_DIAG
  if (p.stamps && tw == 0 && lane == 0) {
    unsigned long long* d = p.stamps + (size_t)blockIdx.x * 16;
    d[0] = t_start; d[1] = __builtin_amdgcn_s_memtime(); d[2] = t_chunks; d[3] = t_seam; d[5] = k_ord; d[6] = t_x8; d[10] = t_gmf; d[11] = t_gml; d[12] = t_c1; d[13] = t_c2;
  }
#endif
}

// =========================================================================================================
// host side: eligibility, tiling, launch
// =========================================================================================================
bool conv_bf16_ws_eligible(int K, int M, int IH, int IW, int B) {
  const int Kb = round_up(K, 16) / 8;
  if (M != 128 || Kb < 8 || (Kb % 4) != 0) return false;   // nchunk = Kb / 2 even and >= 4
  if (IH < 3 || IW < 3 || B <= 0) return false;
  if ((long long)Kb * IH * IW * 16 >= (1ll << 31) || (long long)16 * IH * IW * 16 >= (1ll << 31)) return false;   // per-image descriptors, bit 31 = "out of range"
  return true;
}

// column bands of TW output columns, tiles of 256 consecutive pixels of a band (row-major inside the band): the staged input
// tile of a chunk ([2 channel blocks][rows + 2][TW + 2] units) has to fit the 4 x 256 input slots; among the fitting band
// counts take the one with the fewest tiles per image (MFMA work), then the fewest staged units
bool conv_bf16_ws_plan(ConvWsParams* p, int slots) {
  const int OH = p->IH, OW = p->IW;
  long long best = -1;
  int best_ncb = 0;
  for (int ncb = 1; ncb <= 48 && ncb <= OW; ++ncb) {
#ifdef WS_FORCE_NCB   // (A/B variant builds: the band count of the plan, whatever the score)
    if (ncb != WS_FORCE_NCB) continue;
#endif
    const int TW = cdiv(OW, ncb);
    const int bw_min = OW - (cdiv(OW, TW) - 1) * TW;
    if (cdiv(OW, TW) != ncb || bw_min < 2) continue;
    // rows a 256-pixel tile can touch in a band of width bw: a tile starts anywhere in a row
    auto rows_of = [&](int bw) { return min(OH, 1 + cdiv(WS_PT - 1, bw)) + 2; };
    const int need = 2 * max(rows_of(TW) * (TW + 2), rows_of(bw_min) * (bw_min + 2));
    if (need > slots * WS_NT) continue;
    const long long tiles = (long long)(ncb - 1) * cdiv(OH * TW, WS_PT) + cdiv(OH * bw_min, WS_PT);
    const long long staged = (long long)(ncb - 1) * cdiv(OH * TW, WS_PT) * rows_of(TW) * (TW + 2) + (long long)cdiv(OH * bw_min, WS_PT) * rows_of(bw_min) * (bw_min + 2);
    const long long score = tiles * 100000000ll + staged;
    if (best < 0 || score < best) { best = score; best_ncb = ncb; }
  }
  if (best < 0) return false;
  p->ncb = best_ncb;
  p->TW = cdiv(OW, best_ncb);
  p->tiles_per_band = cdiv(OH * p->TW, WS_PT);
  p->gx = p->ncb * p->tiles_per_band;
  p->ntiles = p->B * p->gx;
  auto magic = [](int d) -> unsigned { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(((1ull << 32) + (unsigned)d - 1ull) / (unsigned)d); };
  const int bw_last = OW - (p->ncb - 1) * p->TW;
  p->m_gx = magic(p->gx); p->m_tpb = magic(p->tiles_per_band);
  p->m_bw = magic(p->TW); p->m_bwl = magic(bw_last);
  p->m_lw = magic(p->TW + 2); p->m_lwl = magic(bw_last + 2);
  p->nchunk = p->Kb / 2;
  // the kernels' divisions by multiplication: tile -> image / band / tile of the band, pixel -> row of its band, staged element -> row
  const int bw_l = OW - (p->ncb - 1) * p->TW;
  if (!magic_div_exact((long long)p->ntiles - 1, p->gx) || !magic_div_exact(p->gx, p->tiles_per_band) || !magic_div_exact((long long)OH * p->TW, p->TW) ||
      !magic_div_exact((long long)OH * bw_l, bw_l) || !magic_div_exact(2 * slots * WS_NT, p->TW + 2) || !magic_div_exact(2 * slots * WS_NT, bw_l + 2))
    return false;
  return true;
}

int launch_conv_bf16_ws(const ConvWsParams& p, hipStream_t stream) {
  // per device, once: the dynamic-LDS attribute of every instantiation and the CU count (a second device in the process, or a
  // first call from two host threads, must not see another device's state)
  static std::mutex mu;
  static int n_cu_of[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    yogo_set_error("conv_bf16_ws: hipGetDevice failed");
    return YOGO_ERR_HIP;
  }
  int n_cu;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (n_cu_of[dev] == 0) {
      hipError_t e = hipSuccess;
#define WS_ATTR(M) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_ws_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES);
      WS_ATTR(0) WS_ATTR(1) WS_ATTR(3) WS_ATTR(4) WS_ATTR(5) WS_ATTR(7)
#undef WS_ATTR
      if (e != hipSuccess) {
        yogo_set_error("conv_bf16_ws: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", WS_LDS_BYTES, hipGetErrorString(e));
        return YOGO_ERR_HIP;
      }
      hipDeviceProp_t prop;
      n_cu_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    n_cu = n_cu_of[dev];
  }
  if (p.ntiles <= 0) return YOGO_OK;
  // one persistent workgroup per CU; a multiple of 8 so that a workgroup's tiles stay inside one XCD's run
  int grid = min(p.ntiles, n_cu);
  if (grid >= 8) grid &= ~7;
  const bool leaky = p.act == ACT_LEAKY, sc = p.chan_scale != nullptr, sg = p.signs != nullptr;
  if (sg && !leaky) {
    yogo_set_error("conv_bf16_ws: a sign map goes with LeakyReLU");
    return YOGO_ERR_ARG;
  }
  const int mode = (leaky ? 1 : 0) | (sg ? 2 : 0) | (sc ? 4 : 0);
#define WS_LAUNCH(M) case M: hipLaunchKernelGGL(conv_bf16_ws_kernel<M>, dim3(grid), dim3(512), WS_LDS_BYTES, stream, p); break;
  switch (mode) { WS_LAUNCH(0) WS_LAUNCH(1) WS_LAUNCH(3) WS_LAUNCH(4) WS_LAUNCH(5) WS_LAUNCH(7) }
#undef WS_LAUNCH
  if (yogo_launch_log_enabled())
    yogo_launch_log("conv_bf16_ws_kernel<%d> | Kb=%d in=%dx%d ncb=%d TW=%d tiles_per_band=%d nchunk=%d ntiles=%d grid=%d lds=%d act=%d signs=%d scale=%d", mode, p.Kb, p.IH,
                    p.IW, p.ncb, p.TW, p.tiles_per_band, p.nchunk, p.ntiles, grid, WS_LDS_BYTES, p.act, p.signs != nullptr, p.chan_scale != nullptr);
  YOGO_CHECK_LAUNCH("conv_bf16_ws");
  return YOGO_OK;
}
