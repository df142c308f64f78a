// Persistent, wavefront-specialised DATA GRADIENT of a stride-2 3x3 convolution into 128 channels (layer 4 of base_model:
// autograd of yogo/model_defns.py:54-56, `loss.backward()` at yogo/train.py:322).  Same arithmetic as conv_bf16_kernel<4,1,8,S2D>
// (conv_bf16.hip): the gradient is decomposed by output parity -- dx[2a+py][2b+px] only receives the taps with ky = py + 1,
// kx = px + 1 (mod 2): 1 + 2 + 2 + 4 tap-GEMMs per 2x2 output quad -- every accumulator sees its (16-channel chunk, tap)
// products in the same order and the epilogue applies the same formula, so outputs are bit-identical (tests/test_gpu_ws.py).
//
// What the tiled kernel loses (profiles/r04_*: 911 us, 0.21 of the MFMA peak, 0.29 of HBM, FETCH 2.08x): one workgroup per ROW
// parity stages the same dy tile a second time, its chunk loop waits a DMA round trip per chunk, and its epilogue -- 128 KB of
// stores per tile -- overlaps with nothing.  Here:
//   * one persistent workgroup per CU walks tiles of 128 output quads; the tile's dy (128 quads + halo, all K channels: <= 64 KB)
//     is staged ONCE into a ring of eight 16-channel slots and serves both row parities: pass A (py = 0: 3 taps) and pass B
//     (py = 1: 6 taps).  Slot c is refilled with the NEXT tile's chunk as soon as pass B is through with it;
//   * wavefronts 0-3 COMPUTE (one per SIMD): 64 channels x 64 quads x both column parities = 8 accumulator tiles; per dy shift
//     the two pixel operands are read once and serve every tap of that shift (0.83 LDS operand reads per MFMA);
//   * wavefronts 4-7 LOAD: the weight slices of the next period (24 KB, L2 hits) by LDS-DMA two periods ahead (three weight buffers), the next
//     tile's dy, the sign bytes / channel scale of the epilogues.  They issue NO stores: vector-memory operations retire in order and
//     a load behind a store's acknowledgement holds the period's barrier;
//   * the compute wavefronts STORE their own output straight from registers at the end of each pass (the half-wave exchange and
//     one 16-byte store per 8 values, as the tiled kernel): they issue no loads, so they never wait for a store, and four
//     wavefronts' store queues hold a tile's output.  (Measured on the way, tools/probes/store_rate.hip + gpurun_out/r5_ws2_*: output
//     staged through LDS and stored by the loaders -- loads stuck behind store acknowledgements, 16 k of 37 k cycles per tile in
//     vmcnt waits; by ONE dedicated store wavefront, also paced through a 48 KB register FIFO -- the eight extra barriers per tile
//     and the wavefront's own issue time cost what the overlap gained: all three forms ran at the tiled kernel's 0.87 ms.)
//   * one s_barrier per MFMA period orders everything (DMA landed: the loaders wait vmcnt first; buffers free).
// Periods of a tile: pass A in 32-channel periods, pass B in 16-channel periods -- every period is 24 MFMAs per compute wavefront
// and 24 KB of weights.
#include "conv_bf16_ws2.h"
#include <mutex>
#include <type_traits>
#include <utility>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ int w2_udivm(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }   // n / d, m = ceil(2^32 / d), d > 1
__device__ __forceinline__ int w2_udivm1(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }
__device__ __forceinline__ int w2_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ i32x4 w2_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
// (scalar operands travel through assignments hipcc cannot always prove uniform: v_readfirstlane, folded away where it can)
__device__ __forceinline__ unsigned w2_u(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ i32x4 w2_u4(i32x4 r) {
  return i32x4{__builtin_amdgcn_readfirstlane(r.x), __builtin_amdgcn_readfirstlane(r.y), __builtin_amdgcn_readfirstlane(r.z), __builtin_amdgcn_readfirstlane(r.w)};
}
// LDS-DMA pieces (64 lanes x 16 bytes -> LDS bytes [m0, m0 + 1024)); every helper is ONE asm statement: M0 is stepped with scalar
// adds, the pieces of a group share the per-lane offset
// two pieces: LDS + 4096, scalar offset + step (the two channel blocks of a dy chunk)
__device__ __forceinline__ void w2_dma2(i32x4 rs, unsigned lds, int voff, unsigned soff, unsigned step) {
  unsigned so;
  rs = w2_u4(rs); lds = w2_u(lds); soff = w2_u(soff); step = w2_u(step);
  asm volatile("s_mov_b32 m0, %3\n\ts_mov_b32 %0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\t"
               "s_add_u32 m0, m0, 4096\n\ts_add_u32 %0, %0, %5\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds"
               : "=&s"(so)
               : "v"(voff), "s"(rs), "s"(lds), "s"(soff), "s"(step)
               : "memory", "scc");
}
// N pieces: LDS + 4096 each, scalar offset + step each (the weight slices of consecutive taps)
#define W2_PN "s_add_u32 m0, m0, 4096\n\ts_add_u32 %0, %0, %5\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\t"
__device__ __forceinline__ void w2_dma3(i32x4 rs, unsigned lds, int voff, unsigned soff, unsigned step) {
  unsigned so;
  rs = w2_u4(rs); lds = w2_u(lds); soff = w2_u(soff); step = w2_u(step);
  asm volatile("s_mov_b32 m0, %3\n\ts_mov_b32 %0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\t" W2_PN W2_PN
               : "=&s"(so)
               : "v"(voff), "s"(rs), "s"(lds), "s"(soff), "s"(step)
               : "memory", "scc");
}
__device__ __forceinline__ void w2_dma6(i32x4 rs, unsigned lds, int voff, unsigned soff, unsigned step) {
  unsigned so;
  rs = w2_u4(rs); lds = w2_u(lds); soff = w2_u(soff); step = w2_u(step);
  asm volatile("s_mov_b32 m0, %3\n\ts_mov_b32 %0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\t" W2_PN W2_PN W2_PN W2_PN W2_PN
               : "=&s"(so)
               : "v"(voff), "s"(rs), "s"(lds), "s"(soff), "s"(step)
               : "memory", "scc");
}
#undef W2_PN
__device__ __forceinline__ void w2_dma1(i32x4 rs, unsigned lds, int voff, unsigned soff) {
  rs = w2_u4(rs); lds = w2_u(lds); soff = w2_u(soff);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" ::"v"(voff), "s"(lds), "s"(rs), "s"(soff) : "memory");
}
// four pieces, contiguous in memory and in LDS (a 4 KB weight group)
__device__ __forceinline__ void w2_dma4c(i32x4 rs, unsigned lds, int voff, unsigned soff) {
  unsigned so;
  rs = w2_u4(rs); lds = w2_u(lds); soff = w2_u(soff);
#define W2_PC "s_add_u32 m0, m0, 1024\n\ts_add_u32 %0, %0, 1024\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\t"
  asm volatile("s_mov_b32 m0, %3\n\ts_mov_b32 %0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\t" W2_PC W2_PC W2_PC
               : "=&s"(so)
               : "v"(voff), "s"(rs), "s"(lds), "s"(soff)
               : "memory", "scc");
#undef W2_PC
}
// 64 lanes x 4 bytes -> LDS bytes [lds, lds + 256)
__device__ __forceinline__ void w2_dma_dword(i32x4 rs, unsigned lds, int voff, unsigned soff) {
  rs = w2_u4(rs); lds = w2_u(lds); soff = w2_u(soff);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dword %0, %2, %3 offen lds" ::"v"(voff), "s"(lds), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void w2_store16(u32x4 data, int voff, i32x4 rs, unsigned soff) {
  rs = w2_u4(rs); soff = w2_u(soff);
  // (s_nop in front: the descriptor may come from v_readfirstlane; behind: a 16-byte store's data registers must not be overwritten by
  //  the next vector instruction -- hipcc does not look inside asm statements)
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(data), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void w2_vmwait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// the period barrier of a compute wavefront: its LDS reads and staging writes are done, then everybody meets
__device__ __forceinline__ void w2_barrier_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void w2_barrier() { asm volatile("s_barrier" ::: "memory"); }

template <class F, int... I>
__device__ __forceinline__ void w2_static_for(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N> using W2IC = std::integral_constant<int, N>;

// the K steps of a 16-channel chunk: weight group inside the period's buffer (slice order of pack mode 2: row parity 0 = slices
// 0 | 1 2, row parity 1 = slices 3 4 | 5 6 7 8), dy shift (0: (a, b), 1: (a, b + 1), 2: (a + 1, b), 3: (a + 1, b + 1)), column parity.
// Ordered by shift, so that a shift's two pixel operands are read once; each accumulator still sees its slices in ascending order
// (px 0: 3, 4; px 1: 5, 6, 7, 8), the order of conv_bf16_kernel's two tap runs.
struct W2Step { int g, sh, px; };
__device__ constexpr W2Step kW2StepsA[3] = {{0, 0, 0}, {1, 0, 1}, {2, 1, 1}};
__device__ constexpr W2Step kW2StepsB[6] = {{0, 0, 0}, {2, 0, 1}, {3, 1, 1}, {1, 2, 0}, {4, 2, 1}, {5, 3, 1}};
__device__ constexpr W2Step w2_step(bool pass_b, int s) { return pass_b ? kW2StepsB[s < 6 ? s : 5] : kW2StepsA[s < 3 ? s : 2]; }


// ---- the asm statements of a compute wavefront's K steps --------------------------------------------------------------------
// The eight 32x32 accumulator tiles are a[0:127], OWNED BY THE ASM STATEMENTS (named literally, listed as clobbers: see
// conv_bf16_ws.hip -- as "+v" operands hipcc copies the 16-register tuples around every statement and spills 400-600 registers;
// build.sh audits that no compiler-generated instruction touches an AGPR).  Tile of (px, mb, n) = 4 px + 2 mb + n.
#define W2_ACC_CLOBBER "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127"
// A K step = one tap x 16 channels for one column parity PX: 4 MFMAs (row blocks 0, 1 x pixel groups 0, 1) on tiles 4 PX .. 4 PX + 3.
// Operands are requested TWO steps ahead: under the load of this kernel (LDS-DMA of 32 KB per period beside 80 KB of operand reads)
// an LDS read takes 200+ cycles, a step's MFMAs 128 -- with the reads of step s + 1 behind the MFMAs of step s every step stalled
// ~100 cycles (1 330 instead of 768 cycles per period: gpurun_out/r5_ws2_st5.log; hipcc's own schedule of the loop ran the same).
// Per period of six steps:   step 0: reads of step 2 | 1: of step 3 | 2: of steps 4, 5 | 3: none, then lgkmcnt(0) + the period's
// BARRIER (every read of the period's buffers is in registers) | 4: the next period's steps 0, 1 | 5: none, then lgkmcnt(0).
// (compile-time ablations for timing runs, results wrong: W2_ABL bit 0 = no MFMAs, 1 = no operand reads, 2 = no weight DMA, 3 = no epilogue arithmetic / staging)
#ifndef W2_ABL
#define W2_ABL 0
#endif
#if W2_ABL & 1
#define W2_MF(K, A, B) "s_nop 0\n\t"
#define W2_MF0(K, A, B) "s_nop 0\n\t"
#else
#define W2_MF(K, A, B) "v_mfma_f32_32x32x16_bf16 a[16*(%[tb]+" #K "):16*(%[tb]+" #K ")+15], %[" #A "], %[" #B "], a[16*(%[tb]+" #K "):16*(%[tb]+" #K ")+15]\n\t"
#define W2_MF0(K, A, B) "v_mfma_f32_32x32x16_bf16 a[16*(%[tb]+" #K "):16*(%[tb]+" #K ")+15], %[" #A "], %[" #B "], 0\n\t"
#endif
#if W2_ABL & 2
#define W2_RDA(D0, D1, P, O) "s_nop 0\n\t"
#define W2_RDB "s_nop 0\n\t"
#else
#define W2_RDA(D0, D1, P, O) "ds_read_b128 %[" #D0 "], %[" #P "] offset:%[" #O "]\n\tds_read_b128 %[" #D1 "], %[" #P "] offset:%[" #O "]+512\n\t"
#define W2_RDB "ds_read_b128 %[bn0], %[pb0] offset:%[bo]\n\tds_read_b128 %[bn1], %[pb1] offset:%[bo]\n\t"
#endif
#define W2_OPS_IN [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [tb] "n"(4 * PX)
// MFMAs + one weight group (AOFF) and one pair of pixel quads (pb0 / pb1 + BIMM); ends when all but these four reads are done
template <int PX, bool ZERO, int AOFF, int BIMM>
__device__ __forceinline__ void w2_k_ab(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, u32x4& an0, u32x4& an1, u32x4& bn0, u32x4& bn1,
                                        unsigned pa, unsigned pb0, unsigned pb1) {
  if constexpr (ZERO)
    asm volatile(W2_MF0(0, a0, b0) W2_RDA(an0, an1, pa, ao) W2_MF0(1, a0, b1) W2_RDB W2_MF0(2, a1, b0) W2_MF0(3, a1, b1) "s_waitcnt lgkmcnt(4)"
                 : [an0] "=&v"(an0), [an1] "=&v"(an1), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1)
                 : W2_OPS_IN, [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1), [ao] "n"(AOFF), [bo] "n"(BIMM)
                 : "memory", W2_ACC_CLOBBER);
  else
    asm volatile(W2_MF(0, a0, b0) W2_RDA(an0, an1, pa, ao) W2_MF(1, a0, b1) W2_RDB W2_MF(2, a1, b0) W2_MF(3, a1, b1) "s_waitcnt lgkmcnt(4)"
                 : [an0] "=&v"(an0), [an1] "=&v"(an1), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1)
                 : W2_OPS_IN, [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1), [ao] "n"(AOFF), [bo] "n"(BIMM)
                 : "memory", W2_ACC_CLOBBER);
}
// MFMAs + two weight groups (AOFF, AOFF2) and one pair of pixel quads; WAIT6: ends when all but these six reads are done (step 2),
// else without a wait (step 4: the operands of step 5 landed in front of the barrier)
template <int PX, bool WAIT6, int AOFF, int AOFF2, int BIMM>
__device__ __forceinline__ void w2_k_aab(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, u32x4& an0, u32x4& an1, u32x4& am0, u32x4& am1,
                                         u32x4& bn0, u32x4& bn1, unsigned pa, unsigned pb0, unsigned pb1) {
  if constexpr (WAIT6)
    asm volatile(W2_MF(0, a0, b0) W2_RDA(an0, an1, pa, ao) W2_MF(1, a0, b1) W2_RDA(am0, am1, pa, ao2) W2_MF(2, a1, b0) W2_RDB W2_MF(3, a1, b1) "s_waitcnt lgkmcnt(6)"
                 : [an0] "=&v"(an0), [an1] "=&v"(an1), [am0] "=&v"(am0), [am1] "=&v"(am1), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1)
                 : W2_OPS_IN, [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1), [ao] "n"(AOFF), [ao2] "n"(AOFF2), [bo] "n"(BIMM)
                 : "memory", W2_ACC_CLOBBER);
  else
    asm volatile(W2_MF(0, a0, b0) W2_RDA(an0, an1, pa, ao) W2_MF(1, a0, b1) W2_RDA(am0, am1, pa, ao2) W2_MF(2, a1, b0) W2_RDB W2_MF(3, a1, b1)
                 : [an0] "=&v"(an0), [an1] "=&v"(an1), [am0] "=&v"(am0), [am1] "=&v"(am1), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1)
                 : W2_OPS_IN, [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1), [ao] "n"(AOFF), [ao2] "n"(AOFF2), [bo] "n"(BIMM)
                 : "memory", W2_ACC_CLOBBER);
}
// MFMAs without reads.  TAIL 0: nothing (step 4 of a pass's last period); 1: lgkmcnt(0) + s_barrier (step 3); 2: lgkmcnt(0) (step 5); 3: the wait
// states between an MFMA and a read of its result (step 5 of a pass's last period; hipcc does not look inside asm statements)
template <int PX, int TAIL>
__device__ __forceinline__ void w2_k_0(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, u32x4& x0, u32x4& x1, u32x4& x2, u32x4& x3, u32x4& x4,
                                       u32x4& x5) {
  // (x0 .. x5: the registers the retired reads wrote -- named "+v" so that their uses stay behind this statement)
#define W2_K0_OUT [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5)
  if constexpr (TAIL == 1)
    asm volatile(W2_MF(0, a0, b0) W2_MF(1, a0, b1) W2_MF(2, a1, b0) W2_MF(3, a1, b1) "s_waitcnt lgkmcnt(0)\n\ts_barrier" : W2_K0_OUT : W2_OPS_IN : "memory", W2_ACC_CLOBBER);
  else if constexpr (TAIL == 2)
    asm volatile(W2_MF(0, a0, b0) W2_MF(1, a0, b1) W2_MF(2, a1, b0) W2_MF(3, a1, b1) "s_waitcnt lgkmcnt(0)" : W2_K0_OUT : W2_OPS_IN : "memory", W2_ACC_CLOBBER);
  else if constexpr (TAIL == 3)
    asm volatile(W2_MF(0, a0, b0) W2_MF(1, a0, b1) W2_MF(2, a1, b0) W2_MF(3, a1, b1) "s_nop 15\n\ts_nop 15" : W2_K0_OUT : W2_OPS_IN : "memory", W2_ACC_CLOBBER);
  else
    asm volatile(W2_MF(0, a0, b0) W2_MF(1, a0, b1) W2_MF(2, a1, b0) W2_MF(3, a1, b1) : W2_K0_OUT : W2_OPS_IN : "memory", W2_ACC_CLOBBER);
#undef W2_K0_OUT
}
// the operands of a pass's steps 0 and 1 (weight groups 0 and AOFF2, pixel quads of the first slot)
template <int AOFF2>
__device__ __forceinline__ void w2_kfirst(u32x4& an0, u32x4& an1, u32x4& am0, u32x4& am1, u32x4& bn0, u32x4& bn1, unsigned pa, unsigned pb0, unsigned pb1) {
  asm volatile("ds_read_b128 %[an0], %[pa]\n\tds_read_b128 %[an1], %[pa] offset:512\n\tds_read_b128 %[bn0], %[pb0]\n\tds_read_b128 %[bn1], %[pb1]\n\t"
               "ds_read_b128 %[am0], %[pa] offset:%[ao2]\n\tds_read_b128 %[am1], %[pa] offset:%[ao2]+512\n\ts_waitcnt lgkmcnt(0)"
               : [an0] "=&v"(an0), [an1] "=&v"(an1), [am0] "=&v"(am0), [am1] "=&v"(am1), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1)
               : [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1), [ao2] "n"(AOFF2)
               : "memory");
}
// eight consecutive accumulator registers -> VGPRs in ONE statement; it clobbers every accumulator register, so hipcc cannot keep a
// value of its own in an AGPR across any part of the epilogue (conv_bf16_ws.hip: ws_acc_read8)
template <int R>
__device__ __forceinline__ void w2_acc_read8(float (&r)[8]) {
  asm volatile(
      "v_accvgpr_read_b32 %0, a[%8]\n\tv_accvgpr_read_b32 %1, a[%8+1]\n\tv_accvgpr_read_b32 %2, a[%8+2]\n\tv_accvgpr_read_b32 %3, a[%8+3]\n\t"
      "v_accvgpr_read_b32 %4, a[%8+4]\n\tv_accvgpr_read_b32 %5, a[%8+5]\n\tv_accvgpr_read_b32 %6, a[%8+6]\n\tv_accvgpr_read_b32 %7, a[%8+7]"
      : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7])
      : "n"(R)
      : W2_ACC_CLOBBER);
}
// the six K steps of a period (pass A: two 16-channel chunks x slices 0 1 2; pass B: one chunk x slices 3 .. 8 ordered by dy shift):
// weight group, dy row (0: a, 1: a + 1) and byte offset of the pixel quads, column parity.  Steps 0, 2, 3, 5 need new pixel quads.
__device__ constexpr int w2_grp(bool pb, int s) { return pb ? (s == 0 ? 0 : s == 1 ? 2 : s == 2 ? 3 : s == 3 ? 1 : s) : s; }
__device__ constexpr int w2_brow(bool pb, int s) { return pb && s >= 3 ? 1 : 0; }
__device__ constexpr int w2_bimm(bool pb, int s) { return ((s == 2 || s == 5) ? 16 : 0) + ((!pb && s >= 3) ? W2_DYS : 0); }
__device__ constexpr bool w2_newb(int s) { return s == 0 || s == 2 || s == 3 || s == 5; }
__device__ constexpr int w2_bset(int s) { return s < 2 ? 0 : (s == 2 ? 1 : (s < 5 ? 2 : 3)); }   // pixel-quad sets: steps 0 1 | 2 | 3 4 | 5
__device__ constexpr int w2_aset(int par, int s) { return (s + 2 * par) & 3; }             // weight-quad sets, by period parity

#ifdef YOGO_DIAG
#define W2_DBG(BIT) (p.dbg & (BIT))
#define W2_STAMP() __builtin_amdgcn_s_memtime()
#else
#define W2_DBG(BIT) 0
#define W2_STAMP() 0ull
#endif

}  // namespace

// SIGNS: the epilogue multiplies by LeakyReLU'(reference) read from the reference's sign map (dx flows into a LeakyReLU block
// without BatchNorm: layer 3 of base_model); otherwise by the channel scale alone.
template <bool SIGNS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_bf16_ws2_kernel(const ConvWs2Params p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
  constexpr unsigned OOB = 0x80000000u;
  unsigned char* const lds = reinterpret_cast<unsigned char*>(smem4);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, tw = wave & 3;   // team 0 computes, team 1 loads; wavefronts tw and tw + 4 share a SIMD
  const int mh = tw >> 1, nh = tw & 1;         // compute wavefront tw: channels mh * 64 ..., quads nh * 64 ... of the tile
  [[maybe_unused]] const unsigned long long t_start = W2_STAMP();
  const int OH = p.OH, OW = p.OW, IH = p.IH, IW = p.IW;
  const int plane = OH * OW, plane16 = plane * 16;
  const int nck = p.nck, np = nck >> 1;

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
      const int b = w2_udivm1(widx, p.gx, p.m_gx);
      const int bx = widx - b * p.gx;
      const int cb = w2_udivm1(bx, p.tiles_per_band, p.m_tpb);
      const int tb = bx - cb * p.tiles_per_band;
      const int j0 = cb * p.TW;
      const int bw = min(p.TW, IW - j0);
      const int NPb = IH * bw;
      const int p0 = tb * W2_PT;
      if (p0 >= NPb) continue;
      t.b = b; t.j0 = j0; t.bw = bw; t.p0 = p0; t.p1 = min(p0 + W2_PT, NPb); t.lastband = cb == p.ncb - 1;
      return true;
    }
  };

  unsigned k_ord = 0;
  TileS T{};
  if (!find_tile(k_ord, T)) return;
#ifdef YOGO_DIAG
  if (W2_DBG(32)) {   // experiment: the workgroups of a CU group start an eighth of a tile apart (are the CUs' store bursts in phase?)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long dt = (unsigned long long)((blockIdx.x >> 3) & 7u) * 4000ull;
    while (__builtin_amdgcn_s_memtime() - t0 < dt) __builtin_amdgcn_s_sleep(8);
  }
#endif
  if (p.chan_scale == nullptr && tid < 256) reinterpret_cast<float*>(lds + W2_ES)[tid] = 1.f;
  __syncthreads();

  if (team == 1) {
    // =====================================================================================================================
    // LOADERS (4): wavefront tw stages quarter tw of every weight group, positions tw * 64 ... of every dy channel block, and the
    // sign bytes of its partner compute wavefront tw
    // =====================================================================================================================
    const int lane = w2_lane();
    const int q31 = lane & 31, hp = lane >> 5;
    const int ttid = tw * 64 + lane;
    const int rowb = IW * 16, kcb = IH * IW * 16;
    const unsigned ibytes = (unsigned)p.Kb * kcb, wbytes = 9u * p.Kb * 2048u;
    const unsigned wstep = (unsigned)p.Kb * 2048u;   // bytes between the slices of the packed weights
    const i32x4 rs_w = w2_rsrc(p.wp, wbytes);
    const int lane16 = W2_DBG(4) ? (int)OOB : lane * 16;
    const bool has_scale = p.chan_scale != nullptr;
    const i32x4 rs_sc = w2_rsrc(p.chan_scale, has_scale ? (unsigned)p.B * 512u : 0u);
    // per-tile lane geometry: the dy element this lane stages (position ttid of the [rows_in][lw] image of a channel block) and the
    // sign-map offsets it fetches for its partner (lane = (half-wave hp, quad q31) of pixel group n)
    struct LaneGeo { int dyoff; int vs[2][2][2]; };
    auto decode = [&](const TileS& t, LaneGeo& g) __attribute__((always_inline)) {
      const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
      const int bw = t.bw;
      const int i_lo = w2_udivm1(t.p0, bw, m_bw), i_hi = w2_udivm1(t.p1 - 1, bw, m_bw);
      const int rows_in = i_hi - i_lo + 2;
      const int lw = bw + 1;
      const unsigned inv_lw = t.lastband ? p.m_lwl : p.m_lw;   // (lw >= 2)
      const int r_ = w2_udivm(ttid, inv_lw), x_ = ttid - r_ * lw;
      const int iy = i_lo + r_, ix = t.j0 + x_;
      g.dyoff = (r_ < rows_in && iy < IH && ix < IW && !W2_DBG(4) && !W2_DBG(8)) ? iy * rowb + ix * 16 : (int)OOB;   // (diagnostic bit 8: no dy loads, weights as usual)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int pp = t.p0 + (nh * 2 + n) * 32 + q31;
        const bool pv = pp < t.p1;
        const int pc = pv ? pp : (t.p1 - 1);
        const int i = w2_udivm1(pc, bw, m_bw), j = pc - i * bw;
        const int pix = 2 * i * OW + 2 * (t.j0 + j);
        const bool vx = 2 * (t.j0 + j) + 1 < OW, vy = 2 * i + 1 < OH;
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px)
            g.vs[py][n][px] = (pv && (py == 0 || vy) && (px == 0 || vx)) ? (hp * plane + pix + py * OW + px) * 8 + mh * 4 : (int)OOB;
      }
    };
    auto rs_in_of = [&](int b) __attribute__((always_inline)) { return w2_rsrc(reinterpret_cast<const unsigned char*>(p.in) + (size_t)b * ibytes, ibytes); };
    auto rs_sg_of = [&](int b) __attribute__((always_inline)) { return w2_rsrc(SIGNS ? p.signs + (size_t)b * plane16 : nullptr, SIGNS ? (unsigned)plane16 : 0u); };
    // weights of a pass-A period (chunks 2 k, 2 k + 1: groups cc * 3 + slice) / a pass-B period (chunk c: slices 3 .. 8) -> buffer wb
    auto req_wA = [&](int k, int wb) __attribute__((always_inline)) {
      if constexpr ((W2_ABL & 4) != 0) return;
      w2_dma3(rs_w, (unsigned)(wb * W2_WB + tw * 1024), lane16, (unsigned)((4 * k) * 2048 + tw * 1024), wstep);
      w2_dma3(rs_w, (unsigned)(wb * W2_WB + 3 * 4096 + tw * 1024), lane16, (unsigned)((4 * k + 2) * 2048 + tw * 1024), wstep);
    };
    auto req_wB = [&](int c, int wb) __attribute__((always_inline)) {
      if constexpr ((W2_ABL & 4) != 0) return;
      w2_dma6(rs_w, (unsigned)(wb * W2_WB + tw * 1024), lane16, (unsigned)(3u * wstep + (unsigned)((2 * c) * 2048 + tw * 1024)), wstep);
    };
    // 16-channel chunk c of the dy tile described by (rs, dyoff) -> slot c (this wavefront's 64 positions of both channel blocks)
    auto req_dy = [&](i32x4 rs, int dyoff, int c) __attribute__((always_inline)) {
      w2_dma2(rs, (unsigned)(W2_DY + c * W2_DYS + tw * 1024), dyoff, (unsigned)(2 * c) * (unsigned)kcb, (unsigned)kcb);
    };
    auto req_signs = [&](i32x4 rs, const int (&vs)[2][2], int area) __attribute__((always_inline)) {
      if constexpr (SIGNS) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int px = 0; px < 2; ++px) w2_dma_dword(rs, (unsigned)(W2_SG + area * 4096 + (tw * 4 + n * 2 + px) * 256), vs[n][px], 0u);
      }
    };
    auto req_scale = [&](int b, int par) __attribute__((always_inline)) {   // [128] floats: two 256-byte pieces, loaders 2 / 3 repeat those of 0 / 1
      if (has_scale) w2_dma_dword(rs_sc, (unsigned)(W2_ES + par * 512 + (tw & 1) * 256), lane * 4, (unsigned)((b * 128 + (tw & 1) * 64) * 4));
    };
    LaneGeo gc{}, gn{};
    decode(T, gc);
    i32x4 rs_in = rs_in_of(T.b), rs_in_n = rs_in;
    // Weight requests run TWO periods ahead of the MFMAs (three buffers): the requests of the two periods behind an epilogue are in the
    // memory pipeline before its first store is issued
    int wreq = 0;   // buffer of the next weight request (rotates 0, 1, 2)
    auto req_w = [&](bool pass_b, int idx) __attribute__((always_inline)) {
      if (pass_b) req_wB(idx, wreq);
      else req_wA(idx, wreq);
      wreq = wreq == W2_NWB - 1 ? 0 : wreq + 1;
    };
    auto vmwait_n = [&](int n) __attribute__((always_inline)) {   // (uniform) all but the n youngest vector-memory operations are done
      switch (n) {
        case 0: w2_vmwait<0>(); break;
        case 1: w2_vmwait<1>(); break;
        case 2: w2_vmwait<2>(); break;
        case 4: w2_vmwait<4>(); break;
        case 5: w2_vmwait<5>(); break;
        case 6: w2_vmwait<6>(); break;
        case 7: w2_vmwait<7>(); break;
        case 8: w2_vmwait<8>(); break;
        case 10: w2_vmwait<10>(); break;
        case 11: w2_vmwait<11>(); break;
        default: w2_vmwait<0>(); break;
      }
    };
    const int n_small = (SIGNS ? 4 : 0) + (has_scale ? 1 : 0);
    // first tile: every chunk but the last (the first period of every tile requests that one) and the first two periods' weights
    for (int c = 0; c + 1 < nck; ++c) req_dy(rs_in, gc.dyoff, c);
    req_w(false, 0);
    req_w(false, 1);   // (np >= 2)
    w2_vmwait<0>();
    w2_barrier();   // (#1)
    int tpar = 0;           // channel-scale buffer of the tile
    bool has_next = true;
    TileS Tn{};
    for (;;) {
      // ---------------- pass A: np >= 2 periods of 32 channels
      for (int k = 0; k < np; ++k) {
        int young = 0;   // operations of this period that may stay in flight behind the barrier
        if (k == 0) req_dy(rs_in, gc.dyoff, nck - 1);   // this tile's last dy chunk (its slot was pass B's last of the previous tile): landed with this barrier
        if (k + 2 < np) req_w(false, k + 2);
        else req_w(true, k + 2 - np);   // (nck >= 4)
        young += 6;
        if (k == 0) {
          // what the tile's first epilogue reads: pass A's sign bytes, the channel scale of the tile's image
          req_signs(rs_sg_of(T.b), gc.vs[0], 0);
          req_scale(T.b, tpar);
          young += n_small;
          // the next tile: looked up and decoded here, behind this period's requests
          unsigned kn = k_ord + 1;
          has_next = find_tile(kn, Tn);
          k_ord = kn;
          if (has_next) {
            decode(Tn, gn);
            rs_in_n = rs_in_of(Tn.b);
          }
        }
        vmwait_n(young);
        w2_barrier();
      }
      // ---------------- pass B: nck periods of 16 channels
      for (int c = 0; c < nck; ++c) {
        int young = 0;
        if (c + 2 < nck) { req_w(true, c + 2); young += 6; }
        else if (has_next) { req_w(false, c + 2 - nck); young += 6; }
        if (c == 0) {
          req_signs(rs_sg_of(T.b), gc.vs[1], 1);   // pass B's sign bytes (their own area: the compute wavefronts may still be in pass A's epilogue)
          young += SIGNS ? 4 : 0;
        } else if (has_next) {
          req_dy(rs_in_n, gn.dyoff, c - 1);   // slot c - 1 is free: the next tile's chunk
          young += 2;
        }
        vmwait_n(young);
        w2_barrier();
      }
      if (!has_next) break;
      T = Tn;
      gc = gn;
      rs_in = rs_in_n;
      tpar ^= 1;
    }
    return;
  }

  // =======================================================================================================================
  // COMPUTE
  // =======================================================================================================================
  const int lane = w2_lane(), l31 = lane & 31, half = lane >> 5;
  const unsigned a_b0 = (unsigned)(half * 128 + mh * 64 + l31) * 16u;   // weight unit [channel block half][channel] of row block mb = 0
  const unsigned obytes = 16u * plane16;
  unsigned pbr[2], lw16;
  int pix16[2];   // byte offset of this lane's 16-byte unit of quad (n)'s pixel (2 i, 2 j) inside the image: + half * plane16 (the upper half-wave stores the next channel block)
  unsigned vmask = 0;   // bit n: the quad exists; bit 2 + n: its odd column exists; bit 4 + n: its odd row exists
  auto decode_pix = [&](const TileS& t) __attribute__((always_inline)) {
    const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
    const int bw = t.bw;
    const int i_lo = w2_udivm1(t.p0, bw, m_bw);
    lw16 = (unsigned)(bw + 1) * 16u;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int pp = t.p0 + (nh * 2 + n) * 32 + l31;
      const int pc = pp < t.p1 ? pp : (t.p1 - 1);
      const int i = w2_udivm1(pc, bw, m_bw), j = pc - i * bw;
      pbr[n] = (unsigned)((i - i_lo) * (bw + 1) + j) * 16u + (unsigned)half * 4096u;
      pix16[n] = (2 * i * OW + 2 * (t.j0 + j)) * 16 + half * plane16;
      vmask = (vmask & ~(0x15u << n)) | ((pp < t.p1 ? 1u : 0u) << n) | ((2 * (t.j0 + j) + 1 < OW ? 1u : 0u) << (2 + n)) | ((2 * i + 1 < OH ? 1u : 0u) << (4 + n));
    }
  };
  decode_pix(T);
  // the operand sets of the K steps: weight quads of step s of a period of parity PAR in A[w2_aset(PAR, s)], pixel quads in B[w2_bset(s)]
  u32x4 A[4][2], B[4][2];
  unsigned pb0[2], pb1[2];   // this lane's pixel-quad addresses in the period's first dy slot: row a, row a + 1
  // one period = six K steps.  FIRST: the pass's first period (the accumulators start from zero); LASTP: its last (no operands
  // of a next period; the MFMA -> vector-ALU wait states).  pa / pan: this lane's weight unit in this / the next period's buffer
  auto period = [&](auto pb_tag, auto first_tag, auto last_tag, auto par_tag, unsigned pa, unsigned pan) __attribute__((always_inline)) {
    constexpr bool PB = decltype(pb_tag)::value, FIRST = decltype(first_tag)::value, LASTP = decltype(last_tag)::value;
    constexpr int PAR = decltype(par_tag)::value;
    constexpr int NSL = PB ? W2_DYS : 2 * W2_DYS;   // bytes from this period's first dy slot to the next period's
#define W2_A(S) A[w2_aset(PAR, S)]
#define W2_B(S) B[w2_bset(S)]
#define W2_PX(S) ((S) % 3 == 0 ? 0 : 1)
#define W2_PB(S, N) (w2_brow(PB, S) ? pb1[N] : pb0[N])
    // step 0: reads of step 2
    w2_k_ab<W2_PX(0), FIRST, w2_grp(PB, 2) * 4096, w2_bimm(PB, 2)>(W2_A(0)[0], W2_A(0)[1], W2_B(0)[0], W2_B(0)[1], W2_A(2)[0], W2_A(2)[1], W2_B(2)[0], W2_B(2)[1], pa,
                                                                  W2_PB(2, 0), W2_PB(2, 1));
    // step 1: reads of step 3
    w2_k_ab<W2_PX(1), FIRST, w2_grp(PB, 3) * 4096, w2_bimm(PB, 3)>(W2_A(1)[0], W2_A(1)[1], W2_B(1)[0], W2_B(1)[1], W2_A(3)[0], W2_A(3)[1], W2_B(3)[0], W2_B(3)[1], pa,
                                                                  W2_PB(3, 0), W2_PB(3, 1));
    // step 2: reads of steps 4 and 5
    w2_k_aab<W2_PX(2), true, w2_grp(PB, 4) * 4096, w2_grp(PB, 5) * 4096, w2_bimm(PB, 5)>(W2_A(2)[0], W2_A(2)[1], W2_B(2)[0], W2_B(2)[1], W2_A(4)[0], W2_A(4)[1], W2_A(5)[0],
                                                                                        W2_A(5)[1], W2_B(5)[0], W2_B(5)[1], pa, W2_PB(5, 0), W2_PB(5, 1));
    // step 3: everything of this period is in registers -> the period's barrier
    w2_k_0<W2_PX(3), 1>(W2_A(3)[0], W2_A(3)[1], W2_B(3)[0], W2_B(3)[1], W2_A(4)[0], W2_A(4)[1], W2_A(5)[0], W2_A(5)[1], W2_B(5)[0], W2_B(5)[1]);
    if constexpr (!LASTP) {
      // step 4: the next period's steps 0 and 1 (its parity is the other one: their weight quads go to the sets of this period's steps 2, 3)
      w2_k_aab<W2_PX(4), false, 0, w2_grp(PB, 1) * 4096, NSL>(W2_A(4)[0], W2_A(4)[1], W2_B(4)[0], W2_B(4)[1], W2_A(2)[0], W2_A(2)[1], W2_A(3)[0], W2_A(3)[1], W2_B(0)[0],
                                                              W2_B(0)[1], pan, pb0[0], pb0[1]);
      w2_k_0<W2_PX(5), 2>(W2_A(5)[0], W2_A(5)[1], W2_B(5)[0], W2_B(5)[1], W2_A(2)[0], W2_A(2)[1], W2_A(3)[0], W2_A(3)[1], W2_B(0)[0], W2_B(0)[1]);
    } else {
      w2_k_0<W2_PX(4), 0>(W2_A(4)[0], W2_A(4)[1], W2_B(4)[0], W2_B(4)[1], W2_A(2)[0], W2_A(2)[1], W2_A(3)[0], W2_A(3)[1], W2_B(0)[0], W2_B(0)[1]);
      w2_k_0<W2_PX(5), 3>(W2_A(5)[0], W2_A(5)[1], W2_B(5)[0], W2_B(5)[1], W2_A(2)[0], W2_A(2)[1], W2_A(3)[0], W2_A(3)[1], W2_B(0)[0], W2_B(0)[1]);
    }
#undef W2_A
#undef W2_B
#undef W2_PX
#undef W2_PB
  };
  // a pass: `nper` periods from dy slot 0 on; the weight buffer rotates (wcur)
  auto run_pass = [&](auto pb_tag, int nper, int& wcur) __attribute__((always_inline)) {
    constexpr bool PB = decltype(pb_tag)::value;
    using TT = std::true_type;
    using FT = std::false_type;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      pb0[n] = (unsigned)W2_DY + pbr[n];
      pb1[n] = pb0[n] + lw16;
    }
    w2_kfirst<w2_grp(PB, 1) * 4096>(A[0][0], A[0][1], A[1][0], A[1][1], B[0][0], B[0][1], (unsigned)(wcur * W2_WB) + a_b0, pb0[0], pb0[1]);
    for (int k = 0; k < nper; ++k) {
      const int wnext = wcur == W2_NWB - 1 ? 0 : wcur + 1;
      const unsigned pa = (unsigned)(wcur * W2_WB) + a_b0, pan = (unsigned)(wnext * W2_WB) + a_b0;
      const bool last = k == nper - 1;   // (uniform)
      if (k == 0) {
        if (last) period(pb_tag, TT{}, TT{}, W2IC<0>{}, pa, pan);
        else period(pb_tag, TT{}, FT{}, W2IC<0>{}, pa, pan);
      } else if (k & 1) {
        if (last) period(pb_tag, FT{}, TT{}, W2IC<1>{}, pa, pan);
        else period(pb_tag, FT{}, FT{}, W2IC<1>{}, pa, pan);
      } else {
        if (last) period(pb_tag, FT{}, TT{}, W2IC<0>{}, pa, pan);
        else period(pb_tag, FT{}, FT{}, W2IC<0>{}, pa, pan);
      }
      wcur = wnext;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        pb0[n] += PB ? W2_DYS : 2 * W2_DYS;
        pb1[n] += PB ? W2_DYS : 2 * W2_DYS;
      }
    }
  };
  // ---- a pass's epilogue: 8 accumulator tiles -> 16 stores.  Channel group (mb, gp) of pixel group n, column parity px: 8 values
  //      per lane (4 channels of channel block cb, 4 of cb + 1) x scale [x LeakyReLU'(sign bit)] -> bf16; the two half-waves exchange one
  //      8-byte group (v_permlane32_swap) so that every lane stores a whole 16-byte unit -- lanes 0-31 block cb, lanes 32-63 block cb + 1
  //      (conv_bf16_kernel's epilogue: the same formula, the same bits).  A compute wavefront issues no loads: its stores are never waited for.
  int tpar = 0;   // channel-scale buffer of the tile
  auto epilogue = [&](auto py_tag, i32x4 rs_o) __attribute__((always_inline)) {
    constexpr int PY = decltype(py_tag)::value;
    if constexpr ((W2_ABL & 8) != 0) return;
    unsigned sg[2][2] = {{0u, 0u}, {0u, 0u}};   // this lane's sign bytes of the pass: [n][px], byte mb * 2 + gp
    if constexpr (SIGNS) {
      const unsigned* sp = reinterpret_cast<const unsigned*>(lds + W2_SG + PY * 4096) + tw * 256 + lane;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int px = 0; px < 2; ++px) sg[n][px] = sp[(n * 2 + px) * 64];
    }
    int vo[2][2];   // [n][px]
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const bool ok = ((vmask >> n) & 1u) && (px == 0 || ((vmask >> (2 + n)) & 1u)) && (PY == 0 || ((vmask >> (4 + n)) & 1u)) && !W2_DBG(1);
        vo[n][px] = ok ? pix16[n] + (PY * OW + px) * 16 : (int)OOB;
      }
    w2_static_for([&](auto q_tag) __attribute__((always_inline)) {
      constexpr int Q = decltype(q_tag)::value, MB = Q >> 1, GP = Q & 1;
      const float* es = reinterpret_cast<const float*>(lds + W2_ES) + tpar * 128 + mh * 64 + MB * 32 + 16 * GP + 4 * half;
      const float4 sA = *reinterpret_cast<const float4*>(es), sB = *reinterpret_cast<const float4*>(es + 8);
      const float sa[8] = {sA.x, sA.y, sA.z, sA.w, sB.x, sB.y, sB.z, sB.w};
      float sl_[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) sl_[i] = LEAKY_SLOPE * sa[i];
      const unsigned so = (unsigned)(mh * 8 + Q * 2) * (unsigned)plane16;   // channel block the lower half-wave stores
      w2_static_for([&](auto u_tag) __attribute__((always_inline)) {   // the group's four units (n, px)
        constexpr int n = decltype(u_tag)::value >> 1, px = decltype(u_tag)::value & 1;
        float v[8], r[8];
        w2_acc_read8<16 * (4 * px + 2 * MB + n) + 8 * GP>(r);
        if constexpr (SIGNS) {
          const unsigned m = sg[n][px] >> (8 * Q);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int t = (int)(m << (31 - i)) >> 31;   // bit i spread over the word (v_bfe_i32) selects scale or 0.01 * scale (v_bfi_b32)
            unsigned f;
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(f) : "v"(t), "v"(sa[i]), "v"(sl_[i]));
            v[i] = r[i] * __builtin_bit_cast(float, f);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = fmaf(r[i], sa[i], 0.f * sa[i]);   // (conv_bf16_epi_groups.inc: fma(acc, scale, bias * scale), bias = 0)
        }
        if (W2_DBG(2)) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = r[i];
        }
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
        const u32x4 w = __builtin_bit_cast(u32x4, o);   // (x, y) = this lane's 4 channels of block cb, (z, w) = of block cb + 1
        // lanes 32-63 hand their block-cb group down, lanes 0-31 hand their block-(cb + 1) group up
        const auto r0 = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);
        const u32x4 st = {r0[0], r1[0], r0[1], r1[1]};
        w2_store16(st, vo[n][px], rs_o, so);
      }, std::make_integer_sequence<int, 4>{});
    }, std::make_integer_sequence<int, 4>{});
  };

  w2_barrier();   // (#1)
  int wcur = 0;   // weight buffer of the MFMA period being computed (rotates 0, 1, 2)
  [[maybe_unused]] unsigned long long t_a = 0, t_ea = 0, t_b = 0, t_eb = 0;
  for (;;) {
    const i32x4 rs_o = w2_rsrc(reinterpret_cast<unsigned char*>(p.out) + (size_t)T.b * obytes, obytes);
    [[maybe_unused]] const unsigned long long s0 = W2_STAMP();
    run_pass(std::false_type{}, np, wcur);
    [[maybe_unused]] const unsigned long long s1 = W2_STAMP();
    epilogue(W2IC<0>{}, rs_o);
    [[maybe_unused]] const unsigned long long s2 = W2_STAMP();
    run_pass(std::true_type{}, nck, wcur);
    [[maybe_unused]] const unsigned long long s3 = W2_STAMP();
    epilogue(W2IC<1>{}, rs_o);
    [[maybe_unused]] const unsigned long long s4 = W2_STAMP();
    t_a += s1 - s0; t_ea += s2 - s1; t_b += s3 - s2; t_eb += s4 - s3;
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
    d[0] = t_start; d[1] = __builtin_amdgcn_s_memtime(); d[2] = t_a; d[3] = t_ea; d[4] = t_b; d[5] = t_eb; d[6] = k_ord;
  }
#endif
}

// =========================================================================================================
// host side: eligibility, tiling, launch
// =========================================================================================================
bool conv_bf16_ws2_eligible(int K, int M, int OH, int OW, int B) {
  if (M != 128 || K < 64 || K > 128 || (K % 32) != 0) return false;   // nck = K / 16 even, <= 8 dy slots, >= 2 pass-A periods
  if (OH < 2 || OW < 2 || B <= 0) return false;
  const long long IH = (OH + 1) / 2, IW = (OW + 1) / 2;
  if ((long long)(K / 8) * IH * IW * 16 >= (1ll << 31) || (long long)16 * OH * OW * 16 >= (1ll << 31)) return false;   // per-image descriptors
  return true;
}

// column bands of TW quads, tiles of 128 consecutive quads of a band (row-major inside the band): the staged dy tile of a channel
// block ([rows + 1][TW + 1] units: one halo row below, one halo column right) has to fit the 256 units of a slot's half; among
// the fitting band counts take the fewest tiles per image (MFMA work), then the fewest staged units
bool conv_bf16_ws2_plan(ConvWs2Params* p) {
  const int QH = p->IH, QW = p->IW;   // the quad grid = the dy grid
  long long best = -1;
  int best_ncb = 0;
  for (int ncb = 1; ncb <= 64 && ncb <= QW; ++ncb) {
    const int TW = cdiv(QW, ncb);
    if (cdiv(QW, TW) != ncb) continue;
    const int bw_min = QW - (ncb - 1) * TW;
    auto rows_of = [&](int bw) { return min(QH, 1 + cdiv(W2_PT - 1, bw)) + 1; };
    const int need = max(rows_of(TW) * (TW + 1), rows_of(bw_min) * (bw_min + 1));
    if (need > 256) continue;
    const long long tiles = (long long)(ncb - 1) * cdiv(QH * TW, W2_PT) + cdiv(QH * bw_min, W2_PT);
    const long long staged = (long long)(ncb - 1) * cdiv(QH * TW, W2_PT) * rows_of(TW) * (TW + 1) + (long long)cdiv(QH * bw_min, W2_PT) * rows_of(bw_min) * (bw_min + 1);
    const long long score = tiles * 100000000ll + staged;
    if (best < 0 || score < best) { best = score; best_ncb = ncb; }
  }
  if (best < 0) return false;
  p->ncb = best_ncb;
  p->TW = cdiv(QW, best_ncb);
  p->tiles_per_band = cdiv(QH * p->TW, W2_PT);
  p->gx = p->ncb * p->tiles_per_band;
  p->ntiles = p->B * p->gx;
  auto magic = [](int d) -> unsigned { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(((1ull << 32) + (unsigned)d - 1ull) / (unsigned)d); };
  const int bw_last = QW - (p->ncb - 1) * p->TW;
  p->m_gx = magic(p->gx); p->m_tpb = magic(p->tiles_per_band);
  p->m_bw = magic(p->TW); p->m_bwl = magic(bw_last);
  p->m_lw = magic(p->TW + 1); p->m_lwl = magic(bw_last + 1);
  p->nck = p->Kb / 2;
  return true;
}

int launch_conv_bf16_ws2(const ConvWs2Params& p, hipStream_t stream) {
  static std::mutex mu;
  static int n_cu_of[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    yogo_set_error("conv_bf16_ws2: hipGetDevice failed");
    return YOGO_ERR_HIP;
  }
  int n_cu;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (n_cu_of[dev] == 0) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_ws2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS_BYTES);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_ws2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS_BYTES);
      if (e != hipSuccess) {
        yogo_set_error("conv_bf16_ws2: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", W2_LDS_BYTES, hipGetErrorString(e));
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
  if (p.signs != nullptr) hipLaunchKernelGGL(conv_bf16_ws2_kernel<true>, dim3(grid), dim3(512), W2_LDS_BYTES, stream, p);
  else hipLaunchKernelGGL(conv_bf16_ws2_kernel<false>, dim3(grid), dim3(512), W2_LDS_BYTES, stream, p);
  if (yogo_launch_log_enabled())
    yogo_launch_log("conv_bf16_ws2_kernel<%s> | Kb=%d dy=%dx%d dx=%dx%d ncb=%d TW=%d tiles_per_band=%d nck=%d ntiles=%d grid=%d lds=%d signs=%d scale=%d",
                    p.signs != nullptr ? "true" : "false", p.Kb, p.IH, p.IW, p.OH, p.OW, p.ncb, p.TW, p.tiles_per_band, p.nck, p.ntiles, grid, W2_LDS_BYTES,
                    p.signs != nullptr, p.chan_scale != nullptr);
  YOGO_CHECK_LAUNCH("conv_bf16_ws2");
  return YOGO_OK;
}
