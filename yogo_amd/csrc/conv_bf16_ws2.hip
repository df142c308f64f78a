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
//   * wavefronts 4-7 LOAD and STORE: the weight slices of the next period (24 KB, L2 hits) by LDS-DMA into the other weight
//     buffer, the next tile's dy, the sign bytes / channel scale of the pass whose epilogue comes next, and the output -- the
//     compute wavefronts write finished 16-byte units to an LDS staging slot (a quarter of a pass at a time), the loaders store
//     them with both column parities of a row in ONE instruction (64 lanes x 16 B contiguous: whole 128-byte lines, where the
//     tiled kernel's two half-filled stores per line relied on the L2 to merge them);
//   * one s_barrier per period orders everything (DMA landed: the loaders wait vmcnt first; staging written; buffers free).
// Periods of a tile: pass A in 32-channel periods, its epilogue in 4 quarters, pass B in 16-channel periods, 4 quarters -- every
// MFMA period is 24 MFMAs per compute wavefront and 24 KB of weights.
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
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(data), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void w2_vmwait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#ifdef YOGO_DIAG
// (the loaders' stamps only with diagnostic bit 64: an s_memtime is a scalar-memory round trip, four of them per period are not free)
#define W2_VMWAIT(N) do { if (W2_DBG(64)) { const unsigned long long v0__ = __builtin_amdgcn_s_memtime(); w2_vmwait<N>(); t_vm += __builtin_amdgcn_s_memtime() - v0__; } else w2_vmwait<N>(); } while (0)
#define W2_LBARRIER() do { if (W2_DBG(64)) { const unsigned long long v0__ = __builtin_amdgcn_s_memtime(); w2_barrier(); t_lb += __builtin_amdgcn_s_memtime() - v0__; } else w2_barrier(); } while (0)
#else
#define W2_VMWAIT(N) w2_vmwait<N>()
#define W2_LBARRIER() w2_barrier()
#endif
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

// ---- the store wavefront's AGPR quarter buffers: 16 staged units (unit u = 4 cw + 2 n + e, LDS bytes cw * 4096 + n * 2048 + e * 512
// from the lane's base) <-> a[BASE + 4 u : BASE + 4 u + 3]
#define W2_RDU(U) "ds_read_b128 a[%[b]+4*" #U ":%[b]+4*" #U "+3], %[ad] offset:(" #U "/4)*4096+((" #U "/2)%%2)*2048+(" #U "%%2)*512\n\t"
template <int BASE>
__device__ __forceinline__ void w2_rd16_a(unsigned addr) {
  asm volatile(W2_RDU(0) W2_RDU(1) W2_RDU(2) W2_RDU(3) W2_RDU(4) W2_RDU(5) W2_RDU(6) W2_RDU(7) W2_RDU(8) W2_RDU(9) W2_RDU(10) W2_RDU(11) W2_RDU(12) W2_RDU(13)
               W2_RDU(14) W2_RDU(15)
               :
               : [ad] "v"(addr), [b] "n"(BASE)
               : "memory", W2_ACC_CLOBBER);
}
#undef W2_RDU
// units 8 H .. 8 H + 7 (compute wavefronts 2 H, 2 H + 1) -> 8 stores: pixel-group offsets vo[s], s = 2 (cw & 1) + n; scalar offsets so0
// (channel block of e = 0) and so0 + dso (e = 1)
template <int BASE, int H>
__device__ __forceinline__ void w2_st8_a(const int (&vo)[4], i32x4 rs, unsigned so0, unsigned dso) {
  rs = w2_u4(rs);
  so0 = w2_u(so0);
  const unsigned so1 = w2_u(so0 + dso);
#define W2_STU(J, V, S) "buffer_store_dwordx4 a[%[b]+4*(8*%[h]+" #J "):%[b]+4*(8*%[h]+" #J ")+3], %[" #V "], %[rs], %[" #S "] offen\n\t"
  asm volatile("s_nop 4\n\t" W2_STU(0, v0, s0) W2_STU(1, v0, s1) W2_STU(2, v1, s0) W2_STU(3, v1, s1) W2_STU(4, v2, s0) W2_STU(5, v2, s1) W2_STU(6, v3, s0)
               W2_STU(7, v3, s1)
               :
               : [v0] "v"(vo[0]), [v1] "v"(vo[1]), [v2] "v"(vo[2]), [v3] "v"(vo[3]), [rs] "s"(rs), [s0] "s"(so0), [s1] "s"(so1), [b] "n"(BASE), [h] "n"(H)
               : "memory");
#undef W2_STU
}

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
  if (p.chan_scale == nullptr && tid < 128) reinterpret_cast<float*>(lds + W2_ES)[tid] = 1.f;
  __syncthreads();

  if (team == 1 && tw < 3) {
    // =====================================================================================================================
    // LOAD wavefronts (3): the weight slices of the next period, the next tile's dy and (wavefront 2) the small inputs of the
    // epilogues -- the sign bytes of a pass, the channel scale of the tile's image.  They issue no stores: vector-memory
    // operations retire in order, and a load that waits behind a store's acknowledgement holds the period's barrier.
    // =====================================================================================================================
    const int lane = w2_lane();
    const int q31 = lane & 31, hp = lane >> 5;
    const int rowb = IW * 16, kcb = IH * IW * 16;
    const unsigned ibytes = (unsigned)p.Kb * kcb, wbytes = 9u * p.Kb * 2048u;
    const unsigned wstep = (unsigned)p.Kb * 2048u;   // bytes between the slices of the packed weights
    const i32x4 rs_w = w2_rsrc(p.wp, wbytes);
    const int lane16 = W2_DBG(4) ? (int)OOB : lane * 16;
    const bool has_scale = p.chan_scale != nullptr;
    const i32x4 rs_sc = w2_rsrc(p.chan_scale, has_scale ? (unsigned)p.B * 512u : 0u);
    [[maybe_unused]] unsigned long long t_vm = 0, t_lb = 0;
    // dy element of position q * 64 + lane of the [rows_in][lw] image of a channel block, q = 0..3
    auto decode = [&](const TileS& t, int (&dyoff)[4]) __attribute__((always_inline)) {
      const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
      const int bw = t.bw;
      const int i_lo = w2_udivm1(t.p0, bw, m_bw), i_hi = w2_udivm1(t.p1 - 1, bw, m_bw);
      const int rows_in = i_hi - i_lo + 2;
      const int lw = bw + 1;
      const unsigned inv_lw = t.lastband ? p.m_lwl : p.m_lw;   // (lw >= 2)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pos = q * 64 + lane;
        const int r_ = w2_udivm(pos, inv_lw), x_ = pos - r_ * lw;
        const int iy = i_lo + r_, ix = t.j0 + x_;
        dyoff[q] = (r_ < rows_in && iy < IH && ix < IW && !W2_DBG(4)) ? iy * rowb + ix * 16 : (int)OOB;
      }
    };
    // sign-map offsets of the tile's four 32-quad groups s (compute wavefront cw owns groups 2 (cw & 1) + n): lane = (half-wave hp,
    // quad q31); + 4 bytes for the upper 64 channels
    auto decode_vs = [&](const TileS& t, int (&vs)[2][4][2]) __attribute__((always_inline)) {
      const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
      const int bw = t.bw;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int pp = t.p0 + s * 32 + q31;
        const bool pv = pp < t.p1;
        const int pc = pv ? pp : (t.p1 - 1);
        const int i = w2_udivm1(pc, bw, m_bw), j = pc - i * bw;
        const int pix = 2 * i * OW + 2 * (t.j0 + j);
        const bool vx = 2 * (t.j0 + j) + 1 < OW, vy = 2 * i + 1 < OH;
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px)
            vs[py][s][px] = (pv && (py == 0 || vy) && (px == 0 || vx)) ? (hp * plane + pix + py * OW + px) * 8 : (int)OOB;
      }
    };
    auto rs_in_of = [&](int b) __attribute__((always_inline)) { return w2_rsrc(reinterpret_cast<const unsigned char*>(p.in) + (size_t)b * ibytes, ibytes); };
    auto rs_sg_of = [&](int b) __attribute__((always_inline)) { return w2_rsrc(SIGNS ? p.signs + (size_t)b * plane16 : nullptr, SIGNS ? (unsigned)plane16 : 0u); };
    // weight groups 2 tw, 2 tw + 1 of a period (4 KB each, contiguous in memory and in LDS).  Pass A, period k: group cc * 3 + t =
    // slice t of chunk 2 k + cc; pass B, chunk c: group g = slice 3 + g
    auto req_wA = [&](int k, int wb) __attribute__((always_inline)) {
      if constexpr ((W2_ABL & 4) != 0) return;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int g = 2 * tw + u, cc = g >= 3 ? 1 : 0, t = g - 3 * cc;
        w2_dma4c(rs_w, (unsigned)(wb * W2_WB + g * 4096), lane16, (unsigned)t * wstep + (unsigned)((4 * k + 2 * cc) * 2048));
      }
    };
    auto req_wB = [&](int c, int wb) __attribute__((always_inline)) {
      if constexpr ((W2_ABL & 4) != 0) return;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int g = 2 * tw + u;
        w2_dma4c(rs_w, (unsigned)(wb * W2_WB + g * 4096), lane16, (unsigned)(3 + g) * wstep + (unsigned)((2 * c) * 2048));
      }
    };
    // 16-channel chunk c of a dy tile -> slot c: 8 pieces (channel block, position quarter); wavefront 0 takes pieces 0-2, 1: 3-5, 2: 6-7
    const int j_lo = tw * 3, nd = tw < 2 ? 3 : 2;
    auto req_dy = [&](i32x4 rs, const int (&dyoff)[4], int c) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (u < nd) {   // (uniform)
          const int j = j_lo + u, q = j & 3, kb = j >> 2;
          const int vo = q == 0 ? dyoff[0] : (q == 1 ? dyoff[1] : (q == 2 ? dyoff[2] : dyoff[3]));
          w2_dma1(rs, (unsigned)(W2_DY + c * W2_DYS + j * 1024), vo, (unsigned)(2 * c + kb) * (unsigned)kcb);
        }
      }
    };
    // the 16 sign dwords of a pass: compute wavefront cw's (n, px) -> [cw][n][px][64 lanes]; its channels' bytes are the 4 at + 4 (cw >> 1)
    auto req_signs = [&](i32x4 rs, const int (&vs)[4][2]) __attribute__((always_inline)) {
      if constexpr (SIGNS) {
#pragma unroll
        for (int cw = 0; cw < 4; ++cw)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int px = 0; px < 2; ++px)
              w2_dma_dword(rs, (unsigned)(W2_SG + (cw * 4 + n * 2 + px) * 256), vs[(cw & 1) * 2 + n][px], (unsigned)((cw >> 1) * 4));
      }
    };
    auto req_scale = [&](int b) __attribute__((always_inline)) {   // [128] floats: two 256-byte pieces
      if (has_scale) {
        w2_dma_dword(rs_sc, (unsigned)W2_ES, lane * 4, (unsigned)(b * 512));
        w2_dma_dword(rs_sc, (unsigned)(W2_ES + 256), lane * 4, (unsigned)(b * 512 + 256));
      }
    };
    const bool small = tw == 2;   // (uniform) this wavefront also fetches the epilogues' small inputs
    int dyo[4], dyo_n[4] = {(int)OOB, (int)OOB, (int)OOB, (int)OOB};
    int vs[2][4][2], vs_n[2][4][2];
    decode(T, dyo);
    if (small) decode_vs(T, vs);
    i32x4 rs_in_n = rs_in_of(T.b);
    for (int c = 0; c < nck; ++c) req_dy(rs_in_n, dyo, c);
    req_wA(0, 0);
    if (small) {
      req_signs(rs_sg_of(T.b), vs[0]);
      req_scale(T.b);
    }
    W2_VMWAIT(0);
    W2_LBARRIER();   // (#1)
    int wpar = 0;           // weight buffer of the MFMA period being computed
    bool has_next = true;
    TileS Tn{};
    for (;;) {
      for (int k = 0; k < np; ++k) {           // pass A
        if (k + 1 < np) req_wA(k + 1, wpar ^ 1);
        else req_wB(0, wpar ^ 1);
        if (k == 0) {   // the next tile: looked up and decoded behind this period's requests
          // (the channel scale is read by the epilogue quarters only: free since the previous tile's last one; it may stay in flight)
          if (small) req_scale(T.b);
          unsigned kn = k_ord + 1;
          has_next = find_tile(kn, Tn);
          k_ord = kn;
          if (has_next) {
            decode(Tn, dyo_n);
            if (small) decode_vs(Tn, vs_n);
            rs_in_n = rs_in_of(Tn.b);
          }
          if (small && has_scale) W2_VMWAIT(2);
          else W2_VMWAIT(0);
        } else {
          W2_VMWAIT(0);
        }
        W2_LBARRIER();
        wpar ^= 1;
      }
      for (int e = 0; e < 4; ++e) W2_LBARRIER();   // epilogue A
      for (int c = 0; c < nck; ++c) {          // pass B
        if (c + 1 < nck) req_wB(c + 1, wpar ^ 1);
        else if (has_next) req_wA(0, wpar ^ 1);
        if (c == 0) {   // pass B's sign bytes (pass A's were read in front of its first quarter); they may stay in flight
          if (small && SIGNS) {
            req_signs(rs_sg_of(T.b), vs[1]);
            W2_VMWAIT(16);
          } else {
            W2_VMWAIT(0);
          }
        } else if (has_next) {   // slot c - 1 is free: the next tile's chunk (it may stay in flight)
          req_dy(rs_in_n, dyo_n, c - 1);
          if (tw < 2) W2_VMWAIT(3);
          else W2_VMWAIT(2);
        } else {
          W2_VMWAIT(0);
        }
        W2_LBARRIER();
        wpar ^= 1;
      }
      for (int e = 0; e < 4; ++e) {            // epilogue B
        if (e == 0 && has_next) req_dy(rs_in_n, dyo_n, nck - 1);
        if (e == 1 && has_next && small) req_signs(rs_sg_of(Tn.b), vs_n[0]);   // (pass B's bytes were read in front of quarter 0)
        if (e == 3) W2_VMWAIT(0);   // (everything of the next tile's first period has landed)
        W2_LBARRIER();
      }
      if (!has_next) break;
      T = Tn;
      if (small) {
#pragma unroll
        for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
          for (int b_ = 0; b_ < 4; ++b_)
#pragma unroll
            for (int c_ = 0; c_ < 2; ++c_) vs[a_][b_][c_] = vs_n[a_][b_][c_];
      }
    }
#ifdef YOGO_DIAG
    if (p.stamps && tw == 0 && lane == 0) {
      unsigned long long* d = p.stamps + (size_t)blockIdx.x * 16;
      d[10] = t_vm; d[11] = t_lb;
    }
#endif
    return;
  }
  if (team == 1) {
    // =====================================================================================================================
    // STORE wavefront: the output.  The compute wavefronts stage a quarter of a pass (16 units of 1 KB) per epilogue period; this
    // wavefront moves every staged quarter into REGISTERS in the following period (the slot is free again) and issues the stores at
    // an even pace -- 8 per period -- over the periods that follow.  Why: the output is 128 KB per tile, ~80 % of what a CU's share
    // of the chip's write bandwidth moves in a tile's time (tools/probes/store_rate.hip: 10.5 B/cycle/CU at saturation, one store
    // wavefront per CU reaches it).  Issued in bursts behind the epilogues the stores back up in the CU's memory pipeline, the weight
    // loads queue behind them and the MFMA passes wait (first forms of this kernel: the dy-load + store skeleton alone took 0.63 ms);
    // and a store wavefront that must finish 16 stores inside an epilogue period holds that period's barrier.  It issues no loads and
    // never waits for a store.  Register FIFO: three quarter buffers -- a[0:63], a[64:127] (this role's AGPRs hold data, not
    // accumulators) and 16 VGPR quads; the schedule below is static (K = 128: 4 + 4 + 8 + 4 periods per tile).
    // =====================================================================================================================
    const int lane = w2_lane();
    const int q31 = lane & 31, hp = lane >> 5;
    const unsigned obytes = 16u * plane16;
    [[maybe_unused]] unsigned long long t_vm = 0, t_lb = 0;
    [[maybe_unused]] unsigned dbg_run = 0;
    // output offsets of the tile's four 32-quad groups s: lane = (column parity hp, quad q31)
    auto decode = [&](const TileS& t, int (&vo)[2][4]) __attribute__((always_inline)) {
      const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
      const int bw = t.bw;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int pp = t.p0 + s * 32 + q31;
        const bool pv = pp < t.p1;
        const int pc = pv ? pp : (t.p1 - 1);
        const int i = w2_udivm1(pc, bw, m_bw), j = pc - i * bw;
        const int pix = 2 * i * OW + 2 * (t.j0 + j);
        const bool vx = 2 * (t.j0 + j) + 1 < OW, vy = 2 * i + 1 < OH;
#pragma unroll
        for (int py = 0; py < 2; ++py) vo[py][s] = (pv && (py == 0 || vy) && (hp == 0 || vx) && !W2_DBG(1)) ? (pix + py * OW + hp) * 16 : (int)OOB;
      }
    };
    auto rs_out_of = [&](int b) __attribute__((always_inline)) { return w2_rsrc(reinterpret_cast<unsigned char*>(p.out) + (size_t)b * obytes, obytes); };
    // staged unit u = 4 cw + 2 n + e of a quarter (mb, gp): compute wavefront cw's pixel group n, channel block e of the pair; a lane
    // reads the column parity hp's 16 bytes of quad q31 -> one store covers 32 quads x both column parities of ONE channel block
    const unsigned stg_rd = (unsigned)(W2_STG + hp * 1024 + q31 * 16);
    // ... immediately (any K): 16 reads, 16 stores
    auto store_quarter = [&](int sl, int quarter, const int (&vo)[4], i32x4 rs_o) __attribute__((always_inline)) {
      const unsigned char* base = lds + stg_rd + sl * W2_SLOT;
#pragma unroll
      for (int cw = 0; cw < 4; ++cw) {
        u32x4 d[4];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int e = 0; e < 2; ++e) d[n * 2 + e] = *reinterpret_cast<const u32x4*>(base + cw * 4096 + n * 2048 + e * 512);
        const int cb0 = (cw >> 1) * 8 + quarter * 2;   // channel block of e = 0 (quarter = mb * 2 + gp)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int e = 0; e < 2; ++e) w2_store16(d[n * 2 + e], vo[(cw & 1) * 2 + n], rs_o, (unsigned)(cb0 + e) * (unsigned)plane16);
      }
    };
    int vo[2][4], vo_n[2][4];
    decode(T, vo);
    i32x4 rs_out = rs_out_of(T.b);
    W2_LBARRIER();   // (#1)
    bool has_next = true;
    TileS Tn{};
    int vo_prev[4] = {(int)OOB, (int)OOB, (int)OOB, (int)OOB};   // pass B of the previous tile (its stores run into this tile's first periods)
    i32x4 rs_prev = rs_out;
    auto next_tile = [&]() __attribute__((always_inline)) {
      unsigned kn = k_ord + 1;
      has_next = find_tile(kn, Tn);
      k_ord = kn;
      if (has_next) decode(Tn, vo_n);
    };
    auto advance = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 4; ++s) vo_prev[s] = vo[1][s];
      rs_prev = rs_out;
      if (has_next) {
        T = Tn;
#pragma unroll
        for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
          for (int s = 0; s < 4; ++s) vo[a_][s] = vo_n[a_][s];
        rs_out = rs_out_of(T.b);
      }
    };
    if (nck == 8) {
      // ---- the paced form.  Quarter buffers: QB0 = a[0:63], QB1 = a[64:127], QB2 = q2[16].  Per tile (periods t = 0..19: P0-3, EA0-3,
      //      Q0-7, EB0-3), stores in front of reads inside a period:
      //        read  (slot -> QB):  t5 A0->0, t6 A1->1, t7 A2->2, t8 A3->0, t17 B0->1, t18 B1->2, t19 B2->0, t0' B3->1
      //        store (8 units):     t6 A0a, t7 A0b, t8 A1a, t9 A1b, t10 A2a, t11 A2b, t12 A3a, t13 A3b,
      //                             t18 B0a, t19 B0b, t0' B1a, t1' B1b, t2' B2a, t3' B2b, t4' B3a, t5' B3b
      //      (every quarter is read one period after it was staged and is out of its buffer before the buffer's next read)
      u32x4 q2[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) q2[u] = u32x4{0u, 0u, 0u, 0u};
      const unsigned rd0 = stg_rd, rd1 = stg_rd + W2_SLOT;
      auto rd_q2 = [&](unsigned addr) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 16; ++u) q2[u] = *reinterpret_cast<const u32x4*>(lds + addr + (u >> 2) * 4096 + ((u >> 1) & 1) * 2048 + (u & 1) * 512);
      };
      auto st_q2 = [&](auto h_tag, int quarter, const int (&vv)[4], i32x4 rs_o) __attribute__((always_inline)) {
        constexpr int Hh = decltype(h_tag)::value;
#pragma unroll
        for (int u = 8 * Hh; u < 8 * Hh + 8; ++u) {
          const int cw = u >> 2, n = (u >> 1) & 1, e = u & 1;
          w2_store16(q2[u], vv[(cw & 1) * 2 + n], rs_o, (unsigned)((cw >> 1) * 8 + quarter * 2 + e) * (unsigned)plane16);
        }
      };
      auto lb = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this period's staging reads are in registers before the slot is written again)
        W2_LBARRIER();
      };
      using H0 = W2IC<0>;
      using H1 = W2IC<1>;
      for (;;) {
        // t0 (P0)
        st_q2(H0{}, 1, vo_prev, rs_prev);            // B1a
        w2_rd16_a<64>(rd1);                          // B3 -> QB1
        next_tile();
        lb();
        st_q2(H1{}, 1, vo_prev, rs_prev); lb();      // t1: B1b
        w2_st8_a<0, 0>(vo_prev, rs_prev, (unsigned)(0 * 8 + 2 * 2) * (unsigned)plane16, (unsigned)plane16); lb();   // t2: B2a
        w2_st8_a<0, 1>(vo_prev, rs_prev, (unsigned)(1 * 8 + 2 * 2) * (unsigned)plane16, (unsigned)plane16); lb();   // t3: B2b
        w2_st8_a<64, 0>(vo_prev, rs_prev, (unsigned)(0 * 8 + 3 * 2) * (unsigned)plane16, (unsigned)plane16); lb();  // t4 (EA0): B3a
        w2_st8_a<64, 1>(vo_prev, rs_prev, (unsigned)(1 * 8 + 3 * 2) * (unsigned)plane16, (unsigned)plane16);        // t5 (EA1): B3b
        w2_rd16_a<0>(rd0); lb();                                                                                     //           A0 -> QB0
        w2_st8_a<0, 0>(vo[0], rs_out, (unsigned)(0 * 8 + 0 * 2) * (unsigned)plane16, (unsigned)plane16);            // t6 (EA2): A0a
        w2_rd16_a<64>(rd1); lb();                                                                                    //           A1 -> QB1
        w2_st8_a<0, 1>(vo[0], rs_out, (unsigned)(1 * 8 + 0 * 2) * (unsigned)plane16, (unsigned)plane16);            // t7 (EA3): A0b
        rd_q2(rd0); lb();                                                                                            //           A2 -> QB2
        w2_st8_a<64, 0>(vo[0], rs_out, (unsigned)(0 * 8 + 1 * 2) * (unsigned)plane16, (unsigned)plane16);           // t8 (Q0): A1a
        w2_rd16_a<0>(rd1); lb();                                                                                     //          A3 -> QB0
        w2_st8_a<64, 1>(vo[0], rs_out, (unsigned)(1 * 8 + 1 * 2) * (unsigned)plane16, (unsigned)plane16); lb();     // t9: A1b
        st_q2(H0{}, 2, vo[0], rs_out); lb();                                                                         // t10: A2a
        st_q2(H1{}, 2, vo[0], rs_out); lb();                                                                         // t11: A2b
        w2_st8_a<0, 0>(vo[0], rs_out, (unsigned)(0 * 8 + 3 * 2) * (unsigned)plane16, (unsigned)plane16); lb();      // t12: A3a
        w2_st8_a<0, 1>(vo[0], rs_out, (unsigned)(1 * 8 + 3 * 2) * (unsigned)plane16, (unsigned)plane16); lb();      // t13: A3b
        lb();                                                                                                        // t14
        lb();                                                                                                        // t15
        lb();                                                                                                        // t16 (EB0)
        w2_rd16_a<64>(rd0); lb();                                                                                    // t17 (EB1): B0 -> QB1
        w2_st8_a<64, 0>(vo[1], rs_out, (unsigned)(0 * 8 + 0 * 2) * (unsigned)plane16, (unsigned)plane16);           // t18 (EB2): B0a
        rd_q2(rd1); lb();                                                                                            //            B1 -> QB2
        w2_st8_a<64, 1>(vo[1], rs_out, (unsigned)(1 * 8 + 0 * 2) * (unsigned)plane16, (unsigned)plane16);           // t19 (EB3): B0b
        w2_rd16_a<0>(rd0); lb();                                                                                     //            B2 -> QB0
        advance();
        if (!has_next) break;
      }
      // the last tile's pass B: B1 (QB2), B2 (QB0) and B3 (slot 1)
      st_q2(H0{}, 1, vo_prev, rs_prev);
      st_q2(H1{}, 1, vo_prev, rs_prev);
      w2_rd16_a<64>(rd1);
      w2_st8_a<0, 0>(vo_prev, rs_prev, (unsigned)(0 * 8 + 2 * 2) * (unsigned)plane16, (unsigned)plane16);
      w2_st8_a<0, 1>(vo_prev, rs_prev, (unsigned)(1 * 8 + 2 * 2) * (unsigned)plane16, (unsigned)plane16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      w2_st8_a<64, 0>(vo_prev, rs_prev, (unsigned)(0 * 8 + 3 * 2) * (unsigned)plane16, (unsigned)plane16);
      w2_st8_a<64, 1>(vo_prev, rs_prev, (unsigned)(1 * 8 + 3 * 2) * (unsigned)plane16, (unsigned)plane16);
      return;
    }
    // ---- any other K: every staged quarter goes out in the period after its staging
    bool pend = false;
    for (;;) {
      for (int k = 0; k < np; ++k) {           // pass A
        if (k == 0) {
          if (pend) store_quarter(1, 3, vo_prev, rs_prev);
          pend = false;
          next_tile();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W2_LBARRIER();
      }
      for (int e = 0; e < 4; ++e) {            // epilogue A
        if (e >= 1) store_quarter((e - 1) & 1, e - 1, vo[0], rs_out);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the staging reads are in registers before the slot is written again)
        W2_LBARRIER();
      }
      for (int c = 0; c < nck; ++c) {          // pass B
        if (c == 0) store_quarter(1, 3, vo[0], rs_out);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W2_LBARRIER();
      }
      for (int e = 0; e < 4; ++e) {            // epilogue B
        if (e >= 1) store_quarter((e - 1) & 1, e - 1, vo[1], rs_out);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W2_LBARRIER();
      }
      pend = true;
      advance();
      if (!has_next) break;
    }
    store_quarter(1, 3, vo_prev, rs_prev);
    return;
  }

  // =======================================================================================================================
  // COMPUTE
  // =======================================================================================================================
  const int lane = w2_lane(), l31 = lane & 31, half = lane >> 5;
  const unsigned a_b0 = (unsigned)(half * 128 + mh * 64 + l31) * 16u;   // weight unit [channel block half][channel] of row block mb = 0
  const unsigned stg_wr = (unsigned)(W2_STG + tw * 4096 + l31 * 16 + half * 8);
  unsigned pbr[2], lw16;
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
  // a pass: `nper` periods from dy slot 0 on; the weight buffer alternates (wpar)
  auto run_pass = [&](auto pb_tag, int nper, int& wpar) __attribute__((always_inline)) {
    constexpr bool PB = decltype(pb_tag)::value;
    using TT = std::true_type;
    using FT = std::false_type;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      pb0[n] = (unsigned)W2_DY + pbr[n];
      pb1[n] = pb0[n] + lw16;
    }
    w2_kfirst<w2_grp(PB, 1) * 4096>(A[0][0], A[0][1], A[1][0], A[1][1], B[0][0], B[0][1], (unsigned)(wpar * W2_WB) + a_b0, pb0[0], pb0[1]);
    for (int k = 0; k < nper; ++k) {
      const unsigned pa = (unsigned)(wpar * W2_WB) + a_b0, pan = (unsigned)((wpar ^ 1) * W2_WB) + a_b0;
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
      wpar ^= 1;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        pb0[n] += PB ? W2_DYS : 2 * W2_DYS;
        pb1[n] += PB ? W2_DYS : 2 * W2_DYS;
      }
    }
  };
  // a quarter of a pass's epilogue: channel group (mb, gp) of the 8 accumulator tiles -> staging slot `sl` (4 units of 1 KB)
  [[maybe_unused]] unsigned long long t_bw = 0, t_ebw = 0;   // (diagnostic build) ticks inside the barrier statements of the MFMA periods / the epilogue quarters
  unsigned sg[2][2] = {{0u, 0u}, {0u, 0u}};   // this lane's sign bytes of the pass: [n][px], byte mb * 2 + gp
  auto epi_quarter = [&](auto q_tag, int sl) __attribute__((always_inline)) {
    constexpr int Q = decltype(q_tag)::value, MB = Q >> 1, GP = Q & 1;
    if constexpr ((W2_ABL & 8) != 0) return;
    const float* es = reinterpret_cast<const float*>(lds + W2_ES) + mh * 64 + MB * 32 + 16 * GP + 4 * half;
    const float4 sA = *reinterpret_cast<const float4*>(es), sB = *reinterpret_cast<const float4*>(es + 8);
    const float sa[8] = {sA.x, sA.y, sA.z, sA.w, sB.x, sB.y, sB.z, sB.w};
    float sl_[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sl_[i] = LEAKY_SLOPE * sa[i];
    unsigned char* dst0 = lds + stg_wr + sl * W2_SLOT;
    w2_static_for([&](auto u_tag) __attribute__((always_inline)) {   // the quarter's four units (n, px)
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
      unsigned char* dst = dst0 + (n * 2 + px) * 1024;
      *reinterpret_cast<u32x2*>(dst) = u32x2{w.x, w.y};
      *reinterpret_cast<u32x2*>(dst + 512) = u32x2{w.z, w.w};
    }, std::make_integer_sequence<int, 4>{});
  };
  auto read_signs = [&]() __attribute__((always_inline)) {
    if constexpr (SIGNS) {
      const unsigned* sp = reinterpret_cast<const unsigned*>(lds + W2_SG) + tw * 256 + lane;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int px = 0; px < 2; ++px) sg[n][px] = sp[(n * 2 + px) * 64];
    }
  };
  auto epilogue = [&]() __attribute__((always_inline)) {
    read_signs();
    w2_static_for([&](auto q_tag) __attribute__((always_inline)) {
      epi_quarter(q_tag, decltype(q_tag)::value & 1);
      [[maybe_unused]] const unsigned long long b0 = W2_STAMP();
      w2_barrier_lgkm();
      t_ebw += W2_STAMP() - b0;
    }, std::make_integer_sequence<int, 4>{});
  };

  w2_barrier();   // (#1)
  int wpar = 0;
  [[maybe_unused]] unsigned long long t_a = 0, t_ea = 0, t_b = 0, t_eb = 0;
  for (;;) {
    [[maybe_unused]] const unsigned long long s0 = W2_STAMP();
    run_pass(std::false_type{}, np, wpar);
    [[maybe_unused]] const unsigned long long s1 = W2_STAMP();
    epilogue();
    [[maybe_unused]] const unsigned long long s2 = W2_STAMP();
    run_pass(std::true_type{}, nck, wpar);
    [[maybe_unused]] const unsigned long long s3 = W2_STAMP();
    epilogue();
    [[maybe_unused]] const unsigned long long s4 = W2_STAMP();
    t_a += s1 - s0; t_ea += s2 - s1; t_b += s3 - s2; t_eb += s4 - s3;
    unsigned kn = k_ord + 1;
    const bool has_next = find_tile(kn, T);
    k_ord = kn;
    if (!has_next) break;
    decode_pix(T);
  }
#ifdef YOGO_DIAG
  if (p.stamps && tw == 0 && lane == 0) {
    unsigned long long* d = p.stamps + (size_t)blockIdx.x * 16;
    d[0] = t_start; d[1] = __builtin_amdgcn_s_memtime(); d[2] = t_a; d[3] = t_ea; d[4] = t_b; d[5] = t_eb; d[6] = k_ord; d[7] = t_bw; d[8] = t_ebw;
  }
#endif
}

// =========================================================================================================
// host side: eligibility, tiling, launch
// =========================================================================================================
bool conv_bf16_ws2_eligible(int K, int M, int OH, int OW, int B) {
  if (M != 128 || K < 32 || K > 128 || (K % 32) != 0) return false;   // nck = K / 16 even, <= 8 dy slots
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
    auto rows_of = [&](int bw) __attribute__((always_inline)) { return min(QH, 1 + cdiv(W2_PT - 1, bw)) + 1; };
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
