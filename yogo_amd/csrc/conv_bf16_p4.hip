// Persistent 4-wavefront bf16 convolution for the stride-1 3x3 layers with 128 output channels (forward of layers 3/5/6 and
// the data gradients of layers 5/6 of base_model: yogo/model_defns.py:49-65 and their autograd) -- the dominant kernel of the
// training step.  Same arithmetic, accumulation order and epilogue formula as conv_bf16_kernel<4,2,8,...,PP> (conv_bf16.hip):
// outputs are bit-identical (tests/test_gpu_p4.py).
//
// Structure (cdna_hip_programming.md "4-wave, one-wave-per-SIMD, persistent structure", MI355X_MICROARCH.md register files):
//   * ONE workgroup of 4 wavefronts per CU, one wavefront per SIMD, each with the whole 512-register file: 192 accumulator
//     registers a[0:191] are OWNED BY THE ASM STATEMENTS below (named literally, listed as clobbers so the kernel descriptor
//     allocates them; hipcc never touches AGPRs here -- audit: build.sh checks .vgpr_spill_count 0 and no compiler v_accvgpr),
//     the arch VGPRs hold two operand sets, the 96 packed output registers of the PREVIOUS tile and addresses;
//   * a wavefront computes 128 channels x 96 pixels (4 x 3 accumulator tiles of 32x32): 7 ds_read_b128 per 12 MFMAs
//     (conv_bf16_kernel's 8-wavefront tiles: 6 per 8), all of them issued in the gaps of the wavefront's own MFMA stream;
//   * the workgroup is PERSISTENT: it walks tiles (image, band, 384 pixels) grid-strided inside its XCD's run of tiles.  The
//     16-channel chunks of the K loop stream through two LDS buffers by LDS-DMA exactly as in the ping-pong kernel, but the
//     stream does not stop at a tile seam: chunk 0 of the next tile is requested during the last chunk of this one;
//   * the EPILOGUE of tile t runs under the MFMAs of tile t + 1: at the seam the accumulators are read out, bias / channel
//     scale / LeakyReLU / bf16 conversion / half-wave exchange are applied and the 24 16-byte units per lane are parked in
//     VGPRs; their stores (1 KB per wavefront instruction) are issued one per K step of the next tile's first three chunks.
//     The CU's store path takes ~14 B/clk (MI355X_MICROARCH.md "store-ISSUE-bound"): 96 KB per tile is 7 k cycles when nothing
//     else runs, and nothing when spread over 27 steps of 384 cycles.
// One barrier per chunk (inside step 8), counted in front of it: vmcnt(0) -- the last DMA piece of a chunk is issued in step 4.
#include "conv_bf16_p4.h"
#include <type_traits>
#include <utility>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define P4_ACC_CLOBBER "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191"

#define P4_XS(x) #x
#define P4_S(x) P4_XS(x)
// accumulator tile (mb, n) = a[16 * (3 mb + n) : +15]
#define P4_MFMA(TILE, AOP, BOP) \
  "v_mfma_f32_32x32x16_bf16 a[16*" #TILE ":16*" #TILE "+15], %[" #AOP "], %[" #BOP "], a[16*" #TILE ":16*" #TILE "+15]\n\t"
#define P4_MFMA0(TILE, AOP, BOP) \
  "v_mfma_f32_32x32x16_bf16 a[16*" #TILE ":16*" #TILE "+15], %[" #AOP "], %[" #BOP "], 0\n\t"
// operand reads of tap T1 (kernel column KX1): weights at pa + T1 * 4 KB + mb * 512 B, input at the row base + KX1 * 16 B
// (T1 / KX1 name "n" operands of the statement, or are literal numbers)
#define P4_RDA(DST, MB, T1) "ds_read_b128 %[" #DST "], %[pa] offset:4096*" T1 "+512*" #MB "\n\t"
#define P4_RDB(DST, SRC, KX1) "ds_read_b128 %[" #DST "], %[" #SRC "] offset:16*" KX1 "\n\t"

namespace {

__device__ __forceinline__ int p4_udivm(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }   // n / d, m = ceil(2^32 / d), d > 1
__device__ __forceinline__ int p4_udivm1(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }
__device__ __forceinline__ i32x4 p4_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}

// one LDS-DMA piece (64 lanes x 16 bytes -> LDS bytes [lds, lds + 1024))
__device__ __forceinline__ void p4_dma1(i32x4 rs, unsigned lds, int voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" ::"v"(voff), "s"(lds), "s"(rs), "s"(soff) : "memory");
}
// 64 lanes x 4 bytes -> LDS bytes [lds, lds + 256)
__device__ __forceinline__ void p4_dma_dword(i32x4 rs, unsigned lds, int voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dword %0, %2, %3 offen lds" ::"v"(voff), "s"(lds), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void p4_store16(u32x4 data, int voff, i32x4 rs, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(data), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

template <class F, int... I>
__device__ __forceinline__ void p4_static_for(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

// ---- the asm statements of a K step (tap) ------------------------------------------------------------------------------
// first half: accumulator rows mb = 0, 1 (6 MFMAs) with the 7 operand reads of the NEXT step in their gaps.  The reads are
// retired by the lgkmcnt(0) that ends the second half (p4_sb), which names their destinations "+v".
template <bool ZERO, int T1, int KX1>
__device__ __forceinline__ void p4_sa(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, const u32x4& b2, u32x4& an0, u32x4& an1,
                                      u32x4& an2, u32x4& an3, u32x4& bn0, u32x4& bn1, u32x4& bn2, unsigned pa, unsigned pb0, unsigned pb1, unsigned pb2) {
#define P4_SA_BODY(M)                                                              \
  M(0, a0, b0) P4_RDA(an0, 0, "%[t1]") M(1, a0, b1) P4_RDB(bn0, pb0, "%[kx1]")               \
  M(2, a0, b2) P4_RDA(an1, 1, "%[t1]") P4_RDB(bn1, pb1, "%[kx1]") M(3, a1, b0) P4_RDA(an2, 2, "%[t1]") P4_RDB(bn2, pb2, "%[kx1]") \
  M(4, a1, b1) P4_RDA(an3, 3, "%[t1]") M(5, a1, b2)
#define P4_SA_OPS                                                                                                                    \
  : [an0] "=&v"(an0), [an1] "=&v"(an1), [an2] "=&v"(an2), [an3] "=&v"(an3), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1), [bn2] "=&v"(bn2)       \
  : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1), [pb2] "v"(pb2), \
    [t1] "n"(T1), [kx1] "n"(KX1)                                                                                                     \
  : "memory", P4_ACC_CLOBBER
  if constexpr (ZERO) asm volatile(P4_SA_BODY(P4_MFMA0) P4_SA_OPS);
  else asm volatile(P4_SA_BODY(P4_MFMA) P4_SA_OPS);
#undef P4_SA_BODY
#undef P4_SA_OPS
}

// second half: rows mb = 2, 3 with up to three LDS-DMA pieces of ONE descriptor (LDS destinations lds, lds + 4 KB, lds + 8 KB)
// and one 16-byte store of a parked output unit in the gaps; ends by retiring the first half's operand reads.
template <bool ZERO, bool DMA, bool STORE>
__device__ __forceinline__ void p4_sb(const u32x4& a2, const u32x4& a3, const u32x4& b0, const u32x4& b1, const u32x4& b2, u32x4& an0, u32x4& an1,
                                      u32x4& an2, u32x4& an3, u32x4& bn0, u32x4& bn1, u32x4& bn2, i32x4 rs, int v0, int v1, int v2, unsigned s0,
                                      unsigned s1, unsigned s2, unsigned lds, const u32x4& sdata, int svo, i32x4 srs, unsigned sso) {
#define P4_DMA_A "s_mov_b32 m0, %[lds]\n\ts_nop 4\n\tbuffer_load_dwordx4 %[v0], %[rs], %[s0] offen lds\n\t"
#define P4_DMA_B "s_add_u32 m0, m0, 4096\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v1], %[rs], %[s1] offen lds\n\t"
#define P4_DMA_C "s_add_u32 m0, m0, 4096\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v2], %[rs], %[s2] offen lds\n\t"
#define P4_ST "buffer_store_dwordx4 %[sd], %[svo], %[srs], %[sso] offen\n\t"
#define P4_SB_OPS                                                                                                                          \
  : [an0] "+v"(an0), [an1] "+v"(an1), [an2] "+v"(an2), [an3] "+v"(an3), [bn0] "+v"(bn0), [bn1] "+v"(bn1), [bn2] "+v"(bn2)                    \
  : [a2] "v"(a2), [a3] "v"(a3), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [rs] "s"(rs), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [s0] "s"(s0), \
    [s1] "s"(s1), [s2] "s"(s2), [lds] "s"(lds), [sd] "v"(sdata), [svo] "v"(svo), [srs] "s"(srs), [sso] "s"(sso)                              \
  : "memory", "scc", P4_ACC_CLOBBER
#define P4_SB_BODY(M, DA, DB, DC, ST)                                                                              \
  M(6, a2, b0) DA M(7, a2, b1) DB M(8, a2, b2) DC M(9, a3, b0) ST M(10, a3, b1) M(11, a3, b2) "s_waitcnt lgkmcnt(0)"
  if constexpr (ZERO) {
    if constexpr (DMA && STORE) asm volatile(P4_SB_BODY(P4_MFMA0, P4_DMA_A, P4_DMA_B, P4_DMA_C, P4_ST) P4_SB_OPS);
    else if constexpr (DMA) asm volatile(P4_SB_BODY(P4_MFMA0, P4_DMA_A, P4_DMA_B, P4_DMA_C, "") P4_SB_OPS);
    else if constexpr (STORE) asm volatile(P4_SB_BODY(P4_MFMA0, "", "", "", P4_ST) P4_SB_OPS);
    else asm volatile(P4_SB_BODY(P4_MFMA0, "", "", "", "") P4_SB_OPS);
  } else {
    if constexpr (DMA && STORE) asm volatile(P4_SB_BODY(P4_MFMA, P4_DMA_A, P4_DMA_B, P4_DMA_C, P4_ST) P4_SB_OPS);
    else if constexpr (DMA) asm volatile(P4_SB_BODY(P4_MFMA, P4_DMA_A, P4_DMA_B, P4_DMA_C, "") P4_SB_OPS);
    else if constexpr (STORE) asm volatile(P4_SB_BODY(P4_MFMA, "", "", "", P4_ST) P4_SB_OPS);
    else asm volatile(P4_SB_BODY(P4_MFMA, "", "", "", "") P4_SB_OPS);
  }
#undef P4_SB_BODY
#undef P4_SB_OPS
}

// step 8, first part: row mb = 0, then "my pieces of the next chunk have landed" + the chunk's barrier
__device__ __forceinline__ void p4_x8(const u32x4& a0, const u32x4& b0, const u32x4& b1, const u32x4& b2) {
  asm volatile(P4_MFMA(0, a0, b0) P4_MFMA(1, a0, b1) P4_MFMA(2, a0, b2) "s_waitcnt vmcnt(0)\n\ts_barrier"
               :
               : [a0] "v"(a0), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2)
               : "memory", P4_ACC_CLOBBER);
}
// step 8, second part: rows mb = 1..3 with the operand reads of the next chunk's step 0 (other buffer / next tile) up front
template <bool STORE>
__device__ __forceinline__ void p4_y8(const u32x4& a1, const u32x4& a2, const u32x4& a3, const u32x4& b0, const u32x4& b1, const u32x4& b2, u32x4& an0,
                                      u32x4& an1, u32x4& an2, u32x4& an3, u32x4& bn0, u32x4& bn1, u32x4& bn2, unsigned pa, unsigned pb0, unsigned pb1,
                                      unsigned pb2, const u32x4& sdata, int svo, i32x4 srs, unsigned sso) {
#define P4_Y8_OPS                                                                                                                    \
  : [an0] "=&v"(an0), [an1] "=&v"(an1), [an2] "=&v"(an2), [an3] "=&v"(an3), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1), [bn2] "=&v"(bn2)       \
  : [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1),   \
    [pb2] "v"(pb2), [sd] "v"(sdata), [svo] "v"(svo), [srs] "s"(srs), [sso] "s"(sso)                                                  \
  : "memory", P4_ACC_CLOBBER
#define P4_Y8_BODY(ST)                                                                                                        \
  P4_MFMA(3, a1, b0) P4_RDA(an0, 0, "0") P4_RDB(bn0, pb0, "0") P4_MFMA(4, a1, b1) P4_RDA(an1, 1, "0") P4_RDB(bn1, pb1, "0")           \
  P4_MFMA(5, a1, b2) P4_RDA(an2, 2, "0") P4_RDB(bn2, pb2, "0") P4_MFMA(6, a2, b0) P4_RDA(an3, 3, "0") P4_MFMA(7, a2, b1) ST         \
  P4_MFMA(8, a2, b2) P4_MFMA(9, a3, b0) P4_MFMA(10, a3, b1) P4_MFMA(11, a3, b2) "s_waitcnt lgkmcnt(0)"
  if constexpr (STORE) asm volatile(P4_Y8_BODY(P4_ST) P4_Y8_OPS);
  else asm volatile(P4_Y8_BODY("") P4_Y8_OPS);
#undef P4_Y8_BODY
#undef P4_Y8_OPS
}
// operand reads of a chunk's step 0 without MFMAs (first tile of a workgroup)
__device__ __forceinline__ void p4_prefetch(u32x4& an0, u32x4& an1, u32x4& an2, u32x4& an3, u32x4& bn0, u32x4& bn1, u32x4& bn2, unsigned pa, unsigned pb0,
                                            unsigned pb1, unsigned pb2) {
  asm volatile(P4_RDA(an0, 0, "0") P4_RDA(an1, 1, "0") P4_RDA(an2, 2, "0") P4_RDA(an3, 3, "0") P4_RDB(bn0, pb0, "0") P4_RDB(bn1, pb1, "0") P4_RDB(bn2, pb2, "0")
               "s_waitcnt lgkmcnt(0)"
               : [an0] "=&v"(an0), [an1] "=&v"(an1), [an2] "=&v"(an2), [an3] "=&v"(an3), [bn0] "=&v"(bn0), [bn1] "=&v"(bn1), [bn2] "=&v"(bn2)
               : [pa] "v"(pa), [pb0] "v"(pb0), [pb1] "v"(pb1), [pb2] "v"(pb2)
               : "memory");
}

template <int IDX>
__device__ __forceinline__ float p4_acc_read() {
  float r;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(r) : "n"(IDX));
  return r;
}

#ifdef YOGO_DIAG
#define P4_DBG(BIT) (p.dbg & (BIT))
#define P4_STAMP() (p.stamps ? __builtin_amdgcn_s_memtime() : 0ull)
#else
#define P4_DBG(BIT) 0
#define P4_STAMP() 0ull
#endif

}  // namespace

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_bf16_p4_kernel(const ConvP4Params p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
  constexpr unsigned OOB = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kbw = wave >> 1, colh = wave & 1;   // this wavefront's weight pieces: channel block of the chunk, column half
  [[maybe_unused]] const unsigned long long t_start = P4_STAMP();
  float* ldsf = reinterpret_cast<float*>(smem4);

  const int OH = p.IH, OW = p.IW;
  const int plane = OH * OW, plane16 = plane * 16;
  const int rowb = p.IW * 16, kcb = p.IH * p.IW * 16;
  const unsigned ibytes = (unsigned)p.Kb * kcb, obytes = 16u * plane16, wbytes = 9u * p.Kb * 2048u;
  const unsigned so_i = 2u * kcb;            // bytes between the 16-channel chunks of an image
  const unsigned wstep = (unsigned)p.Kb * 2048u;   // bytes between the taps of the packed weights
  const i32x4 rs_w = p4_rsrc(p.wp, wbytes);
  const int lane16 = P4_DBG(4) ? (int)0x80000000u : lane * 16;   // (diagnostic bit 4: no weight DMA)
  const unsigned a_b0 = (unsigned)(P4_LDSW_OFF + half * 128 + l31) * 16u;

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
      const int b = p4_udivm1(widx, p.gx, p.m_gx);
      const int bx = widx - b * p.gx;
      const int cb = p4_udivm1(bx, p.tiles_per_band, p.m_tpb);
      const int tb = bx - cb * p.tiles_per_band;
      const int j0 = cb * p.TW;
      const int bw = min(p.TW, OW - j0);
      const int NPb = OH * bw;
      const int p0 = tb * P4_PT;
      if (p0 >= NPb) continue;
      t.b = b; t.j0 = j0; t.bw = bw; t.p0 = p0; t.p1 = min(p0 + P4_PT, NPb); t.lastband = cb == p.ncb - 1;
      return true;
    }
  };
  // per-lane part of a tile: DMA source offsets of the 6 input slots, LDS byte address of the operand row 0 of each pixel
  // group (buffer 0), output byte offsets; (uniform) the row pitch of the staged tile in bytes
  auto decode = [&](const TileS& t, int (&voff)[P4_NI], unsigned (&pbr)[P4_NW], int (&vo)[P4_NW], unsigned& lw16) {
    const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
    const int bw = t.bw;
    const int i_lo = p4_udivm1(t.p0, bw, m_bw), i_hi = p4_udivm1(t.p1 - 1, bw, m_bw);
    const int rows_in = i_hi - i_lo + 3;
    const int iy0 = i_lo - 1, ix0 = t.j0 - 1;
    const int lw = bw + 2;
    const int per_kb = rows_in * lw;
    const unsigned inv_lw = t.lastband ? p.m_lwl : p.m_lw;
    const unsigned inv_perkb = 0xFFFFFFFFu / (unsigned)per_kb + 1u;
    lw16 = (unsigned)lw * 16u;
#pragma unroll
    for (int n = 0; n < P4_NW; ++n) {
      const int pp = t.p0 + (wave * P4_NW + n) * 32 + l31;
      const bool pv = pp < t.p1;
      const int pc = pv ? pp : (t.p1 - 1);
      const int i = p4_udivm1(pc, bw, m_bw), j = pc - i * bw;
      pbr[n] = (unsigned)((i - i_lo) * lw + j + half * per_kb) * 16u;
      vo[n] = pv ? (i * OW + t.j0 + j) * 16 + half * plane16 : (int)OOB;
    }
    // element tid + i * 256 of the flattened [2][rows_in][lw] tile -> (channel block, row, column): slot 0 by division, the
    // others by stepping with two carries
    const int skc = p4_udivm(P4_NT, inv_perkb);
    const int srm = P4_NT - skc * per_kb;
    const int sr = p4_udivm(srm, inv_lw);
    const int sx = srm - sr * lw;
    int kc_ = p4_udivm(tid, inv_perkb);
    const int rm0 = tid - kc_ * per_kb;
    int r_ = p4_udivm(rm0, inv_lw);
    int x_ = rm0 - r_ * lw;
#pragma unroll
    for (int i = 0; i < P4_NI; ++i) {
      const int iy_ = iy0 + r_, ix_ = ix0 + x_;
      const bool ok = (kc_ < 2) && (iy_ >= 0) && (iy_ < p.IH) && (ix_ >= 0) && (ix_ < p.IW);
      voff[i] = (ok && !P4_DBG(4)) ? kc_ * kcb + iy_ * rowb + ix_ * 16 : (int)OOB;
      x_ += sx; r_ += sr; kc_ += skc;
      if (x_ >= lw) { x_ -= lw; ++r_; }
      if (r_ >= rows_in) { r_ -= rows_in; ++kc_; }
    }
  };

  unsigned k_ord = 0;
  TileS T{};
  if (!find_tile(k_ord, T)) return;

  // bias (the same for every tile) and a unit channel scale when there is none
  if (tid < 128) {
    ldsf[P4_EB / 4 + tid] = p.bias != nullptr ? p.bias[tid] : 0.f;
    if (p.chan_scale == nullptr) {
      ldsf[P4_ES / 4 + tid] = 1.f;
      ldsf[P4_ES / 4 + 128 + tid] = 1.f;
    }
  }
  __syncthreads();
  const bool has_scale = p.chan_scale != nullptr;
  const i32x4 rs_sc = p4_rsrc(p.chan_scale, has_scale ? (unsigned)p.B * 512u : 0u);
  auto issue_scale = [&](int b, int par) {   // [128] channel scale of image b -> es[par] (wavefronts 0 and 1, 64 floats each)
    if (has_scale && wave < 2)
      p4_dma_dword(rs_sc, (unsigned)__builtin_amdgcn_readfirstlane(P4_ES + par * 512 + wave * 256), lane * 4,
                   (unsigned)__builtin_amdgcn_readfirstlane((b * 128 + wave * 64) * 4));
  };

  int dvoff[P4_NI];          // input slot offsets of the tile whose chunks are being REQUESTED
  unsigned pbr[P4_NW];       // operand row-0 addresses of the tile being COMPUTED
  int vo[P4_NW];             // output offsets of the tile being computed
  unsigned lw16;
  decode(T, dvoff, pbr, vo, lw16);
  i32x4 rs_in = p4_rsrc(reinterpret_cast<const unsigned char*>(p.in) + (size_t)T.b * ibytes, ibytes);
  int tpar = 0;              // parity of the tile being computed (channel-scale buffer)
  issue_scale(T.b, 0);
  // chunk 0 of the first tile -> buffer 0
  if (!P4_DBG(4)) {
#pragma unroll
    for (int i = 0; i < P4_NI; ++i) p4_dma1(rs_in, (unsigned)((i * P4_NT + wave * 64) * 16), dvoff[i], 0u);
#pragma unroll
    for (int j = 0; j < 9; ++j)
      p4_dma1(rs_w, (unsigned)((P4_LDSW_OFF + (j * 2 + kbw) * 128 + colh * 64) * 16), lane16, (unsigned)j * wstep + (unsigned)((kbw * 128 + colh * 64) * 16));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  u32x4 A0[4], B0[3], A1[4], B1[3];   // the two operand sets
  p4_prefetch(A0[0], A0[1], A0[2], A0[3], B0[0], B0[1], B0[2], a_b0, pbr[0], pbr[1], pbr[2]);

  u32x4 hold[24];            // parked output units of the PREVIOUS tile: hold[(2 mb + gp) * 3 + n]
#pragma unroll
  for (int i = 0; i < 24; ++i) hold[i] = u32x4{0u, 0u, 0u, 0u};
  int vo_prev[P4_NW] = {(int)OOB, (int)OOB, (int)OOB};
  i32x4 rs_out_prev = p4_rsrc(p.out, 0u);

  // state of the request stream: the chunk that the steps of the chunk being computed request
  unsigned nx_soff_in = 0, nx_wbase = 0;     // scalar offsets of that chunk (input descriptor, weight descriptor)
  unsigned pbn[P4_NW] = {0u, 0u, 0u};        // operand row-0 addresses of the NEXT tile (valid during the last chunk)
  int von[P4_NW] = {(int)OOB, (int)OOB, (int)OOB};
  unsigned lw16n = 0;
  TileS Tn{};
  bool has_next = false;
  const bool leaky = p.act == ACT_LEAKY;
  const bool write_signs = p.signs != nullptr;
  [[maybe_unused]] unsigned long long t_chunks = 0, t_seam = 0, t_prep = 0, t_x8 = 0;

  // one 16-channel chunk: 9 K steps.  P = parity of the chunk (LDS buffer AND operand set of step 0); FIRST: the tile's first
  // chunk (step 0 starts the accumulators from zero); STC = 0..2: the steps carry the stores of hold[8 STC .. 8 STC + 7]
  auto chunk = [&](auto first_tag, auto stc_tag, auto p_tag, unsigned pa_nx, const unsigned (&pb_nx)[P4_NW]) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int STC = decltype(stc_tag)::value;
    constexpr int P = decltype(p_tag)::value;
    const unsigned pa = a_b0 + P * P4_BUFB;
    const unsigned ldsn = (1 - P) * P4_BUFB;   // the buffer the requests go to
    auto one = [&](auto s_tag) {
      constexpr int S = decltype(s_tag)::value;
      constexpr bool EVEN = ((P + S) & 1) == 0;   // operand set of this step: 0 when even
      u32x4(&Ac)[4] = EVEN ? A0 : A1;
      u32x4(&Bc)[3] = EVEN ? B0 : B1;
      u32x4(&An)[4] = EVEN ? A1 : A0;
      u32x4(&Bn)[3] = EVEN ? B1 : B0;
      // the store this step carries: steps 0..6 and 8 of chunks 0..2
      constexpr bool STORE = STC >= 0 && S != 7;
      constexpr int HI = STC >= 0 ? STC * 8 + (S < 7 ? S : 7) : 0;
      constexpr int HQ = HI / 3, HN = HI % 3;           // (2 mb + gp, n)
      const unsigned sso = (unsigned)(2 * HQ) * (unsigned)plane16;   // channel block 4 mb + 2 gp = 2 (2 mb + gp)
      if constexpr (S < 8) {
        constexpr int T1 = S + 1, KY1 = T1 / 3, KX1 = T1 % 3;
        const unsigned rowo = (unsigned)KY1 * lw16 + P * P4_BUFB;
        p4_sa<FIRST && S == 0, T1, KX1>(Ac[0], Ac[1], Bc[0], Bc[1], Bc[2], An[0], An[1], An[2], An[3], Bn[0], Bn[1], Bn[2], pa, pbr[0] + rowo,
                                         pbr[1] + rowo, pbr[2] + rowo);
        constexpr bool DMA = S < 5;
        if constexpr (S < 2) {   // input slots 3 S .. 3 S + 2
          p4_sb<FIRST && S == 0, DMA, STORE>(Ac[2], Ac[3], Bc[0], Bc[1], Bc[2], An[0], An[1], An[2], An[3], Bn[0], Bn[1], Bn[2], rs_in, dvoff[3 * S],
                                            dvoff[3 * S + 1], dvoff[3 * S + 2], nx_soff_in, nx_soff_in, nx_soff_in,
                                            ldsn + (unsigned)((3 * S * P4_NT + wave * 64) * 16), hold[HI], vo_prev[HN], rs_out_prev, sso);
        } else {                 // weight slices of taps 3 (S - 2) .. + 2 (nothing from step 5 on)
          constexpr int J0 = S < 5 ? 3 * (S - 2) : 0;
          const unsigned sw = nx_wbase + (unsigned)J0 * wstep;
          p4_sb<false, DMA, STORE>(Ac[2], Ac[3], Bc[0], Bc[1], Bc[2], An[0], An[1], An[2], An[3], Bn[0], Bn[1], Bn[2], rs_w, lane16, lane16, lane16, sw,
                                   sw + wstep, sw + 2 * wstep, ldsn + (unsigned)((P4_LDSW_OFF + (J0 * 2 + kbw) * 128 + colh * 64) * 16), hold[HI],
                                   vo_prev[HN], rs_out_prev, sso);
        }
      } else {
        [[maybe_unused]] const unsigned long long tx0 = P4_STAMP();
        p4_x8(Ac[0], Bc[0], Bc[1], Bc[2]);
        t_x8 += P4_STAMP() - tx0;
        p4_y8<STORE>(Ac[1], Ac[2], Ac[3], Bc[0], Bc[1], Bc[2], An[0], An[1], An[2], An[3], Bn[0], Bn[1], Bn[2], pa_nx, pb_nx[0], pb_nx[1], pb_nx[2],
                     hold[HI], vo_prev[HN], rs_out_prev, sso);
      }
    };
    p4_static_for(one, std::make_integer_sequence<int, 9>{});
  };

  using TT = std::true_type;
  using FT = std::false_type;
  for (;;) {
    // ---- the chunks of tile T.  Requests: chunk c + 1 of T, or (during the last chunk) chunk 0 of the next tile.
    [[maybe_unused]] const unsigned long long tc0 = P4_STAMP();
    const int nck = p.nchunk;
    auto same_tile_next = [&](int c, unsigned (&pb_nx)[P4_NW], int P) {   // requests / step-8 prefetch for chunk c + 1 of this tile
      nx_soff_in = (unsigned)(c + 1) * so_i;
      nx_wbase = (unsigned)(((2 * (c + 1) + kbw) * 128 + colh * 64) * 16);
#pragma unroll
      for (int n = 0; n < P4_NW; ++n) pb_nx[n] = pbr[n] + (unsigned)(1 - P) * P4_BUFB;
    };
    unsigned pbx[P4_NW];
    same_tile_next(0, pbx, 0);
    chunk(TT{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, a_b0 + P4_BUFB, pbx);
    same_tile_next(1, pbx, 1);
    chunk(FT{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, a_b0, pbx);
    same_tile_next(2, pbx, 0);
    chunk(FT{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, a_b0 + P4_BUFB, pbx);
    for (int c = 3; c < nck; c += 2) {
      // chunk c (parity 1); the last chunk of the tile has this parity (nchunk is even)
      if (c == nck - 1) {
        [[maybe_unused]] const unsigned long long tp0 = P4_STAMP();
        unsigned kn = k_ord + 1;
        has_next = find_tile(kn, Tn);
        k_ord = kn;
        if (has_next) {
          decode(Tn, dvoff, pbn, von, lw16n);
          rs_in = p4_rsrc(reinterpret_cast<const unsigned char*>(p.in) + (size_t)Tn.b * ibytes, ibytes);
          issue_scale(Tn.b, tpar ^ 1);
        } else {
#pragma unroll
          for (int i = 0; i < P4_NI; ++i) dvoff[i] = (int)OOB;
#pragma unroll
          for (int n = 0; n < P4_NW; ++n) { pbn[n] = pbr[n]; von[n] = (int)OOB; }
          lw16n = lw16;
        }
        nx_soff_in = 0u;
        nx_wbase = (unsigned)((kbw * 128 + colh * 64) * 16);
#pragma unroll
        for (int n = 0; n < P4_NW; ++n) pbx[n] = pbn[n];
        t_prep += P4_STAMP() - tp0;
      } else {
        same_tile_next(c, pbx, 1);
      }
      chunk(FT{}, std::integral_constant<int, -1>{}, std::integral_constant<int, 1>{}, a_b0, pbx);
      if (c + 1 < nck) {
        same_tile_next(c + 1, pbx, 0);
        chunk(FT{}, std::integral_constant<int, -1>{}, std::integral_constant<int, 0>{}, a_b0 + P4_BUFB, pbx);
      }
    }
    [[maybe_unused]] const unsigned long long ts0 = P4_STAMP();
    t_chunks += ts0 - tc0;
    // ---- seam: the accumulators of T -> parked bf16 units (every store of the previous tile has been issued)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results are in the accumulator file
    {
      const float* eb = ldsf + P4_EB / 4;
      const float* es = ldsf + P4_ES / 4 + tpar * 128;
      unsigned sg[P4_NW][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
      auto group = [&](auto q_tag) {
        constexpr int Q = decltype(q_tag)::value, MB = Q >> 1, GP = Q & 1;
        const int cl = MB * 32 + 16 * GP + 4 * half;   // local channel of group A; group B = cl + 8
        const float4 bA = *reinterpret_cast<const float4*>(eb + cl), bB = *reinterpret_cast<const float4*>(eb + cl + 8);
        const float4 sA = *reinterpret_cast<const float4*>(es + cl), sB = *reinterpret_cast<const float4*>(es + cl + 8);
        const float ba[8] = {bA.x, bA.y, bA.z, bA.w, bB.x, bB.y, bB.z, bB.w};
        const float sa[8] = {sA.x, sA.y, sA.z, sA.w, sB.x, sB.y, sB.z, sB.w};
        float bs[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bs[i] = ba[i] * sa[i];
        auto pix = [&](auto n_tag) {
          constexpr int N = decltype(n_tag)::value;
          constexpr int R0 = (MB * 3 + N) * 16 + 8 * GP;
          float v[8];
          v[0] = fmaf(p4_acc_read<R0 + 0>(), sa[0], bs[0]); v[1] = fmaf(p4_acc_read<R0 + 1>(), sa[1], bs[1]);
          v[2] = fmaf(p4_acc_read<R0 + 2>(), sa[2], bs[2]); v[3] = fmaf(p4_acc_read<R0 + 3>(), sa[3], bs[3]);
          v[4] = fmaf(p4_acc_read<R0 + 4>(), sa[4], bs[4]); v[5] = fmaf(p4_acc_read<R0 + 5>(), sa[5], bs[5]);
          v[6] = fmaf(p4_acc_read<R0 + 6>(), sa[6], bs[6]); v[7] = fmaf(p4_acc_read<R0 + 7>(), sa[7], bs[7]);
          if (leaky) {   // max(v, 0.01 v) as a bare v_max_f32 (the arithmetic of conv_bf16_epi_groups.inc's lean order)
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
              typedef float f32x2_t __attribute__((ext_vector_type(2)));
              const f32x2_t sv = (f32x2_t){v[i], v[i + 1]} * (f32x2_t){LEAKY_SLOPE, LEAKY_SLOPE};
              asm("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(sv.x));
              asm("v_max_f32 %0, %1, %2" : "=v"(v[i + 1]) : "v"(v[i + 1]), "v"(sv.y));
            }
          }
          if (write_signs) {
            unsigned mA = 0;
#pragma unroll
            for (int i = 7; i >= 0; --i)
              asm("v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(mA) : "v"(v[i]) : "vcc");
            sg[N][Q >> 2] |= mA << (8 * (Q & 3));
          }
          bf16x8 o;
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
          const u32x4 w = __builtin_bit_cast(u32x4, o);
          const auto r0 = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);
          hold[Q * 3 + N] = u32x4{r0[0], r1[0], r0[1], r1[1]};
        };
        p4_static_for(pix, std::make_integer_sequence<int, P4_NW>{});
      };
      if (!P4_DBG(2)) p4_static_for(group, std::make_integer_sequence<int, 8>{});
      if (write_signs) {   // the 8 sign bytes of a pixel (this lane's half of the 128 channels) go out together
        const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)(p.signs + (size_t)T.b * plane * 16), (short)0, plane * 16, 0x00020000);
#pragma unroll
        for (int n = 0; n < P4_NW; ++n) {
          const int vs = vo[n] < 0 ? (int)OOB : (vo[n] >> 4) * 8;
          const u32x2 tsg = {sg[n][0], sg[n][1]};
          __builtin_amdgcn_raw_buffer_store_b64(tsg, rs_s, vs, 0, 0);
        }
      }
    }
#pragma unroll
    for (int n = 0; n < P4_NW; ++n) vo_prev[n] = P4_DBG(1) ? (int)OOB : vo[n];
    rs_out_prev = p4_rsrc(reinterpret_cast<unsigned char*>(p.out) + (size_t)T.b * obytes, obytes);
    t_seam += P4_STAMP() - ts0;
    if (!has_next) break;
    T = Tn;
#pragma unroll
    for (int n = 0; n < P4_NW; ++n) { pbr[n] = pbn[n]; vo[n] = von[n]; }
    lw16 = lw16n;
    tpar ^= 1;
  }
  // ---- the last tile's parked units
#pragma unroll
  for (int i = 0; i < 24; ++i) p4_store16(hold[i], vo_prev[i % 3], rs_out_prev, (unsigned)(2 * (i / 3)) * (unsigned)plane16);
#ifdef YOGO_DIAG
  if (p.stamps && tid == 0) {
    unsigned long long* d = p.stamps + (size_t)blockIdx.x * 16;
    d[0] = t_start; d[1] = __builtin_amdgcn_s_memtime(); d[2] = t_chunks; d[3] = t_seam; d[4] = t_prep; d[5] = k_ord; d[6] = t_x8;
  }
#endif
}

// =========================================================================================================
// host side: eligibility, tiling, launch
// =========================================================================================================
bool conv_bf16_p4_eligible(int K, int M, int IH, int IW, int B) {
  const int Kb = round_up(K, 16) / 8;
  if (M != 128 || Kb < 8 || (Kb % 4) != 0) return false;   // nchunk = Kb / 2 even and >= 4
  if (IH < 3 || IW < 3 || B <= 0) return false;
  if ((long long)Kb * IH * IW * 16 >= (1ll << 31) || (long long)16 * IH * IW * 16 >= (1ll << 31)) return false;   // per-image descriptors, bit 31 = "out of range"
  return true;
}

// column bands of TW output columns, tiles of 384 consecutive pixels of a band (row-major inside the band): the staged input
// tile of a chunk ([2 channel blocks][rows + 2][TW + 2] units) has to fit the 6 x 256 input slots; among the fitting band
// counts take the one that stages the fewest units per image
bool conv_bf16_p4_plan(ConvP4Params* p) {
  const int OH = p->IH, OW = p->IW;
  long long best = -1;
  int best_ncb = 0;
  for (int ncb = 1; ncb <= 32 && ncb <= OW; ++ncb) {
    const int TW = cdiv(OW, ncb);
    const int bw_min = OW - (cdiv(OW, TW) - 1) * TW;
    if (cdiv(OW, TW) != ncb || bw_min <= 0) continue;
    // rows a 384-pixel tile can touch in a band of width bw: a tile starts anywhere in a row
    auto rows_of = [&](int bw) { return min(OH, 1 + cdiv(P4_PT - 1, bw)) + 2; };
    const int need = 2 * max(rows_of(TW) * (TW + 2), rows_of(bw_min) * (bw_min + 2));
    if (need > P4_NI * P4_NT) continue;
    const long long staged = (long long)(ncb - 1) * cdiv(OH * TW, P4_PT) * rows_of(TW) * (TW + 2) + (long long)cdiv(OH * bw_min, P4_PT) * rows_of(bw_min) * (bw_min + 2);
    if (best < 0 || staged < best) { best = staged; best_ncb = ncb; }
  }
  if (best < 0) return false;
  p->ncb = best_ncb;
  p->TW = cdiv(OW, best_ncb);
  p->tiles_per_band = cdiv(OH * p->TW, P4_PT);
  p->gx = p->ncb * p->tiles_per_band;
  p->ntiles = p->B * p->gx;
  auto magic = [](int d) -> unsigned { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(((1ull << 32) + (unsigned)d - 1ull) / (unsigned)d); };
  const int bw_last = OW - (p->ncb - 1) * p->TW;
  p->m_gx = magic(p->gx); p->m_tpb = magic(p->tiles_per_band);
  p->m_bw = magic(p->TW); p->m_bwl = magic(bw_last);
  p->m_lw = magic(p->TW + 2); p->m_lwl = magic(bw_last + 2);
  p->nchunk = p->Kb / 2;
  return true;
}

int launch_conv_bf16_p4(const ConvP4Params& p, hipStream_t stream) {
  static int n_cu = 0;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_p4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, P4_LDS_BYTES);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
    attr_set = true;
  }
  if (p.ntiles <= 0) return YOGO_OK;
  // one persistent workgroup per CU; a multiple of 8 so that a workgroup's tiles stay inside one XCD's run
  int grid = min(p.ntiles, n_cu);
  if (grid >= 8) grid &= ~7;
  hipLaunchKernelGGL(conv_bf16_p4_kernel, dim3(grid), dim3(P4_NT), P4_LDS_BYTES, stream, p);
  if (yogo_launch_log_enabled())
    yogo_launch_log("conv_bf16_p4_kernel | Kb=%d in=%dx%d ncb=%d TW=%d tiles_per_band=%d nchunk=%d ntiles=%d grid=%d lds=%d act=%d signs=%d scale=%d", p.Kb, p.IH,
                    p.IW, p.ncb, p.TW, p.tiles_per_band, p.nchunk, p.ntiles, grid, P4_LDS_BYTES, p.act, p.signs != nullptr, p.chan_scale != nullptr);
  YOGO_CHECK_LAUNCH("conv_bf16_p4");
  return YOGO_OK;
}
