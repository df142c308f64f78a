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
#define W2_VMWAIT(N) do { const unsigned long long v0__ = __builtin_amdgcn_s_memtime(); w2_vmwait<N>(); t_vm += __builtin_amdgcn_s_memtime() - v0__; } while (0)
#define W2_LBARRIER() do { const unsigned long long v0__ = __builtin_amdgcn_s_memtime(); w2_barrier(); t_lb += __builtin_amdgcn_s_memtime() - v0__; } while (0)
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
  if (p.chan_scale == nullptr && tid < 128) reinterpret_cast<float*>(lds + W2_ES)[tid] = 1.f;
  __syncthreads();

  if (team == 1) {
    // =====================================================================================================================
    // LOADERS
    // =====================================================================================================================
    const int lane = w2_lane();
    const int q31 = lane & 31, hp = lane >> 5;
    const int ttid = tw * 64 + lane;
    const int rowb = IW * 16, kcb = IH * IW * 16;
    const unsigned ibytes = (unsigned)p.Kb * kcb, obytes = 16u * plane16, wbytes = 9u * p.Kb * 2048u;
    const unsigned wstep = (unsigned)p.Kb * 2048u;   // bytes between the slices of the packed weights
    const i32x4 rs_w = w2_rsrc(p.wp, wbytes);
    const int lane16 = W2_DBG(4) ? (int)OOB : lane * 16;
    const bool has_scale = p.chan_scale != nullptr;
    const i32x4 rs_sc = w2_rsrc(p.chan_scale, has_scale ? (unsigned)p.B * 512u : 0u);

    // per-tile lane geometry: the dy element this lane stages (position ttid of the [rows_in][lw] image of a channel block), the
    // output offsets of the quads it stores (lane = (column parity hp, quad q31) of pixel group n), and the sign-map offsets it
    // fetches for its partner compute wavefront (lane = (half-wave hp, quad q31))
    struct LaneGeo { int dyoff; int vo[2][2]; int vs[2][2][2]; };
    auto decode = [&](const TileS& t, LaneGeo& g) __attribute__((always_inline)) {
      const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
      const int bw = t.bw;
      const int i_lo = w2_udivm1(t.p0, bw, m_bw), i_hi = w2_udivm1(t.p1 - 1, bw, m_bw);
      const int rows_in = i_hi - i_lo + 2;
      const int lw = bw + 1;
      const unsigned inv_lw = t.lastband ? p.m_lwl : p.m_lw;   // (lw >= 2)
      const int r_ = w2_udivm(ttid, inv_lw), x_ = ttid - r_ * lw;
      const int iy = i_lo + r_, ix = t.j0 + x_;
      g.dyoff = (r_ < rows_in && iy < IH && ix < IW && !W2_DBG(4)) ? iy * rowb + ix * 16 : (int)OOB;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int pp = t.p0 + (nh * 2 + n) * 32 + q31;
        const bool pv = pp < t.p1;
        const int pc = pv ? pp : (t.p1 - 1);
        const int i = w2_udivm1(pc, bw, m_bw), j = pc - i * bw;
        const int pix = 2 * i * OW + 2 * (t.j0 + j);
        const bool vx = 2 * (t.j0 + j) + 1 < OW, vy = 2 * i + 1 < OH;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          const bool okr = pv && (py == 0 || vy);
          g.vo[py][n] = (okr && (hp == 0 || vx) && !W2_DBG(1)) ? (pix + py * OW + hp) * 16 : (int)OOB;
#pragma unroll
          for (int px = 0; px < 2; ++px)
            g.vs[py][n][px] = (okr && (px == 0 || vx)) ? (hp * plane + pix + py * OW + px) * 8 + mh * 4 : (int)OOB;
        }
      }
    };
    auto rs_in_of = [&](int b) __attribute__((always_inline)) { return w2_rsrc(reinterpret_cast<const unsigned char*>(p.in) + (size_t)b * ibytes, ibytes); };
    auto rs_out_of = [&](int b) __attribute__((always_inline)) { return w2_rsrc(reinterpret_cast<unsigned char*>(p.out) + (size_t)b * obytes, obytes); };
    auto rs_sg_of = [&](int b) __attribute__((always_inline)) { return w2_rsrc(SIGNS ? p.signs + (size_t)b * plane16 : nullptr, SIGNS ? (unsigned)plane16 : 0u); };
    // requests ----------------------------------------------------------------------------------------------------------
    // weights of a pass-A period (chunks 2 k, 2 k + 1: groups cc * 3 + slice) / a pass-B period (chunk c: slices 3 .. 8) -> buffer wb
    auto req_wA = [&](int k, int wb) __attribute__((always_inline)) {
      w2_dma3(rs_w, (unsigned)(wb * W2_WB + tw * 1024), lane16, (unsigned)((4 * k) * 2048 + tw * 1024), wstep);
      w2_dma3(rs_w, (unsigned)(wb * W2_WB + 3 * 4096 + tw * 1024), lane16, (unsigned)((4 * k + 2) * 2048 + tw * 1024), wstep);
    };
    auto req_wB = [&](int c, int wb) __attribute__((always_inline)) {
      w2_dma6(rs_w, (unsigned)(wb * W2_WB + tw * 1024), lane16, (unsigned)(3u * wstep + (unsigned)((2 * c) * 2048 + tw * 1024)), wstep);
    };
    // 16-channel chunk c of the dy tile described by (rs, dyoff) -> slot c (this wavefront's 64 positions of both channel blocks)
    auto req_dy = [&](i32x4 rs, int dyoff, int c) __attribute__((always_inline)) {
      w2_dma2(rs, (unsigned)(W2_DY + c * W2_DYS + tw * 1024), dyoff, (unsigned)(2 * c) * (unsigned)kcb, (unsigned)kcb);
    };
    auto req_signs = [&](i32x4 rs, const int (&vs)[2][2]) {
      if constexpr (SIGNS) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int px = 0; px < 2; ++px) w2_dma_dword(rs, (unsigned)(W2_SG + (tw * 4 + n * 2 + px) * 256), vs[n][px], 0u);
      }
    };
    auto req_scale = [&](int b) __attribute__((always_inline)) {   // [128] floats: two 256-byte pieces, loaders 2 / 3 repeat those of 0 / 1 (the same counts in every wavefront)
      if (has_scale) w2_dma_dword(rs_sc, (unsigned)(W2_ES + (tw & 1) * 256), lane * 4, (unsigned)((b * 128 + (tw & 1) * 64) * 4));
    };
    // the staged quarter (mb, gp) of this wavefront's partner: 4 units (n, px) of 1 KB in slot `sl` -> 4 stores, each one channel
    // block of 32 quads x both column parities (lanes 0-31: px 0, lanes 32-63: px 1)
    auto store_quarter = [&](int sl, int quarter, const int (&vo)[2], i32x4 rs_o) {
      const unsigned char* base = lds + W2_STG + sl * W2_SLOT + (tw * 4 + hp) * 1024 + q31 * 16;
      u32x4 d[4];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 2; ++e) d[n * 2 + e] = *reinterpret_cast<const u32x4*>(base + n * 2048 + e * 512);
      const int cb0 = mh * 8 + quarter * 2;   // channel block of e = 0 (quarter = mb * 2 + gp)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 2; ++e) w2_store16(d[n * 2 + e], vo[n], rs_o, (unsigned)(cb0 + e) * (unsigned)plane16);
    };

    [[maybe_unused]] unsigned long long t_vm = 0, t_lb = 0;
    LaneGeo gc{}, gn{};
    decode(T, gc);
    i32x4 rs_in = rs_in_of(T.b), rs_out = rs_out_of(T.b), rs_sg = rs_sg_of(T.b);
    // first tile: everything of its first period
    for (int c = 0; c < nck; ++c) req_dy(rs_in, gc.dyoff, c);
    req_wA(0, 0);
    req_signs(rs_sg, gc.vs[0]);
    W2_VMWAIT(0);
    W2_LBARRIER();   // (#1)
    int wpar = 0;           // weight buffer of the MFMA period being computed
    bool pend = false;      // a staged quarter of the previous epilogue is waiting in slot 1 (its last quarter)
    int vo_pend[2] = {(int)OOB, (int)OOB};
    i32x4 rs_pend = rs_out;
    bool has_next = true;
    TileS Tn{};
    i32x4 rs_in_n = rs_in;
    for (;;) {
      // ---------------- pass A: np periods of 32 channels
      for (int k = 0; k < np; ++k) {
        if (k + 1 < np) req_wA(k + 1, wpar ^ 1);
        else req_wB(0, wpar ^ 1);
        if (k == 0) {
          const bool hadp = pend;
          if (pend) {   // the last quarter of the previous tile's pass B
            store_quarter(1, 3, vo_pend, rs_pend);
            pend = false;
          }
          req_scale(T.b);   // (the channel scale is read by the epilogue quarters only: free since the previous tile's last one)
          // the next tile: looked up and decoded here, behind this period's requests
          unsigned kn = k_ord + 1;
          has_next = find_tile(kn, Tn);
          k_ord = kn;
          if (has_next) {
            decode(Tn, gn);
            rs_in_n = rs_in_of(Tn.b);
          }
          // (vector-memory operations retire in order: everything but what was issued behind the weights has to be done)
          if (hadp && has_scale) W2_VMWAIT(5);
          else if (hadp) W2_VMWAIT(4);
          else if (has_scale) W2_VMWAIT(1);
          else W2_VMWAIT(0);
        } else {
          W2_VMWAIT(0);
        }
        W2_LBARRIER();
        wpar ^= 1;
      }
      // ---------------- epilogue A: 4 quarters
      for (int e = 0; e < 4; ++e) {
        if (e >= 1) store_quarter((e - 1) & 1, e - 1, gc.vo[0], rs_out);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the staging reads are in registers before the slot is written again)
        W2_LBARRIER();
      }
      // ---------------- pass B: nck periods of 16 channels
      for (int c = 0; c < nck; ++c) {
        if (c + 1 < nck) req_wB(c + 1, wpar ^ 1);
        else if (has_next) req_wA(0, wpar ^ 1);
        if (c == 0) {
          store_quarter(1, 3, gc.vo[0], rs_out);
          req_signs(rs_sg, gc.vs[1]);
          if constexpr (SIGNS) W2_VMWAIT(8);
          else W2_VMWAIT(4);
        } else if (has_next) {
          req_dy(rs_in_n, gn.dyoff, c - 1);
          W2_VMWAIT(2);
        } else {
          W2_VMWAIT(0);
        }
        W2_LBARRIER();
        wpar ^= 1;
      }
      // ---------------- epilogue B
      for (int e = 0; e < 4; ++e) {
        if (e == 0 && has_next) req_dy(rs_in_n, gn.dyoff, nck - 1);
        if (e >= 1) store_quarter((e - 1) & 1, e - 1, gc.vo[1], rs_out);
        if (e == 3 && has_next) req_signs(rs_sg_of(Tn.b), gn.vs[0]);   // (the sign area is free: pass B's bytes were read in front of quarter 0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (e == 3) W2_VMWAIT(0);   // (everything of the next tile's first period has landed; its P0 weights were waited for in the last B period)
        W2_LBARRIER();
      }
      pend = true;
      vo_pend[0] = gc.vo[1][0]; vo_pend[1] = gc.vo[1][1];
      rs_pend = rs_out;
      if (!has_next) break;
      T = Tn;
      gc = gn;
      rs_in = rs_in_n;
      rs_out = rs_out_of(T.b);
      rs_sg = rs_sg_of(T.b);
    }
    store_quarter(1, 3, vo_pend, rs_pend);
#ifdef YOGO_DIAG
    if (p.stamps && ttid == 0) {
      unsigned long long* d = p.stamps + (size_t)blockIdx.x * 16;
      d[10] = t_vm; d[11] = t_lb; d[12] = __builtin_amdgcn_s_memtime();
    }
#endif
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
  f32x16 acc[2][2][2];   // [px][mb][n]
  auto acc_zero = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;
  };
  // one 16-channel chunk: weight groups at wbase (this lane's unit of row block 0), dy slot at dbase; NS steps of 4 MFMAs
  auto run16 = [&](auto passb_tag, unsigned wbase, unsigned dbase) __attribute__((always_inline)) {
    constexpr bool PB = decltype(passb_tag)::value;
    constexpr int NS = PB ? 6 : 3;
    u32x4 Bv[2];
    w2_static_for([&](auto s_tag) __attribute__((always_inline)) {
      constexpr int S = decltype(s_tag)::value;
      constexpr W2Step st = w2_step(PB, S);
      constexpr int prev_sh = S == 0 ? -1 : w2_step(PB, S > 0 ? S - 1 : 0).sh;
      if constexpr (st.sh != prev_sh) {
        const unsigned so = (unsigned)((st.sh & 1) * 16) + ((st.sh & 2) ? lw16 : 0u);
#pragma unroll
        for (int n = 0; n < 2; ++n) Bv[n] = *reinterpret_cast<const u32x4*>(lds + dbase + pbr[n] + so);
      }
      u32x4 Av[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) Av[mb] = *reinterpret_cast<const u32x4*>(lds + wbase + st.g * 4096 + mb * 512);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[st.px][mb][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Av[mb]), __builtin_bit_cast(bf16x8, Bv[n]),
                                                                      acc[st.px][mb][n], 0, 0, 0);
    }, std::make_integer_sequence<int, NS>{});
  };
  // a quarter of a pass's epilogue: channel group (mb, gp) of the 8 accumulator tiles -> staging slot `sl` (4 units of 1 KB)
  [[maybe_unused]] unsigned long long t_bw = 0, t_ebw = 0;   // (diagnostic build) ticks inside the barrier statements of the MFMA periods / the epilogue quarters
  unsigned sg[2][2] = {{0u, 0u}, {0u, 0u}};   // this lane's sign bytes of the pass: [n][px], byte mb * 2 + gp
  auto epi_quarter = [&](auto q_tag, int sl) __attribute__((always_inline)) {
    constexpr int Q = decltype(q_tag)::value, MB = Q >> 1, GP = Q & 1;
    const float* es = reinterpret_cast<const float*>(lds + W2_ES) + mh * 64 + MB * 32 + 16 * GP + 4 * half;
    const float4 sA = *reinterpret_cast<const float4*>(es), sB = *reinterpret_cast<const float4*>(es + 8);
    const float sa[8] = {sA.x, sA.y, sA.z, sA.w, sB.x, sB.y, sB.z, sB.w};
    float sl_[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sl_[i] = LEAKY_SLOPE * sa[i];
    unsigned char* dst0 = lds + stg_wr + sl * W2_SLOT;
    w2_static_for([&](auto u_tag) __attribute__((always_inline)) {   // the quarter's four units (n, px)
      constexpr int n = decltype(u_tag)::value >> 1, px = decltype(u_tag)::value & 1;
      float v[8];
      if constexpr (SIGNS) {
        const unsigned m = sg[n][px] >> (8 * Q);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int t = (int)(m << (31 - i)) >> 31;   // bit i spread over the word (v_bfe_i32) selects scale or 0.01 * scale (v_bfi_b32)
          unsigned f;
          asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(f) : "v"(t), "v"(sa[i]), "v"(sl_[i]));
          v[i] = acc[px][MB][n][8 * GP + i] * __builtin_bit_cast(float, f);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(acc[px][MB][n][8 * GP + i], sa[i], 0.f * sa[i]);   // (conv_bf16_epi_groups.inc: fma(acc, scale, bias * scale), bias = 0)
      }
      if (W2_DBG(2)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = acc[px][MB][n][8 * GP + i];
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
    acc_zero();
    for (int k = 0; k < np; ++k) {
      const unsigned wb = (unsigned)(wpar * W2_WB) + a_b0;
      run16(std::false_type{}, wb, (unsigned)(W2_DY + (2 * k) * W2_DYS));
      run16(std::false_type{}, wb + 3 * 4096, (unsigned)(W2_DY + (2 * k + 1) * W2_DYS));
      [[maybe_unused]] const unsigned long long b0 = W2_STAMP();
      w2_barrier_lgkm();
      t_bw += W2_STAMP() - b0;
      wpar ^= 1;
    }
    [[maybe_unused]] const unsigned long long s1 = W2_STAMP();
    epilogue();
    [[maybe_unused]] const unsigned long long s2 = W2_STAMP();
    acc_zero();
    for (int c = 0; c < nck; ++c) {
      run16(std::true_type{}, (unsigned)(wpar * W2_WB) + a_b0, (unsigned)(W2_DY + c * W2_DYS));
      [[maybe_unused]] const unsigned long long b0 = W2_STAMP();
      w2_barrier_lgkm();
      t_bw += W2_STAMP() - b0;
      wpar ^= 1;
    }
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
