// conv_bf16_ws16_kernel: the persistent wavefront-specialised stride-1 3x3 bf16 convolution (128 output channels) on
// v_mfma_f32_16x16x32_bf16 -- the plain-epilogue member of the conv_bf16_ws family (layer 5's forward, the data gradients of
// layers 5 / 6 of base_model: yogo/model_defns.py:54-65 and their autograd).
//
// Why a second member.  conv_bf16_ws_kernel runs its matrix pipes 0.755 busy at 1 627 MHz (profiles/r05_mfma_util.txt): it is limited by
// the clock the chip holds under its power cap, and that clock depends on the MFMA shape (MI355X_MICROARCH.md "DVFS give-back" item 7).
// Swapping the shape in situ at equal FLOPs and LDS reads (WS_ABL = 256, profiles/r06_ws_shape16_insitu_ab.log) gave -5.3 % on layer 5.
// A 16x16x32 MFMA contracts 32 channels, so this kernel works on PAIRS of 16-channel chunks: lane l of a wavefront (column c16 = l & 15,
// K group g = l >> 4) holds, per operand quad, the 8 channels of channel block g & 1 of chunk g >> 1 of the pair -- the packed weight
// slices and the staged input tiles keep the 16-byte units of the family unchanged.  Both chunks of a pair have to be resident, which
// two whole-chunk weight buffers (2 x 72 KB) would not fit: the weights stream by KERNEL ROW instead.  A period = (chunk pair, kernel
// row) = 3 taps x 8 row blocks x 4 pixel blocks = 96 MFMAs of 16 cycles per compute wavefront (1 536 cycles), 24 KB of weight slices
// (two slots by period parity, one period ahead), and the pair's two input tiles (32 KB, two slots by pair parity, requested while the
// previous pair is multiplied).  Roles as in the family: wavefronts 0-3 compute (128 channels x 64 pixels = 32 accumulator tiles of
// 16x16 in a[0:127]), wavefronts 4-7 load (LDS-DMA, next-tile decode, output stores out of the LDS staging area).
//
// The compute wavefronts' instruction stream is GENERATED (tools/gen_ws16.py -> conv_bf16_ws16_asm.inc) and is ONE asm statement: tile
// loop, period loop and tile seam, every state-carrying register fixed and clobbered (operand quads as "+v" operands of one statement per
// period made hipcc shuffle and spill them), every s_waitcnt lgkmcnt counted by the generator.  Operand reads run ahead in a ring of
// eight weight quads (six row-block slots ahead) and two sets of four pixel quads; a period's text starts with the previous period's
// last 16 MFMAs (operands in registers) while the new period's first reads are in flight, and ends with lgkmcnt(0) + s_barrier.  A tile's first and last period run group-major (row blocks 0-3, then 4-7): the output
// of rows 0-3 leaves the accumulators in the gaps of rows 4-7's MFMAs of the last period, that of rows 4-7 in the gaps of rows 0-3's
// MFMAs of the NEXT tile's first period (behind barrier X: the loaders have taken the first half out of the 32 KB staging area).
// Summation order: the two chunks of a pair are summed inside one MFMA, so results differ from conv_bf16_ws_kernel / the tiled kernel
// in the last fp32 bits: contract = one bf16 ulp on a small fraction of the outputs + the CPU fp64 reference (tests/test_gpu_ws.py).
#include "conv_bf16_ws16.h"
#ifndef W16_ASM_INC
#define W16_ASM_INC "conv_bf16_ws16_asm.inc"
#endif
#include W16_ASM_INC
// timing-only ablations of the loaders (build.sh variant TAG conv_bf16_ws16 -DW16_ABL=bits; wrong results): 1 = the weight slices are
// requested for a tile's first two periods only, 2 = no output stores, 4 = no barrier X, 8 = no input requests behind the first tile's
#ifndef W16_ABL
#define W16_ABL 0
#endif
#include <mutex>
#include <type_traits>
#include <utility>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// the accumulators a[0:127], the bias a[128:159] and the arch registers v16..v95 are named literally by the generated statement (its inputs
// live in v0..v15: the kernel is held to 96 arch VGPRs so that 160 accumulator registers fit a wavefront's 256)
#define W16_CLOBBER \
  "memory", "scc", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", \
  "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159"

namespace {

__device__ __forceinline__ int w16_udivm(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }   // n / d, m = ceil(2^32 / d), d > 1
__device__ __forceinline__ int w16_udivm1(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }
__device__ __forceinline__ int w16_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ i32x4 w16_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
// four LDS-DMA pieces (64 lanes x 16 bytes each) of one descriptor with one scalar offset: LDS destinations lds + k * 4 KB
__device__ __forceinline__ void w16_dma4(i32x4 rs, unsigned lds, int v0, int v1, int v2, int v3, unsigned soff) {
  asm volatile(
      "s_mov_b32 m0, %5\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %4, %6 offen lds\n\t"
      "s_add_u32 m0, m0, 4096\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %4, %6 offen lds\n\t"
      "s_add_u32 m0, m0, 4096\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, %6 offen lds\n\t"
      "s_add_u32 m0, m0, 4096\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, %6 offen lds"
      ::"v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(rs), "s"(lds), "s"(soff) : "memory", "scc");
}
// six pieces of one descriptor with one per-lane offset: LDS destinations lds + k * 4 KB, scalar offsets soff, + sj, + sk, + sj, + sk, + sj
// (the weight slices [3 kx][2 chunks] of a period for this wavefront's channel block and column half)
__device__ __forceinline__ void w16_dma6(i32x4 rs, unsigned lds, int voff, unsigned soff, unsigned sj, unsigned sk) {
  unsigned so;
#define W16_PJ "s_add_u32 m0, m0, 4096\n\ts_add_u32 %0, %0, %5\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\t"
#define W16_PK "s_add_u32 m0, m0, 4096\n\ts_add_u32 %0, %0, %6\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\t"
  asm volatile("s_mov_b32 m0, %3\n\ts_mov_b32 %0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds\n\t" W16_PJ W16_PK W16_PJ W16_PK W16_PJ
               : "=&s"(so)
               : "v"(voff), "s"(rs), "s"(lds), "s"(soff), "s"(sj), "s"(sk)
               : "memory", "scc");
#undef W16_PJ
#undef W16_PK
}
__device__ __forceinline__ void w16_store16(u32x4 data, int voff, i32x4 rs, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(data), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

}  // namespace

template <bool BIAS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2), amdgpu_num_vgpr(96))) void conv_bf16_ws16_kernel(const ConvWsParams p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
  constexpr unsigned OOB = 0x80000000u;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, tw = wave & 3;   // team 0 computes, team 1 loads; wavefronts tw and tw + 4 share a SIMD
  float* ldsf = reinterpret_cast<float*>(smem4);
  unsigned char* lds = reinterpret_cast<unsigned char*>(smem4);

  const int OH = p.IH, OW = p.IW;
  const int plane = OH * OW, plane16 = plane * 16;
  const int npairs = p.nchunk >> 1, nper = 3 * npairs;   // (nchunk a multiple of 4: npairs and nper are even)

  // ---- tile walk (conv_bf16_ws_kernel's): virtual block lin = slot + k * G, an XCD's workgroups share a contiguous run of tiles
  const unsigned NV = (unsigned)p.ntiles, G = gridDim.x, slot = blockIdx.x;
  const unsigned xq = NV >> 3, xr = NV & 7;
  struct TileS { int b, j0, bw, p0, p1, lastband; };
  auto find_tile = [&](unsigned& k, TileS& t) -> bool {   // (uniform) next non-empty tile of this workgroup from ordinal k on
    for (;; ++k) {
      const unsigned lin = slot + k * G;
      if (lin >= NV) return false;
      const unsigned xcd = lin & 7;
      const int widx = (int)((xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3));
      const int b = w16_udivm1(widx, p.gx, p.m_gx);
      const int bx = widx - b * p.gx;
      const int cb = w16_udivm1(bx, p.tiles_per_band, p.m_tpb);
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
  // pixel q (0..255) of tile t: byte offset of its unit inside a channel block of the staged tile, and inside a channel block of the output image
  auto pix_geom = [&](const TileS& t, int q, unsigned& pbase, int& vo) {
    const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
    const int bw = t.bw;   // >= 2 (planner)
    const int i_lo = w16_udivm(t.p0, m_bw);
    const int pp = t.p0 + q;
    const bool pv = pp < t.p1;
    const int pc = pv ? pp : (t.p1 - 1);
    const int i = w16_udivm(pc, m_bw), j = pc - i * bw;
    pbase = (unsigned)((i - i_lo) * (bw + 2) + j) * 16u;
    vo = pv ? (i * OW + t.j0 + j) * 16 : (int)OOB;
  };
  auto tile_pitch = [&](const TileS& t, unsigned& lw16, unsigned& perkb16) {   // (uniform) staged row pitch, bytes of a channel block of the staged tile
    const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
    const int i_lo = w16_udivm(t.p0, m_bw), i_hi = w16_udivm(t.p1 - 1, m_bw);
    lw16 = (unsigned)(t.bw + 2) * 16u;
    perkb16 = (unsigned)(i_hi - i_lo + 3) * lw16;
  };

  unsigned k_ord = 0;
  TileS T{};
  if (!find_tile(k_ord, T)) return;

  if (tid < 128) ldsf[W16_EB / 4 + tid] = (BIAS && p.bias != nullptr) ? p.bias[tid] : 0.f;
  __syncthreads();

  if (team == 1) {
    // =====================================================================================================================
    // LOADERS.  (Ablations of this arrangement, gpurun_out/r6_w16_abl1.log: without the epilogue -2.4 ... -4 %, without the weight requests
    // -5 %, without the output stores -8 %, without the input requests -10 %: what keeps the matrix pipes at ~0.75 busy is the CU's
    // vector-memory work -- ~40 kilobyte-pieces per 1 536-cycle period through four wavefronts --, not one stream blocking another:
    // two weight-only + two input / output wavefronts measured +2 ... +5 % (profiles/r06_ws16_split_loaders_ab.log), three weight (+ input)
    // wavefronts + one store-only wavefront +0.5 ... +1 % (profiles/r06_ws16_3plus1_loaders_ab.log); both are in the history of this file.)
    // Per period: [period 0: the staged half of the previous tile -> registers, barrier X] the NEXT period's weight
    // slices (6 pieces per wavefront), chunk r of the NEXT pair's input tiles in the pair's periods r = 0, 1 (4 pieces), four output
    // stores in each of the periods 0..3, [period 0: next-tile lookup + decode; period 1: the mailbox], counted vmcnt, barrier.
    // =====================================================================================================================
    const int lane = w16_lane();
    const int ttid = tw * 64 + lane;
    const int rowb = p.IW * 16, kcb = p.IH * p.IW * 16;
    const unsigned ibytes = (unsigned)p.Kb * kcb, obytes = 16u * plane16, wbytes = 9u * p.Kb * 2048u;
    const unsigned so_i = 2u * kcb;                  // bytes between the 16-channel chunks of an image
    const unsigned wstep = (unsigned)p.Kb * 2048u;   // bytes between the taps of the packed weights
    const i32x4 rs_w = w16_rsrc(p.wp, wbytes);
    const int lane16 = lane * 16;
    const int kbw = tw >> 1, colh = tw & 1;          // this wavefront's weight pieces: channel block of the chunk, column half
    // DMA source offsets of the 4 input slots: element ttid + i * 256 of the flattened [2][rows_in][lw] tile -> (channel block, row, column)
    auto decode_slots = [&](const TileS& t, int (&voff)[WS_NI]) {
      const unsigned m_bw = t.lastband ? p.m_bwl : p.m_bw;
      const int bw = t.bw;
      const int i_lo = w16_udivm(t.p0, m_bw), i_hi = w16_udivm(t.p1 - 1, m_bw);
      const int rows_in = i_hi - i_lo + 3;
      const int iy0 = i_lo - 1, ix0 = t.j0 - 1;
      const int lw = bw + 2;
      const int per_kb = rows_in * lw;
      const unsigned inv_lw = t.lastband ? p.m_lwl : p.m_lw;
      const unsigned inv_perkb = 0xFFFFFFFFu / (unsigned)per_kb + 1u;
      const int skc = w16_udivm(WS_NT, inv_perkb);
      const int srm = WS_NT - skc * per_kb;
      const int sr = w16_udivm(srm, inv_lw);
      const int sx = srm - sr * lw;
      int kc_ = w16_udivm(ttid, inv_perkb);
      const int rm0 = ttid - kc_ * per_kb;
      int r_ = w16_udivm(rm0, inv_lw);
      int x_ = rm0 - r_ * lw;
#pragma unroll
      for (int i = 0; i < WS_NI; ++i) {
        const int iy_ = iy0 + r_, ix_ = ix0 + x_;
        const bool ok = (kc_ < 2) && (iy_ >= 0) && (iy_ < p.IH) && (ix_ >= 0) && (ix_ < p.IW);
        voff[i] = ok ? kc_ * kcb + iy_ * rowb + ix_ * 16 : (int)OOB;
        x_ += sx; r_ += sr; kc_ += skc;
        if (x_ >= lw) { x_ -= lw; ++r_; }
        if (r_ >= rows_in) { r_ -= rows_in; ++kc_; }
      }
    };
    static_assert(WS_NI == 4, "w16_dma4 issues the four input slots");
    // weight slices of period cn (pair cn / 3, kernel row cn % 3) -> weight slot cn & 1
    auto req_w = [&](int P, int r, int par) {
      const unsigned soff = (unsigned)(((3 * r) * p.Kb + 4 * P + kbw) * 2048 + colh * 1024);
      w16_dma6(rs_w, (unsigned)(W16_W0 + par * W16_WSLOT + kbw * 2048 + colh * 1024), lane16, soff, 4096u, wstep - 4096u);
    };
    // input tile of chunk cn of the image behind (rs, voff) -> pair slot ps, chunk half j
    auto req_i = [&](i32x4 rs, const int (&voff)[WS_NI], int cn, int ps, int j) {
      w16_dma4(rs, (unsigned)(W16_I0 + ps * W16_ISLOT + j * WS_IB + tw * 64 * 16), voff[0], voff[1], voff[2], voff[3], (unsigned)cn * so_i);
    };
    // Output hand-over: the staged half (8 channel blocks x this wavefront's 64 pixels) -> registers, four stores per period
    const unsigned stg_rd = (unsigned)(W16_STG + tw * 8192 + lane * 16);
    u32x4 fifo[8];
    int f_vo = (int)OOB;
    i32x4 f_rs = w16_rsrc(p.out, 0u);
    int f_cb0 = 0;
    auto fifo_fill = [&](int cb0, int vop, i32x4 rs_o) {
#pragma unroll
      for (int u = 0; u < 8; ++u) fifo[u] = *reinterpret_cast<const u32x4*>(lds + stg_rd + u * 1024);
      f_vo = vop; f_rs = rs_o; f_cb0 = cb0;
    };
    auto fifo_store4 = [&](auto h_tag) {   // units 4 h .. 4 h + 3 of the FIFO
      constexpr int Hh = decltype(h_tag)::value;
      const i32x4 rs = {__builtin_amdgcn_readfirstlane(f_rs.x), __builtin_amdgcn_readfirstlane(f_rs.y), __builtin_amdgcn_readfirstlane(f_rs.z),
                        __builtin_amdgcn_readfirstlane(f_rs.w)};
      const int cb0 = __builtin_amdgcn_readfirstlane(f_cb0);
#pragma unroll
      for (int u = 4 * Hh; u < 4 * Hh + 4; ++u) w16_store16(fifo[u], (W16_ABL & 2) ? (int)OOB : f_vo, rs, (unsigned)(cb0 + u) * (unsigned)plane16);
    };
    using IC0 = std::integral_constant<int, 0>;
    using IC1 = std::integral_constant<int, 1>;

    int voff[WS_NI];
    decode_slots(T, voff);
    unsigned pbase_ = 0, lw16_ = 0, perkb16_ = 0;
    int vo = (int)OOB, vo_prev = (int)OOB;
    pix_geom(T, ttid, pbase_, vo);
    i32x4 rs_in = w16_rsrc(reinterpret_cast<const unsigned char*>(p.in) + (size_t)T.b * ibytes, ibytes);
    i32x4 rs_out = w16_rsrc(reinterpret_cast<unsigned char*>(p.out) + (size_t)T.b * obytes, obytes);
    i32x4 rs_out_prev = w16_rsrc(p.out, 0u);
    req_i(rs_in, voff, 0, 0, 0);
    req_i(rs_in, voff, 1, 0, 1);
    req_w(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // (#1) pair 0 and the weight slices of period 0 of the first tile have landed
    bool has_next = true;
    TileS Tn{};
    int voff_n[WS_NI] = {(int)OOB, (int)OOB, (int)OOB, (int)OOB}, vo_n = (int)OOB;
    i32x4 rs_in_n = rs_in, rs_out_n = rs_out;
    while (has_next) {
      int P = 0, r = 0;   // pair and kernel row of period c
      for (int c = 0; c < nper; ++c) {
        // oldest first: the weight slices of the NEXT period (needed at this period's barrier; a period is 1 536 MFMA cycles, the six
        // pieces take ~600 to issue and an L2 round trip to land) ...
        {
          const int rn = r == 2 ? 0 : r + 1, Pn = r == 2 ? P + 1 : P;
          if ((W16_ABL & 1) && c >= 1) {
          } else if (c + 1 < nper) req_w(Pn, rn, (c + 1) & 1);
          else if (has_next) req_w(0, 0, 0);
        }
        if (c == 0) {   // the half of the previous tile's output the compute wavefronts staged in its last period
          fifo_fill(0, vo_prev, rs_out_prev);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (!(W16_ABL & 4)) __builtin_amdgcn_s_barrier();   // (X) the staging area is free for the other half
        }
        // ... then chunk r of the next pair's input tiles (needed at the barrier of this pair's last period: they may stay in flight)
        bool req = false;   // (uniform)
        if (r < 2 && !(W16_ABL & 8)) {
          if (P + 1 < npairs) {
            req_i(rs_in, voff, 2 * (P + 1) + r, (P + 1) & 1, r);
            req = true;
          } else if (has_next) {   // the request stream crosses into the next tile (pair 0 -> pair slot 0)
            if (r == 0) {
#pragma unroll
              for (int i = 0; i < WS_NI; ++i) voff[i] = voff_n[i];
              rs_in = rs_in_n;
            }
            req_i(rs_in, voff, r, 0, r);
            req = true;
          }
        }
        if (c == 0) fifo_store4(IC0{});
        else if (c == 1) fifo_store4(IC1{});
        else if (c == 2) { fifo_fill(8, vo_prev, rs_out_prev); fifo_store4(IC0{}); }   // (the other half: staged during period 0)
        else if (c == 3) fifo_store4(IC1{});
        const bool dr = c < 4;   // (uniform) four stores were issued
        if (c == 0) {   // the NEXT tile: looked up and decoded behind this period's requests and stores
          unsigned kn = k_ord + 1;
          has_next = find_tile(kn, Tn);
          k_ord = kn;
          if (has_next) {
            decode_slots(Tn, voff_n);
            pix_geom(Tn, ttid, pbase_, vo_n);
            tile_pitch(Tn, lw16_, perkb16_);
            rs_in_n = w16_rsrc(reinterpret_cast<const unsigned char*>(p.in) + (size_t)Tn.b * ibytes, ibytes);
            rs_out_n = w16_rsrc(reinterpret_cast<unsigned char*>(p.out) + (size_t)Tn.b * obytes, obytes);
          }
        }
        // the NEXT tile's operand addresses for the compute wavefronts: written in period 1 (they read the previous message at their
        // tile seam, in front of this tile's period 0) and read at the next seam
        if (c == 1) {
          *reinterpret_cast<unsigned*>(lds + W16_MB + tw * 256 + ((lane & 15) * 4 + (lane >> 4)) * 4) = pbase_;
          if (tw == 0 && lane == 0) *reinterpret_cast<u32x4*>(lds + W16_MBS) = u32x4{has_next ? 1u : 0u, lw16_, perkb16_, 0u};
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // vector-memory operations retire in order: everything but this period's input request (4) and stores (4) has to be done
        if (dr && req) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (dr || req) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (++r == 3) { r = 0; ++P; }
      }
      vo_prev = vo;
      rs_out_prev = rs_out;
      if (has_next) {
        vo = vo_n;
        rs_out = rs_out_n;
      }
    }
    // the last tile: the staged half, barrier X, (compute: the other half), barrier, the other half
    fifo_fill(0, vo_prev, rs_out_prev);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(W16_ABL & 4)) __builtin_amdgcn_s_barrier();
    fifo_store4(IC0{});
    fifo_store4(IC1{});
    __builtin_amdgcn_s_barrier();
    fifo_fill(8, vo_prev, rs_out_prev);
    fifo_store4(IC0{});
    fifo_store4(IC1{});
    return;
  }

  // =======================================================================================================================
  // COMPUTE
  // =======================================================================================================================
  __builtin_amdgcn_s_setprio(3);
  const int lane = w16_lane(), c16 = lane & 15, g = lane >> 4;
  const unsigned pa_lane = (unsigned)(W16_W0 + (g >> 1) * 4096 + (g & 1) * 2048 + c16 * 16);
  const unsigned stg = (unsigned)(W16_STG + tw * 8192 + (g >> 1) * 1024 + c16 * 16 + (g & 1) * 8);
  const unsigned eb = (unsigned)(W16_EB + g * 16);
  unsigned lw16, perkb16;
  tile_pitch(T, lw16, perkb16);
  unsigned pbl[4];   // this lane's operand address of pixel block pb: unit of its pixel + K group (chunk half, channel block)
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    unsigned pbase;
    int vo_unused;
    pix_geom(T, tw * 64 + pb * 16 + c16, pbase, vo_unused);
    pbl[pb] = pbase + (unsigned)(g & 1) * perkb16 + (unsigned)(g >> 1) * (unsigned)WS_IB + (unsigned)W16_I0;
  }
  __builtin_amdgcn_s_barrier();   // (#1)
  // K-group part of a pixel-block address as the seam recomputes it: (bytes of a channel block & kmask) + kconst
  const unsigned kmask = (g & 1) ? 0xFFFFFFFFu : 0u, kconst = (unsigned)(g >> 1) * (unsigned)WS_IB + (unsigned)W16_I0;
  const unsigned mba = (unsigned)(W16_MB + tw * 256 + c16 * 16), mbs = (unsigned)W16_MBS;
  const unsigned npp = (unsigned)((nper - 2) >> 1);   // pairs of tap-major periods between a tile's first and last period
  if constexpr (BIAS)
    asm volatile(W16_TXT_ROLEB
                 :
                 : [pa] "v"(pa_lane), [stg] "v"(stg), [eb] "v"(eb), [q0] "v"(pbl[0]), [q1] "v"(pbl[1]), [q2] "v"(pbl[2]), [q3] "v"(pbl[3]), [mba] "v"(mba),
                   [mbs] "v"(mbs), [kmask] "v"(kmask), [kconst] "v"(kconst), [lw16] "s"(lw16), [npp] "s"(npp)
                 : W16_CLOBBER);
  else
    asm volatile(W16_TXT_ROLE
                 :
                 : [pa] "v"(pa_lane), [stg] "v"(stg), [eb] "v"(eb), [q0] "v"(pbl[0]), [q1] "v"(pbl[1]), [q2] "v"(pbl[2]), [q3] "v"(pbl[3]), [mba] "v"(mba),
                   [mbs] "v"(mbs), [kmask] "v"(kmask), [kconst] "v"(kconst), [lw16] "s"(lw16), [npp] "s"(npp)
                 : W16_CLOBBER);
}

// =========================================================================================================
// host side
// =========================================================================================================
bool conv_bf16_ws16_eligible(int K, int M, int IH, int IW, int B) {
  const int Kb = round_up(K, 16) / 8;
  if (!conv_bf16_ws_eligible(K, M, IH, IW, B)) return false;
  return Kb >= 8 && (Kb % 8) == 0;   // chunk pairs, an even number of them (the period / pair parities repeat from tile to tile)
}

int launch_conv_bf16_ws16(const ConvWsParams& p, hipStream_t stream) {
  static std::mutex mu;
  static int n_cu_of[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    yogo_set_error("conv_bf16_ws16: hipGetDevice failed");
    return YOGO_ERR_HIP;
  }
  int n_cu;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (n_cu_of[dev] == 0) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_ws16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, W16_LDS_BYTES);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_ws16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, W16_LDS_BYTES);
      if (e != hipSuccess) {
        yogo_set_error("conv_bf16_ws16: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", W16_LDS_BYTES, hipGetErrorString(e));
        return YOGO_ERR_HIP;
      }
      hipDeviceProp_t prop;
      n_cu_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    n_cu = n_cu_of[dev];
  }
  if (p.ntiles <= 0) return YOGO_OK;
  if (p.act != ACT_NONE || p.signs != nullptr || p.chan_scale != nullptr || (p.nchunk & 3) != 0) {
    yogo_set_error("conv_bf16_ws16: plain epilogue and whole chunk-pair pairs only");
    return YOGO_ERR_ARG;
  }
  int grid = min(p.ntiles, n_cu);
  if (grid >= 8) grid &= ~7;
  if (p.bias != nullptr) hipLaunchKernelGGL(conv_bf16_ws16_kernel<true>, dim3(grid), dim3(512), W16_LDS_BYTES, stream, p);
  else hipLaunchKernelGGL(conv_bf16_ws16_kernel<false>, dim3(grid), dim3(512), W16_LDS_BYTES, stream, p);
  if (yogo_launch_log_enabled())
    yogo_launch_log("conv_bf16_ws16_kernel<%s> | Kb=%d in=%dx%d ncb=%d TW=%d tiles_per_band=%d periods=%d ntiles=%d grid=%d lds=%d", p.bias ? "true" : "false", p.Kb,
                    p.IH, p.IW, p.ncb, p.TW, p.tiles_per_band, 3 * (p.nchunk / 2), p.ntiles, grid, W16_LDS_BYTES);
  YOGO_CHECK_LAUNCH("conv_bf16_ws16");
  return YOGO_OK;
}
