// Batched objectness threshold + box convert + greedy NMS + class-confidence filter.
// Replaces the per-image Python loop around format_preds (yogo/utils/prediction_formatting.py:23-93, called from
// yogo/infer.py:45,73) and, inside it, torchvision.ops.box_convert / torchvision.ops.nms (third-party,
// torchvision>=0.14.1; CPU kernel semantics restated in oracle/yogo_oracle.py).  SURVEY.md K17, rows a8-a10.
//
// One 1024-lane workgroup per image, whole pipeline in one launch:
//   1. ordered compaction of cells with objectness > obj_thresh (ballot + LDS scan; cell order preserved)
//   2. score = max(class) * objectness; 64-bit key = (monotone(score) << 32) | ~candidate  -> bitonic sort in LDS
//      (descending score, ties -> lower cell first == torch's stable descending sort)
//   3. greedy suppression in sorted order, 64 candidates (one wavefront) at a time: the owning wavefront resolves
//      the chunk serially over its *alive* members with ballot/readlane broadcasts, publishes the survivors in LDS,
//      then all 16 wavefronts apply them to every later candidate held in registers.  The result is exactly the
//      sequential greedy keep list.
//   4. ordered output compaction (+ max-class > min_class_confidence filter), rows gathered from the prediction.
//
// Bit-exactness contract: fp32 arithmetic in the same operation order as the CPU reference, no FMA contraction
// (-ffp-contract=off for this TU), correctly rounded division, IoU compared in double against the double threshold,
// objectness / class thresholds compared in float32 (torch casts the Python scalar to the tensor dtype).
//
// DEC = true (yogo_decode_format_preds_batched) takes the head's RAW output and decodes a value where it is loaded (the
// arithmetic of decode_fwd_kernel, decode_loss.hip: same expressions, same -ffp-contract=off, so each value has the bits the
// separate decode pass would have stored): the decoded [B, 5+C, Sy, Sx] tensor of `yogo infer` (yogo/model.py:277-313 ->
// yogo/infer.py:45,73) is never written or read back, and the kept rows / cells / counts are identical.
#include "common.h"
#include <cmath>
#include <cstring>

#ifndef NMS_UNROLL
#define NMS_UNROLL 4   // (A/B builds) unroll factor of the suppression loop over a chunk's kept boxes
#endif
#define NMS_THREADS 1024
#define NMS_WAVES 16
#define NMS_SLOTS 16
#define NMS_MAX_CELLS (NMS_THREADS * NMS_SLOTS)  // 16384 >= 97*129

struct NmsParams {
  const float* pred;      // [B][P][cells]
  float* out_rows;        // [B][cap][P]
  long long* out_cells;   // [B][cap]   kept cell index (y*Sx + x), in output order
  int* out_count;         // [B]
  // workspace, per image
  int* ws_cells;          // [B][cells] compacted candidate -> cell
  float4* ws_box;         // [B][cells] sorted xyxy
  float* ws_area;         // [B][cells]
  int B, P, cells, cap;
  float obj_thresh;       // float32(obj_thresh)
  double iou_thresh;
  int iou_half;           // iou_thresh == 0.5: the division-free form of the suppression test (see THE SUPPRESSION TEST below)
  float min_cls;          // float32(min_class_confidence_threshold)
  int do_nms, use_cls_filter, xyxy;
  // DEC only: the decode's operands (yogo_decode_fwd)
  const float* cxs;       // [cells]
  const float* cys;       // [cells]
  float inv_sx, inv_sy, anchor_w, anchor_h, wmul, hmul;
  int inference;          // class channels: softmax (1) or raw logits (0)
};

__device__ __forceinline__ float nms_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// One image's prediction as the kernel sees it: the decoded tensor itself (DEC = false) or the raw head output decoded per load.
template <bool DEC>
struct PredView {
  const float* pred;   // [P][cells] of this image
  const NmsParams& p;
  __device__ __forceinline__ float at(int ch, int cell) const { return pred[(size_t)ch * p.cells + cell]; }
  __device__ __forceinline__ float obj(int cell) const { return DEC ? nms_sigmoid(at(4, cell)) : at(4, cell); }
  __device__ __forceinline__ void box(int cell, float& cx, float& cy, float& w, float& h) const {
    if (DEC) {
      cx = p.inv_sx * nms_sigmoid(at(0, cell)) + p.cxs[cell];
      cy = p.inv_sy * nms_sigmoid(at(1, cell)) + p.cys[cell];
      w = p.anchor_w * expf(fminf(at(2, cell), 80.f)) * p.wmul;
      h = p.anchor_h * expf(fminf(at(3, cell), 80.f)) * p.hmul;
    } else {
      cx = at(0, cell);
      cy = at(1, cell);
      w = at(2, cell);
      h = at(3, cell);
    }
  }
  // softmax normalisation of the cell (decode_fwd_kernel's mx / sum), only for DEC && inference
  __device__ __forceinline__ void softmax_terms(int cell, float& mx, float& sum) const {
    const int C = p.P - 5;
    mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, at(5 + c, cell));
    sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(at(5 + c, cell) - mx);
  }
  // NaN-propagating max over the class channels, like torch.max
  __device__ __forceinline__ float max_cls(int cell) const {
    const int C = p.P - 5;
    float smx = 0.f, ssum = 1.f;
    const bool sm = DEC && p.inference;
    if (sm) softmax_terms(cell, smx, ssum);
    float mx = sm ? expf(at(5, cell) - smx) / ssum : at(5, cell);
    for (int c = 1; c < C; ++c) {
      const float v = sm ? expf(at(5 + c, cell) - smx) / ssum : at(5 + c, cell);
      mx = (v > mx || v != v) ? v : mx;
      if (mx != mx) break;
    }
    return mx;
  }
  // channels 4.. of the output row
  __device__ __forceinline__ void tail(int cell, float* dst) const {
    const int P = p.P;
    dst[4] = obj(cell);
    if (DEC && p.inference) {
      float smx, ssum;
      softmax_terms(cell, smx, ssum);
      for (int c = 5; c < P; ++c) dst[c] = expf(at(c, cell) - smx) / ssum;
    } else {
      for (int c = 5; c < P; ++c) dst[c] = at(c, cell);
    }
  }
};

// THE SUPPRESSION TEST.  torchvision's CPU kernel (restated in oracle/yogo_oracle.py:324-361) suppresses a box when
// (double) fl32(inter / union) > iou_thresh: an fp32 division (~10 instructions), a float -> double conversion and an fp64 compare per
// pair of boxes -- half of the suppression loop of the dense case (every cell fires: ~47 M pairs per image).  For the threshold every caller
// uses, 0.5 (yogo/utils/prediction_formatting.py:27 iou_thresh default, yogo/infer.py), the same predicate in four fp32 instructions:
// the floats just above 0.5 are 0.5 + k 2^-24, so with round-to-nearest-even (monotone; a quotient exactly half way rounds DOWN to 0.5,
// whose mantissa is even)
//     fl32(inter / union) > 0.5  <=>  inter / union > 0.5 + 2^-25  <=>  inter - 0.5 union > 2^-25 union        (union > 0)
// where 0.5 union and 2^-25 union are exact (union >= 2^-100) and d = fl32(inter - 0.5 union) is exact whenever inter lies within
// [0.25, 1] union (Sterbenz) and keeps its sign and more than 2^-25 union of magnitude outside that range; an infinite union makes both
// sides infinite or NaN (false, like the quotient); for a zero union the test reads inter > 0 (the quotient is +inf, or NaN for 0 / 0); a
// negative or NaN union never suppresses.  Only 0 < union < 2^-100 and every other threshold take the division: same truth value for every input
// (tests/test_gpu_parity.py, test_gpu_infer_fused.py: torch.equal to the oracle, degenerate boxes and thresholds 0.01 / 0.999 included).
// Measured on the way (dense batch of 256, one box, gpurun_out/r6_nms_ab*.log; 16.2 ms at the start): the general-threshold form of the
// idea in double precision (inter >= M * union with M the rounding boundary: exact too) 20.6 ms -- two conversions and an fp64 multiply
// cost more than the fp32 division here; the kept boxes in the outer loop (one LDS read per kept box, not per pair) 17.0 ms; a per-lane
// select between the two forms 18.8 ms (hipcc computes both); the uniform loop below with bare v_max / v_min 14.1 ms; the division
// path out of line (inlined into the 16 slots it spilled 250 registers) and the cell array out of the loop's live range: 12.0 ms.

__device__ __forceinline__ unsigned monotone_key(float s) {
  if (s != s) return 0xFFFFFFFFu;  // NaN sorts first (torch: NaN is the largest)
  s = s + 0.0f;                    // -0 -> +0
  const unsigned u = __float_as_uint(s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// exclusive block scan of a 0/1 flag; returns this thread's rank and the block total (all threads)
__device__ __forceinline__ int block_rank(bool flag, int* sh_wave /*[NMS_WAVES+1]*/, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(flag);
  const int wrank = __popcll(bal & ((1ull << lane) - 1ull));
  __syncthreads();  // protect sh_wave reuse
  if (lane == 0) sh_wave[wave] = __popcll(bal);
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NMS_WAVES; ++w) {
    const int c = sh_wave[w];
    if (w < wave) base += c;
    tot += c;
  }
  *total = tot;
  return base + wrank;
}

template <bool DEC>
__global__ __launch_bounds__(NMS_THREADS) void nms_batched_kernel(const NmsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem_raw);  // [npow2] during the sort
  __shared__ int sh_wave[NMS_WAVES + 1];
  __shared__ int sh_cnt[NMS_SLOTS][NMS_WAVES];
  __shared__ float4 kept_box[64];
  __shared__ float kept_area[64];
  __shared__ int kept_n;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int cells = p.cells, P = p.P;
  const PredView<DEC> pv{p.pred + (size_t)b * P * cells, p};
  int* ws_cells = p.ws_cells + (size_t)b * cells;
  float4* ws_box = p.ws_box + (size_t)b * cells;
  float* ws_area = p.ws_area + (size_t)b * cells;

  // ---- 1. ordered compaction of candidates --------------------------------------------------------------------
  // every lane's objectness loads are issued before the first use (one round trip instead of one per 1024 cells), the 16 x 16
  // per-(slot, wavefront) counts go to LDS in one step, and each lane sums the prefix of its own slots from there
  unsigned fire = 0u;
  {
    float ob[NMS_SLOTS];
#pragma unroll
    for (int k = 0; k < NMS_SLOTS; ++k) {
      const int cell = k * NMS_THREADS + tid;
      ob[k] = (cell < cells) ? pv.at(4, cell) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NMS_SLOTS; ++k) {
      const int cell = k * NMS_THREADS + tid;
      const float o = DEC ? nms_sigmoid(ob[k]) : ob[k];
      const bool f = (cell < cells) && (o > p.obj_thresh);
      const unsigned long long bal = __ballot(f);
      if (f) fire |= 1u << k;
      if (lane == 0) sh_cnt[k][wave] = __popcll(bal);
    }
  }
  __syncthreads();
  int n = 0;
#pragma unroll
  for (int k = 0; k < NMS_SLOTS; ++k) {
    if (k * NMS_THREADS >= cells) break;  // uniform
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NMS_WAVES; ++w) {
      const int c = sh_cnt[k][w];
      if (w < wave) base += c;
      tot += c;
    }
    const bool f = ((fire >> k) & 1u) != 0u;
    const unsigned long long bal = __ballot(f);
    // rank inside the wavefront = fired lanes below this one in slot k
    if (f) ws_cells[n + base + __popcll(bal & ((1ull << lane) - 1ull))] = k * NMS_THREADS + tid;
    n += tot;
  }
  __syncthreads();
  // (global writes of ws_cells are read back by other lanes of this workgroup)
  __threadfence_block();
  __syncthreads();

  float bx1[NMS_SLOTS], by1[NMS_SLOTS], bx2[NMS_SLOTS], by2[NMS_SLOTS], bar[NMS_SLOTS];
  unsigned alive = 0;  // bit k: slot k holds a live candidate

  if (p.do_nms && n > 0) {
    // ---- 2. keys + bitonic sort (descending) ----------------------------------------------------------------
    int npow2 = 64;
    while (npow2 < n) npow2 <<= 1;
    for (int i = tid; i < npow2; i += NMS_THREADS) {
      unsigned long long key = 0ull;
      if (i < n) {
        const int cell = ws_cells[i];
        const float score = pv.max_cls(cell) * pv.obj(cell);
        key = ((unsigned long long)monotone_key(score) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
      }
      keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < npow2; i += NMS_THREADS) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long a = keys[i], c = keys[ixj];
            const bool desc = (i & k) == 0;  // descending blocks first -> overall descending
            if (desc ? (a < c) : (a > c)) {
              keys[i] = c;
              keys[ixj] = a;
            }
          }
        }
        __syncthreads();
      }
    }
    // ---- sorted candidates -> registers + global sorted arrays ------------------------------------------------
#pragma unroll
    for (int k = 0; k < NMS_SLOTS; ++k) {
      const int j = tid + k * NMS_THREADS;
      bx1[k] = by1[k] = bx2[k] = by2[k] = bar[k] = 0.f;
      if (j < n) {
        const unsigned ci = 0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull);
        const int cell = ws_cells[ci];
        float cx, cy, w, h;
        pv.box(cell, cx, cy, w, h);
        bx1[k] = cx - 0.5f * w;
        by1[k] = cy - 0.5f * h;
        bx2[k] = cx + 0.5f * w;
        by2[k] = cy + 0.5f * h;
        bar[k] = (bx2[k] - bx1[k]) * (by2[k] - by1[k]);
        ws_box[j] = make_float4(bx1[k], by1[k], bx2[k], by2[k]);
        ws_area[j] = bar[k];
        alive |= 1u << k;
      }
    }
    __threadfence_block();
    __syncthreads();

    // ---- 3. greedy suppression, one 64-candidate chunk at a time ------------------------------------------------
    const int nchunk = (n + 63) >> 6;
    for (int c = 0; c < nchunk; ++c) {
      const int ow = c & (NMS_WAVES - 1);  // owner wavefront
      const int ok = c >> 4;               // its slot
      if (wave == ow) {
        const int j = c * 64 + lane;
        bool live = ((alive >> ok) & 1u) != 0u;  // implies j < n
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
        float ar = 0.f;
        if (j < n) {
          bb = ws_box[j];
          ar = ws_area[j];
        }
        unsigned long long kept = 0ull;
        unsigned long long todo = __ballot(live);
        while (todo != 0ull) {
          const int i = __builtin_ctzll(todo);
          kept |= 1ull << i;
          const float ix1 = __shfl(bb.x, i, 64), iy1 = __shfl(bb.y, i, 64);
          const float ix2 = __shfl(bb.z, i, 64), iy2 = __shfl(bb.w, i, 64);
          const float iar = __shfl(ar, i, 64);
          if (live && lane > i) {
            const float xx1 = fmaxf(ix1, bb.x), yy1 = fmaxf(iy1, bb.y);
            const float xx2 = fminf(ix2, bb.z), yy2 = fminf(iy2, bb.w);
            const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
            const float inter = w * h;
            // (the reference's own form here: this serial part is latency-bound, the division-free test pays in the throughput-bound part below)
            if ((double)(inter / (iar + ar - inter)) > p.iou_thresh) live = false;
          }
          const unsigned long long above = (i == 63) ? 0ull : (~0ull << (i + 1));
          todo = __ballot(live) & above;
        }
        if (!live) alive &= ~(1u << ok);
        if ((kept >> lane) & 1ull) {
          const int pos = __popcll(kept & ((1ull << lane) - 1ull));
          kept_box[pos] = bb;
          kept_area[pos] = ar;
        }
        if (lane == 0) kept_n = __popcll(kept);
      }
      __syncthreads();
      const int kn = kept_n;
      const int jmin = (c + 1) * 64;
      unsigned odd_slots = 0u;   // slots of this lane whose candidate met a union outside the division-free test's range in this chunk
      if (kn > 0) {
        // Every later live candidate against the chunk's kept boxes.  The loop over the kept boxes is UNIFORM (no per-lane break: with
        // one, hipcc wraps every iteration in exec-mask bookkeeping -- 45 instructions per pair for 18 of arithmetic) and the four
        // max / min of the intersection are bare v_max_f32 / v_min_f32 (fmaxf / fminf canonicalise both operands first -- four extra
        // instructions per pair for values that come out of arithmetic and are never signalling NaNs; for every other input the
        // instruction IS fmaxf / fminf).  A suppressed candidate stays suppressed: the surplus tests change nothing.
        // (measured: skipping the rest when inter == 0 -- exact, the threshold is positive -- makes the dense case 14 % SLOWER: some lane
        //  of the 64 nearly always overlaps, so the wave pays both sides of the branch)
#pragma unroll
        for (int k = 0; k < NMS_SLOTS; ++k) {
          if ((k + 1) * NMS_THREADS <= jmin || k * NMS_THREADS >= n) continue;   // uniform: nobody behind this chunk in slot k
          const int j = tid + k * NMS_THREADS;
          const bool mine = ((alive >> k) & 1u) && j >= jmin;
          if (__ballot(mine) == 0ull) continue;   // uniform
          if (!p.iou_half) {   // (uniform) another threshold than 0.5: the division for every pair, below
            if (mine) odd_slots |= 1u << k;
            continue;
          }
          const float cx1 = bx1[k], cy1 = by1[k], cx2 = bx2[k], cy2 = by2[k], ca = bar[k];
          // the hot loop carries the division-free test only; a lane that meets a union outside its range (degenerate boxes: never in
          // a decoded prediction) is marked and goes through its candidate again with the division, below
          bool dead = false, odd = false;
          // where the division-free test is exact: union >= 2^-100 -- an infinite union too (both sides of its comparison become -inf /
          // +inf or NaN: false, like the quotient 0 or NaN) -- and union == 0 (it reads inter > 0: the quotient +inf, or NaN for 0 / 0).  A
          // negative or NaN union never suppresses (the quotient is <= -0 or NaN).  What is left for the division is 0 < union < 2^-100.
          // (A random-init network's predictions hold zero-area and infinite boxes: with those on the division path nearly every chunk
          //  sent whole wavefronts through both loops -- 18.8 instead of 8.3 ms per batch of 256.)
          const float u_lo = 0x1p-100f;
#pragma unroll NMS_UNROLL
          for (int i = 0; i < kn; ++i) {
            const float4 kb = kept_box[i];
            const float ka = kept_area[i];
            float xx1, yy1, xx2, yy2;
            asm("v_max_f32 %0, %1, %2" : "=v"(xx1) : "v"(kb.x), "v"(cx1));
            asm("v_max_f32 %0, %1, %2" : "=v"(yy1) : "v"(kb.y), "v"(cy1));
            asm("v_min_f32 %0, %1, %2" : "=v"(xx2) : "v"(kb.z), "v"(cx2));
            asm("v_min_f32 %0, %1, %2" : "=v"(yy2) : "v"(kb.w), "v"(cy2));
            const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
            const float inter = w * h;
            const float uni = ka + ca - inter;
            const bool big = uni >= u_lo;
            dead |= (big || uni == 0.f) && (inter - 0.5f * uni > 0x1p-25f * uni);
            odd |= uni > 0.f && !big;
          }
          if (mine && dead) alive &= ~(1u << k);
          if (mine && odd) odd_slots |= 1u << k;
        }
        // the marked candidates again, with the division (ONE copy of this loop, the candidate's box out of the sorted arrays: inlined
        // into each of the 16 slots above it cost the kernel 250 spilled registers)
        if (__ballot(odd_slots != 0u) != 0ull) {   // (uniform)
#pragma nounroll
          for (int k = 0; k < NMS_SLOTS; ++k) {
            if (!((odd_slots >> k) & 1u)) continue;
            const int j = tid + k * NMS_THREADS;
            const float4 cb = ws_box[j];
            const float ca = ws_area[j];
            bool dead = false;
            for (int i = 0; i < kn; ++i) {
              const float4 kb = kept_box[i];
              const float xx1 = fmaxf(kb.x, cb.x), yy1 = fmaxf(kb.y, cb.y);
              const float xx2 = fminf(kb.z, cb.z), yy2 = fminf(kb.w, cb.w);
              const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
              const float inter = w * h;
              dead |= (double)(inter / (kept_area[i] + ca - inter)) > p.iou_thresh;
            }
            if (dead) alive &= ~(1u << k);
          }
        }
      }
      __syncthreads();
    }
  } else {
    // no NMS: candidates stay in cell order
#pragma unroll
    for (int k = 0; k < NMS_SLOTS; ++k) {
      const int j = tid + k * NMS_THREADS;
      if (j < n) alive |= 1u << k;
    }
  }

  // ---- 4. class-confidence filter + ordered output --------------------------------------------------------------
  int nout = 0;
  float* rows = p.out_rows + (size_t)b * p.cap * P;
  long long* ocells = p.out_cells + (size_t)b * p.cap;
#pragma unroll
  for (int k = 0; k < NMS_SLOTS; ++k) {
    if (k * NMS_THREADS >= n) break;  // uniform
    bool keep = ((alive >> k) & 1u) != 0u;
    // (the candidate's cell is looked up again here -- sorted position -> candidate -> cell -- instead of living in 16 registers per lane
    //  through the suppression loop, whose 5 x 16 box registers leave the 128-register budget of a 1024-lane workgroup no slack)
    const int jj = tid + k * NMS_THREADS;
    int cell = -1;
    if (jj < n) cell = (p.do_nms && n > 0) ? ws_cells[0xFFFFFFFFu - (unsigned)(keys[jj] & 0xFFFFFFFFull)] : ws_cells[jj];
    if (keep && p.use_cls_filter) keep = pv.max_cls(cell) > p.min_cls;
    int tot;
    const int r = block_rank(keep, sh_wave, &tot);
    if (keep) {
      const int pos = nout + r;
      if (pos < p.cap) {
        float* dst = rows + (size_t)pos * P;
        float cx, cy, w, h;
        pv.box(cell, cx, cy, w, h);
        if (p.xyxy) {
          dst[0] = cx - 0.5f * w;
          dst[1] = cy - 0.5f * h;
          dst[2] = cx + 0.5f * w;
          dst[3] = cy + 0.5f * h;
        } else {
          dst[0] = cx;
          dst[1] = cy;
          dst[2] = w;
          dst[3] = h;
        }
        pv.tail(cell, dst);
        ocells[pos] = (long long)cell;
      }
    }
    nout += tot;
  }
  if (tid == 0) p.out_count[b] = nout;
}

// =========================================================================================================
// C ABI
// =========================================================================================================
extern "C" int yogo_format_preds_workspace_bytes(int B, int Sy, int Sx, size_t* bytes) {
  YOGO_CHECK_ARG(bytes && B > 0 && Sy > 0 && Sx > 0, "format_preds_workspace_bytes: bad arguments");
  const size_t cells = (size_t)Sy * Sx;
  *bytes = (size_t)B * cells * (sizeof(int) + sizeof(float4) + sizeof(float)) + 64;
  return YOGO_OK;
}

template <bool DEC>
static int launch_format_preds(NmsParams p, void* workspace, double obj_thresh, double iou_thresh, int box_format,
                               double min_class_confidence_threshold, hipStream_t stream, const char* what) {
  const int B = p.B, cells = p.cells;
  char* ws = reinterpret_cast<char*>(workspace);
  // float4 array first (16-byte alignment), then floats, then ints
  size_t off = (16 - (reinterpret_cast<uintptr_t>(ws) & 15)) & 15;
  p.ws_box = reinterpret_cast<float4*>(ws + off);
  off += (size_t)B * cells * sizeof(float4);
  p.ws_area = reinterpret_cast<float*>(ws + off);
  off += (size_t)B * cells * sizeof(float);
  p.ws_cells = reinterpret_cast<int*>(ws + off);
  p.obj_thresh = (float)obj_thresh;
  p.iou_thresh = iou_thresh;
  p.iou_half = iou_thresh == 0.5;
  p.min_cls = (float)min_class_confidence_threshold;
  p.do_nms = iou_thresh > 0.0;
  p.use_cls_filter = min_class_confidence_threshold > 0.0;
  p.xyxy = box_format;
  int npow2 = 64;
  while (npow2 < cells) npow2 <<= 1;
  const size_t lds = p.do_nms ? (size_t)npow2 * sizeof(unsigned long long) : 0;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_batched_kernel<DEC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              NMS_MAX_CELLS * 8);
    attr_set = true;
  }
  yogo_launch_log("nms_batched_kernel<%s> | B=%d cells=%d P=%d nms=%d", DEC ? "true" : "false", B, cells, p.P, p.do_nms);
  hipLaunchKernelGGL(nms_batched_kernel<DEC>, dim3(B), dim3(NMS_THREADS), lds, stream, p);
  YOGO_CHECK_LAUNCH(what);
  return YOGO_OK;
}

// pred [B][5+C][Sy][Sx] fp32 -> per image: count, kept rows [cap][5+C] in reference order, kept cell indices.
// box_format: 0 = cxcywh, 1 = xyxy.  iou_thresh <= 0 disables NMS, min_class_confidence <= 0 disables the filter.
extern "C" int yogo_format_preds_batched(const float* pred, float* out_rows, long long* out_cells, int* out_count,
                                         void* workspace, int B, int P, int Sy, int Sx, int cap, double obj_thresh,
                                         double iou_thresh, int box_format, double min_class_confidence_threshold,
                                         hipStream_t stream) {
  YOGO_CHECK_ARG(pred && out_rows && out_cells && out_count && workspace, "format_preds_batched: null pointer");
  YOGO_CHECK_ARG(B > 0 && P > 5 && Sy > 0 && Sx > 0 && cap > 0, "format_preds_batched: bad shape");
  YOGO_CHECK_ARG(box_format == 0 || box_format == 1, "invalid box format %d; valid box formats are 0 (cxcywh), 1 (xyxy)", box_format);
  const int cells = Sy * Sx;
  YOGO_CHECK_ARG(cells <= NMS_MAX_CELLS, "format_preds_batched: Sy*Sx = %d exceeds the supported %d cells", cells, NMS_MAX_CELLS);
  NmsParams p{};
  p.pred = pred; p.out_rows = out_rows; p.out_cells = out_cells; p.out_count = out_count;
  p.B = B; p.P = P; p.cells = cells; p.cap = cap;
  return launch_format_preds<false>(p, workspace, obj_thresh, iou_thresh, box_format, min_class_confidence_threshold, stream,
                                    "format_preds_batched");
}

// raw [B][5+C][Sy][Sx] fp32 = the head's output BEFORE the decode; everything else as yogo_format_preds_batched, whose result on
// yogo_decode_fwd(raw) this reproduces bit for bit (rows, cells, counts) without the decoded tensor going through memory.
extern "C" int yogo_decode_format_preds_batched(const float* raw, const float* cxs, const float* cys, float* out_rows,
                                                long long* out_cells, int* out_count, void* workspace, int B, int P, int Sy,
                                                int Sx, int cap, float anchor_w, float anchor_h, float width_multiplier,
                                                float height_multiplier, int inference, double obj_thresh, double iou_thresh,
                                                int box_format, double min_class_confidence_threshold, hipStream_t stream) {
  YOGO_CHECK_ARG(raw && cxs && cys && out_rows && out_cells && out_count && workspace, "decode_format_preds_batched: null pointer");
  YOGO_CHECK_ARG(B > 0 && P > 5 && Sy > 0 && Sx > 0 && cap > 0, "decode_format_preds_batched: bad shape");
  YOGO_CHECK_ARG(box_format == 0 || box_format == 1, "invalid box format %d; valid box formats are 0 (cxcywh), 1 (xyxy)", box_format);
  const int cells = Sy * Sx;
  YOGO_CHECK_ARG(cells <= NMS_MAX_CELLS, "decode_format_preds_batched: Sy*Sx = %d exceeds the supported %d cells", cells, NMS_MAX_CELLS);
  NmsParams p{};
  p.pred = raw; p.out_rows = out_rows; p.out_cells = out_cells; p.out_count = out_count;
  p.B = B; p.P = P; p.cells = cells; p.cap = cap;
  p.cxs = cxs; p.cys = cys;
  p.inv_sx = (float)(1.0 / Sx); p.inv_sy = (float)(1.0 / Sy);   // as yogo_decode_fwd
  p.anchor_w = anchor_w; p.anchor_h = anchor_h; p.wmul = width_multiplier; p.hmul = height_multiplier;
  p.inference = inference;
  return launch_format_preds<true>(p, workspace, obj_thresh, iou_thresh, box_format, min_class_confidence_threshold, stream,
                                   "decode_format_preds_batched");
}
