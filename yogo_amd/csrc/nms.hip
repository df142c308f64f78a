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
#include "common.h"

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
  float min_cls;          // float32(min_class_confidence_threshold)
  int do_nms, use_cls_filter, xyxy;
};

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

__global__ __launch_bounds__(NMS_THREADS) void nms_batched_kernel(const NmsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem_raw);  // [npow2] during the sort
  __shared__ int sh_wave[NMS_WAVES + 1];
  __shared__ float4 kept_box[64];
  __shared__ float kept_area[64];
  __shared__ int kept_n;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int cells = p.cells, P = p.P, C = P - 5;
  const float* pred = p.pred + (size_t)b * P * cells;
  int* ws_cells = p.ws_cells + (size_t)b * cells;
  float4* ws_box = p.ws_box + (size_t)b * cells;
  float* ws_area = p.ws_area + (size_t)b * cells;

  // ---- 1. ordered compaction of candidates --------------------------------------------------------------------
  int n = 0;
  for (int base = 0; base < cells; base += NMS_THREADS) {
    const int cell = base + tid;
    const bool f = (cell < cells) && (pred[(size_t)4 * cells + cell] > p.obj_thresh);
    int tot;
    const int r = block_rank(f, sh_wave, &tot);
    if (f) ws_cells[n + r] = cell;
    n += tot;
  }
  __syncthreads();
  // (global writes of ws_cells are read back by other lanes of this workgroup)
  __threadfence_block();
  __syncthreads();

  int mycell[NMS_SLOTS];
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
        float mx = pred[(size_t)5 * cells + cell];
        for (int c = 1; c < C; ++c) {
          const float v = pred[(size_t)(5 + c) * cells + cell];
          mx = (v > mx || v != v) ? v : mx;  // NaN-propagating max like torch.max
          if (mx != mx) break;
        }
        const float score = mx * pred[(size_t)4 * cells + cell];
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
      mycell[k] = -1;
      bx1[k] = by1[k] = bx2[k] = by2[k] = bar[k] = 0.f;
      if (j < n) {
        const unsigned ci = 0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull);
        const int cell = ws_cells[ci];
        mycell[k] = cell;
        const float cx = pred[cell], cy = pred[(size_t)cells + cell];
        const float w = pred[(size_t)2 * cells + cell], h = pred[(size_t)3 * cells + cell];
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
            const float ovr = inter / (iar + ar - inter);
            if ((double)ovr > p.iou_thresh) live = false;
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
      if (kn > 0) {
#pragma unroll
        for (int k = 0; k < NMS_SLOTS; ++k) {
          const int j = tid + k * NMS_THREADS;
          if (((alive >> k) & 1u) && j >= jmin) {
            for (int i = 0; i < kn; ++i) {
              const float4 kb = kept_box[i];
              const float xx1 = fmaxf(kb.x, bx1[k]), yy1 = fmaxf(kb.y, by1[k]);
              const float xx2 = fminf(kb.z, bx2[k]), yy2 = fminf(kb.w, by2[k]);
              const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
              const float inter = w * h;
              const float ovr = inter / (kept_area[i] + bar[k] - inter);
              if ((double)ovr > p.iou_thresh) {
                alive &= ~(1u << k);
                break;
              }
            }
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
      mycell[k] = -1;
      if (j < n) {
        mycell[k] = ws_cells[j];
        alive |= 1u << k;
      }
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
    const int cell = mycell[k];
    if (keep && p.use_cls_filter) {
      float mx = pred[(size_t)5 * cells + cell];
      for (int c = 1; c < C; ++c) {
        const float v = pred[(size_t)(5 + c) * cells + cell];
        mx = (v > mx || v != v) ? v : mx;
        if (mx != mx) break;
      }
      keep = mx > p.min_cls;
    }
    int tot;
    const int r = block_rank(keep, sh_wave, &tot);
    if (keep) {
      const int pos = nout + r;
      if (pos < p.cap) {
        float* dst = rows + (size_t)pos * P;
        const float cx = pred[cell], cy = pred[(size_t)cells + cell];
        const float w = pred[(size_t)2 * cells + cell], h = pred[(size_t)3 * cells + cell];
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
        for (int c = 4; c < P; ++c) dst[c] = pred[(size_t)c * cells + cell];
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
  char* ws = reinterpret_cast<char*>(workspace);
  // float4 array first (16-byte alignment), then floats, then ints
  size_t off = (16 - (reinterpret_cast<uintptr_t>(ws) & 15)) & 15;
  p.ws_box = reinterpret_cast<float4*>(ws + off);
  off += (size_t)B * cells * sizeof(float4);
  p.ws_area = reinterpret_cast<float*>(ws + off);
  off += (size_t)B * cells * sizeof(float);
  p.ws_cells = reinterpret_cast<int*>(ws + off);
  p.B = B; p.P = P; p.cells = cells; p.cap = cap;
  p.obj_thresh = (float)obj_thresh;
  p.iou_thresh = iou_thresh;
  p.min_cls = (float)min_class_confidence_threshold;
  p.do_nms = iou_thresh > 0.0;
  p.use_cls_filter = min_class_confidence_threshold > 0.0;
  p.xyxy = box_format;
  int npow2 = 64;
  while (npow2 < cells) npow2 <<= 1;
  const size_t lds = p.do_nms ? (size_t)npow2 * sizeof(unsigned long long) : 0;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_batched_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        NMS_MAX_CELLS * 8);
    attr_set = true;
  }
  hipLaunchKernelGGL(nms_batched_kernel, dim3(B), dim3(NMS_THREADS), lds, stream, p);
  YOGO_CHECK_LAUNCH("format_preds_batched");
  return YOGO_OK;
}
