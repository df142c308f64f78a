// The step in front of the hot path, on the device: the label rasteriser (yogo/data/yogo_dataset.py:24-46,
// format_labels_tensor) and the batch flips with their bounding boxes (yogo/data/data_transforms.py:51-98), which the
// reference runs per image / per batch inside DataLoader workers.  Both are pure HBM traffic.
#include "common.h"

namespace {

constexpr int BAD_CELL = -2147483647 - 1;

// One workgroup per image.  The reference writes its labels one after the other, so a later label of the same cell
// overwrites an earlier one: the winner of a cell is the label with the LARGEST index (atomicMax in LDS), then only winners
// write.  Cell of a label: i = floor((x1 + x2) * Sx / 2), j likewise -- torch's float floor-division by 2 is exact, and so
// is floorf(a * 0.5f); negative indices wrap once like Python's, anything else is an IndexError in the reference and sets
// *status to 1 + the label's global index (first offender wins, atomicCAS).
__global__ __launch_bounds__(256) void labels_rasterize_kernel(const float* __restrict__ labels, const int* __restrict__ offsets,
                                                               float* __restrict__ out, int* __restrict__ status, int Sx, int Sy,
                                                               int cxcywh) {
  extern __shared__ int win[];
  const int b = blockIdx.x, tid = threadIdx.x, cells = Sx * Sy;
  const int l0 = offsets[b], l1 = offsets[b + 1];
  float* o = out + (size_t)b * 6 * cells;
  for (int e = tid; e < 6 * cells; e += blockDim.x) o[e] = 0.f;
  for (int e = tid; e < cells; e += blockDim.x) win[e] = -1;
  __syncthreads();
  for (int pass = 0; pass < 2; ++pass) {
    for (int l = l0 + tid; l < l1; l += blockDim.x) {
      const float* r = labels + (size_t)l * 5;
      float x1 = r[1], y1 = r[2], x2 = r[3], y2 = r[4];
      if (cxcywh) {  // torchvision.ops.box_convert(cxcywh -> xyxy), yogo_dataset.py:132
        const float cx = x1, cy = y1, hw = 0.5f * x2, hh = 0.5f * y2;
        x1 = cx - hw; y1 = cy - hh; x2 = cx + hw; y2 = cy + hh;
      }
      const float fi = floorf((x1 + x2) * (float)Sx * 0.5f), fj = floorf((y1 + y2) * (float)Sy * 0.5f);
      // (NaN and out-of-range values fail the comparisons below)
      int i = (fi >= -(float)Sx && fi < (float)Sx) ? (int)fi : BAD_CELL;
      int j = (fj >= -(float)Sy && fj < (float)Sy) ? (int)fj : BAD_CELL;
      if (i == BAD_CELL || j == BAD_CELL) {
        if (pass == 0) atomicCAS(status, 0, l + 1);
        continue;
      }
      if (i < 0) i += Sx;
      if (j < 0) j += Sy;
      const int cell = j * Sx + i;
      if (pass == 0) {
        atomicMax(&win[cell], l);
      } else if (win[cell] == l) {
        o[cell] = 1.f;
        o[cells + cell] = x1;
        o[2 * cells + cell] = y1;
        o[3 * cells + cell] = x2;
        o[4 * cells + cell] = y2;
        o[5 * cells + cell] = r[0];
      }
    }
    __syncthreads();
  }
}

// images: [planes][H][W] elements of EB bytes, planes = B * C.  One lane per 8-byte group of a row when the row is a whole
// number of groups (772 x 1032 uint8: 129 groups), else per element.
template <int EB>
__global__ __launch_bounds__(256) void flip_images_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                          long long planes, int H, int W, int hflip, int vflip) {
  const long long rowb = (long long)W * EB;
  if (rowb % 8 == 0) {
    const int G = (int)(rowb / 8);
    const long long total = planes * H * G;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
      const int g = (int)(e % G);
      const long long row = e / G;
      const int y = (int)(row % H);
      const long long pl = row / H;
      const int sy = vflip ? H - 1 - y : y, sg = hflip ? G - 1 - g : g;
      unsigned long long v = *reinterpret_cast<const unsigned long long*>(in + ((pl * H + sy) * G + sg) * 8);
      if (hflip) {
        if (EB == 1) v = __builtin_bswap64(v);
        else v = (v << 32) | (v >> 32);
      }
      *reinterpret_cast<unsigned long long*>(out + e * 8) = v;
    }
  } else {
    const long long total = planes * H * W;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
      const int x = (int)(e % W);
      const long long row = e / W;
      const int y = (int)(row % H);
      const long long pl = row / H;
      const int sy = vflip ? H - 1 - y : y, sx = hflip ? W - 1 - x : x;
      const unsigned char* s = in + ((pl * H + sy) * W + sx) * EB;
      unsigned char* d = out + e * EB;
#pragma unroll
      for (int k = 0; k < EB; ++k) d[k] = s[k];
    }
  }
}

// labels [B][6][Sy][Sx]: a horizontal flip mirrors the grid along x and maps (x1, x2) -> (1 - x2, 1 - x1) in EVERY cell (the
// reference does not look at the mask: empty cells end up holding 1.0), a vertical flip does the same along y.
__global__ __launch_bounds__(256) void flip_labels_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int Sy, int Sx,
                                                          int hflip, int vflip) {
  const int cells = Sx * Sy;
  const long long total = (long long)B * 6 * cells;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(e % Sx);
    const int j = (int)((e / Sx) % Sy);
    const int ch = (int)((e / cells) % 6);
    const long long b = e / (6LL * cells);
    int sch = ch;
    bool comp = false;
    if (hflip && (ch == 1 || ch == 3)) { sch = 4 - ch; comp = true; }
    if (vflip && (ch == 2 || ch == 4)) { sch = 6 - ch; comp = true; }
    const int sj = vflip ? Sy - 1 - j : j, si = hflip ? Sx - 1 - i : i;
    const float v = in[((b * 6 + sch) * Sy + sj) * Sx + si];
    out[e] = comp ? 1.f - v : v;
  }
}

}  // namespace

// format_labels_tensor (yogo/data/yogo_dataset.py:24-46) for a whole batch.  labels: [N][5] fp32 rows (class, x1, y1, x2, y2)
// -- or (class, xc, yc, w, h) with box_format = 1, converted like label_file_to_tensor does (:132) -- of all images back to
// back; offsets: [B + 1] int32 row ranges; out: [B][6][Sy][Sx] fp32, written completely (mask, x1, y1, x2, y2, class).
// status: one device int32 the caller zeroes; left at 0 or set to 1 + the index of a label whose cell lies outside the grid
// (the reference raises IndexError there).
extern "C" int yogo_labels_rasterize(const float* labels, const int* offsets, float* out, int* status, int B, int Sx, int Sy,
                                     int box_format, hipStream_t stream) {
  YOGO_CHECK_ARG(offsets && out && status && B >= 0 && Sx > 0 && Sy > 0 && (box_format == 0 || box_format == 1),
                 "labels_rasterize: bad arguments");
  const size_t lds = (size_t)Sx * Sy * sizeof(int);
  YOGO_CHECK_ARG(lds <= 160 * 1024, "labels_rasterize: grid of %d x %d cells does not fit the LDS", Sx, Sy);
  if (B == 0) return YOGO_OK;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&labels_rasterize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(labels_rasterize_kernel, dim3(B), dim3(256), lds, stream, labels, offsets, out, status, Sx, Sy, box_format);
  YOGO_CHECK_LAUNCH("labels_rasterize");
  return YOGO_OK;
}

// RandomHorizontalFlipWithBBs / RandomVerticalFlipWithBBs (yogo/data/data_transforms.py:51-98) applied to a batch, both flips
// in one pass; the caller draws the two decisions.  img: [B][C][H][W] of elem_bytes 1 (uint8) or 4 (float32); lab:
// [B][6][Sy][Sx] fp32.  Out of place (in != out); either pair may be NULL.
extern "C" int yogo_flip_batch(const void* img_in, void* img_out, int elem_bytes, const float* lab_in, float* lab_out, int B, int C,
                               int H, int W, int Sy, int Sx, int hflip, int vflip, hipStream_t stream) {
  YOGO_CHECK_ARG((img_in == nullptr) == (img_out == nullptr) && (lab_in == nullptr) == (lab_out == nullptr), "flip_batch: unpaired pointers");
  YOGO_CHECK_ARG(img_in == nullptr || (img_in != img_out && (elem_bytes == 1 || elem_bytes == 4) && C > 0 && H > 0 && W > 0),
                 "flip_batch: bad image arguments");
  YOGO_CHECK_ARG(lab_in == nullptr || (lab_in != lab_out && Sy > 0 && Sx > 0), "flip_batch: bad label arguments");
  YOGO_CHECK_ARG(B >= 0, "flip_batch: bad batch size");
  if (B == 0) return YOGO_OK;
  if (img_in != nullptr) {
    const long long planes = (long long)B * C;
    const long long work = planes * H * ((long long)W * elem_bytes % 8 == 0 ? (long long)W * elem_bytes / 8 : W);
    const int grid = (int)((work + 255) / 256 < 65536 ? (work + 255) / 256 : 65536);
    if (elem_bytes == 1)
      hipLaunchKernelGGL((flip_images_kernel<1>), dim3(grid), dim3(256), 0, stream, (const unsigned char*)img_in, (unsigned char*)img_out,
                         planes, H, W, hflip, vflip);
    else
      hipLaunchKernelGGL((flip_images_kernel<4>), dim3(grid), dim3(256), 0, stream, (const unsigned char*)img_in, (unsigned char*)img_out,
                         planes, H, W, hflip, vflip);
    YOGO_CHECK_LAUNCH("flip_batch (images)");
  }
  if (lab_in != nullptr) {
    const long long work = (long long)B * 6 * Sy * Sx;
    const int grid = (int)((work + 255) / 256 < 65536 ? (work + 255) / 256 : 65536);
    hipLaunchKernelGGL(flip_labels_kernel, dim3(grid), dim3(256), 0, stream, lab_in, lab_out, B, Sy, Sx, hflip, vflip);
    YOGO_CHECK_LAUNCH("flip_batch (labels)");
  }
  return YOGO_OK;
}
