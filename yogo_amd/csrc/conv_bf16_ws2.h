// Persistent, wavefront-specialised data gradient of a stride-2 3x3 convolution with 128 input channels (conv_bf16_ws2.hip):
// dx [B][128][OH][OW] from dy [B][K][ceil(OH/2)][ceil(OW/2)], K = 32 ... 128 (autograd of yogo/model_defns.py:54-56).
#pragma once
#include "common.h"

// 8 wavefronts per workgroup, one workgroup per CU: wavefronts 0-3 COMPUTE and store (one per SIMD), wavefronts 4-7 LOAD (w + 4
// shares w's SIMD).  A workgroup tile is 128 consecutive output QUADS (2x2 output pixels = one dy pixel) of a column band, all
// 128 channels; a compute wavefront owns 64 channels x 64 quads x both column parities of ONE row parity at a time (8 accumulator
// tiles of 32x32) and runs the tile's two row parities one after the other off ONE staged dy tile.
#define W2_PT 128                      // quads per workgroup tile
#define W2_WB (6 * 4096)               // a period's weight slices: 6 groups of [2 channel blocks][128 channels] units (24 576)
#define W2_DYS (2 * 256 * 16)          // a dy slot: one 16-channel chunk of the tile, [2 channel blocks][256 units] (8 192)
#define W2_NSLOT 8                     // dy slots = 16-channel chunks of the contraction (K <= 128)
#define W2_NWB 3                       // weight buffers: a period's slices are requested TWO periods ahead
#define W2_DY (W2_NWB * W2_WB)         // first dy slot
#define W2_SG (W2_DY + W2_NSLOT * W2_DYS)   // LeakyReLU sign bytes: [pass A | pass B][4 wavefronts][n][px][64 lanes] dwords (2 x 4 KB)
#define W2_ES (W2_SG + 2 * 4096)       // [tile parity][128] fp32 channel scale of the tile's image
#define W2_LDS_BYTES (W2_ES + 1024)    // 148 480

struct ConvWs2Params {
  const void* in;     // dy: bf16 NCHW8c [B][Kb][IH][IW] units, IH = ceil(OH / 2), IW = ceil(OW / 2)
  const void* wp;     // packed weights, mode 2: [9 slices in parity-class order][Kb][128] units
  void* out;          // dx: bf16 NCHW8c [B][16][OH][OW] units
  const unsigned char* signs;   // optional LeakyReLU sign map of the block output dx flows into (ConvBf16Params::signs), or null
  const float* chan_scale;      // optional [B][128]
  int B, Kb, IH, IW, OH, OW;
  int ncb, TW, tiles_per_band, gx, ntiles;
  unsigned m_gx, m_tpb, m_bw, m_bwl, m_lw, m_lwl;
  int nck;            // 16-channel chunks of the contraction (Kb / 2; even, <= 8)
#ifdef YOGO_DIAG
  int dbg;
  unsigned long long* stamps;   // [workgroups][16]
#endif
};

// true when the kernel takes the launch (M = 128 output channels of the gradient, K = 64, 96 or 128)
bool conv_bf16_ws2_eligible(int K, int M, int OH, int OW, int B);
// fills the tiling part of `p`; false when no band count fits the fixed dy slot
bool conv_bf16_ws2_plan(ConvWs2Params* p);
int launch_conv_bf16_ws2(const ConvWs2Params& p, hipStream_t stream);
