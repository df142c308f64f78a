// Persistent, wavefront-specialised FORWARD of a stride-2 3x3 convolution with 128 output channels (conv_bf16_ws3.hip): layer 4 of
// base_model (yogo/model_defns.py:54-56).
#pragma once
#include "common.h"

// 8 wavefronts per workgroup, one workgroup per CU: wavefronts 0-3 COMPUTE and store (one per SIMD; 64 channels x 64 pixels = 4
// accumulator tiles of 32x32 each), wavefronts 4-7 LOAD.  A workgroup tile is 128 channels x 128 consecutive output pixels of a
// column band (at most 4 output rows of at most 47 columns).  The input tile of a 16-channel chunk is staged with its even and odd
// columns DE-INTERLEAVED -- [2 channel blocks][9 rows][even columns: 48 units | odd columns: 48 units] -- so that a tap's 32
// consecutive output pixels read 32 consecutive 16-byte units, at offsets that are compile-time immediates of the ds_read.
#define W3_PT 128                      // output pixels per workgroup tile
#define W3_ROWS 9                      // staged input rows (2 x 4 output rows + 1)
#define W3_PL 48                       // units of a column-parity plane of a staged row (band width <= 47)
#define W3_LW (2 * W3_PL)              // units of a staged row (1 536 bytes)
#define W3_KBU (W3_ROWS * W3_LW)       // units of a staged channel block (864)
#define W3_IB (2 * W3_KBU * 16)        // bytes of an input buffer: one 16-channel chunk (27 648)
#define W3_NI 7                        // input slots: 16-byte elements per loader lane and chunk (7 x 256 >= 1 728)
#define W3_WB (9 * 4096)               // bytes of a chunk's weight slices [tap][2 channel blocks][128 channels] (36 864)
#define W3_I0 (2 * W3_WB)              // first input buffer (ring of three: the input comes from HBM, requested two periods ahead)
#define W3_EB (W3_I0 + 3 * W3_IB)      // [128] fp32 bias
#define W3_ES (W3_EB + 512)            // [2][128] fp32 channel scale (by tile parity)
#define W3_LDS_BYTES (W3_ES + 1024)    // 158 208

struct ConvWs3Params {
  const void* in;     // bf16 NCHW8c [B][Kb][IH][IW] units
  const void* wp;     // packed weights (forward): [9][Kb][128] units
  const float* bias;  // [128] or null
  void* out;          // bf16 NCHW8c [B][16][OH][OW] units, OH = (IH - 1) / 2 + 1
  unsigned char* signs;     // optional LeakyReLU sign map of the output (ConvBf16Params::signs)
  const float* chan_scale;  // optional [B][128]
  int B, Kb, IH, IW, OH, OW;
  int ncb, TW, tiles_per_band, gx, ntiles;
  int PT;             // output pixels of a tile: 128, fewer in bands so narrow that 128 consecutive pixels would touch more than 4 rows
  unsigned m_gx, m_tpb, m_bw, m_bwl;
  int nchunk, act;
#ifdef YOGO_DIAG
  int dbg;
  unsigned long long* stamps;   // [workgroups][16]
#endif
};

// true when the kernel takes the launch (M = 128, K a multiple of 32 and >= 64, ACT_NONE / ACT_LEAKY)
bool conv_bf16_ws3_eligible(int K, int M, int IH, int IW, int B);
// fills the tiling part of `p`; false when no band width fits the fixed staging layout
bool conv_bf16_ws3_plan(ConvWs3Params* p);
int launch_conv_bf16_ws3(const ConvWs3Params& p, hipStream_t stream);
