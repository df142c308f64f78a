// Persistent, wavefront-specialised form of the stride-1 3x3 bf16 convolution with 128 GEMM rows (conv_bf16_ws.hip).
#pragma once
#include "common.h"

// 8 wavefronts per workgroup, one workgroup per CU: wavefronts 0-3 COMPUTE (one per SIMD; 128 output channels x 64 pixels each
// = 8 accumulator tiles of 32x32 in 128 asm-owned AGPRs), wavefronts 4-7 LOAD (LDS-DMA of the next chunk, global stores of the
// previous tile's output) -- wavefronts w and w + 4 share a SIMD.  A workgroup tile is 128 channels x 256 consecutive pixels
// of a column band.
#define WS_PT 256                     // pixels per workgroup tile
#define WS_NT 256                     // lanes of a team
#define WS_NI 4                       // input slots (16-byte elements per loader lane and 16-channel chunk)
// LDS: two weight buffers (chunk parity), a ring of three input buffers (the input tile comes from HBM and is requested two
// chunk periods ahead, the weight slices are L2 hits and are requested one period ahead), the output staging area, bias / scale
#define WS_WB (9 * 2 * 128 * 16)      // bytes of a chunk's weight slices [tap][2 channel blocks][128 channels] (36 864)
#define WS_IB (WS_NI * WS_NT * 16)    // bytes of an input buffer (16 384)
#define WS_I0 (2 * WS_WB)             // first input buffer
#define WS_STG (WS_I0 + 3 * WS_IB)    // output staging: [2 regions][4 wavefronts][4 units][64 lanes] x 16 B = 32 KB
#define WS_EB (WS_STG + 32768)        // [128] fp32 bias
#define WS_ES (WS_EB + 512)           // [2][128] fp32 channel scale (by tile parity)
#define WS_MB (WS_ES + 1024)           // mailbox loader -> compute: [4 wavefronts][64 lanes] x 16 B (operand row addresses, output offsets of the NEXT tile)
#define WS_MBS (WS_MB + 4096)         // ... and its scalars: {there is a next tile, staged row pitch in bytes, image}
#define WS_LDS_BYTES (WS_MBS + 16)    // 161 296

struct ConvWsParams {
  const void* in;     // bf16 NCHW8c [B][Kb][IH][IW] units
  const void* wp;     // packed weights [9][Kb][128] units
  const float* bias;  // [128] or null
  void* out;          // bf16 NCHW8c [B][16][IH][IW] units
  unsigned char* signs;     // optional LeakyReLU sign map of the output (see ConvBf16Params::signs)
  const float* chan_scale;  // optional [B][128]
  int B, Kb, IH, IW;
  int ncb, TW, tiles_per_band, gx, ntiles;
  unsigned m_gx, m_tpb, m_bw, m_bwl, m_lw, m_lwl;
  int nchunk, act;
#ifdef YOGO_DIAG
  int dbg;                      // 1 = no output stores, 2 = no epilogue arithmetic, 4 = no DMA
  unsigned long long* stamps;   // [workgroups][16]
#endif
};

// true when the kernel takes the launch (stride 1, 3x3, M = 128, K a multiple of 32 and >= 64, lean epilogue)
bool conv_bf16_ws_eligible(int K, int M, int IH, int IW, int B);
// fills the tiling part of `p` (ncb, TW, ..., magic numbers); false when no tiling fits the kernel's fixed LDS layout
// (slots: 256-element DMA slots of a chunk's staged input tile the kernel has -- WS_NI for conv_bf16_ws_kernel)
bool conv_bf16_ws_plan(ConvWsParams* p, int slots = WS_NI);
int launch_conv_bf16_ws(const ConvWsParams& p, hipStream_t stream);
