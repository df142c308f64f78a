// Persistent 4-wavefront form of the stride-1 3x3 bf16 convolution with 128 GEMM rows (conv_bf16_p4.hip).
#pragma once
#include "common.h"

// geometry of the persistent kernel: 4 wavefronts (one per SIMD), each 128 output channels x 96 pixels = 12 accumulator
// tiles of 32x32 in 192 asm-owned AGPRs; a workgroup tile is 128 channels x 384 consecutive pixels of a column band
#define P4_NWV 4
#define P4_NW 3
#define P4_PT (32 * P4_NWV * P4_NW)   // 384 pixels per workgroup tile
#define P4_NT (64 * P4_NWV)           // 256 lanes
#define P4_NI 6                       // input slots (16-byte elements per lane and 16-channel chunk)
#define P4_LDSW_OFF (P4_NI * P4_NT)   // first unit of the weight slices inside a chunk buffer
#define P4_BUFU (P4_LDSW_OFF + 9 * 2 * 128)
#define P4_BUFB (P4_BUFU * 16)        // bytes per chunk buffer (61 440)
#define P4_EB (2 * P4_BUFB)           // [128] fp32 bias
#define P4_ES (P4_EB + 512)           // [2][128] fp32 channel scale (by tile parity)
#define P4_LDS_BYTES (P4_ES + 1024)

struct ConvP4Params {
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

// true when the persistent kernel takes the launch (stride 1, 3x3, M = 128, K a multiple of 32 and >= 64, lean epilogue)
bool conv_bf16_p4_eligible(int K, int M, int IH, int IW, int B);
// fills the tiling part of `p` (ncb, TW, ..., magic numbers); false when no tiling fits the kernel's fixed LDS layout
bool conv_bf16_p4_plan(ConvP4Params* p);
int launch_conv_bf16_p4(const ConvP4Params& p, hipStream_t stream);
