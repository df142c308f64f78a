// The persistent wavefront-specialised stride-1 3x3 convolution on v_mfma_f32_16x16x32_bf16 (conv_bf16_ws16.hip): the member of the
// conv_bf16_ws family for the launches with the plain epilogue (convolution [+ bias], bf16 out: layer 5's forward and the data
// gradients of layers 5 / 6 of base_model, yogo/model_defns.py:54-65 and their autograd).
#pragma once
#include "conv_bf16_ws.h"

// Same tiles, tile walk and staged input image as conv_bf16_ws_kernel (ConvWsParams + conv_bf16_ws_plan).  What differs is the K
// step: one MFMA contracts 32 channels = a PAIR of 16-channel chunks (the lane's K group picks the chunk and its channel block),
// so both chunks of a pair are resident and the weight slices stream by KERNEL ROW: a period = (chunk pair, kernel row) =
// 3 taps x 32 MFMAs of 16 cycles per compute wavefront.  A period is 1 536 MFMA cycles -- too short for weight slices requested at its
// start to have landed at its end (first form: two slots, one period ahead; without the weight requests -5 %): there are three slots and
// the slices of period c + 2 are requested in period c, which the LDS affords because the tile plan keeps the staged input tile of a chunk
// within 768 units (three DMA pieces per loader instead of four: a quarter less input staging work as well).
#define W16_WSLOT (3 * 2 * 2 * 128 * 16)   // bytes of a period's weight slices [3 kx][2 chunks][2 channel blocks][128 channels] (24 576)
#define W16_W0 0                            // THREE weight slots (slot = kernel row): the slices are requested two periods ahead
#define W16_NI 3                            // input slots (16-byte elements per loader lane and 16-channel chunk): the staged tile has <= 768 units
#define W16_IB (W16_NI * WS_NT * 16)        // bytes of a chunk's input buffer (12 288)
#define W16_ISLOT (2 * W16_IB)              // bytes of a pair's input tiles [2 chunks][768 units] (24 576)
#define W16_I0 (3 * W16_WSLOT)              // two pair slots (pair parity)
#define W16_STG (W16_I0 + 2 * W16_ISLOT)    // output staging: [4 wavefronts][8 channel blocks][64 pixels] x 16 B = 32 KB (one half of the channels)
#define W16_EB (W16_STG + 32768)            // [128] fp32 bias
#define W16_MB (W16_EB + 512)               // mailbox loader -> compute: [4 wavefronts][16 pixel columns][4 pixel blocks] x 4 B (operand row addresses of the NEXT tile)
#define W16_MBS (W16_MB + 1024)             // ... and its scalars: {there is a next tile, staged row pitch in bytes, bytes of a channel block of the staged tile}
#define W16_LDS_BYTES (W16_MBS + 16)        // 157 200

// true when the kernel takes the launch (stride 1, 3x3, M = 128, K a multiple of 64, no activation / sign map / channel scale)
bool conv_bf16_ws16_eligible(int K, int M, int IH, int IW, int B);
int launch_conv_bf16_ws16(const ConvWsParams& p, hipStream_t stream);
