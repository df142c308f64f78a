// The persistent wavefront-specialised stride-1 3x3 convolution on v_mfma_f32_16x16x32_bf16 (conv_bf16_ws16.hip): the member of the
// conv_bf16_ws family for the launches with the plain epilogue (convolution [+ bias], bf16 out: layer 5's forward and the data
// gradients of layers 5 / 6 of base_model, yogo/model_defns.py:54-65 and their autograd).
#pragma once
#include "conv_bf16_ws.h"

// Same tiles, tile walk and staged input image as conv_bf16_ws_kernel (ConvWsParams + conv_bf16_ws_plan).  What differs is the K
// step: one MFMA contracts 32 channels = a PAIR of 16-channel chunks (the lane's K group picks the chunk and its channel block),
// so both chunks of a pair are resident and the weight slices stream by KERNEL ROW: a period = (chunk pair, kernel row) =
// 3 taps x 32 MFMAs of 16 cycles per compute wavefront.
#define W16_WSLOT (3 * 2 * 2 * 128 * 16)   // bytes of a period's weight slices [3 kx][2 chunks][2 channel blocks][128 channels] (24 576)
#define W16_W0 0                            // two weight slots (period parity)
#define W16_ISLOT (2 * WS_IB)               // bytes of a pair's input tiles [2 chunks][1024 units] (32 768)
#define W16_I0 (2 * W16_WSLOT)              // two pair slots (pair parity)
#define W16_STG (W16_I0 + 2 * W16_ISLOT)    // output staging: [4 wavefronts][8 channel blocks][64 pixels] x 16 B = 32 KB (one half of the channels)
#define W16_EB (W16_STG + 32768)            // [128] fp32 bias
#define W16_MB (W16_EB + 512)               // mailbox loader -> compute: [4 wavefronts][16 pixel columns][4 pixel blocks] x 4 B (operand row addresses of the NEXT tile)
#define W16_MBS (W16_MB + 1024)             // ... and its scalars: {there is a next tile, staged row pitch in bytes, bytes of a channel block of the staged tile}
#define W16_LDS_BYTES (W16_MBS + 16)        // 149 008

// true when the kernel takes the launch (stride 1, 3x3, M = 128, K a multiple of 64, no activation / sign map / channel scale)
bool conv_bf16_ws16_eligible(int K, int M, int IH, int IW, int B);
int launch_conv_bf16_ws16(const ConvWsParams& p, hipStream_t stream);
