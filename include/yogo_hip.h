/* yogo_hip.h -- C ABI of libyogo_hip.so, the MI355X (gfx950) implementation of the YOGO hot path.
 *
 * The reference (czbiohub-sf/yogo @ 2024_08_07) has no native code and no FFI: its "operators" are torch ATen /
 * cuDNN / torchvision calls made from Python.  Each entry point below names the reference call site it replaces
 * (file:line relative to the reference root).  How a maintainer binds them (ctypes) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless noted; tensors are contiguous NCHW fp32 unless noted
 *   - nothing is allocated or freed here: the caller passes outputs and workspaces (sizes from *_bytes / *_rows)
 *   - hipStream_t is passed as an opaque pointer; all work is enqueued on it, no host synchronisation
 *   - return value 0 = ok; non-zero = error, message from yogo_hip_last_error() (thread-local)
 *   - re-entrant; one host thread per device
 *   - activation codes: 0 none, 1 LeakyReLU(0.01), 2 SiLU
 */
#ifndef YOGO_HIP_H
#define YOGO_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* yogo_stream_t; /* hipStream_t */

const char* yogo_hip_last_error(void);
int yogo_hip_abi_version(void);
/* Launch log (test / profiling aid, no counterpart in the reference): while enabled, every entry point that launches a
 * convolution / weight-gradient kernel appends one line "<kernel instantiation as rocprofv3 names it> | <planner parameters>".
 * enable = 1 clears and starts recording, 0 stops.  yogo_hip_launch_log_read copies the text into a HOST buffer (NUL
 * terminated, truncated to cap) and reports the bytes needed. */
int yogo_hip_launch_log(int enable);
int yogo_hip_launch_log_read(char* buf, size_t cap, size_t* needed);

/* ---- convolution, fp32 matrix cores ---------------------------------------------------------------------------
 * nn.Conv2d(cin, cout, 3, stride=1|2, padding=1) and nn.Conv2d(cin, cout, 1): yogo/model_defns.py:34-67,
 * executed at yogo/model.py:275 (forward) and by loss.backward() at yogo/train.py:322 (dgrad / wgrad).          */

/* bytes of the packed weight buffer. mode 0: forward, mode 1: dgrad */
int yogo_conv_packed_bytes(int Cin, int Cout, int ksize, int stride, int mode, size_t* bytes);
/* OIHW weights -> packed [tap][Kpad][Mpad] */
int yogo_conv_pack_f32(const float* w_oihw, float* packed, int Cin, int Cout, int ksize, int stride, int mode,
                       yogo_stream_t stream);
/* rows / row stride (floats/2) of the BatchNorm partial-sum buffer a forward launch fills when stats_part != NULL */
int yogo_conv2d_fwd_stats_shape(int B, int Cin, int Cout, int IH, int IW, int ksize, int stride, int* rows, int* mpad);
/* out = chan_scale[b][c] * act(conv(in) + bias); out_pre (optional) receives conv(in)+bias;
 * stats_part (optional) receives per-workgroup (sum, sumsq) per channel for BatchNorm training statistics.
 * Fuses Conv2d + bias + LeakyReLU/SiLU + Dropout2d channel mask (model_defns.py:39-41 etc.).                      */
int yogo_conv2d_fwd_f32(const float* in, const float* packed, const float* bias, float* out, float* out_pre,
                        const float* chan_scale, float* stats_part, int B, int Cin, int Cout, int IH, int IW, int ksize,
                        int stride, int act, yogo_stream_t stream);
/* dx = conv_transpose(dy) * act'(act_ref) * chan_scale   (act_ref / chan_scale optional).
 * (IH, IW) are the forward conv's INPUT dims.  act_ref: block output for LeakyReLU, pre-activation for SiLU.      */
int yogo_conv2d_dgrad_f32(const float* dy, const float* packed_dgrad, float* dx, const float* act_ref, int ref_act,
                          const float* chan_scale, int B, int Cin, int Cout, int IH, int IW, int ksize, int stride,
                          yogo_stream_t stream);
int yogo_conv2d_wgrad_workspace_bytes(int B, int Cin, int Cout, int IH, int IW, int ksize, int stride, size_t* bytes);
/* dw (OIHW) and optional db from the layer input x and the gradient g w.r.t. the conv output; both are clamped to
 * +-clip when clip > 0 (the per-parameter hook of yogo/model.py:76-77).  Deterministic (no float atomics).        */
int yogo_conv2d_wgrad_f32(const float* x, const float* g, float* dw, float* db, void* workspace, int B, int Cin, int Cout,
                          int IH, int IW, int ksize, int stride, float clip, yogo_stream_t stream);

/* ---- first convolution: Cin = 1|3, uint8 or fp32 input (model_defns.py:34 + the cast at model.py:272-273) --------- */
int yogo_conv_first_stats_rows(int B, int IH, int IW, int stride, int* rows);
/* in_dtype: 0 = uint8, 1 = float32.  stats_part rows: [rows][Cout][2] */
int yogo_conv_first_fwd(const void* in, int in_dtype, const float* w_oihw, const float* bias, float* out, float* out_pre,
                        const float* chan_scale, float* stats_part, int B, int Cin, int Cout, int IH, int IW, int stride,
                        int act, yogo_stream_t stream);
/* partial gradients [rows][Cout][Cin*9 + 1] (last column = bias), rows from yogo_conv_first_wgrad_rows; finish with
 * yogo_partials_reduce */
int yogo_conv_first_wgrad_rows(int B, int IH, int IW, int stride, int* rows);
int yogo_conv_first_wgrad(const void* in, int in_dtype, const float* dy, float* part, int B, int Cin, int Cout, int IH,
                          int IW, int stride, yogo_stream_t stream);

/* ---- BatchNorm2d (model_defns.py:35,55,60): batch statistics, apply + activation, backward ------------------------ */
int yogo_bn_finalize(const float* part, int rows, int row_stride, int C, long long count, float eps, float momentum,
                     float* mean_out, float* invstd_out, float* running_mean, float* running_var,
                     long long* num_batches_tracked, yogo_stream_t stream);
int yogo_bn_apply_act(const float* z, float* y, const float* mean, const float* invstd_or_var, int stat_is_var, float eps,
                      const float* gamma, const float* beta, int B, int C, int HW, int act, yogo_stream_t stream);
int yogo_bn_invstd(const float* var, float eps, float* invstd, int C, yogo_stream_t stream);
int yogo_bn_bwd_rows(int B, int HW, int* rows);
/* g: gradient w.r.t. the block OUTPUT; the activation derivative (act) is applied inside from the recomputed
 * pre-activation, then dz, dgamma, dbeta (clamped to +-clip when clip > 0) */
int yogo_bn_bwd(const float* g, const float* z, float* dz, const float* mean, const float* invstd, const float* gamma,
                const float* beta, int act, float* dgamma, float* dbeta, float* part, float* sums, int B, int C, int HW,
                int training, float clip, yogo_stream_t stream);
/* `part` is folded in place (used as scratch) */
int yogo_partials_reduce(float* part, int rows, int N, float clip, float* out, yogo_stream_t stream);
int yogo_channel_sum(const float* g, int B, int C, int HW, float clip, float* out, yogo_stream_t stream);

/* ---- box decode: YOGO.forward, yogo/model.py:277-313 ------------------------------------------------------------------ */
int yogo_decode_fwd(const float* raw, float* out, const float* cxs, const float* cys, int B, int P, int Sy, int Sx,
                    float anchor_w, float anchor_h, float width_multiplier, float height_multiplier, int inference,
                    yogo_stream_t stream);
int yogo_decode_bwd(const float* raw, const float* out, const float* gout, float* graw, int B, int P, int Sy, int Sx,
                    int inference, yogo_stream_t stream);

/* the same, gradient written as bf16 NCHW8c [B][2 * ceil(P / 16)][Sy][Sx][8] for the bf16 backward pass of the head convolution */
int yogo_decode_bwd_bf16(const float* raw, const float* out, const float* gout, void* graw8c, int B, int P, int Sy, int Sx,
                         int inference, yogo_stream_t stream);

/* ---- loss: YOGOLoss.forward, yogo/yogo_loss.py:38-129 (+ its autograd) ------------------------------------------------- */
int yogo_loss_workspace_bytes(int B, int Sy, int Sx, size_t* bytes);
/* loss_out: 4 device floats {total, iou_loss, objectness_loss, classification_loss}; grad = d total / d pred */
int yogo_loss_fwd_bwd(const float* pred, const float* label, float* grad, float* loss_out, void* workspace, int B, int P,
                      int Sy, int Sx, float no_obj_weight, float iou_weight, float classify_weight, float label_smoothing,
                      yogo_stream_t stream);

/* The trainer's fused form of the three calls above (yogo_decode_fwd + yogo_loss_fwd_bwd + yogo_decode_bwd_bf16) for TRAINING mode
 * (class logits pass through the decode): replaces YOGO.forward's decode (yogo/model.py:277-313), YOGOLoss.forward
 * (yogo/yogo_loss.py:38-129) and the autograd of both inside Trainer.train's step (yogo/train.py:309-322).  One pass over the
 * cells; the decoded prediction and its gradient never go to memory.  graw8c / loss_out / workspace as in the separate calls. */
int yogo_decode_loss_bwd_bf16(const float* raw, const float* label, const float* cxs, const float* cys, void* graw8c, float* loss_out,
                              void* workspace, int B, int P, int Sy, int Sx, float anchor_w, float anchor_h, float width_multiplier,
                              float height_multiplier, float no_obj_weight, float iou_weight, float classify_weight,
                              float label_smoothing, yogo_stream_t stream);

/* ---- post-process: format_preds, yogo/utils/prediction_formatting.py:23-93, batched over images ------------------------ */
int yogo_format_preds_workspace_bytes(int B, int Sy, int Sx, size_t* bytes);
/* out_rows [B][cap][P], out_cells [B][cap] (int64 cell index y*Sx+x), out_count [B] (int32).
 * box_format 0 = cxcywh, 1 = xyxy.  Row order per image is the reference's: descending max(class)*objectness
 * (ties: lower cell first) when iou_thresh > 0, cell order otherwise.                                                    */
int yogo_format_preds_batched(const float* pred, float* out_rows, long long* out_cells, int* out_count, void* workspace,
                              int B, int P, int Sy, int Sx, int cap, double obj_thresh, double iou_thresh, int box_format,
                              double min_class_confidence_threshold, yogo_stream_t stream);
/* The inference driver's fused form of yogo_decode_fwd + yogo_format_preds_batched (SURVEY.md 8(b) `decode_nms_batched`): `raw` is
 * the head's output BEFORE YOGO.forward's decode (yogo/model.py:277-313); the kernel decodes each value where it loads it, so the
 * decoded [B, 5+C, Sy, Sx] tensor that `yogo infer` hands from the model to format_preds (yogo/infer.py:45,73) never goes through
 * memory.  Rows, cells and counts are bit-identical to the two separate calls.  cxs / cys / anchor / multipliers / inference as
 * yogo_decode_fwd, the rest as yogo_format_preds_batched (same workspace query).                                              */
int yogo_decode_format_preds_batched(const float* raw, const float* cxs, const float* cys, float* out_rows, long long* out_cells,
                                     int* out_count, void* workspace, int B, int P, int Sy, int Sx, int cap, float anchor_w,
                                     float anchor_h, float width_multiplier, float height_multiplier, int inference,
                                     double obj_thresh, double iou_thresh, int box_format, double min_class_confidence_threshold,
                                     yogo_stream_t stream);
/* Eval-mode `model(x)` (yogo/model.py:275-313: self.model(x), then the decode) with the 1x1 head and the box decode in ONE launch
 * (SURVEY.md 8(b) `head1x1_decode_fwd`): x = the last 3x3 block's output, bf16 NCHW8c [B][Cin][Sy][Sx] (Cin a multiple of 16, <= 128),
 * packed = yogo_conv_bf16_pack(w [P][Cin][1][1], mode 0), bias [P] or null; out fp32 [B][P][Sy][Sx] = what yogo_conv2d_fwd_bf16 with an
 * fp32 output followed by yogo_decode_fwd writes, bit for bit -- the raw head output never goes through memory.  6 <= P <= 16.      */
int yogo_head1x1_decode_fwd_bf16(const void* x, const void* packed, const float* bias, float* out, const float* cxs, const float* cys,
                                 int B, int Cin, int P, int Sy, int Sx, float anchor_w, float anchor_h, float width_multiplier,
                                 float height_multiplier, int inference, yogo_stream_t stream);

/* ---- bf16 path: the bf16-autocast forward of `yogo infer` (yogo/infer.py:313-317) and half-precision training
 * (yogo/train.py:315-318, --half) -------------------------------------------------------------------------------------------
 * Activations and activation gradients in "NCHW8c" = [B][C/8][H][W][8] bf16 (C padded to a multiple of 16; padding
 * channels hold zeros); weights, BatchNorm statistics, parameter gradients, optimiser state stay fp32.
 * Inference folds eval-mode BatchNorm into `scale` (weights) and `bias` on the caller's side.                               */
int yogo_bf16_channel_blocks(int C);
/* mode 0: forward packing (optional per-output-channel scale); mode 1: dgrad packing (roles swapped, taps flipped);
 * mode 2: dgrad packing for a stride-2 3x3 convolution (as 1, tap slices in output-parity-class order) */
int yogo_conv_bf16_packed_bytes(int Cin, int Cout, int ksize, int mode, size_t* bytes);
int yogo_conv_bf16_pack(const float* w_oihw, const float* scale, void* packed, int Cin, int Cout, int ksize, int mode,
                        yogo_stream_t stream);
int yogo_conv2d_fwd_bf16_stats_shape(int B, int Cin, int Cout, int IH, int IW, int ksize, int stride, int* rows, int* mpad);
/* Plan (no counterpart in the reference; cuDNN picks its algorithm behind torch.backends.cudnn.benchmark, yogo/train.py:36): the 3x3
 * bf16 convolutions with 128 GEMM rows and 64 / 128 / ... contraction channels (forward of yogo/model_defns.py:49-65's 128-channel
 * blocks and their data gradients) run on the persistent wavefront-specialised kernels (conv_bf16_ws_kernel; 8 wavefronts per CU: 4
 * compute, 4 load / store), everything else on the tiled conv_bf16_kernel.  The two families agree bit for bit where the tiled plan
 * also steps the contraction in 16-channel chunks (every shape of the training step), within one bf16 ulp otherwise.  The library has
 * no run-time plan switch and no mutable global state besides its lazily-loaded module handles and the launch log: the A/B switches
 * used by tests/ and tools/ (yogo_hook_*) exist only in libyogo_hip_hooks.so (build.sh, -DYOGO_TEST_HOOKS), which the package never loads. */
/* y = chan_scale * act(conv(x) + bias); out: bf16 NCHW8c, or fp32 NCHW when out_f32 != NULL (the head);
 * stats_part (optional): BatchNorm partial (sum, sumsq) of the fp32 pre-activation */
int yogo_conv2d_fwd_bf16(const void* in, const void* packed, const float* bias, void* out, float* out_f32,
                         const float* chan_scale, float* stats_part, int B, int Cin, int Cout, int IH, int IW, int ksize,
                         int stride, int act, yogo_stream_t stream);
/* as above plus a second bf16 NCHW8c output out_pre = conv + bias before the activation (SiLU blocks without BatchNorm keep it:
 * the activation derivative of yogo/model_defns.py's nn.SiLU is a function of the pre-activation) */
int yogo_conv2d_fwd_bf16_pre(const void* in, const void* packed, const float* bias, void* out, void* out_pre,
                             const float* chan_scale, int B, int Cin, int Cout, int IH, int IW, int ksize, int stride, int act,
                             yogo_stream_t stream);
/* yogo_conv_bf16_pack for a list of tensors in one launch (a training step repacks every layer twice).  table: device int64
 * [n][8] = {w pointer, scale pointer or 0, packed pointer, Cin, Cout, ksize, mode, first block}; entry k owns
 * yogo_conv_bf16_pack_blocks blocks starting at its first block; total_blocks = their sum */
int yogo_conv_bf16_pack_blocks(int Cin, int Cout, int ksize, int mode, int* blocks);
int yogo_conv_bf16_pack_multi(const void* table, int n, int total_blocks, yogo_stream_t stream);
/* LeakyReLU sign map of a bf16 NCHW8c tensor: [B][2][H][W][Cpad/16] bytes (Cpad = C rounded up to 32 / 64 / a multiple of 128
 * for C <= 32 / <= 64 / more); byte (h, pixel, q), bit i + 4e = (channel 4h + i of channel block 2q + e > 0), h, e in {0, 1},
 * i < 4; bytes of channel blocks beyond kb(C) are unspecified -- all the data gradient needs of a LeakyReLU block's output (torch's
 * leaky_relu_backward reads the whole tensor), at 1/16 of its bytes */
int yogo_bf16_signs_bytes(int B, int C, int H, int W, size_t* bytes);
/* yogo_conv2d_fwd_bf16 (bf16 output) that also writes the sign map of its output */
int yogo_conv2d_fwd_bf16_signs(const void* in, const void* packed, const float* bias, void* out, void* signs,
                               const float* chan_scale, int B, int Cin, int Cout, int IH, int IW, int ksize, int stride, int act,
                               yogo_stream_t stream);
/* yogo_conv2d_dgrad_bf16 for a LeakyReLU reference given as its sign map: dx = conv_transpose(dy) * (bit ? 1 : 0.01) * chan_scale.
 * Contract against yogo_conv2d_dgrad_bf16 with the bf16 reference tensor: the same bits wherever both run on the same kernel; the
 * stride-2 3x3 shapes conv_bf16_s2d_direct_kernel takes (<= 32 or 65..128 input channels of the forward convolution) step the
 * contraction 16 channels at a time where the tiled kernel takes 32 / 64: equal to ONE bf16 rounding step on < 0.5 % of the values
 * (tests/test_gpu_bf16.py, tests/test_gpu_ws.py).  The same one-step contract holds between conv_bf16_ws16_kernel (the plain-epilogue
 * stride-1 launches with 128 output channels and a multiple of 64 input channels: two 16-channel chunks summed inside one MFMA) and
 * the 32x32x16 kernels. */
int yogo_conv2d_dgrad_bf16_signs(const void* dy, const void* packed_dgrad, void* dx, const void* signs,
                                 const float* chan_scale, int B, int Cin, int Cout, int IH, int IW, int ksize, int stride,
                                 yogo_stream_t stream);
/* dx = conv_transpose(dy) * act'(act_ref) * chan_scale, everything bf16 NCHW8c; (IH, IW) = forward INPUT dims */
int yogo_conv2d_dgrad_bf16(const void* dy, const void* packed_dgrad, void* dx, const void* act_ref, int ref_act,
                           const float* chan_scale, int B, int Cin, int Cout, int IH, int IW, int ksize, int stride,
                           yogo_stream_t stream);
/* fp32 dw/db from bf16 NCHW8c x and g (exact fp32 MFMA on the widened values); workspace as yogo_conv2d_wgrad_f32 */
int yogo_conv2d_wgrad_bf16in(const void* x, const void* g, float* dw, float* db, void* workspace, int B, int Cin, int Cout,
                             int IH, int IW, int ksize, int stride, float clip, yogo_stream_t stream);
/* the same on the bf16 matrix cores (contraction over pixels through transposed LDS reads), fp32 accumulation */
int yogo_conv2d_wgrad_bf16_workspace_bytes(int B, int Cin, int Cout, int IH, int IW, int ksize, int stride, size_t* bytes);
int yogo_conv2d_wgrad_bf16(const void* x, const void* g, float* dw, float* db, void* workspace, int B, int Cin, int Cout,
                           int IH, int IW, int ksize, int stride, float clip, yogo_stream_t stream);
/* ... with the split-K reduction deferred (ABI 5): recorded in `queue` and run by yogo_wgrad_reduce_flush together with every other
 * recorded one -- one launch for the weight gradients of a whole backward pass (the per-layer launches are ~21 us each, mostly launch and
 * tail latency).  dw / db / workspace stay valid until the flush; results are bit-identical to yogo_conv2d_wgrad_bf16.  A queue holds up
 * to 16 reductions (a 17th flushes first, on the stream of the call that records it) and belongs to one host thread AND one stream:
 * record and flush a queue on the same stream. */
int yogo_wgrad_reduce_queue_create(void** queue_out);
int yogo_wgrad_reduce_queue_destroy(void* queue);
int yogo_wgrad_reduce_queue_reset(void* queue);   /* forget the recorded reductions without running them */
int yogo_wgrad_reduce_flush(void* queue, yogo_stream_t stream);
int yogo_conv2d_wgrad_bf16_deferred(const void* x, const void* g, float* dw, float* db, void* workspace, int B, int Cin, int Cout, int IH,
                                    int IW, int ks, int stride, float clip, void* queue, yogo_stream_t stream);
int yogo_conv_first_fwd_bf16(const void* in, int in_dtype, const float* w, const float* bias, void* out, int B, int Cin,
                             int Cout, int IH, int IW, int stride, int act, yogo_stream_t stream);
int yogo_conv_first_fwd_train_bf16(const void* in, int in_dtype, const float* w, const float* bias, void* out_bf16,
                                   const float* chan_scale, float* stats_part, int B, int Cin, int Cout, int IH, int IW,
                                   int stride, int act, yogo_stream_t stream);
int yogo_conv_first_wgrad_bf16g(const void* in, int in_dtype, const void* dy_bf16, float* part, int B, int Cin, int Cout,
                                int IH, int IW, int stride, yogo_stream_t stream);
/* Layer-0 backward in one pass: BatchNorm backward (model_defns.py:35) + activation derivative + weight gradient of the first
 * convolution; g = gradient w.r.t. the block output, z = saved conv output, both bf16 NCHW8c.  part: rows
 * (yogo_conv_first_wgrad_rows) x cols (yogo_conv_first_bn_wgrad_cols) floats; then yogo_partials_reduce(part, rows, cols, 0,
 * sums) and yogo_conv_first_bn_wgrad_finalize (dw OIHW, dgamma, dbeta, clamped to +-clip). */
int yogo_conv_first_bn_wgrad_cols(int Cin, int Cout, int* cols);
int yogo_conv_first_bn_wgrad_bf16(const void* in, int in_dtype, const void* g, const void* z, const float* mean,
                                  const float* invstd, const float* gamma, const float* beta, float* part, int B, int Cin,
                                  int Cout, int IH, int IW, int stride, int act, yogo_stream_t stream);
int yogo_conv_first_bn_wgrad_finalize(const float* sums, const float* mean, const float* invstd, const float* gamma,
                                      const float* w_oihw, float* dw, float* dgamma, float* dbeta, int B, int Cin, int Cout,
                                      int IH, int IW, int stride, int training, float clip, yogo_stream_t stream);
int yogo_bn_apply_act_bf16(const void* z, void* y, const float* mean, const float* invstd_or_var, int stat_is_var, float eps,
                           const float* gamma, const float* beta, int B, int C, int HW, int act, yogo_stream_t stream);
int yogo_bn_bwd_bf16_rows(int B, int HW, int* rows);
/* batch statistics of a stored bf16 NCHW8c tensor: part [rows][C][2] (rows = yogo_bn_bwd_bf16_rows), then yogo_bn_finalize */
int yogo_bn_stats_bf16(const void* z, float* part, int B, int C, int HW, yogo_stream_t stream);
int yogo_bn_bwd_bf16(const void* g, const void* z, void* dz, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, int act, float* dgamma, float* dbeta, float* part, float* sums, int B, int C, int HW,
                     int training, float clip, yogo_stream_t stream);
/* ... of the block under a 1x1 convolution with P <= 16 outputs (the detection head, yogo/model.py:150-155) WITHOUT that convolution's data
 * gradient in memory (ABI 7): gh = gradient w.r.t. the head's output (bf16 NCHW8c [B][2][HW], channels >= P zero), head_w = the head's
 * weights [P][C]; g = bf16(sum_k gh[k] * bf16(w[k][c])) is computed inside both sweeps (one MFMA per 16 pixels x 16 channels).  C a
 * multiple of 16; dz must not alias gh.  Replaces the head's yogo_conv2d_dgrad_bf16 + yogo_bn_bwd_bf16. */
int yogo_bn_bwd_bf16_head(const void* gh, const float* head_w, int P, const void* z, void* dz, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, int act, float* dgamma, float* dbeta, float* part, float* sums, int B,
                          int C, int HW, int training, float clip, yogo_stream_t stream);
int yogo_nchw_f32_to_bf16_8c(const float* in, void* out, int B, int C, int HW, yogo_stream_t stream);
int yogo_bf16_8c_to_nchw_f32(const void* in, float* out, int B, int C, int HW, yogo_stream_t stream);

/* ---- layer 0 of the bf16 training path on the matrix cores (uint8 image, Cin = 1, Cout <= 16, stride 2, even H and W):
 * Conv2d + the uint8 -> float cast (yogo/model_defns.py:34, yogo/model.py:272) and the BatchNorm2d + activation behind it
 * (model_defns.py:35).  Weights are rounded to bf16 inside (autocast, yogo/train.py:315-318).  Outputs, each optional:
 * stats_part = partial (sum, sumsq) of conv + bias in fp32, [rows][16][2] with rows from yogo_conv_first_mfma_stats_rows ->
 * yogo_bn_finalize(part, rows, 16, ...); z = conv + bias, bf16 NCHW8c; y = act((z - mean) * invstd * gamma + beta), bf16
 * NCHW8c; without y the activation is applied to z (inference, BatchNorm folded into w / bias).  Training: statistics (this
 * kernel or yogo_conv_first_gram), then z + y in one sweep instead of conv + a separate BatchNorm pass over the activations. */
int yogo_conv_first_mfma_supported(int in_dtype, int Cin, int Cout, int IH, int IW, int stride);
int yogo_conv_first_mfma_stats_rows(int B, int IH, int IW, int* rows);
int yogo_conv_first_mfma(const void* in, const float* w, const float* bias, void* z, void* y, const float* mean,
                         const float* invstd, const float* gamma, const float* beta, float* stats_part, int B, int Cout, int IH,
                         int IW, int act, yogo_stream_t stream);

/* Batch statistics of that layer WITHOUT the convolution: z = w . patch + b is linear in the 3x3 patch, so the statistics of all
 * channels follow from P = sum patch (9) and G = sum patch (x) patch (9 x 9) over the batch -- accumulated exactly in integers.
 * part: scratch, yogo_conv_first_gram_rows x 54 uint32; gram: double[90], gram_f32: float[90] (P[9], then G[9][9]).
 * yogo_bn_stats_from_gram = yogo_bn_finalize on those sums (w rounded to bf16 as the convolution does); the backward pass
 * reuses gram_f32 (yogo_conv_first_bn_wgrad_bf16_xg / _finalize_xg) instead of sweeping the image again. */
int yogo_conv_first_gram_rows(int B, int IH, int IW, int* rows);
int yogo_conv_first_gram(const void* in, void* part, double* gram, float* gram_f32, int B, int IH, int IW, yogo_stream_t stream);
int yogo_bn_stats_from_gram(const double* gram, const float* w, const float* bias, int Cout, long long count, float eps,
                            float momentum, float* mean_out, float* invstd_out, float* running_mean, float* running_var,
                            long long* num_batches_tracked, yogo_stream_t stream);
int yogo_conv_first_bn_wgrad_bf16_xg(const void* in, int in_dtype, const void* g, const void* z, const float* mean,
                                     const float* invstd, const float* gamma, const float* beta, float* part, int B, int Cin,
                                     int Cout, int IH, int IW, int stride, int act, yogo_stream_t stream);
int yogo_conv_first_bn_wgrad_finalize_xg(const float* sums, const float* gram, const float* mean, const float* invstd,
                                         const float* gamma, const float* w_oihw, float* dw, float* dgamma, float* dbeta, int B,
                                         int Cin, int Cout, int IH, int IW, int stride, int training, float clip,
                                         yogo_stream_t stream);
/* The same pair of sweeps WITHOUT the layer's conv output z in memory (ABI 5).  What the backward pass needs of z is (a) the sign of
 * the BatchNorm output for the LeakyReLU derivative and (b) sum g * xhat, which is linear in z = W . patch and follows from the
 * weight-gradient sums themselves.  yogo_conv_first_mfma_signs = yogo_conv_first_mfma that also writes (a) as a bit map, signs =
 * [B][OH*OW][2] bytes (byte h of a pixel: bit i = channel 4h + i, bit 4 + i = channel 8 + 4h + i; z may then be NULL);
 * yogo_conv_first_bn_wgrad_bf16_xs reads it in place of z (_xs_supported: uint8 one-channel image, stride 2, even sizes, Cout 8 or
 * 16, ACT_NONE / ACT_LEAKY, no conv bias); yogo_conv_first_bn_wgrad_finalize_xs derives (b); w_oihw there = the bf16-rounded weights
 * the forward pass multiplied with.  Replaces the same reference lines as the _xg pair (model_defns.py:34-35 under autograd). */
int yogo_conv_first_mfma_signs(const void* in, const float* w, const float* bias, void* z, void* y, void* signs, const float* mean,
                               const float* invstd, const float* gamma, const float* beta, int B, int Cout, int IH, int IW, int act,
                               yogo_stream_t stream);
int yogo_conv_first_bn_wgrad_xs_supported(int in_dtype, int Cin, int Cout, int IH, int IW, int stride, int act);
/* (both sweeps take two adjacent pixels per lane where the output width is even, one otherwise) */
int yogo_conv_first_bn_wgrad_bf16_xs(const void* in, int in_dtype, const void* g, const void* signs, const float* mean,
                                     const float* invstd, const float* gamma, const float* beta, float* part, int B, int Cin,
                                     int Cout, int IH, int IW, int stride, int act, yogo_stream_t stream);
int yogo_conv_first_bn_wgrad_finalize_xs(const float* sums, const float* gram, const float* mean, const float* invstd,
                                         const float* gamma, const float* w_oihw, float* dw, float* dgamma, float* dbeta, int B,
                                         int Cin, int Cout, int IH, int IW, int stride, int training, float clip,
                                         yogo_stream_t stream);
/* Layer 1's data gradient and the sweep of yogo_conv_first_bn_wgrad_bf16_xs in ONE launch (ABI 7): the gradient w.r.t. layer 0's output
 * is folded into layer 0's backward sums tile by tile and never written (layer 0 has no data gradient of its own).  g = gradient
 * w.r.t. layer 1's conv output, bf16 NCHW8c [B][Cout1 / 8][H][W]; packed = layer 1's yogo_conv_bf16_pack mode-1 packing; image = uint8
 * [B][2H][2W]; signs as above (NULL with ACT_NONE); part: rows (yogo_conv2d_dgrad_first_bwd_rows, on the current device) x cols
 * (yogo_conv_first_bn_wgrad_cols) floats; finish with yogo_partials_reduce and yogo_conv_first_bn_wgrad_finalize_xs exactly as after the
 * unfused sweep.  _supported: 3x3 stride-1 layer 1 with 16 -> 32 channels, even W.  Results agree with yogo_conv2d_dgrad_bf16 followed by
 * yogo_conv_first_bn_wgrad_bf16_xs to fp32 rounding of the sums (every element is rounded to bf16 as the stored gradient would have been).
 * Replaces autograd of yogo/model_defns.py:34-41 (the backward of the first two blocks of base_model). */
int yogo_conv2d_dgrad_first_bwd_supported(int Cmid, int Cout1, int H, int W, int B, int act0);
int yogo_conv2d_dgrad_first_bwd_rows(int B, int H, int W, int with_wgrad, int* rows);
int yogo_conv2d_dgrad_bf16_first_bwd(const void* g, const void* packed, const void* image, const void* signs, float* part, int B, int Cmid,
                                     int Cout1, int H, int W, int act0, yogo_stream_t stream);
/* ... and layer 1's WEIGHT gradient in the same sweep (it reads the same g; x = layer 1's input = layer 0's output, bf16 NCHW8c): dw (OIHW)
 * and db (may be NULL) clamped to +-clip, as yogo_conv2d_wgrad_bf16 / _deferred deliver them (queue: NULL = reduce now, else a
 * yogo_wgrad_reduce_queue); part rows: yogo_conv2d_dgrad_first_bwd_rows(..., 1, ...).  Replaces autograd of model_defns.py:34-41 except
 * layer 1's own data-gradient input. */
int yogo_conv2d_dgrad_wgrad_first_bwd_workspace_bytes(int B, int H, int W, size_t* bytes);
int yogo_conv2d_dgrad_wgrad_bf16_first_bwd(const void* g, const void* packed, const void* x, const void* image, const void* signs, float* part,
                                           float* dw, float* db, void* workspace, int B, int Cmid, int Cout1, int H, int W, int act0,
                                           float clip, void* queue, yogo_stream_t stream);

/* ---- data-parallel exchange over RCCL / xGMI (replaces torch DDP: init_process_group("nccl") + DistributedDataParallel,
 * yogo/train.py:155-159; gradient all-reduce overlapped with backward, buffers broadcast from rank 0).  One process per GPU.
 * librccl.so is opened lazily.  yogo_comm_unique_id fills a HOST buffer of yogo_comm_unique_id_bytes() bytes on rank 0; the host
 * hands it to every rank (any rendez-vous) and each rank calls yogo_comm_init (collective).  The all-reduce is an in-place SUM
 * of fp32 values enqueued on `stream`; the mean's 1 / world is folded into yogo_adamw_step (grad_scale). */
int yogo_comm_unique_id_bytes(void);
int yogo_comm_unique_id(void* id_out_host);
int yogo_comm_init(int rank, int world, const void* unique_id_host, void** comm_out);
int yogo_comm_allreduce_flat(void* comm, float* buf, size_t count, yogo_stream_t stream);
int yogo_comm_broadcast_flat(void* comm, void* buf, size_t bytes, int root, yogo_stream_t stream);
int yogo_comm_destroy(void* comm);

/* ---- the step in front of the path: label rasteriser and batch flips ----------------------------- */
/* format_labels_tensor, yogo/data/yogo_dataset.py:24-46, for a whole batch.  labels: [N][5] fp32 rows (class, x1, y1, x2, y2),
 * or (class, xc, yc, w, h) with box_format = 1 (converted as label_file_to_tensor does, :132), all images back to back;
 * offsets: [B + 1] int32 row ranges; out: [B][6][Sy][Sx] fp32 = (mask, x1, y1, x2, y2, class), written completely; a later
 * label of a cell overwrites an earlier one, negative cell indices wrap once (Python indexing).  status: one device int32 the
 * caller zeroes; set to 1 + the row index of a label whose cell is outside the grid (IndexError in the reference). */
int yogo_labels_rasterize(const float* labels, const int* offsets, float* out, int* status, int B, int Sx, int Sy,
                          int box_format, yogo_stream_t stream);
/* RandomHorizontalFlipWithBBs + RandomVerticalFlipWithBBs, yogo/data/data_transforms.py:51-98, applied to a batch in one pass
 * (the caller draws the decisions).  img: [B][C][H][W], elem_bytes 1 (uint8) or 4 (float32); lab: [B][6][Sy][Sx] fp32; out of
 * place; either pair may be NULL.  Box coordinates become 1 - x in EVERY cell, as in the reference. */
int yogo_flip_batch(const void* img_in, void* img_out, int elem_bytes, const float* lab_in, float* lab_out, int B, int C, int H,
                    int W, int Sy, int Sx, int hflip, int vflip, yogo_stream_t stream);

/* ---- optimiser: torch.optim.AdamW over one flat buffer, yogo/train.py:213-217,324 ---------------------------------------- */
int yogo_adamw_step(float* p, const float* g, float* m, float* v, long long n, int step, double lr, double beta1,
                    double beta2, double eps, double weight_decay, double grad_scale, yogo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* YOGO_HIP_H */
