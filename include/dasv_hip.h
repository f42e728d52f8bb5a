/*
 * dasv_hip.h — C ABI of libdasv_hip.so: the MI355X (gfx950) hot path of
 * fedecosta/DoubleAttentionSpeakerVerification.
 *
 * The reference has no FFI layer; its boundary for this path is the Python
 * nn.Module API (scripts/poolings.py, scripts/loss.py).  Every entry point
 * below replaces the chain of eager torch ops named next to it, takes raw
 * device pointers and sizes, never allocates, never synchronises, never
 * throws, and launches on the caller's stream (hipStream_t passed as void*).
 * Return value: 0 on success, a negative DASV_E_* code for argument errors,
 * or a positive hipError_t from the launch.
 *
 * All pointers are device pointers unless stated otherwise.  Tensors are
 * borrowed for the duration of the enqueued work; nothing is retained.
 */
#ifndef DASV_HIP_H
#define DASV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DASV_ABI_VERSION 3

#define DASV_OK            0
#define DASV_E_NULL       -1  /* a required pointer is NULL */
#define DASV_E_SHAPE      -2  /* B/T/H/dh out of range or inconsistent */
#define DASV_E_WORKSPACE  -3  /* workspace missing or too small */
#define DASV_E_STRIDE     -4  /* negative stride / overlapping rows */
#define DASV_E_DTYPE      -5  /* unknown element type code */

/* element type of the frame embeddings x / grad_x */
#define DASV_F32  0
#define DASV_BF16 1
#define DASV_F16  2

typedef void* dasv_stream_t; /* hipStream_t */

int dasv_abi_version(void);
const char* dasv_error_string(int code);

/* ------------------------------------------------------------------------
 * Double multi-head attention pooling, forward.
 * Replaces poolings.py:73-80 (innerKeyValueAttention), :100-109
 * (MultiHeadAttention.forward), :45-51/:61-71 (HeadAttention narrow path) and
 * :126-129 (DoubleMHA.forward) — about ten eager kernels and two x-sized
 * temporaries — with ONE launch that reads x once.
 *
 *   x        [B,T,D=H*dh]  element (b,t,d) at x + b*x_stride_b + t*x_stride_t + d
 *                          (strides in ELEMENTS; innermost dim contiguous)
 *   query    [dh,H] f32    as stored by MultiHeadAttention.query (poolings.py:90)
 *   att      [dh]   f32    HeadAttention.att (poolings.py:35); NULL => stop after
 *                          the per-head stage (pooling_method 'MHA', model.py:37-38;
 *                          with H=1 and query=att[D,1] this is poolings.py:14-27)
 *   head_keep[B,H]  u8     train-time head mask (poolings.py:39-43), 1 = keep,
 *                          0 = score forced to -inf; NULL => keep all (eval)
 *   lengths  [B]    i32    valid frames per utterance (<= T); NULL => all T.
 *                          Frames >= length get alignment 0 and do not contribute.
 *   out      [B,dh] f32    DoubleMHA output (NULL allowed iff att is NULL)
 *   align    [B,T,H] f32   softmax-over-time alignments (poolings.py:77); NULL => skipped
 *   head_w   [B,H]  f32    softmax-over-heads weights (poolings.py:50); NULL => skipped
 *   ctx      [B,H,dh] f32  per-head context vectors (poolings.py:79): the output when att is
 *                          NULL, what the backward launch needs otherwise; NULL => skipped
 *                          (allowed iff att is given: inference)
 *   ml       [B,H,2] f32   softmax statistics saved for backward: (max, sum) of the
 *                          log2-domain scores; NULL => skipped
 *   workspace              dasv_dmha_workspace_bytes() bytes, zeroed ONCE by
 *                          dasv_dmha_workspace_init(); every launch leaves its arrival
 *                          counters zeroed, and they sit in a region of FIXED size at
 *                          the start, so one buffer may serve launches of different
 *                          B / T / H (any shape whose dasv_dmha_workspace_bytes() fits).
 *                          Not shareable between launches that may overlap in time.
 *
 * Score scale is 1/sqrt(H) (poolings.py:75: d_k = query.size(-1)).
 * A row whose heads are all masked yields NaN, as the reference does.
 * ------------------------------------------------------------------------ */
size_t dasv_dmha_workspace_bytes(int B, int T, int H, int dh);
int dasv_dmha_workspace_init(void* workspace, size_t workspace_bytes, dasv_stream_t stream);

int dasv_dmha_fwd(int x_dtype, const void* x, const float* query, const float* att,
                  const uint8_t* head_keep, const int32_t* lengths,
                  float* out, float* align, float* head_w, float* ctx, float* ml,
                  void* workspace, size_t workspace_bytes,
                  int B, int T, int H, int dh,
                  int64_t x_stride_b, int64_t x_stride_t, dasv_stream_t stream);

/* typed conveniences (same semantics) */
int dasv_dmha_fwd_f32(const float* x, const float* query, const float* att,
                      const uint8_t* head_keep, const int32_t* lengths,
                      float* out, float* align, float* head_w, float* ctx, float* ml,
                      void* workspace, size_t workspace_bytes,
                      int B, int T, int H, int dh,
                      int64_t x_stride_b, int64_t x_stride_t, dasv_stream_t stream);
int dasv_dmha_fwd_bf16(const void* x, const float* query, const float* att,
                       const uint8_t* head_keep, const int32_t* lengths,
                       float* out, float* align, float* head_w, float* ctx, float* ml,
                       void* workspace, size_t workspace_bytes,
                       int B, int T, int H, int dh,
                       int64_t x_stride_b, int64_t x_stride_t, dasv_stream_t stream);
int dasv_dmha_fwd_f16(const void* x, const float* query, const float* att,
                      const uint8_t* head_keep, const int32_t* lengths,
                      float* out, float* align, float* head_w, float* ctx, float* ml,
                      void* workspace, size_t workspace_bytes,
                      int B, int T, int H, int dh,
                      int64_t x_stride_b, int64_t x_stride_t, dasv_stream_t stream);

/* ------------------------------------------------------------------------
 * Double multi-head attention pooling, backward (what torch autograd derives
 * from poolings.py:73-80, 45-51, 61-71).  One pass: read x, write grad_x.
 *
 *   grad_out [B,dh] f32   gradient w.r.t. `out` (or, when att is NULL, grad_ctx
 *                         [B,H,dh] is passed here and the head stage is skipped)
 *   head_w, ctx, ml, out  as produced by the forward launch (out is unused when att is NULL)
 *   grad_x   [B,T,D]      same dtype and strides as x
 *   grad_query [dh,H] f32, grad_att [dh] f32: OVERWRITTEN (not accumulated)
 *   workspace             dasv_dmha_bwd_workspace_bytes() bytes, no init needed
 * ------------------------------------------------------------------------ */
size_t dasv_dmha_bwd_workspace_bytes(int B, int T, int H, int dh);

int dasv_dmha_bwd(int x_dtype, const void* x, const float* query, const float* att,
                  const uint8_t* head_keep, const int32_t* lengths,
                  const float* head_w, const float* ctx, const float* ml, const float* out,
                  const float* grad_out,
                  void* grad_x, float* grad_query, float* grad_att,
                  void* workspace, size_t workspace_bytes,
                  int B, int T, int H, int dh,
                  int64_t x_stride_b, int64_t x_stride_t, dasv_stream_t stream);

/* ------------------------------------------------------------------------
 * conv3x3 (stride 1, pad 1) + bias + ReLU on NHWC fp32 activations, as an implicit GEMM on the matrix cores.
 * Replaces `F.relu(self.convXY(x))` of CNNs.py:38-48 / 72-86 for layers with Cin % 64 == 0 and
 * Cout % 128 == 0 (conv12 ... conv42 at the reference's kernel sizes; conv11 has Cin = 1 and stays in torch).
 *
 *   dasv_conv3x3_pack_weights: torch-layout weights [Cout,Cin,3,3] fp32 -> two bf16 planes [Cout][9][Cin]
 *                              (hi = bf16(w), lo = bf16(w - hi)), each dasv_conv3x3_packed_weight_bytes() long.
 *                              Call again whenever the weights change.
 *   dasv_conv3x3_bias_relu_nhwc: x [B,H,W,Cin] fp32 -> y [B,H,W,Cout] fp32 (y = relu?(conv(x) + bias)).
 *       mode 0: split-bf16 operands (x = hi + lo, three MFMAs per product, fp32 accumulate): ~2^-16 relative error
 *               per product, fp32-grade for this model's 1e-3 embedding bar.
 *       mode 1: plain bf16 operands (one MFMA per product).
 *       mode 2: fp16 operands (one MFMA per product, 11 mantissa bits; |activation| must stay below 65504): w_hi is the
 *               plane written by dasv_conv3x3_pack_weights_f16, w_lo is ignored (the "fp16 MFMA path" of BASELINE
 *               config 5 — forward only; gradients go through mode 0 or 1).
 * ------------------------------------------------------------------------ */
size_t dasv_conv3x3_packed_weight_bytes(int Cout, int Cin);
int dasv_conv3x3_pack_weights(const float* w, void* w_hi, void* w_lo, int Cout, int Cin, dasv_stream_t stream);
/* (bias: the layer's bias or NULL — behind the fp16 plane sit { max_n sum_k |w[n][k]|, max |bias| }, the bound on the layer's
 * output that scales a half-precision output tensor; w_f16 holds dasv_conv3x3_packed_weight_bytes_f16(Cout, Cin) bytes) */
size_t dasv_conv3x3_packed_weight_bytes_f16(int Cout, int Cin);
int dasv_conv3x3_pack_weights_f16(const float* w, const float* bias, void* w_f16, int Cout, int Cin, dasv_stream_t stream);
/*       mode 3 ("split8"): fp32-grade on fewer matrix-pipe cycles, maps up to 80 columns (forward here, data gradient through
 *               dasv_conv3x3_masked_x8_nhwc, weight gradient through dasv_conv3x3_wgrad_x8_nhwc): the
 *               operand c = v * 2^E is split c = hi + r with hi = fp16(c); hi*hi on the fp16 MFMA, both cross terms hi*r' + r*hi'
 *               on one block-scaled e4m3 MFMA per 64 input channels (residuals are 2^-11 of the value, 4 significant bits
 *               suffice).  ~1e-5 of the output scale against a float64 convolution (mode 0: 5e-6, mode 2: 3e-4) at ANY
 *               magnitude of activations and weights: E is an exact power of two per tensor, taken on the device from the
 *               tensor's RANGE RECORD (below), weights are scaled per output channel when they are packed, and the epilogue
 *               undoes both.  Inf / NaN propagate.  Mode 3 is reached through dasv_conv3x3_bias_relu_x8_nhwc (the
 *               mode argument of the generic entry points returns DASV_E_NULL for it: they carry no range records).
 *               w_hi = the fp16 plane and w_lo = the e4m3 planes + scales of dasv_conv3x3_pack_weights_x8 (bias: the layer's
 *               bias or NULL — its maximum bounds the layer's output); w_lo holds dasv_conv3x3_packed_weight_bytes_x8(Cout, Cin)
 *               bytes.
 *               dasv_conv3x3_pack_weights_x8_dgrad packs, straight from the FORWARD layer's weight w [Cout][Cin][3][3], the planes
 *               of that layer's data-gradient convolution (Cin outputs, Cout inputs: the weights transposed over (out, in) and
 *               flipped over (ky, kx) — no transposed copy is made): w_f16 / w_e4m3 sized for (Cin, Cout). */
size_t dasv_conv3x3_packed_weight_bytes_x8(int Cout, int Cin);
int dasv_conv3x3_pack_weights_x8(const float* w, const float* bias, void* w_f16, void* w_e4m3, int Cout, int Cin,
                                 dasv_stream_t stream);
int dasv_conv3x3_pack_weights_x8_dgrad(const float* w, void* w_f16, void* w_e4m3, int Cout, int Cin, dasv_stream_t stream);
/* the same planes through `scratch` (Cout * Cin * 9 floats; NULL = the call above): a tiled transpose-and-flip into scratch, then the
 * row-wise pack — contiguous reads and writes throughout.  The call above stages each gathered row (9 * Cout floats) in LDS and is the
 * faster one whenever that fits 64 KB (Cout <= 1818); this form is for longer rows */
int dasv_conv3x3_pack_weights_x8_dgrad_ws(const float* w, float* scratch, void* w_f16, void* w_e4m3, int Cout, int Cin, dasv_stream_t stream);

/* RANGE RECORDS.  A range record is uint32_t[2] in device memory beside an activation tensor: [0] = the bit pattern of
 * max |v| over the tensor (kept by the kernel that writes the tensor with an atomic maximum; the caller zeroes the record
 * before that kernel runs; a NaN anywhere is recorded as a NaN), [1] = for a PACKED tensor the exponent E it was packed with.
 * Every forward kernel below that writes an activation tensor takes an optional `y_range`; the split8 convolution requires
 * the `x_range` of its input.  For a tensor this library did not produce (the input features, a caller's own activations):
 * dasv_range_absmax_f32 fills a (zeroed) record from n floats (x 16-byte aligned).  No host synchronisation anywhere. */
int dasv_range_absmax_f32(const float* x, size_t n, uint32_t* range, dasv_stream_t stream);
int dasv_conv3x3_bias_relu_nhwc(int mode, const float* x, const void* w_hi, const void* w_lo, const float* bias, float* y,
                                int B, int H, int W, int Cin, int Cout, int relu, dasv_stream_t stream);
/* Same with a caller-provided workspace of dasv_conv3x3_workspace_bytes() bytes: when the map has too few output tiles to
 * fill the chip (single utterances — getEmbeddingExample.py, the B=1 scoring loop of train.py:117-133) the channel
 * slabs of each tile are divided over several workgroups (split-K) and a second launch sums the partial tiles in a fixed
 * order and applies bias / ReLU.  workspace_bytes() is 0 when one pass already fills the chip (then this is the call above). */
size_t dasv_conv3x3_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int dasv_conv3x3_bias_relu_nhwc_ws(int mode, const float* x, const void* w_hi, const void* w_lo, const float* bias, float* y,
                                   void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int relu,
                                   dasv_stream_t stream);
/* y = conv3x3(x) where mask > 0, else 0 (mask [B,H,W,Cout]).  The data gradient of `F.relu(convX2(F.relu(convX1(h))))`
 * w.r.t. convX1's pre-activation in one launch: x = gradient at convX2's output (already masked), weights = convX2's
 * transposed and flipped, mask = convX2's input (= convX1's ReLU output).  Saves autograd's threshold_backward pass.
 * colsum_partial (optional, maps up to 80 columns): [dasv_conv3x3_colsum_rows(mode,B,H,W,ragged)][Cout] per-channel sums of the
 * rows of y each workgroup wrote (one row per pixel tile of the launch — the tile shape follows mode and map width, hence the
 * query; ragged = 1 for the *_ragged_* entry points, where B = 1 and H = the stacked rows); their sum over the first axis is
 * convX1's bias gradient (autograd's separate `sum` pass). */
int dasv_conv3x3_masked_nhwc(int mode, const float* x, const void* w_hi, const void* w_lo, const float* mask, float* y,
                             float* colsum_partial, int B, int H, int W, int Cin, int Cout, dasv_stream_t stream);
int dasv_conv3x3_colsum_rows(int mode, int B, int H, int W, int ragged);
/* The same in the split8 mode (mode 3 above), which the range records make usable for GRADIENTS too: x (the gradient at the upper
 * layer's output, any magnitude — 1e-8 is common) is scaled into the operand planes' range by its record, the transposed weights
 * per channel.  x_range required, y_range optional (receives max |y|: the next layer down scales ITS gradient by it). */
int dasv_conv3x3_masked_x8_nhwc(const float* x, const uint32_t* x_range, const void* w_f16, const void* w_e4m3, const float* mask,
                                float* y, uint32_t* y_range, float* colsum_partial, int B, int H, int W, int Cin, int Cout,
                                dasv_stream_t stream);

/* Two layers in one launch: y = relu(conv3x3(relu(conv3x3_1ch(mel) + b1)) + bias) — the first block of the VGG front-end
 * (`F.relu(self.conv12(F.relu(self.conv11(x))))`, CNNs.py:38-39 / 72-73).  The one-channel layer's activations are
 * computed while the second layer's input window is staged, so its [B,H,W,Cmid] output never exists in HBM.
 * mel [B,H,W] fp32, w1 [Cmid,1,3,3] (torch layout), b1 [Cmid] or NULL, w_hi/w_lo: packed weights of the second layer
 * ([Cout][9][Cmid]); Cmid % 64 == 0, Cout % 128 == 0, W <= 80. */
int dasv_conv3x3_c1_conv3x3_bias_relu_nhwc(int mode, const float* mel, const float* w1, const float* b1, const void* w_hi,
                                           const void* w_lo, const float* bias, float* y, int B, int H, int W, int Cmid,
                                           int Cout, dasv_stream_t stream);

/* Weight gradient of the same convolution: gw[Cout,Cin,3,3] (torch layout, OVERWRITTEN) from x [B,H,W,Cin] and
 * g [B,H,W,Cout] = gradient w.r.t. the convolution output (ReLU mask applied by the caller), both NHWC fp32.
 * Replaces autograd's weight gradient of `F.relu(self.convXY(x))` (CNNs.py:38-48 / 72-86; loss.backward(), train.py:220).
 * Cin % 64 == 0, Cout % 64 == 0; modes as above; workspace: dasv_conv3x3_wgrad_workspace_bytes(), no init needed.
 * Deterministic (fixed-order split-K reduction, no atomics). */
size_t dasv_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int dasv_conv3x3_wgrad_nhwc(int mode, const float* x, const float* g, float* gw, void* workspace, size_t workspace_bytes,
                            int B, int H, int W, int Cin, int Cout, dasv_stream_t stream);
/* The same in the fp16 + e4m3 form of the `split8` forward mode: fp16 x fp16 products on the fp16 MFMA, both cross terms of the
 * fp16 residuals on one block-scaled e4m3 MFMA (64 matrix-pipe cycles per 16 x 16 x 64 instead of 96), operands scaled by powers
 * of two from their range records (x_range, g_range: [0] = bits of max |v|, an upper bound is enough; both mandatory). */
int dasv_conv3x3_wgrad_x8_nhwc(const float* x, const uint32_t* x_range, const float* g, const uint32_t* g_range, float* gw,
                               void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, dasv_stream_t stream);

/* conv3x3 with ONE input channel (+ bias + ReLU): x [B,H,W] fp32 -> y [B,H,W,Cout] NHWC fp32.  Replaces
 * `F.relu(self.conv11(x))` (CNNs.py:38 / 72); w is the torch weight [Cout,1,3,3].  Cout % 4 == 0, (Cout/4) | 256. */
int dasv_conv3x3_c1_bias_relu_nhwc(const float* x, const float* w, const float* bias, float* y, uint32_t* y_range, int B, int H,
                                   int W, int Cout, int relu, dasv_stream_t stream);
/* 2x2 stride-2 ceil-mode max-pool on NHWC fp32: x [B,H,W,C] -> y [B,ceil(H/2),ceil(W/2),C].  Replaces
 * `F.max_pool2d(x, 2, stride=2, ceil_mode=True)` (CNNs.py:40,44,48 / 74,78,82,86).  C % 4 == 0. */
/* x_range / y_range (optional, together): the pooled map's range record is its input's (every input lies in a window). */
int dasv_maxpool2x2_ceil_nhwc(const float* x, const uint32_t* x_range, float* y, uint32_t* y_range, int B, int H, int W, int C,
                              dasv_stream_t stream);

/* ------------------------------------------------------------------------
 * Ragged batches (inference): utterances of DIFFERENT lengths processed as one tall image.  The reference can only run
 * unequal utterances one per forward (train.py:117-133); zero-padding them into a batch is not equivalent (no masks
 * anywhere in the model), but the convolutions and pools are position-independent except at the utterance borders, so the
 * utterances are stacked along the time axis ([rows = sum T_i, W, C], NHWC) and the kernels take the borders from tables:
 *   row_span [rows][2] int32: first row and one-past-last row of the utterance that owns each row;
 *   pool_src [rows_out][2] int32: for each output row of the ceil-mode pool, its first input row and whether a second
 *                                 input row exists (it does not for the last row of an odd-length utterance).
 * Results equal the one-utterance-per-forward results (same arithmetic per output pixel).
 * ------------------------------------------------------------------------ */
int dasv_conv3x3_bias_relu_ragged_nhwc(int mode, const float* x, const void* w_hi, const void* w_lo, const float* bias, float* y,
                                       const int* row_span, void* workspace, size_t workspace_bytes, int rows, int W, int Cin,
                                       int Cout, int relu, dasv_stream_t stream);
int dasv_conv3x3_c1_bias_relu_ragged_nhwc(const float* x, const float* w, const float* bias, float* y, uint32_t* y_range,
                                          const int* row_span, int rows, int W, int Cout, int relu, dasv_stream_t stream);
int dasv_maxpool2x2_ceil_ragged_nhwc(const float* x, const uint32_t* x_range, float* y, uint32_t* y_range, const int* pool_src,
                                     int rows_in, int rows_out, int W, int C, dasv_stream_t stream);
/* The same three in the split8 mode, optionally on PACKED activations (see dasv_conv3x3_bias_relu_x8_nhwc). */
int dasv_conv3x3_bias_relu_ragged_x8_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                          const float* bias, void* y, int y_packed, uint32_t* y_range, const int* row_span, void* workspace,
                                          size_t workspace_bytes, int rows, int W, int Cin, int Cout, int relu, dasv_stream_t stream);
int dasv_conv3x3_c1_bias_relu_ragged_x8_nhwc(const float* x, const uint32_t* x_range, const float* w, const float* bias,
                                             void* y_packed, uint32_t* y_range, const int* row_span, int rows, int W, int Cout,
                                             int relu, dasv_stream_t stream);
int dasv_maxpool2x2_ceil_ragged_x8_nhwc(const float* x, const uint32_t* x_range, void* y_packed, uint32_t* y_range,
                                        const int* pool_src, int rows_in, int rows_out, int W, int C, dasv_stream_t stream);

/* Backward of the max-pool above: gx [B,H,W,C] (OVERWRITTEN) from x [B,H,W,C] and gy [B,ceil(H/2),ceil(W/2),C]; the
 * arg-max is recomputed (first maximum in row-major window order wins, as ATen does).  Replaces autograd's max_pool2d
 * backward for CNNs.py:40,44,48 / 74,78,82,86. */
int dasv_maxpool2x2_ceil_bwd_nhwc(const float* x, const float* gy, float* gx, int B, int H, int W, int C, dasv_stream_t stream);
/* (the scattered gradient's range record is gy's: max |gx| <= max |gy|, a bound, which is what a consumer's scale needs) */
/* Same, for x = a ReLU's output (`F.relu(self.convX2(..))` feeding the pool): the gradient is also passed through that
 * ReLU (zero where the window maximum is not > 0), which saves the separate mask pass over [B,H,W,C]. */
int dasv_maxpool2x2_ceil_relu_bwd_nhwc(const float* x, const float* gy, float* gx, float* colsum_partial, int B, int H, int W,
                                       int C, dasv_stream_t stream);
/* Training form of the pool: the forward launch also writes `idx` [B,Ho,Wo,C] bytes {arg-max position 0-3 (row-major in the window,
 * the first maximum wins as in ATen) | 4 if the maximum is > 0}, and the backward launch reads those bytes and gy only — not the
 * full-resolution input again.  relu_mask != 0: the pooled map was a ReLU's output, the gradient passes that ReLU too (the bit);
 * colsum_partial as in dasv_maxpool2x2_ceil_relu_bwd_nhwc (or NULL).  Results identical to the index-free pair. */
int dasv_maxpool2x2_ceil_idx_nhwc(const float* x, const uint32_t* x_range, float* y, uint32_t* y_range, uint8_t* idx, int B, int H,
                                  int W, int C, dasv_stream_t stream);
int dasv_maxpool2x2_ceil_bwd_idx_nhwc(const uint8_t* idx, const float* gy, float* gx, float* colsum_partial, int relu_mask, int B,
                                      int H, int W, int C, dasv_stream_t stream);
/* The TRAINING forms over a ragged batch — utterances of different lengths in one minibatch (BASELINE.json north_star; the reference
 * draws one window length per batch, train.py:205-207, and can only run unequal utterances one per forward): the pool with its index
 * bytes and its backward, the masked split8 data gradient, conv11's backward — the dense launches above with the border tables.  The
 * weight gradient of a ragged batch is dasv_conv3x3_wgrad_x8_nhwc on the stacked image (B = 1, H = rows) MINUS the same launch on the
 * 2-row images {last row of utterance i, first row of utterance i+1} that carry exactly the products across the borders
 * (doubleattentionspeakerverification_amd/conv.py, conv3x3_wgrad(..., span_borders=...)). */
int dasv_maxpool2x2_ceil_idx_ragged_nhwc(const float* x, const uint32_t* x_range, float* y, uint32_t* y_range, uint8_t* idx,
                                         const int* pool_src, int rows_in, int rows_out, int W, int C, dasv_stream_t stream);
int dasv_maxpool2x2_ceil_bwd_idx_ragged_nhwc(const uint8_t* idx, const float* gy, float* gx, float* colsum_partial, int relu_mask,
                                             const int* pool_src, int rows_in, int rows_out, int W, int C, dasv_stream_t stream);
int dasv_maxpool2x2_bwd_colsum_rows_ragged(int rows_out, int W, int C);
int dasv_conv3x3_masked_ragged_x8_nhwc(const float* x, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                       const float* mask, float* y, uint32_t* y_range, float* colsum_partial, const int* row_span,
                                       int rows, int W, int Cin, int Cout, dasv_stream_t stream);
/* ... and on PACKED activations (the packed training chain, below, over a ragged batch): the pool's packed output beside the arg-max
 * bytes, its packed scattered gradient, and the masked data gradient with packed x / packed mask. */
int dasv_maxpool2x2_ceil_idx_ragged_x8_nhwc(const float* x, const uint32_t* x_range, void* y_packed, uint32_t* y_range, uint8_t* idx,
                                            const int* pool_src, int rows_in, int rows_out, int W, int C, dasv_stream_t stream);
int dasv_maxpool2x2_ceil_bwd_idx_ragged_x8_nhwc(const uint8_t* idx, const float* gy, const uint32_t* gy_range, void* gx_packed,
                                                uint32_t* gx_range, float* colsum_partial, int relu_mask, const int* pool_src, int rows_in,
                                                int rows_out, int W, int C, dasv_stream_t stream);
int dasv_conv3x3_masked_ragged_x8p_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                        const void* mask_packed, float* y, uint32_t* y_range, float* colsum_partial, const int* row_span,
                                        int rows, int W, int Cin, int Cout, dasv_stream_t stream);
int dasv_conv3x3_c1_bwd_ragged_nhwc(const float* x, const float* y, const float* gy, float* gw, float* gb, void* workspace,
                                    size_t workspace_bytes, const int* row_span, int rows, int W, int Cout, dasv_stream_t stream);
/* Border tables of a ragged batch (utterances of different lengths stacked along time: what the reference does one utterance per
 * forward, train.py:117-133), made on the device in ONE launch from the B utterance lengths (lengths_dev int32 on the device;
 * lengths_host the same values on the host: the sizes below are host arithmetic).  rows_l = sum_b ceil(len_b / 2^l).
 * tables (int32, 2 * sum_{l < levels} (rows_l + rows_{l+1}) entries), level by level:
 *   spans_l [rows_l][2]    = {first row, one past the last row} of the utterance that owns the row (row_span of the `_ragged` entries)
 *   pools_l [rows_{l+1}][2] = {first input row, 1 if the 2x2 window has a second row} per pooled row (pool_src)
 * frame_index (optional, int64 [2][rows_levels]): {owner utterance, row inside it} of every output frame.  B <= 4096, levels <= 8. */
int dasv_ragged_tables(const int* lengths_dev, const int* lengths_host, int B, int levels, int* tables, long long* frame_index,
                       dasv_stream_t stream);
/* rows of the optional colsum_partial [rows][C] above (per-workgroup channel sums of gx = that conv's bias gradient after a
 * sum over the first axis); 0 when C does not allow it ((C/4) must divide 256). */
int dasv_maxpool2x2_bwd_colsum_rows(int B, int H, int W, int C);
/* Backward of the one-input-channel layer: from x [B,H,W], its output y [B,H,W,Cout] and gy (gradient w.r.t. y, before
 * the ReLU mask): gw [Cout,1,3,3] and gb [Cout] (or NULL), both OVERWRITTEN; fixed-order reduction.  Replaces autograd's
 * backward of `F.relu(self.conv11(x))` (CNNs.py:38 / 72).  workspace: dasv_conv3x3_c1_bwd_workspace_bytes().
 * y == NULL: gy already carries the mask (the data-gradient kernel of the layer above applied it, dasv_conv3x3_masked_*): the
 * launch then reads gy only. */
size_t dasv_conv3x3_c1_bwd_workspace_bytes(int B, int H, int W, int Cout);
int dasv_conv3x3_c1_bwd_nhwc(const float* x, const float* y, const float* gy, float* gw, float* gb, void* workspace,
                             size_t workspace_bytes, int B, int H, int W, int Cout, dasv_stream_t stream);

/* ------------------------------------------------------------------------
 * The split8 convolution (mode 3 above) and PACKED activations (inference chains of split8 convolutions).
 *   dasv_conv3x3_bias_relu_x8_nhwc: y = relu?(conv(x) + bias).  x_range REQUIRED (its [0] for an fp32 x, its [1] for a packed x);
 *   y_range optional for an fp32 y (receives max |y|), REQUIRED for a packed y.
 * A packed tensor has the geometry of the fp32 NHWC tensor it replaces — [B,H,W,C] floats, C % 64 == 0 — but every 256-byte
 * slab of 64 channels holds { fp16(c) x 64 | e4m3(c) x 64 | e4m3((c - hi) * 2^12) x 64 }, c = v * 2^E: what the convolution makes
 * of fp32 values while it stages its window, made once by the producer, whose record carries E: the consumer's staging becomes
 * a copy.  E comes from the tensor's maximum where the producer knows it before it writes (the max-pool: its input's
 * maximum), else from a bound on it (conv11 and the convolution itself: |y| <= max|x| * max_n sum_k |w[n][k]| + max|b|, a few
 * binades above the true maximum — the fp16 plane has 30 binades).  Producers: this convolution (y_packed), conv11
 * (dasv_conv3x3_c1_bias_relu_x8_nhwc; x_range = the record of the input features), the max-pool (dasv_maxpool2x2_ceil_x8_nhwc:
 * fp32 in, packed out).  Replaces nothing new in the reference: these are the calls above with one conversion moved
 * (CNNs.py:72-86 in eval mode).
 * ------------------------------------------------------------------------ */
int dasv_conv3x3_bias_relu_x8_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                   const float* bias, void* y, int y_packed, uint32_t* y_range, void* workspace, size_t workspace_bytes,
                                   int B, int H, int W, int Cin, int Cout, int relu, dasv_stream_t stream);
/* conv3x3 + bias + ReLU + 2x2 stride-2 ceil-mode max-pool in ONE launch: `F.max_pool2d(F.relu(self.convX2(h)), 2, stride=2,
 * ceil_mode=True)` (CNNs.py:39-40, 43-44, 47-48 / 73-74, 77-78, 81-82, 85-86).  y is the POOLED tensor
 * [B, ceil(H/2), ceil(W/2), Cout] — fp32, packed or fp16 like the un-pooled entry points' y; the full-resolution map never reaches
 * memory.  A workgroup's tile is a whole number of row pairs of one or more images, the pooled pixels are formed from the tile's
 * fp32 values before the output conversion (so the pooled values are exactly those of the two-launch form; a PACKED / fp16 output is
 * scaled by this layer's output BOUND like the un-pooled packed output — the stand-alone pool knows the exact maximum — which changes
 * the operand planes of the next layer by less than their own rounding).  y_range as above (its [0] = max |y|).
 * dasv_conv3x3_pool_supported(mode 2 | 3, ...): 1 when the launch exists for the shape (maps up to 80 columns, enough tiles that the
 * channel slabs need not be divided over workgroups); callers use the un-pooled entry point + dasv_maxpool2x2_ceil_* otherwise.
 * tile_pixels: 0 = the tile shape that wastes the fewest pixel rows (160 pixels = whole row pairs of 80- / 40- / 20- / 10-column
 * maps, else 256 / 192); 160 or 256 / 192 force one (A/B measurements; DASV_E_SHAPE when it does not exist for the shape). */
int dasv_conv3x3_pool_supported(int mode, int B, int H, int W, int Cin, int Cout);
int dasv_conv3x3_bias_relu_pool_x8_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                        const float* bias, void* y, int y_packed, uint32_t* y_range, int B, int H, int W, int Cin,
                                        int Cout, int tile_pixels, dasv_stream_t stream);
int dasv_conv3x3_bias_relu_pool_f16_nhwc(const void* x, int x_f16, const uint32_t* x_range, const void* w_f16, const float* bias,
                                         void* y, int y_f16, uint32_t* y_range, int B, int H, int W, int Cin, int Cout,
                                         int tile_pixels, dasv_stream_t stream);
/* The TRAINING form (train.py:215-220 through the same lines of CNNs.py): x and the pooled y fp32 or PACKED (x_packed / y_packed as in
 * dasv_conv3x3_bias_relu_pool_x8_nhwc), and idx [B, ceil(H/2), ceil(W/2), Cout]
 * bytes {arg-max position 0-3 (row-major in the window, the first maximum wins — ATen's tie-breaking) | 4 if the maximum is > 0}, the
 * bytes dasv_maxpool2x2_ceil_idx_nhwc writes: dasv_maxpool2x2_ceil_bwd_idx_nhwc(relu_mask = 1) turns the pooled gradient into the
 * gradient at convX2's pre-activation from them alone, so the full-resolution activation map is neither written nor kept. */
int dasv_conv3x3_bias_relu_pool_idx_x8_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                            const float* bias, void* y, int y_packed, uint32_t* y_range, uint8_t* idx, int B, int H, int W,
                                            int Cin, int Cout, dasv_stream_t stream);
/* The training chain on PACKED activations (every activation of the forward pass in the split8 operand form, nothing in fp32 except
 * the last pooled map): the backward's two consumers of an activation tensor read the packed form directly —
 *   dasv_conv3x3_masked_x8p_nhwc  the data gradient with the ReLU mask taken from a packed tensor (the sign of its fp16 plane);
 *                                 workspace as dasv_conv3x3_bias_relu_nhwc_ws (small batches; not with colsum_partial)
 *   dasv_conv3x3_wgrad_x8p_nhwc   the weight gradient with x packed (its pieces are parked in LDS as fetched: no conversion work
 *                                 in the producer waves; x_range[1] = the exponent the tensor was packed with). */
int dasv_conv3x3_masked_x8p_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                 const void* mask_packed, float* y, uint32_t* y_range, float* colsum_partial, void* workspace,
                                 size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, dasv_stream_t stream);
/*   dasv_maxpool2x2_ceil_bwd_idx_x8_nhwc  the pool's backward writing the scattered gradient PACKED (scaled by the exponent of gy's
 *                                 recorded maximum, which the scattered tensor shares; gx_range receives {maximum, exponent}): the
 *                                 x operand of convX2's data gradient (x_packed above) and the g operand of its weight gradient
 *                                 (g_packed below) are then copied, not converted */
int dasv_maxpool2x2_ceil_bwd_idx_x8_nhwc(const uint8_t* idx, const float* gy, const uint32_t* gy_range, void* gx_packed, uint32_t* gx_range,
                                         float* colsum_partial, int relu_mask, int B, int H, int W, int C, dasv_stream_t stream);
/* The data gradient of the SECOND layer with the FIRST layer's whole backward in its write-out pass (train.py:220 through CNNs.py:38-39 /
 * 72-73): conv11 has one input channel, so its weight gradient gw[c][tap] = sum_pix g1[pix][c] * mel[pix + tap] (g1 = this launch's
 * masked output) costs 9 FMAs per element while the tile is in hand, and its bias gradient is the column sum.  The gradient at
 * conv11's output (786 MB for a [64,300,80] batch) is then written and read by nobody.  c1_partial: [dasv_conv3x3_colsum_rows(3, B,
 * H, W, 0)][Cout][10] floats {nine taps, bias} per pixel tile; dasv_conv3x3_c1_bwd_reduce_parts sums the rows in a fixed order into
 * gw [Cout][1][3][3] and gb [Cout] (gb may be NULL).  mask: conv11's ReLU output, fp32 or packed (mask_packed). */
int dasv_conv3x3_masked_c1_x8p_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3, const void* mask,
                                    int mask_packed, const float* mel, float* c1_partial, int B, int H, int W, int Cin, int Cout,
                                    dasv_stream_t stream);
/* out[c] = sum over the rows of a [rows][C] partial-sum buffer (colsum_partial of the entry points above), in a fixed order. */
int dasv_colsum_reduce(const float* partial, int rows, int C, float* out, dasv_stream_t stream);
/* ... over a ragged batch (B = 1, H = the stacked rows, borders from row_span; partial rows: dasv_conv3x3_colsum_rows(3, 1, rows, W, 1)) */
int dasv_conv3x3_masked_c1_ragged_x8p_nhwc(const void* x, int x_packed, const uint32_t* x_range, const void* w_f16, const void* w_e4m3,
                                           const void* mask, int mask_packed, const float* mel, float* c1_partial, const int* row_span, int rows,
                                           int W, int Cin, int Cout, dasv_stream_t stream);
int dasv_conv3x3_c1_bwd_reduce_parts(const float* partial, int rows, float* gw, float* gb, int Cout, dasv_stream_t stream);
int dasv_conv3x3_wgrad_x8p_nhwc(const void* x_packed, const uint32_t* x_range, const void* g, int g_packed, const uint32_t* g_range, float* gw,
                                void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, dasv_stream_t stream);
int dasv_conv3x3_c1_bias_relu_x8_nhwc(const float* x, const uint32_t* x_range, const float* w, const float* bias, void* y_packed,
                                      uint32_t* y_range, int B, int H, int W, int Cout, int relu, dasv_stream_t stream);
int dasv_maxpool2x2_ceil_x8_nhwc(const float* x, const uint32_t* x_range, void* y_packed, uint32_t* y_range, int B, int H, int W,
                                 int C, dasv_stream_t stream);

/* ------------------------------------------------------------------------
 * Half-precision activation chain (mode 2, fp16 operands: BASELINE config 3 "bf16" / config 5 "fp16 MFMA path" name 16-bit
 * arithmetic).  Between the layers of an inference front-end the activations are stored as fp16 NHWC tensors (2 bytes per
 * element: half the HBM traffic of every layer, and the convolution's window staging becomes a copy) holding v * 2^E with E in
 * the tensor's range record — the same device-side scaling as the packed split8 tensors, so fp16's range (65504) is never the
 * limit: E comes from the bound on the layer's output, mapped to [2^14, 2^15).  x_f16 / y_f16: 0 = fp32 tensor, 1 = fp16 tensor.
 *   dasv_conv3x3_bias_relu_f16_nhwc   x_range required when x_f16 or y_f16, y_range when y_f16 (optional otherwise)
 *   dasv_conv3x3_c1_bias_relu_f16_nhwc  conv11 -> fp16 tensor (x_range: the record of the input features)
 *   dasv_maxpool2x2_ceil_f16_nhwc     fp16 tensor in; fp16 out (same exponent: halves are compared and copied) or fp32 out (unscaled)
 * Replaces nothing new in the reference: CNNs.py:72-86 in eval mode with 16-bit storage.
 * ------------------------------------------------------------------------ */
int dasv_conv3x3_bias_relu_f16_nhwc(const void* x, int x_f16, const uint32_t* x_range, const void* w_f16, const float* bias, void* y,
                                    int y_f16, uint32_t* y_range, void* workspace, size_t workspace_bytes, int B, int H, int W,
                                    int Cin, int Cout, int relu, dasv_stream_t stream);
int dasv_conv3x3_c1_bias_relu_f16_nhwc(const float* x, const uint32_t* x_range, const float* w, const float* bias, void* y_f16,
                                       uint32_t* y_range, int B, int H, int W, int Cout, int relu, dasv_stream_t stream);
int dasv_maxpool2x2_ceil_f16_nhwc(const void* x_f16, const uint32_t* x_range, void* y, int y_f16, uint32_t* y_range, int B, int H,
                                  int W, int C, dasv_stream_t stream);
/* the same three over ragged batches (row_span / pool_src as above): BASELINE config 5 is "variable-length batches ... fp16 MFMA path" */
int dasv_conv3x3_bias_relu_ragged_f16_nhwc(const void* x, int x_f16, const uint32_t* x_range, const void* w_f16, const float* bias,
                                           void* y, int y_f16, uint32_t* y_range, const int* row_span, void* workspace,
                                           size_t workspace_bytes, int rows, int W, int Cin, int Cout, int relu, dasv_stream_t stream);
int dasv_conv3x3_c1_bias_relu_ragged_f16_nhwc(const float* x, const uint32_t* x_range, const float* w, const float* bias, void* y_f16,
                                              uint32_t* y_range, const int* row_span, int rows, int W, int Cout, int relu,
                                              dasv_stream_t stream);
int dasv_maxpool2x2_ceil_ragged_f16_nhwc(const void* x_f16, const uint32_t* x_range, void* y, int y_f16, uint32_t* y_range,
                                         const int* pool_src, int rows_in, int rows_out, int W, int C, dasv_stream_t stream);

/* ------------------------------------------------------------------------
 * The same two launches on the front-end's own NHWC map: x / grad_x are [B, T, F, C] fp32 (what a channels_last
 * [B, C, T, F] tensor is in memory), i.e. the reference's `transpose(1, 2).contiguous().view(B, T, C*F)` (CNNs.py:88-89)
 * is folded into the kernels' addressing: feature d = c*F + f, D = C*F, dh = D / H = (C/H)*F.  All other tensors
 * (query [dh,H], att, out, align, head_w, ctx, ml, gradients of query / att) are exactly those of dasv_dmha_fwd / _bwd.
 * dasv_dmha_nhwc_supported: 1 for F == 5 (VGG4L on 80 mel bins) with 16, 32 or 64 channels per head; callers permute
 * and use dasv_dmha_fwd otherwise.  Workspaces: dasv_dmha_workspace_bytes / dasv_dmha_bwd_workspace_bytes of (B,T,H,dh).
 * ------------------------------------------------------------------------ */
int dasv_dmha_nhwc_supported(int F, int C, int H);
int dasv_dmha_fwd_nhwc_f32(const float* x, const float* query, const float* att, const uint8_t* head_keep,
                           const int32_t* lengths, float* out, float* align, float* head_w, float* ctx, float* ml,
                           void* workspace, size_t workspace_bytes, int B, int T, int F, int C, int H, dasv_stream_t stream);
int dasv_dmha_bwd_nhwc_f32(const float* x, const float* query, const float* att, const int32_t* lengths, const float* head_w,
                           const float* ctx, const float* ml, const float* out, const float* grad_out, float* grad_x,
                           float* grad_query, float* grad_att, void* workspace, size_t workspace_bytes, int B, int T, int F,
                           int C, int H, dasv_stream_t stream);

/* ------------------------------------------------------------------------
 * AM-softmax head + cross-entropy (loss.py:37-52 AMSoftmax.forward and train.py:219 nn.CrossEntropyLoss), forward and backward:
 * cos(theta)[b][n] = <x_b, W_n> / (max(|x_b|, 1e-12) max(|W_n|, 1e-12)) as one fp32 product with the norms taken from the same
 * tiles, the margin by a label compare on the device (the reference builds a [B,N] one-hot on the HOST and copies it back,
 * loss.py:46-48), logits = scale * (cos - margin * onehot / (1 + alpha)) (alpha: loss.py:28-35, 0 without annealing), and —
 * when loss / lse are given — the mean cross-entropy of the logits with the row log-sum-exps, without a [B,N] softmax.
 *   x [B,E], W [E,N], label [B] int64;  costh [B,N] (always written), logits [B,N] or NULL, loss [1] + lse [B] or both NULL,
 *   xinv [B], winv [N] (the reciprocal clamped norms, saved for backward).
 * Backward: gx [B,E], gW [E,N] (OVERWRITTEN) from either the fused loss (lse given; g_loss [1] = upstream gradient of the loss
 * or NULL for 1) or upstream gradients of the two module outputs (lse NULL; g_costh / g_logits [B,N], either may be NULL).
 * Deterministic (fixed-order reductions).  workspace: dasv_amsoftmax_workspace_bytes(), no init needed.
 * Labels must lie in [0, N): the reference raises on anything else (scatter_, loss.py:47; CrossEntropyLoss, train.py:219); a launch
 * cannot, so an out-of-range label makes the fused loss NaN (the row has no target column) — check labels on the host if a finite
 * loss must mean valid labels; costh / logits of such a row are the margin-free values.
 * ------------------------------------------------------------------------ */
size_t dasv_amsoftmax_workspace_bytes(int B, int E, int N);
int dasv_amsoftmax_fwd(const float* x, const float* W, const int64_t* label, float* costh, float* logits, float* loss, float* lse,
                       float* xinv, float* winv, void* workspace, size_t workspace_bytes, int B, int E, int N, float margin,
                       float scale, float alpha, dasv_stream_t stream);
int dasv_amsoftmax_bwd(const float* x, const float* W, const int64_t* label, const float* costh, const float* lse, const float* xinv,
                       const float* winv, const float* g_loss, const float* g_costh, const float* g_logits, float* gx, float* gW,
                       void* workspace, size_t workspace_bytes, int B, int E, int N, float margin, float scale, float alpha,
                       dasv_stream_t stream);

/* name of the kernel variant the last dasv_dmha_fwd call on this thread selected (introspection: smoke tests and bench.py
 * record it; e.g. "vec<dtype=0,G=16,NI=1,R=4> nw=8 wpu=4 upw=2 lds=11200") */
const char* dasv_dmha_last_variant(void);

#ifdef DASV_DEBUG
/* ------------------------------------------------------------------------
 * Developer surface — NOT in libdasv_hip.so.  The build also links libdasv_hip_dbg.so from the same sources with
 * -DDASV_DEBUG; only that library exports what follows (benchmark sweeps and the read-floor probe of bench.py).
 * Process-global, not thread-safe.
 * waves_per_group: wavefronts per workgroup (0 = auto); waves_per_head: wavefronts
 * cooperating on one (utterance, head) (0 = auto); lds_pad_bytes: extra LDS per
 * workgroup to cap workgroups per CU (0 = none); force_generic: 1 = always take
 * the shape-agnostic kernel; rows_per_pass: rows a lane keeps in flight per pass
 * (0 = default for the shape; ignored when no such variant is compiled).
 * ------------------------------------------------------------------------ */
void dasv_debug_set_tuning(int waves_per_group, int waves_per_head, int lds_pad_bytes, int force_generic,
                           int rows_per_pass);
/* Measurement aid: reads `bytes` (16-byte aligned, whole 16-byte vectors) at x once with the simplest possible kernel and
 * adds per-wave sums into scratch[0..2047] (>= 2048 floats, caller-zeroed).  bench.py times it beside the fused pooling
 * kernel: no kernel that streams x once can be faster than this on the same buffers. */
int dasv_debug_read_floor(const void* x, size_t bytes, float* scratch, dasv_stream_t stream);
/* A/B aid: 0 = the convolutions never take the patch-tile geometry (16 x 16 pixel tiles on maps whose width is a multiple of 16);
 * 1 (default) = automatic. */
void dasv_debug_conv_patch_tiles(int on);
#endif /* DASV_DEBUG */

/* ------------------------------------------------------------------------
 * bf16 STORAGE chain for the bf16-operand mode (mode 1; BASELINE config 3 names bf16): activations between the layers as bf16 NHWC
 * tensors — plain round-to-nearest-even values: no scale, no range record (bf16 has fp32's exponent range) — written by the producer
 * and copied into the operand window by the consumer, like the fp16 chain above.  x_bf16 / y_bf16: that tensor is bf16; w_bf16 = the
 * `hi` plane of dasv_conv3x3_pack_weights.  Split-K (workspace) only with an fp32 y.  The inference chain of `CNNs.VGG*` in
 * conv_mode "bf16" (`half_activations`): conv11 -> bf16 -> conv12 -> bf16 -> pool -> bf16 -> ... -> last pool -> fp32 map.
 * ------------------------------------------------------------------------ */
int dasv_conv3x3_bias_relu_bf16_nhwc(const void* x, int x_bf16, const void* w_bf16, const float* bias, void* y, int y_bf16, void* workspace,
                                     size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int relu, dasv_stream_t stream);
int dasv_conv3x3_c1_bias_relu_bf16_nhwc(const float* x, const float* w, const float* bias, void* y_bf16, int B, int H, int W, int Cout,
                                        int relu, dasv_stream_t stream);
int dasv_maxpool2x2_ceil_bf16_nhwc(const void* x_bf16, void* y, int y_bf16, int B, int H, int W, int C, dasv_stream_t stream);

/* ------------------------------------------------------------------------
 * The optimizer step of the training loop (train.py:224 `optimizer.step()` with train.py:82-84 `optim.Adam(net.parameters(), lr,
 * weight_decay)`) over FLAT fp32 buffers: parameters, gradients and both moments share one layout of 64-element-aligned
 * slots (dist.FlatGradReducer's), so the whole model is one HBM pass.  torch's Adam arithmetic: grad += weight_decay * param;
 * exp_avg = b1 exp_avg + (1 - b1) grad; exp_avg_sq = b2 exp_avg_sq + (1 - b2) grad^2; param -= lr / (1 - b1^step) * exp_avg /
 * (sqrt(exp_avg_sq) / sqrt(1 - b2^step) + eps).  active_blocks (optional): one byte per 64 elements, 0 = leave that block alone —
 * parameters that received no gradient get no update and no weight decay, as with `grad is None` in the reference (model.py:46,50:
 * b1.*, b3.*).  zero_grad: clear the gradient on the way (the next step accumulates into it).  n % 64 == 0, step >= 1.
 * ------------------------------------------------------------------------ */
int dasv_adam_step_f32(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* active_blocks, long long n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, long long step, int zero_grad, dasv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DASV_HIP_H */
