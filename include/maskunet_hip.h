/* maskunet_hip.h -- C ABI of libmaskunet_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the MaskAttn-UNet forward/backward hot path.  The reference
 * (Belis0811/MaskUnet) has no FFI of its own: its hot path is a chain of torch aten ops invoked from
 * nn.Module.forward (code/ade20k/ade_semantic.py:152-314) and their autograd backward (:400).  Each
 * entry point below replaces one such op group; the reference line it stands for is cited.  The
 * Python binding that a maintainer would add is the ctypes table in maskunet_amd/_lib.py (shown in
 * INTEGRATION.md).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all memory
 *     (outputs and workspaces included), nothing is allocated or freed inside, no host sync;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it (capturable);
 *   - activations are NHWC rows: a tensor [B,H,W,C] is M = B*H*W rows of C elements with row
 *     stride `ld` (elements); internal channel counts are multiples of 32 (callers zero-pad);
 *   - dtype: MU_F32 (0) = fp32 storage + exact-fp32 MFMA, MU_F16 (1) = fp16 storage + fp32 accumulate, MU_F32X (2) = fp32 storage +
 *     split-bf16 matrix products (matrix entry points only, see below);
 *     per-channel parameters, statistics and all parameter gradients are always fp32;
 *   - return value: 0 on success, <0 on error (MU_ERR_*); no exceptions cross the boundary.
 */
#ifndef MASKUNET_HIP_H
#define MASKUNET_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MU_OK 0
#define MU_ERR_ARG (-1)       /* null pointer / non-positive size / unknown enum            */
#define MU_ERR_SHAPE (-2)     /* shape not supported by the kernels (alignment, channel set) */
#define MU_ERR_LAUNCH (-3)    /* HIP launch failure                                           */
#define MU_ERR_WORKSPACE (-4) /* workspace smaller than mu_*_workspace_bytes()                */

#define MU_F32 0
#define MU_F16 1
/* fp32 storage, matrix products on the 16-bit matrix cores with (hi, lo) splits of the fp32 operands, fp32 accumulate (~1e-5 relative per
 * product or better; torch's float32_matmul_precision "high").  Accepted by the matrix entry points (mu_conv_fwd, mu_conv_fwd_fused,
 * mu_conv1x1_fwd_add, mu_conv_wgrad, mu_attn_*); every other entry point takes MU_F32 for the same tensors.
 * With MU_F32X the MATRIX OPERANDS of those entry points are passed ENCODED (same size, same strides; the split then costs one pass per
 * tensor instead of VALU work per fragment per wave):
 *   - 1x1 layers / Linear (taps = 1) -- x and w of the convolutions, x and dy of the weight gradient: every aligned 16-byte chunk of four
 *     fp32 values re-written as [4 x bf16 hi | 4 x bf16 lo] by mu_split_encode; three bf16 MFMAs per product;
 *   - 3x3 layers (taps = 9; round 6) -- x and w of mu_conv_fwd / mu_conv_fwd_stats / mu_conv_fwd_fused: every aligned 16-byte chunk
 *     re-written as [4 x fp16 hi | 4 x fp16 lo] by mu_split_encode_h4 (|value| < 65504; 22 mantissa bits down to an absolute floor of
 *     2^-25), the weights under a static shift of 2^6 that the epilogues undo (mu_prep_weight / mu_prep_weights_multi with MU_F32X
 *     write every layer's layouts in the right encoding); three fp16 MFMAs per product.  Their BACKWARD has its own entry points:
 *     mu_conv_dgrad_h / mu_conv_wgrad_h take dy as ONE power-of-two-scaled fp16 operand (mu_bn_act_bwd_h, mu_bn_pair_bwd_h or
 *     mu_dy_encode_h) against the fp16 pair of the partner: two MFMAs per product.  mu_conv_wgrad with taps = 9 and MU_F32X serves
 *     only the <= 3-channel first layer (a plain-FMA kernel on plain fp32 operands);
 *   - qkv of the attention sweeps (mu_attn_*): every aligned 32-byte group of eight fp32 values re-written as
 *     [8 x fp16 hi | 8 x fp16 lo] by mu_split_encode_h (|value| < 65504; the softmax probabilities and dS then enter the matrix core as
 *     single fp16 operands: two MFMAs per P V / dS K / dS^T Q / P^T dO product, three for Q K^T and dO V^T).
 * Outputs, biases, residual / addend tensors, x / oattn / grad_out / dY of the attention block are plain fp32.  Producers that write an encoded form directly and save the encoding pass: mu_bn_act_fwd with MU_F32X writes
 * y in the 3x3 operand encoding (for a y that only feeds a 3x3 convolution), mu_bn_act_bwd_h / mu_bn_pair_bwd_h write dx as the scaled fp16
 * dy of the 3x3 convolution in front of the BatchNorm; their inputs, dres and all statistics stay plain fp32. */
#define MU_F32X 2

#define MU_ACT_NONE 0
#define MU_ACT_GELU 1 /* exact erf GELU, ade_semantic.py:201,208 */
#define MU_ACT_RELU 2 /* ade_semantic.py:286                      */

/* library / build identification ("gfx950"); host pointer to a static string */
const char* mu_version_host(void);

/* ---- layout -------------------------------------------------------------------------------- */
/* dst[b][c][r] = (dst_dtype) src[b][r][c], b < batch, r < R, c < C.  Replaces x.view(B,C,HW).permute
 * (ade_semantic.py:168), the `.view(B,C,H,W)` re-interpretation of [B,N,C] (:190) and the
 * NCHW<->NHWC conversions at the module boundary. */
int mu_transpose(const void* src, int src_dtype, long src_ld, void* dst, int dst_dtype, long dst_ld, int batch, int R, int C,
                 void* stream);
/* as mu_transpose, and additionally dst[b][c][r] = 0 for R <= r < R_pad <= dst_ld (the zero channel padding of the NHWC
 * layout). */
int mu_transpose_pad(const void* src, int src_dtype, long src_ld, void* dst, int dst_dtype, long dst_ld, int batch, int R, int C,
                     int R_pad, void* stream);
/* fp32x operand encoding (see MU_F32X): n_elems fp32 values (a multiple of 4, 16-byte aligned, contiguous rows) -> the chunk-encoded
 * operand; dst may be src (in place). */
int mu_split_encode(const void* src, void* dst, long n_elems, void* stream);
/* fp32x 3x3-CONVOLUTION operand encoding (see MU_F32X): n_elems fp32 values (a multiple of 4, 16-byte aligned, contiguous rows) ->
 * [4 fp16 hi | 4 fp16 lo] per chunk of four, hi = fp16(x), lo = fp16(x - hi); dst may be src (in place).  The input of a 3x3 layer
 * (nn.Conv2d k = 3 in ConvBlock, ade_semantic.py:199,202) that no producer wrote encoded. */
int mu_split_encode_h4(const void* src, void* dst, long n_elems, void* stream);
/* mu_split_encode_h4 with a second output: dst16 = the fp16 rounding of the values (n_elems halves: the hi halves as plain rows) -- the
 * saved-input form of the one-term weight gradient mu_conv_wgrad_h1.  Out of place. */
int mu_split_encode_h4x(const void* src, void* dst, void* dst16, long n_elems, void* stream);
/* A plain fp32 gradient (n_elems values, a multiple of 4) -> ONE power-of-two-scaled fp16 operand dy_h (n_elems halves; must not alias
 * dy) + dy_scale = {S, 1 / S} on the device (S max|dy| in [2^13, 2^14); no host sync): the dy form of mu_conv_dgrad_h / mu_conv_wgrad_h
 * for a 3x3 layer whose dy does not come out of mu_bn_act_bwd_h (a conv without a BatchNorm behind it, city_instance.py:243). */
long mu_dy_encode_h_workspace_bytes(void);
int mu_dy_encode_h(const void* dy, void* dy_h, float* dy_scale, long n_elems, void* workspace, long ws_bytes, void* stream);
/* fp32x ATTENTION operand encoding (see MU_F32X): n_elems fp32 values (a multiple of 8, 32-byte aligned) -> [8 fp16 hi | 8 fp16 lo] per
 * group of eight, hi = fp16(x), lo = fp16(x - hi); dst may be src (in place).  qkv of mu_attn_fwd / mu_attn_bwd* with MU_F32X. */
int mu_split_encode_h(const void* src, void* dst, long n_elems, void* stream);
/* y[M][Cout] = x[M][Cin] w^T + bias with y written DIRECTLY in the attention operand encoding (the mu_split_encode_h form): the q/k/v
 * projection nn.Linear(C, C) x 3 (ade_semantic.py:170-172) of the MU_F32X attention block as one 1x1 layer, without the separate
 * encoding pass over qkv.  x, w chunk-encoded MU_F32X operands (mu_split_encode), bias fp32 or NULL; Cin % 32 == 0, Cout % 64 == 0. */
int mu_conv1x1_fwd_enc_h(const void* x, const void* w, const float* bias, void* y, long M, int Cin, int Cout, long x_ld, long y_ld,
                         void* stream);
/* elementwise dtype conversion of n elements */
int mu_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, void* stream);
/* OIHW fp32 parameter -> tap-major compute layout [taps][rows_pad][cols_pad].
 * mode 0: forward weights (rows=O, cols=I); mode 1: data-gradient weights (taps flipped, rows=I, cols=O);
 * mode 2: both in one launch, dst = the mode-0 block followed by the mode-1 block [taps][cols_pad][rows_pad].
 * dtype MU_F32X: fp32-sized blocks in their operand encodings (see MU_F32X) -- taps = 1: both blocks bf16 chunk-encoded; taps = 9: the
 * forward block fp16 chunk-encoded (x 2^6), the data-gradient block as "HL" rows for mu_conv_dgrad_h: each row (tap, in) of the padded
 * output-channel count n becomes [n fp16 lo | n fp16 hi] of 2^6 w (the same bytes). */
int mu_prep_weight(const float* w_oihw, void* dst, int dtype, int O, int I, int taps, int rows_pad, int cols_pad, int mode,
                   void* stream);

/* mu_prep_weight for MANY layers in one launch (a training forward re-lays every conv weight of the model: nn.Conv2d keeps OIHW fp32
 * masters, ade_semantic.py:199,202,284).  jobs = DEVICE array of njobs x 10 int64:
 *   { address of the OIHW fp32 weight, dst offset in elements from dst_base, first_tile, O, I, taps, rows_pad, cols_pad, mode, 0 }
 * a layer occupies the 32 x 32 tiles [first_tile, first_tile + rows_pad/32 * cols_pad/32) of the launch (rows = O, cols = I, both padded
 * to multiples of 32; mode 0, 1 or 2 as above with rows/cols always meaning out/in); ntiles = the sum over layers.  njobs <= 128.
 * Same values, bit for bit, as njobs mu_prep_weight calls. */
int mu_prep_weights_multi(const void* jobs, int njobs, long ntiles, void* dst_base, int dtype, void* stream);

/* The q/k/v projection of one attention block -- three nn.Linear(C, C) with bias (ade_semantic.py:157-159,170-172) -- as ONE [3C, C]
 * 1x1 layer: dst = the forward block [3C][C] followed by the data-gradient block [C][3C] (mu_prep_weight mode 2 of the concatenated
 * weight), bias[3C] = the concatenated fp32 biases.  C % 32 == 0. */
int mu_prep_qkv(const float* wq, const float* wk, const float* wv, const float* bq, const float* bk, const float* bv, void* dst,
                float* bias, int dtype, int C, void* stream);

/* ---- convolution / linear (MFMA implicit GEMM) --------------------------------------------- */
/* y[p][co] = bias[co] + sum_{tap,ci} x[p+shift(tap)][ci] * w[tap][co][ci]; taps = 9 (3x3, pad 1) or 1.
 * Replaces nn.Conv2d in ConvBlock (ade_semantic.py:199,202), the 1x1 heads (:284, city_instance.py:243-249)
 * and nn.Linear query/key/value (:170-172, as one [3C,C] weight).  Called with mode-1 weights it is the
 * data-gradient of the same layer. */
int mu_conv_fwd(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int taps, long x_ld,
                long y_ld, int dtype, void* stream);
/* y = conv1x1(x, w) + addend over M rows (addend has y's row stride): the two gradients of Mask2FormerAttention's input -- the
 * data-gradient of the q/k/v projection and the residual branch of `attention_output += x` (ade_semantic.py:187) -- joined in the
 * projection's epilogue.  Only where mu_conv1x1_add_supported(...) == 1; MU_ERR_SHAPE otherwise (callers then use mu_conv_fwd + mu_add). */
int mu_conv1x1_add_supported(int Cin, int Cout, int dtype);
int mu_conv1x1_fwd_add(const void* x, const void* w, const void* addend, void* y, long M, int Cin, int Cout, long x_ld, long y_ld, int dtype,
                       void* stream);
/* Inference epilogue: y = act(conv(x, w) * scale[co] + shift[co] + res) with per-channel fp32 scale / shift (either may be NULL = 1 / 0),
 * res (may be NULL) a tensor with y's shape and row stride, act = MU_ACT_*.  One launch for Conv2d -> BatchNorm2d(eval) [-> BatchNorm2d(eval)]
 * [-> + x] -> GELU / ReLU of ConvBlock and the heads (ade_semantic.py:199-208, 283-287; validation loop :443-471) with (scale, shift)
 * from mu_bn_eval_fold; every shape mu_conv_fwd accepts. */
int mu_conv_fwd_fused(const void* x, const void* w, const float* scale, const float* shift, const void* res, int act, void* y, int B, int H,
                      int W, int Cin, int Cout, int taps, long x_ld, long y_ld, int dtype, void* stream);
/* mu_conv_fwd that also leaves per-tile BatchNorm statistics of its (rounded) output: stat_part[rows][Cout][2] floats =
 * (sum, sum of squares) per output channel, rows = mu_conv_stats_rows(...) (0 = this shape has no statistics epilogue;
 * stat_part must then be NULL).  Feeds mu_bn_train_stats_rows and saves the separate statistics sweep of
 * conv -> BatchNorm2d (ade_semantic.py:199-200, 202-204). */
int mu_conv_stats_rows(int B, int H, int W, int Cin, int Cout, int taps, int dtype);
int mu_conv_fwd_stats(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int taps,
                      long x_ld, long y_ld, int dtype, float* stat_part, void* stream);
/* fp32x 3x3 layers, two-term data gradient (round 6; autograd of nn.Conv2d k = 3, ade_semantic.py:199,202,400): dx[p][ci] =
 * (1 / S) sum_{tap,co} dy_h[p - shift(tap)][co] * w[co][ci][tap] with dy_h = fp16(S dy) ONE scaled fp16 operand (rows of Cin halves,
 * stride dy_ld halves; Cin = the layer's OUTPUT channels) and w_hl the HL data-gradient block of mu_prep_weight(MU_F32X, taps 9);
 * dx plain fp32 rows of Cout floats (the layer's INPUT channels, stride dx_ld floats); dy_scale = {S, 1 / S} on the device. */
int mu_conv_dgrad_h(const void* dy_h, const void* w_hl, const float* dy_scale, void* dx, int B, int H, int W, int Cin, int Cout, long dy_ld,
                    long dx_ld, void* stream);
/* fp32x 3x3 layers, two-term weight gradient: dw_oihw as mu_conv_wgrad from x = the layer's input in the 3x3 operand encoding
 * (mu_split_encode_h4 form, Cin channels, stride x_ld floats) and dy_h / dy_scale as above (Cout halves per row, stride dy_ld halves).
 * cin_valid > 3 (the first layer keeps mu_conv_wgrad on plain operands). */
long mu_conv_wgrad_h_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int mu_conv_wgrad_h(const void* x, const void* dy_h, const float* dy_scale, float* dw_oihw, int B, int H, int W, int Cin, int Cout,
                    int cin_valid, int cout_valid, long x_ld, long dy_ld, void* workspace, long ws_bytes, void* stream);
/* ... and its ONE-term form: x_h = the fp16 rounding of the input (rows of Cin halves, stride x_ld halves: the second output of
 * mu_bn_act_fwd_enc / mu_split_encode_h4x).  One MFMA per product on the fp16 kernels; dW sums over every pixel of the batch and the
 * roundings of x are random-signed: the sum carries ~2^-12 of the root-sum-square of its terms (no change of any gradient metric in the
 * oracle sizing).  Workspace: mu_conv_wgrad_workspace_bytes(B, H, W, Cin, Cout, 9). */
int mu_conv_wgrad_h1(const void* x_h, const void* dy_h, const float* dy_scale, float* dw_oihw, int B, int H, int W, int Cin, int Cout,
                     int cin_valid, int cout_valid, long x_ld, long dy_ld, void* workspace, long ws_bytes, void* stream);
/* dw_oihw[o][i][tap] = sum_p dy[p][o] * x[p+shift(tap)][i] for o < cout_valid, i < cin_valid (fp32, OIHW). */
long mu_conv_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int taps);
int mu_conv_wgrad(const void* x, const void* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout, int taps, int cin_valid,
                  int cout_valid, long x_ld, long dy_ld, void* workspace, long ws_bytes, int dtype, void* stream);
/* mu_conv_wgrad that also returns the bias gradient db[cout_valid] = sum over pixels of dy (nn.Linear query/key/value biases
 * :170-172, final_layer / head Conv2d biases :284) from the same sweep: the dy tiles are multiplied with one more column of ones.
 * Only where mu_conv_wgrad_bias_supported(...) == 1 (fp16 1x1 layers served by the wide tiles); MU_ERR_SHAPE otherwise --
 * callers then use mu_conv_wgrad + mu_colsum.  Same workspace size. */
int mu_conv_wgrad_bias_supported(int Cin, int Cout, int taps, int dtype);
int mu_conv_wgrad_bias(const void* x, const void* dy, float* dw_oihw, float* db, int B, int H, int W, int Cin, int Cout, int taps,
                       int cin_valid, int cout_valid, long x_ld, long dy_ld, void* workspace, long ws_bytes, int dtype, void* stream);
/* out[c] = sum_r x[r][c]  (bias gradients); dtype MU_F32X: x is chunk-encoded (mu_split_encode form) */
long mu_colsum_workspace_bytes(int C);
int mu_colsum(const void* x, long M, int C, long ld, float* out, void* workspace, long ws_bytes, int dtype, void* stream);

/* ---- BatchNorm2d (+activation, +residual) --------------------------------------------------- */
/* nn.BatchNorm2d training statistics (ade_semantic.py:200,204,219,240,285): biased batch variance ->
 * mean/rstd; running stats updated with the unbiased variance and `momentum` for c < c_valid
 * (running_* may be NULL); *num_batches_tracked (int64 on the device, may be NULL) is incremented. */
long mu_bn_workspace_bytes(int C);
int mu_bn_train_stats(const void* x, long M, int C, long ld, float* mean, float* rstd, float* running_mean, float* running_var,
                      long* num_batches_tracked, int c_valid, float momentum, float eps, void* workspace, long ws_bytes, int dtype,
                      void* stream);
/* the same from the rows written by mu_conv_fwd_stats (M = pixels per channel behind those rows) */
int mu_bn_train_stats_rows(const float* stat_part, int rows, long M, int C, float* mean, float* rstd, float* running_mean,
                           float* running_var, long* num_batches_tracked, int c_valid, float momentum, float eps, void* workspace,
                           long ws_bytes, void* stream);
/* eval mode: mean/rstd from the running statistics */
int mu_bn_eval_stats(const float* running_mean, const float* running_var, float eps, float* mean, float* rstd, int C, int c_valid,
                     void* stream);
/* eval mode, folded: the affine map of one BatchNorm2d with running statistics, or of two applied back to back (DownSample / UpSample
 * tails, :218-219, 239-240), behind a conv with optional bias: scale[c], shift[c] for mu_conv_fwd_fused; entries c >= c_valid are (0, 0).
 * gamma / beta / conv_bias and the whole second layer may be NULL. */
int mu_bn_eval_fold(const float* running_mean1, const float* running_var1, const float* gamma1, const float* beta1, float eps1,
                    const float* running_mean2, const float* running_var2, const float* gamma2, const float* beta2, float eps2,
                    const float* conv_bias, float* scale, float* shift, int C, int c_valid, void* stream);
/* y = act( res + (x-mean)*rstd*gamma + beta ); res may be NULL.  ConvBlock's BN+GELU (:200-201), BN (+x, GELU)
 * (:204,208) and final_layer's BN+ReLU (:285-286). */
int mu_bn_act_fwd(const void* x, const void* res, void* y, long M, int C, long ld, const float* mean, const float* rstd,
                  const float* gamma, const float* beta, int act, int dtype, void* stream);
/* fp32x: mu_bn_act_fwd(MU_F32X) -- y written in the 3x3 operand encoding -- with a second output y16 = the fp16 rounding of y as plain rows
 * of C halves (row stride C; contiguous input rows): the form of y the backward of the convolution behind keeps (mu_conv_wgrad_h1). */
int mu_bn_act_fwd_enc(const void* x, const void* res, void* y_enc, void* y16, long M, int C, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, int act, void* stream);
/* backward of mu_bn_act_fwd: dx (wrt x), dres (wrt res, iff res given), dgamma, dbeta.  training=1 uses the
 * batch-statistics formula, 0 treats mean/rstd as constants. */
int mu_bn_act_bwd(const void* x, const void* res, const void* grad_out, void* dx, void* dres, long M, int C, long ld,
                  const float* mean, const float* rstd, const float* gamma, const float* beta, int act, int training, float* dgamma,
                  float* dbeta, void* workspace, long ws_bytes, int dtype, void* stream);
/* BatchNorm pair: DownSample / UpSample apply nn.BatchNorm2d directly to the output of ConvBlock's last nn.BatchNorm2d
 * (ade_semantic.py:216-219, 237-240; maxpool_conv[2..3], conv[1..2]).  In training mode the second layer's batch statistics
 * follow from the first's (mean = beta1, biased var = gamma1^2 * var1/(var1+eps1)), so the pair is one normalisation of the conv
 * output: mu_bn_pair_compose turns (rstd1, gamma1, beta1, gamma2) into gamma_eff for mu_bn_act_fwd(.., gamma_eff, beta2), updates the
 * second layer's running statistics / step counter, and leaves the per-channel factors of the backward:
 * mu_bn_act_bwd_scaled(.., gamma_eff, .., xhat_scale) returns dx, dbeta = dbeta2 and dgamma = A;
 * dgamma2 = dgamma2_coef * A, dgamma1 = dgamma1_coef * A, dbeta1 = 0.  All vectors have C (padded) entries. */
int mu_bn_pair_compose(const float* rstd1, const float* gamma1, const float* beta1, const float* gamma2, int C, int c_valid, long M,
                       float eps1, float eps2, float momentum2, float* running_mean2, float* running_var2,
                       long* num_batches_tracked2, float* gamma_eff, float* xhat_scale, float* dgamma2_coef, float* dgamma1_coef,
                       void* stream);
/* mu_bn_act_bwd with the xhat term of the batch-statistics formula scaled per channel (xhat_scale NULL = mu_bn_act_bwd) */
int mu_bn_act_bwd_scaled(const void* x, const void* res, const void* grad_out, void* dx, void* dres, long M, int C, long ld,
                         const float* mean, const float* rstd, const float* gamma, const float* beta, int act, int training,
                         float* dgamma, float* dbeta, const float* xhat_scale, void* workspace, long ws_bytes, int dtype, void* stream);

/* The pair's whole backward in one call (training mode, no residual, no activation): mu_bn_act_bwd_scaled(.., gamma_eff, beta2, ..,
 * xhat_scale) plus the three parameter gradients that follow from A, written by the same finalize kernel:
 * pair_grads[0][c] = dgamma2_coef[c] * A[c] (dgamma2), pair_grads[1][c] = dgamma1_coef[c] * A[c] (dgamma1), pair_grads[2][c] = 0 (dbeta1);
 * dbeta2 as mu_bn_act_bwd's dbeta.  pair_grads: 3 * C floats. */
int mu_bn_pair_bwd(const void* x, const void* grad_out, void* dx, long M, int C, long ld, const float* mean, const float* rstd,
                   const float* gamma_eff, const float* beta2, const float* xhat_scale, const float* dgamma2_coef,
                   const float* dgamma1_coef, float* pair_grads, float* dbeta2, void* workspace, long ws_bytes, int dtype, void* stream);

/* fp32x (round 6): mu_bn_act_bwd / mu_bn_pair_bwd on fp32 storage (contiguous rows, ld = C) with dx written as ONE power-of-two-scaled
 * fp16 operand -- dx of a BatchNorm is the dy of the 3x3 convolution in front of it and of nothing else (ConvBlock wiring,
 * ade_semantic.py:199-204).  dx_h: M rows of C halves (row stride C); dy_scale = {S, 1 / S}, two floats written on the device from
 * maxima the statistics sweep collects (|S dx| < 2^14; no host sync), handed to mu_conv_dgrad_h / mu_conv_wgrad_h.  dres (residual
 * form) stays plain fp32.  The BatchNorm's own parameter gradients are computed from the fp32 operands as always. */
int mu_bn_act_bwd_h(const void* x, const void* res, const void* grad_out, void* dx_h, void* dres, long M, int C, const float* mean,
                    const float* rstd, const float* gamma, const float* beta, int act, int training, float* dgamma, float* dbeta,
                    float* dy_scale, void* workspace, long ws_bytes, void* stream);
int mu_bn_pair_bwd_h(const void* x, const void* grad_out, void* dx_h, long M, int C, const float* mean, const float* rstd,
                     const float* gamma_eff, const float* beta2, const float* xhat_scale, const float* dgamma2_coef,
                     const float* dgamma1_coef, float* pair_grads, float* dbeta2, float* dy_scale, void* workspace, long ws_bytes,
                     void* stream);

/* ---- per-sample LayerNorm with full-shape affine: nn.LayerNorm([64,128,128]) (:281,311) ------ */
long mu_ln_sample_workspace_bytes(int B);
int mu_ln_sample_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int B, long L, float eps,
                     void* workspace, long ws_bytes, int dtype, void* stream);
int mu_ln_sample_bwd(const void* x, const void* dy, const float* w, const float* mean, const float* rstd, void* dx, float* dw,
                     float* db, int B, long L, void* workspace, long ws_bytes, int dtype, void* stream);

/* ---- pooling / resampling -------------------------------------------------------------------- */
/* nn.MaxPool2d(2) (:216); backward recomputes the arg-max (first maximum in scan order) */
int mu_maxpool2_fwd(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
int mu_maxpool2_bwd(const void* x, const void* dy, void* dx, int B, int H, int W, int C, int dtype, void* stream);
/* mu_maxpool2_bwd with gradient joins that autograd would otherwise run as separate elementwise kernels: dy2 (may be NULL, pooled
 * resolution) is a second gradient of the pooled tensor -- the residual branch of the ConvBlock behind the pool, F.gelu(x + block(x))
 * (:208, 216-217) -- and dx_add (may be NULL, input resolution) a gradient the pool's input received from its other consumer, the skip
 * connection into UpSample (:253, 303-309):  dx = scatter(dy + dy2) + dx_add. */
int mu_maxpool2_bwd_acc(const void* x, const void* dy, const void* dy2, const void* dx_add, void* dx, int B, int H, int W, int C, int dtype,
                        void* stream);
/* y = cat([skip, bilinear_x2(x, align_corners=True)], channel)  (:235,250-253); x [B,h,w,Cx], skip [B,2h,2w,Cs] */
int mu_upcat_fwd(const void* x, const void* skip, void* y, int B, int h, int w, int Cx, int Cs, int dtype, void* stream);
int mu_upcat_bwd(const void* dy, void* dx, void* dskip, int B, int h, int w, int Cx, int Cs, int dtype, void* stream);
/* mu_upcat_bwd of (dy + dy2): dy2 (may be NULL) is the residual-branch gradient of the ConvBlock behind the concat (:208, 237-238) */
int mu_upcat_bwd_acc(const void* dy, const void* dy2, void* dx, void* dskip, int B, int h, int w, int Cx, int Cs, int dtype, void* stream);
/* the same for channel counts that are not multiples of the 32-channel padding (stand-alone UpSample, :231-256): x holds
 * Cx_valid of Cx_ld stored channels, skip Cs_valid of Cs_ld; y rows are [skip valid | up valid | zeros] with Ct_ld stored channels.
 * The backward writes exact zeros into the padded channels of dx / dskip. */
int mu_upcat_compact_fwd(const void* x, const void* skip, void* y, int B, int h, int w, int Cx_ld, int Cx_valid, int Cs_ld, int Cs_valid,
                         int Ct_ld, int dtype, void* stream);
int mu_upcat_compact_bwd(const void* dy, void* dx, void* dskip, int B, int h, int w, int Cx_ld, int Cx_valid, int Cs_ld, int Cs_valid,
                         int Ct_ld, int dtype, void* stream);
/* nn.Dropout (:273,304,307): y = x*keep/(1-p); keep from `mask` (uint8, may be NULL) or the (seed,index) generator.
 * The backward is the same call on the gradient with the same seed/mask. mask_out (may be NULL) receives keep. */
int mu_dropout(const void* x, void* y, long n, float p, unsigned long long seed, const unsigned char* mask, unsigned char* mask_out,
               int dtype, void* stream);
/* the same with a device-resident step counter mixed into the seed (seed_step[0], read by the kernel): a training step captured in
 * a HIP graph bakes `seed` into the launch, the counter -- bumped inside the graph -- gives every replay its own mask, and the
 * backward launch of the same replay regenerates the same one.  seed_step NULL = mu_dropout. */
int mu_dropout_step(const void* x, void* y, long n, float p, unsigned long long seed, const unsigned long long* seed_step,
                    const unsigned char* mask, unsigned char* mask_out, int dtype, void* stream);

/* out = a + b over n elements (joins the residual and projection gradients of the attention block, :187) */
int mu_add(const void* a, const void* b, void* out, long n, int dtype, void* stream);

/* ---- masked attention (flash-style) ---------------------------------------------------------- */
/* The key mask of Mask2FormerAttention (`torch.randint(0, 2, (B,H,W))` -> {0,-inf} per KEY, ade_semantic.py:177-183) as the index list
 * the kernels below iterate over: kidx[b] = the visible keys in ascending order followed by the masked keys in ascending order (a whole
 * permutation of 0..N-1 == torch.argsort(keep, descending=True, stable=True): the MU_ATTN_KIDX_PERMUTATION form), kcnt[b] = visible
 * keys.  keep: [B][N], 1-byte or 8-byte integers (keep_elem_bytes), non-zero = visible; keep8 (may be NULL) receives the {0,1} bytes. */
int mu_compact_keys(const void* keep, int keep_elem_bytes, int B, int N, int* kidx, int* kcnt, unsigned char* keep8, void* stream);
/* Mask2FormerAttention core (ade_semantic.py:174-188) on projected qkv [B,N,3C]:
 *   out = LayerNorm_C( softmax_keys( q k^T / sqrt(C) + keymask ) v + x ), token-major [B,N,C].
 * The {0,-inf} key mask is given as the compacted list of kept keys: kidx[b][0..kcnt[b]) (int32, row
 * stride nkmax).  Also returns what the backward needs: oattn (PV/l, pre-residual), lse2 (log2-domain
 * log-sum-exp of the scaled scores), LayerNorm mean/rstd per token.  C in {32,64,128,256}.
 * An image without a visible key (kcnt[b] == 0) yields NaN in out / oattn and, in the backward, NaN dY and NaN dQ / dK / dV rows --
 * what the reference's softmax over -inf only produces (:183-185).  fp16: the sweep first runs without per-tile running-max tracking and verifies the row sums; a score
 * that exceeds the first 64 kept keys' maximum by more than 16 (log2 units) makes the block repeat its sweep with exact tracking. */
int mu_attn_fwd(const void* qkv, const void* x, const int* kidx, const int* kcnt, const float* gamma, const float* beta, void* out,
                void* oattn, float* lse2, float* ln_mean, float* ln_rstd, int B, int N, int C, int nkmax, float eps, int dtype,
                void* stream);
/* backward: grad_out [B,N,C] (wrt `out`) -> dY (grad wrt the pre-LayerNorm sum, i.e. the residual branch),
 * dqkv [B,N,3C] (masked keys get exact zeros), dgamma, dbeta.  delta [B,N] is scratch output. */
long mu_attn_bwd_workspace_bytes(int B, int N, int C);
int mu_attn_bwd(const void* qkv, const void* x, const void* oattn, const void* grad_out, const int* kidx, const int* kcnt,
                const float* lse2, const float* ln_mean, const float* ln_rstd, const float* gamma, void* dY, float* delta, void* dqkv,
                float* dgamma, float* dbeta, int B, int N, int C, int nkmax, void* workspace, long ws_bytes, int dtype, void* stream);
/* Channel counts that are not one of the kernels' widths (32, 64, 128, 256) -- `Mask2FormerAttention(channels, size)` accepts any
 * (ade_semantic.py:153-161) -- run zero-padded: C = the padded width every tensor here is stored with (qkv, x, out, ... and gamma /
 * beta, all zero in the pad channels: zero-padded projection weights give zero Q/K/V pads), c_valid = the true channel count.  The
 * scores are scaled by 1/sqrt(c_valid), the LayerNorm spans the first c_valid channels, pad channels of every output are zero.
 * c_valid == C is exactly mu_attn_fwd / mu_attn_bwd_phases. */
int mu_attn_fwd_padded(const void* qkv, const void* x, const int* kidx, const int* kcnt, const float* gamma, const float* beta, void* out,
                       void* oattn, float* lse2, float* ln_mean, float* ln_rstd, int B, int N, int C, int c_valid, int nkmax, float eps,
                       int dtype, void* stream);
int mu_attn_bwd_phases_padded(const void* qkv, const void* x, const void* oattn, const void* grad_out, const int* kidx, const int* kcnt,
                              const float* lse2, const float* ln_mean, const float* ln_rstd, const float* gamma, void* dY, float* delta,
                              void* dqkv, float* dgamma, float* dbeta, int B, int N, int C, int c_valid, int nkmax, void* workspace,
                              long ws_bytes, int dtype, int phases, void* stream);
/* ---- generic path for channel counts above the sweeps' widths (C > 256) ------------------------------------------
 * `Mask2FormerAttention(channels, size)` accepts any `channels` (ade_semantic.py:153-161); the model uses 64 / 128 / 256.  For wider
 * blocks the products of one image run as GEMMs on mu_conv_fwd / mu_conv_wgrad (taps = 1) over the KEPT key rows -- S = Q Kg^T,
 * O = P Vg, dP = dO Vg^T, dQ = dS Kg, dKg = dS^T Q, dVg = P^T dO, an N x Nk score tile per image does exist on this path -- and the
 * row kernels below sit in between (maskunet_amd/ops.py `_WideMaskAttention`).  dtype MU_F16 or MU_F32 (MU_F32X is taken as MU_F32:
 * plain, un-encoded fp32 tensors); cnt = a device pointer to ONE int (kcnt + b), idx = kidx + b * nkmax.
 *   mu_gather_rows : dst[j][0..C) = j < *cnt ? src[idx[j]][0..C) : 0 for j < rows_out (src rows src_ld elements apart, dst dense)
 *   mu_scatter_rows: dst[idx[j]][0..C) = (T) src[j][0..C) for j < *cnt (fp32 src dense, dst rows dst_ld apart; other rows untouched);
 *                    *cnt == 0: all dst_rows rows NaN (the reference's dK / dV of an image without a visible key)
 *   mu_softmax_rows: S[r][j] <- softmax_j(scale * S[r][j]) over j < *cnt, 0 for *cnt <= j < ld (:174-184; no visible key: NaN rows,
 *                    as the reference's softmax of an all -inf row)
 *   mu_attn_wide_ds: dP[r][j] <- P[r][j] * (dP[r][j] - sum_k P[r][k] dP[r][k]) * scale for j < *cnt, 0 behind (softmax backward)
 *   mu_ln_rows_fwd : out = LayerNorm over the first c_valid of C stored channels of (o + x), * gamma + beta (:187-188); mean / rstd
 *                    [rows] saved for the backward; pad channels 0
 *   mu_ln_rows_bwd : dY = d(o + x) of the above; g_xhat = grad_out * xhat (its column sums are dgamma, those of grad_out dbeta:
 *                    mu_colsum) */
int mu_gather_rows(const void* src, long src_ld, const int* idx, const int* cnt, void* dst, int rows_out, int C, int dtype, void* stream);
int mu_scatter_rows(const float* src, const int* idx, const int* cnt, void* dst, long dst_ld, int dst_rows, int C, int dtype, void* stream);
int mu_softmax_rows(void* S, int N, int ld, const int* cnt, float scale, int dtype, void* stream);
int mu_attn_wide_ds(const void* P, void* dP, int N, int ld, const int* cnt, float scale, int dtype, void* stream);
int mu_ln_rows_fwd(const void* o, const void* x, const float* gamma, const float* beta, void* out, float* mean, float* rstd, long rows,
                   int C, int c_valid, float eps, int dtype, void* stream);
int mu_ln_rows_bwd(const void* grad_out, const void* o, const void* x, const float* mean, const float* rstd, const float* gamma, void* dY,
                   void* g_xhat, long rows, int C, int c_valid, int dtype, void* stream);
/* the same, one phase group at a time (bit mask): 1 = zero dqkv + LayerNorm backward / delta / dgamma,dbeta,
 * 2 = dQ sweep, 4 = dK/dV sweep.  Phases 2 and 4 need phase 1's dY, delta and workspace contents.
 * 8 (MU_ATTN_KIDX_PERMUTATION, OR-ed into every call of one backward) = a promise about kidx: nkmax == N and every row is a whole
 * permutation of 0..N-1 with the masked keys listed after the kept ones (what a stable descending argsort of the keep mask
 * gives).  The dK/dV sweep then writes the masked keys' zero rows itself and phase 1 skips the memset of the whole dqkv buffer. */
#define MU_ATTN_KIDX_PERMUTATION 8
/* 16 (MU_ATTN_DQKV_ENCODED, OR-ed into every call of one backward; MU_F32X only, ignored otherwise): dqkv is written CHUNK-ENCODED
 * ([4 bf16 hi | 4 bf16 lo] per 16 bytes, the mu_split_encode form) -- the operand form mu_conv_fwd / mu_conv1x1_fwd_add / mu_conv_wgrad
 * take for the q/k/v projection's data- and weight-gradient, and mu_colsum(MU_F32X) for its bias gradient. */
#define MU_ATTN_DQKV_ENCODED 16
int mu_attn_bwd_phases(const void* qkv, const void* x, const void* oattn, const void* grad_out, const int* kidx, const int* kcnt,
                       const float* lse2, const float* ln_mean, const float* ln_rstd, const float* gamma, void* dY, float* delta,
                       void* dqkv, float* dgamma, float* dbeta, int B, int N, int C, int nkmax, void* workspace, long ws_bytes,
                       int dtype, int phases, void* stream);

/* ---- "next" rows: the steps either side of the path ------------------------------ */
/* f1: nn.CrossEntropyLoss (ade_semantic.py:377,399; ignore_index: city_semantic.py:341) on NHWC logits [M, Cp] (C valid
 * channels): mean loss over counted pixels -> loss[0], count[0]; lse[M] is saved for the backward. */
long mu_ce_workspace_bytes(void);
int mu_ce_fwd(const void* logits, const long* labels, long M, int Cp, int C, long ignore_index, float* lse, float* loss, float* count,
              void* workspace, long ws_bytes, int dtype, void* stream);
/* dlogits = (softmax - onehot) * grad_out[0] * grad_scale / count[0]; zeros for ignored pixels and padded channels */
int mu_ce_bwd(const void* logits, const long* labels, const float* lse, const float* count, const float* grad_out, float grad_scale,
              long M, int Cp, int C, long ignore_index, void* dlogits, int dtype, void* stream);
/* f1 on the module's own output: the same loss on NCHW logits [B, C, HW] (fp32 or fp16) exactly as the reference calls it,
 * criterion(outputs, labels) (ade_semantic.py:399; city_semantic.py:341,362); labels [B, HW]; lse [B*HW]; workspace as mu_ce_fwd */
int mu_ce_nchw_fwd(const void* logits, const long* labels, int B, int C, long HW, long ignore_index, float* lse, float* loss,
                   float* count, void* workspace, long ws_bytes, int dtype, void* stream);
int mu_ce_nchw_bwd(const void* logits, const long* labels, const float* lse, const float* count, const float* grad_out,
                   float grad_scale, int B, int C, long HW, long ignore_index, void* dlogits, int dtype, void* stream);
/* f3: mean_iou (ade_semantic.py:128-146) without host syncs.  Element (pixel r, class c) is read at
 * logits[(r / inner) * outer_stride + c * c_stride + (r % inner) * p_stride]; counts is scratch [3*C] uint32; out[0] = mean IoU. */
int mu_mean_iou(const void* logits, const long* labels, long M, int C, long inner, long outer_stride, long c_stride, long p_stride,
                float smooth, unsigned int* counts, float* out, int dtype, void* stream);
/* f1b: InstanceContrastiveLoss (ade_panoptic.py:390-418, city_instance.py:279-307) on the device, no host round trips.
 * feat: fp32 NCHW [B,C,H,W]; mask: int64 [B,H,W] instance ids in [0, id_cap) (others ignored); ignore_label < 0: none;
 * u[k] in [0,1): the k-th instance (ids ascending) that reaches the reference's torch.randint draw takes negative pixel
 * floor(u[k] * n_neg); at most max_inst instances.  loss[0] = mean triplet margin loss (0 without instances).  The backward
 * re-uses the workspace of the forward and writes d loss / d feat * grad_out[0] (dfeat is zero-filled first). */
long mu_inst_triplet_workspace_bytes(int id_cap, int max_inst);
int mu_inst_triplet_fwd(const float* feat, const long* mask, int B, int C, int H, int W, int ignore_label, float margin,
                        const float* u, int id_cap, int max_inst, void* workspace, long ws_bytes, float* loss, void* stream);
int mu_inst_triplet_bwd(const float* feat, int B, int C, int H, int W, const void* workspace, int id_cap, int max_inst,
                        const float* grad_out, float* dfeat, void* stream);
/* f4: uint8 HWC image bytes [npix, C] -> [0,1] floats in the NHWC compute layout [npix, Cp] (ToTensor, ade_semantic.py:85) */
int mu_u8_to_nhwc(const unsigned char* src, void* dst, long npix, int C, int Cp, int dtype, void* stream);
/* f4, resize half: the sample preparation of the reference datasets on the device.  src: decoded image bytes [B][Hs][Ws][C] (C <= 4, as
 * cv2.imread leaves them) of any size -> cv2.resize(.., (Wd, Hd), interpolation=cv2.INTER_LINEAR) (ade_semantic.py:72; OpenCV 4.10's
 * 8-bit fixed-point algorithm incl. its 2x2 INTER_AREA shortcut, restated: cv2 is not vendored by the reference), channels 0 and 2
 * swapped when swap_rb != 0 (cv2.COLOR_BGR2RGB, :65), then ToTensor (:85): dst [B][Hd][Wd][Cp] = byte / 255 in the NHWC compute layout,
 * zero channel padding.  u8_out (may be NULL) receives the resized bytes [B][Hd][Wd][C] themselves. */
int mu_resize_u8_nhwc(const unsigned char* src, int B, int Hs, int Ws, int C, int swap_rb, void* dst, unsigned char* u8_out, int Hd, int Wd,
                      int Cp, int dtype, void* stream);
/* the label map: uint8 [B][Hs][Ws] -> int64 [B][Hd][Wd] = torch.from_numpy(cv2.resize(mask, (Wd, Hd), interpolation=cv2.INTER_NEAREST)).long()
 * (:73,78): source index min(floor(d * scale), n - 1) */
int mu_resize_nearest_u8(const unsigned char* src, int B, int Hs, int Ws, long* dst, int Hd, int Wd, void* stream);
/* f2: optim.AdamW step (ade_semantic.py:379,401) for every parameter in one launch.  table: device array of `ntensors` entries
 * {float* p; const float* g; float* m; float* v; long n; long step;} (48 bytes; g may be NULL = no gradient this step; step = that
 * tensor's own count of attempted updates including this one, torch keeps one counter per parameter); block_tensor/block_chunk:
 * per-block (tensor index, chunk index) with chunks of mu_adamw_chunk() elements.  Gradients are multiplied by grad_scale_inv, or by
 * 1 / *grad_scale when grad_scale (a DEVICE float, torch.cuda.amp.GradScaler's scale) is given.
 * Overflow protocol, all on the device (no host sync): found_inf (device float, may be NULL) != 0 makes the launch update NOTHING;
 * with check_finite == 1 the entry point first sets *found_inf itself (0, then 1 if any gradient element is inf / NaN); check_finite == 2
 * only ORs this table's gradients into *found_inf and updates nothing (several parameter groups: check all, then update all);
 * skipped (device int[ntensors], may be NULL) counts the skipped steps per tensor and is subtracted from `step` in the bias
 * corrections, so a skipped step leaves parameters, moments AND the effective step count untouched. */
int mu_adamw_chunk(void);
int mu_adamw_multi(const void* table, const int* block_tensor, const int* block_chunk, int nblocks, int ntensors, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float grad_scale_inv, const float* grad_scale, float* found_inf,
                   int check_finite, int* skipped, void* stream);

/* ---- measurement aid (bench.py; not on the model's path) ------------------------------------- */
/* Clock / matrix-rate probe: one launch of `nblk` 256-thread blocks, every wave issuing iters * 16 register-only
 * v_mfma_f32_16x16x32_f16 on random operands, stamped once around the loop.  stamps [nblk][2] uint64 = {d s_memtime (shader
 * cycles), d s_memrealtime (100 MHz ticks)} -> clock the chip holds under a dense fp16 MFMA load = 100 MHz * [0]/[1]
 * (MI355X_MICROARCH.md "DVFS give-back" item 6).  sink: nblk * 256 floats of scratch.  The reference has no counterpart: it is what
 * makes two bench lines from two boxes of a pool comparable. */
int mu_clock_probe(void* stamps, void* sink, int nblk, int iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MASKUNET_HIP_H */
