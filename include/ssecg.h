/*
 * ssecg.h - C ABI of the MI355X (gfx950) hot path for SemiSegECG training.
 *
 * The reference (bakqui/semi-seg-ecg) has no FFI layer: its arithmetic is
 * whatever ATen/cuDNN runs under torch.nn.  These entry
 * points are what a binding for that path would call; each one cites the
 * reference call site whose arithmetic it replaces (paths relative to the
 * reference root).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm is
 *     used only as the allocator); the library never allocates, never syncs the
 *     host, never throws, keeps no mutable global state -> thread-safe across
 *     distinct streams; one process per GPU.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - tensors are contiguous fp32 in PyTorch layout (N, C, L), L fastest;
 *     labels / pseudo-labels are int64 (N, L).
 *   - return value: 0 = ok, <0 = invalid argument (SSECG_E_*), >0 = hipError_t.
 */
#ifndef SSECG_H
#define SSECG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSECG_ABI_VERSION 11

#define SSECG_E_INVAL   (-1)  /* bad shape / null pointer / unsupported parameter */
#define SSECG_E_WORKSPACE (-2) /* caller-provided workspace too small */

int ssecg_abi_version(void);
/* gfx arch string the device code was built for ("gfx950"). */
const char *ssecg_build_arch(void);

/* ------------------------------------------------------------------------
 * Conv1d as implicit GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces nn.Conv1d forward / backward at src/models/backbones/resnet.py:31-49,
 * 245-256, 287-293 and src/models/decode_heads/fcn_head.py:39-47,81.
 * --------------------------------------------------------------------- */

/* number of rows the per-channel statistics workspace of ssecg_conv1d_fwd must have for this
 * shape (each row = Cout x {sum, sumsq} floats; rows no workgroup uses are written as zeros). */
int ssecg_conv1d_stats_parts(int N, int Cin, int Cout, int Lout, int ksize);

/* y[n,co,l] = sum_{ci,t} w[co,ci,t] * x[n,ci,l*stride + t*dil - pad]
 * epilogue (each optional, applied in this order):
 *   stats_partial : [stats_parts][Cout][2] sum / sum-of-squares of the RAW conv output
 *                   (train-mode BatchNorm statistics, fused into the producer);
 *                   stats_parts >= ssecg_conv1d_stats_parts(...), else SSECG_E_WORKSPACE
 *   scale, shift  : per-Cout  y = y*scale + shift   (eval-mode BN folded; or
 *                   scale==NULL, shift=bias for the classifier conv)
 *   residual      : same shape as y, added
 *   relu          : max(y,0)
 * prologue (optional, both or neither; body/head shapes only, Cin <= 512): the input is taken as
 *   relu(x*in_scale[ci] + in_shift[ci]) with zero padding applied AFTER it - the producer's train-mode
 *   BatchNorm + ReLU (scale/shift from ssecg_bn_stats_finalize) fused into the gather, so that
 *   activation is never written to HBM.                                        */
int ssecg_conv1d_fwd(const float *x, const float *w, float *y,
                     int N, int Cin, int Lin, int Cout, int Lout,
                     int ksize, int stride, int pad, int dil,
                     const float *scale, const float *shift, const float *residual, int relu,
                     float *stats_partial, int stats_parts,
                     const float *in_scale, const float *in_shift,
                     float *split_ws, size_t split_ws_bytes, void *stream);
/* K split of small launches (ABI 9; the reference's shipped batch_size is 16, configs/base/resnet18/fixmatch.yaml:86).  A launch with
 * fewer tiles than workgroup slots leaves most of the chip idle while each workgroup contracts the whole Cin*ksize axis.  With a
 * caller-owned split_ws of at least ssecg_conv1d_fwd_split_workspace(...) bytes (0 = this shape is not split; NULL = never split)
 * up to 8 workgroup columns contract disjoint channel ranges into partial planes, and a finishing pass adds them in a fixed
 * order, applies the epilogue above and emits the BatchNorm sums.  Same contract, different summation order (fp32 rounding). */
size_t ssecg_conv1d_fwd_split_workspace(int N, int Cin, int Lin, int Cout, int Lout, int ksize);

/* operand layout for the dgrad GEMM: wt[ci][co][t] = w[co][ci][t]; for a 3-tap stride-2 conv the two
 * output-parity phases are packed separately ([ci][co] of tap 1, then [ci][co][2] of taps 0 and 2).
 * wt has Cout*Cin*ksize floats either way; pass the same stride to ssecg_conv1d_dgrad.            */
int ssecg_conv1d_transpose_weight(const float *w, float *wt, int Cout, int Cin, int ksize, int stride, void *stream);

/* dx[n,ci,m] = sum_{co,t} w[co,ci,t] * dy[n,co,(m + pad - t*dil)/stride]   (terms with a
 * non-integer or out-of-range index vanish);  wt from ssecg_conv1d_transpose_weight.
 * epilogue: accumulate (optional, same shape as dx) is added - the residual
 * branch's gradient joins here.                                              */
int ssecg_conv1d_dgrad(const float *dy, const float *wt, float *dx,
                       int N, int Cin, int Lin, int Cout, int Lout,
                       int ksize, int stride, int pad, int dil,
                       const float *accumulate, float *split_ws, size_t split_ws_bytes, void *stream);
/* split_ws: as for ssecg_conv1d_fwd (K split over Cout; the stride-2 phase launches share one workspace) */
size_t ssecg_conv1d_dgrad_split_workspace(int N, int Cin, int Lin, int Cout, int Lout, int ksize, int stride);

/* bytes of workspace ssecg_conv1d_wgrad needs for these shapes */
size_t ssecg_conv1d_wgrad_workspace(int N, int Cin, int Lin, int Cout, int Lout, int ksize);

/* dw[co,ci,t] = sum_{n,l} dy[n,co,l] * x[n,ci,l*stride + t*dil - pad]
 * split over the (n,l) axis into slabs in `workspace`, summed in a fixed order
 * (bitwise reproducible; no float atomics).  x_scale/x_shift (optional): x is taken as
 * relu(x*x_scale[ci] + x_shift[ci]), as in ssecg_conv1d_fwd's prologue.       */
int ssecg_conv1d_wgrad(const float *dy, const float *x, float *dw,
                       int N, int Cin, int Lin, int Cout, int Lout,
                       int ksize, int stride, int pad, int dil,
                       void *workspace, size_t workspace_bytes,
                       const float *x_scale, const float *x_shift, void *stream);

/* ------------------------------------------------------------------------
 * BatchNorm1d (train mode = batch statistics over N*L per channel; eps inside
 * the sqrt; running_var gets the unbiased variance).  Replaces
 * nn.BatchNorm1d / SyncBatchNorm at src/models/backbones/resnet.py:41,50,58,62
 * and the SyncBN conversion at src/algorithms/fixmatch.py:290-291.
 * --------------------------------------------------------------------- */

/* sums[c][0..1] (double) = sum over parts of partial[part][c][0..1]; fixed order.
 * dgamma/dbeta (optional, both or neither): dbeta[c] = sums[c][0], dgamma[c] = sums[c][1] - the BatchNorm
 * parameter gradients when `partial` comes from ssecg_bn_bwd_reduce (rank-local sums).        */
int ssecg_bn_reduce_partials(const float *partial, int parts, int C, double *sums,
                             float *dgamma, float *dbeta, void *stream);

/* single-GPU shortcut: ssecg_bn_reduce_partials + ssecg_bn_finalize in one launch */
int ssecg_bn_stats_finalize(const float *partial, int parts, int C, double count, float eps, float momentum,
                            float *mean, float *invstd, float *running_mean, float *running_var,
                            const float *gamma, const float *beta, float *aff_scale, float *aff_shift,
                            void *stream);

/* from (global) sums + count: mean, invstd = 1/sqrt(var_biased + eps); if
 * running_mean != NULL: running = (1-momentum)*running + momentum*{mean, var_unbiased}.
 * With SyncBN the caller all-reduces `sums` (and count) over ranks in between.
 * aff_scale/aff_shift (optional, need gamma/beta): invstd*gamma and beta - mean*invstd*gamma, the
 * per-channel affine a consumer conv applies in its gather (ssecg_conv1d_fwd prologue).         */
int ssecg_bn_finalize(const double *sums, int C, double count, float eps, float momentum,
                      float *mean, float *invstd, float *running_mean, float *running_var,
                      const float *gamma, const float *beta, float *aff_scale, float *aff_shift,
                      void *stream);

/* eval mode: scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale */
int ssecg_bn_fold(const float *gamma, const float *beta, const float *running_mean,
                  const float *running_var, int C, float eps, float *scale, float *shift,
                  void *stream);

/* the same for every BatchNorm of a model in ONE launch (the eval-mode teacher / evaluate() pass folds 21 layers):
 * table = nlayers rows of 8 words { gamma*, beta*, running_mean*, running_var*, scale*, shift*, C, eps as float bits }. */
int ssecg_bn_fold_multi(const int64_t *table, int nlayers, int max_channels, void *stream);

/* 1 when the packed ReLU mask below is available for this shape (N*C*L a multiple of 8, L >= 4), else 0. */
int ssecg_bn_mask_supported(int N, int C, int L);
/* y = [relu]( (x-mean)*invstd*gamma + beta [+ residual] ).
 * mask_bits (optional, ABI 4; needs relu and ssecg_bn_mask_supported): ceil(N*C*L / 8) bytes, bit (e & 7) of byte (e >> 3)
 * = (y[e] > 0) over the flat element index e - the ReLU mask of a unit with a residual (src/models/backbones/resnet.py:68-70:
 * out += identity; out = relu(out)) for the two backward passes, which then read 1/32 of what the saved activation costs. */
int ssecg_bn_apply_fwd(const float *x, float *y, int N, int C, int L,
                       const float *mean, const float *invstd, const float *gamma, const float *beta,
                       const float *residual, int relu, unsigned char *mask_bits, void *stream);
/* The same with the residual given as the RAW output of the block's 1x1 downsample convolution (src/models/backbones/resnet.py:
 * 64-66: identity = self.downsample(x) = BatchNorm(conv1x1(x))): res_{mean, invstd, gamma, beta} = that branch's BatchNorm, applied
 * while the residual is read - y = [relu]( bn(x) + bn_res(residual) ), the normalised identity tensor is never written (ABI 11,
 * round 6; the same fp32 operations as the two-pass form, bit for bit).  res_* all NULL: ssecg_bn_apply_fwd. */
int ssecg_bn_apply_fwd_resbn(const float *x, float *y, int N, int C, int L,
                             const float *mean, const float *invstd, const float *gamma, const float *beta,
                             const float *residual, const float *res_mean, const float *res_invstd, const float *res_gamma,
                             const float *res_beta, int relu, unsigned char *mask_bits, void *stream);

int ssecg_bn_bwd_parts(int N, int C, int L);
/* pass 1: dz = dy masked by the ReLU; partial[part][c] = { sum dz, sum dz*xhat },  xhat = (x-mean)*invstd.
 * ReLU mask: y != NULL -> (y > 0) from the saved activation (needed when a residual was added before the ReLU);
 * relu_recompute != 0 (and y == NULL) -> ((x-mean)*invstd*gamma + beta > 0) recomputed from the BN input, one
 * tensor less to read; mask_bits != NULL (y == NULL, no recompute) -> the packed mask ssecg_bn_apply_fwd wrote;
 * none of them -> no ReLU.                                                                                    */
int ssecg_bn_bwd_reduce(const float *dy, const float *y, const float *x,
                        const float *mean, const float *invstd, const float *gamma, const float *beta,
                        int relu_recompute, int N, int C, int L, float *partial, const unsigned char *mask_bits,
                        void *stream);
/* pass 2: dx = gamma*invstd*(dz - sums[c][0]/count - xhat*sums[c][1]/count);
 * dz_out (optional) receives dz (gradient of the residual branch).           */
int ssecg_bn_bwd_apply(const float *dy, const float *y, const float *x,
                       const float *mean, const float *invstd, const float *gamma, const float *beta,
                       int relu_recompute, const double *sums, double count, int N, int C, int L,
                       float *dx, float *dz_out, const unsigned char *mask_bits, void *stream);
/* Two BatchNorms behind ONE masked gradient (ABI 11, round 6): a downsample block ends in out = relu(bn2(conv2(..)) +
 * bn_d(conv1x1(x))) (src/models/backbones/resnet.py:64-70), so bn2 and bn_d both receive dz = dout * [out > 0].  _reduce_pair /
 * _apply_pair do for the pair what ssecg_bn_bwd_reduce / _apply do for one: dy and the block's ReLU mask (y = saved output, or the
 * packed mask_bits of ssecg_bn_apply_fwd - exactly one of them) are read once; partial / partial2 and dx / dx2 are what the two
 * single launches write, bit for bit (same kernels, same per-thread order).  _supported: shapes the 16-byte paths take. */
int ssecg_bn_bwd_pair_supported(int N, int C, int L);
int ssecg_bn_bwd_reduce_pair(const float *dy, const float *y, const unsigned char *mask_bits, const float *x, const float *mean,
                             const float *invstd, const float *x2, const float *mean2, const float *invstd2, int N, int C, int L,
                             float *partial, float *partial2, void *stream);
int ssecg_bn_bwd_apply_pair(const float *dy, const float *y, const unsigned char *mask_bits, const float *x, const float *mean,
                            const float *invstd, const float *gamma, const double *sums, const float *x2, const float *mean2,
                            const float *invstd2, const float *gamma2, const double *sums2, double count, int N, int C, int L,
                            float *dx, float *dx2, void *stream);
/* dgamma[c] = sums[c][1], dbeta[c] = sums[c][0]  (rank-local sums) */
int ssecg_bn_param_grads(const double *sums, int C, float *dgamma, float *dbeta, void *stream);

/* per-channel sum over (n,l): bias gradient of the classifier conv (src/models/decode_heads/fcn_head.py:96, nn.Conv1d bias).
 * scratch: caller-owned, >= C * min(N, 32) floats (NOT required to be zeroed): up to 32 sample slabs per channel are summed in
 * parallel and added in slab order by a second launch; NULL: one workgroup per channel (slower, same contract).  ABI 9.  */
int ssecg_channel_sum(const float *x, int N, int C, int L, float *out, float *scratch, size_t scratch_bytes, void *stream);

/* ------------------------------------------------------------------------
 * MaxPool1d(k,stride,pad) with -inf padding, first-maximum-wins gradient routing
 * (src/models/backbones/resnet.py:257); F.interpolate(mode="linear")
 * (src/models/encoder_decoder.py:102-107); nn.Dropout (fcn_head.py:84-87,94-95).
 * --------------------------------------------------------------------- */
int ssecg_maxpool1d_fwd(const float *x, float *y, int rows, int Lin, int Lout,
                        int ksize, int stride, int pad, void *stream);
int ssecg_maxpool1d_bwd(const float *x, const float *dy, float *dx, int rows, int Lin, int Lout,
                        int ksize, int stride, int pad, void *stream);

/* The stem convolution Conv1d(C -> 64, k = 7, stride 2, pad 3), 1 <= C <= 16, as dedicated kernels (weights and the input
 * rows of a 256-position tile resident in LDS; src/models/backbones/resnet.py:245-257, 354-355).  w is (64, C, 7).
 *   ssecg_stem_fwd           c (N, 64, Lout = (L-1)/2+1) and, when stats_partial != NULL, [ssecg_stem_parts(N, L)][64][2]
 *                            per-workgroup {sum, sum of squares} rows for ssecg_bn_stats_finalize (train mode)
 *   ssecg_stem_fwd_eval_pool eval mode: maxpool_3,2,1( relu( conv(x) * scale + shift ) ) -> pooled (N, 64, (Lout-1)/2+1) in
 *                            one launch; the conv output is never written
 *   ssecg_stem_wgrad         dw (64, C, 7) from dc = gradient of the conv output; workspace >= ssecg_stem_wgrad_workspace
 *                            bytes, caller-owned; slabs summed in a fixed order (bitwise reproducible)                     */
int ssecg_stem_supported(int N, int C, int L);
int ssecg_stem_parts(int N, int L);
int ssecg_stem_fwd(const float *x, const float *w, float *c, int N, int C, int L, float *stats_partial,
                   int stats_parts, void *stream);
int ssecg_stem_fwd_eval_pool(const float *x, const float *w, const float *scale, const float *shift, float *pooled,
                             int N, int C, int L, void *stream);
size_t ssecg_stem_wgrad_workspace(int N, int C, int L);
int ssecg_stem_wgrad(const float *dc, const float *x, float *dw, int N, int C, int L, void *workspace,
                     size_t workspace_bytes, void *stream);
/* Two-source forms (ABI 8): samples [0, n1) are read from x, samples [n1, N) from x2 - the student batch of the semi-supervised
 * plugins is torch.cat((labelled, strongly augmented unlabelled)) in the reference (src/algorithms/fixmatch.py:98-100,
 * mean_teacher.py:99-101, cps.py:116-118, stpp.py:160-162), read once by this convolution and its weight gradient: reading the
 * two tensors where they lie saves the concatenated copy.  x2 == NULL: the one-source forms above.
 * lp (ABI 9; 0 in the one-source forms): the reference's default use_amp: true runs this convolution under autocast
 * (src/algorithms/fixmatch.py:97) - 16-bit operands, 16-bit stored output.  lp != 0: x and w are rounded to bf16 while staged and
 * the output is rounded to bf16 before the BatchNorm sums and the store (fp32 containers; bf16 x bf16 products are exact in the
 * fp32 MFMA, accumulation fp32); the weight gradient rounds x and dc the same way.
 * lp == 2 (ABI 10): the same values with c - and dc, as ssecg_bn_relu_maxpool_bwd_apply(lp = 2) writes it - STORED as bf16, planar
 * (N, 64, Lout) 16-bit values (Lout % 8 == 0, 16-byte aligned bases): half the bytes of every pass over the two largest tensors of
 * the stem; results identical to lp == 1 bit for bit.  ssecg_stem_c16_supported (ABI 11): 1 iff BOTH entry points take lp == 2 for
 * this shape (L % 4 == 0 and Lout % 8 == 0 - the weight gradient reads bf16 dc through 16-byte loads only), so that a forward
 * never stores a bf16 c whose backward would be refused (window lengths L = 16m - 1 pass the forward's Lout % 8 test alone). */
int ssecg_stem_c16_supported(int N, int C, int L);
int ssecg_stem_fwd2(const float *x, const float *x2, int n1, const float *w, void *c, int N, int C, int L, float *stats_partial,
                    int stats_parts, int lp, void *stream);
int ssecg_stem_wgrad2(const void *dc, const float *x, const float *x2, int n1, float *dw, int N, int C, int L, void *workspace,
                      size_t workspace_bytes, int lp, void *stream);

/* Stem fusion: y = maxpool_k,s,pad( relu( bn(x) ) ) without materialising the activation.
 * train mode: mean/invstd/gamma/beta; eval mode: mean == invstd == NULL and gamma/beta = folded scale/shift.
 * Backward recomputes the activation to route the pooled gradient (first maximum wins) and apply the ReLU mask:
 * _bwd_reduce -> partial[ssecg_bn_bwd_parts(N,C,Lin)][C][2] = {sum dz, sum dz*xhat}; _bwd_apply -> dx (BN input grad).
 * lp != 0 (use_amp, 16-bit stem): the gradient is routed on the bf16-ROUNDED activation, as the reference under autocast,
 * which pools BatchNorm's bf16 output, routes it (ties between rounded neighbours go to the first).  lp == 2 (the stem's own
 * shape only: k 3, stride 2, pad 1, Lin % 4 == 0): x - bf16-valued under use_amp - is STORED as bf16 (planar, N*C*Lin 16-bit
 * values) and dx is written as bf16 (what the stem's weight gradient rounds it to anyway): same results bit for bit, half the bytes.
 * (src/models/backbones/resnet.py:254-257, 354-355)                                                              */
int ssecg_bn_relu_maxpool_fwd(const float *x, float *y, int N, int C, int Lin, int Lout,
                              int ksize, int stride, int pad, const float *mean, const float *invstd,
                              const float *gamma, const float *beta, void *stream);
int ssecg_bn_relu_maxpool_bwd_reduce(const float *dy, const void *x, const float *mean, const float *invstd,
                                     const float *gamma, const float *beta, int N, int C, int Lin, int Lout,
                                     int ksize, int stride, int pad, float *partial, int lp, void *stream);
int ssecg_bn_relu_maxpool_bwd_apply(const float *dy, const void *x, const float *mean, const float *invstd,
                                    const float *gamma, const float *beta, const double *sums, double count,
                                    int N, int C, int Lin, int Lout, int ksize, int stride, int pad,
                                    void *dx, int lp, void *stream);

int ssecg_interp_linear_fwd(const float *x, float *y, int rows, int Lin, int Lout,
                            int align_corners, void *stream);
int ssecg_interp_linear_bwd(const float *dy, float *dx, int rows, int Lin, int Lout,
                            int align_corners, void *stream);

/* keep-mask drawn from a counter-based generator keyed by (seed, element index):
 * mask[i] = u(seed,i) >= p;  y = x*mask/(1-p) */
/* seed_dev (optional, device): read instead of `seed` - a launch captured in a HIP graph takes this step's seed from
 * device memory (ssecg/graph.py), so that a replay draws what the eager step would have drawn. */
int ssecg_dropout_fwd(const float *x, float *y, uint8_t *mask, size_t n, float p,
                      uint64_t seed, const uint64_t *seed_dev, void *stream);
/* y = x * mask * scale  (dropout with a given mask; dropout backward) */
int ssecg_mask_scale(const float *x, const uint8_t *mask, float *y, size_t n, float scale, void *stream);

/* ------------------------------------------------------------------------
 * Pseudo-labels and losses.
 * --------------------------------------------------------------------- */

/* conf = softmax(logits,1).max(1), mask = argmax(logits,1) (first max wins)
 * (src/algorithms/fixmatch.py:90-91); prob (optional) = softmax(logits,1)
 * (src/algorithms/mean_teacher.py:92).                                       */
int ssecg_softmax_conf_argmax(const float *logits, int N, int num_classes, int L,
                              float *conf, int64_t *mask, float *prob, void *stream);

int ssecg_ce_parts(int N, int L);
/* Hard-label cross-entropy over (N, num_classes, L) logits, summed over positions:
 *   loss_i = (logsumexp(x_i) - x_i[target_i]) * w_i,   w_i = 1 if conf==NULL else (conf_i >= thresh)
 *   dlogits = (softmax(x_i) - onehot(target_i)) * w_i * grad_scale
 * partial[part] = { sum loss_i, sum w_i }.  F.cross_entropy at
 * src/algorithms/fixmatch.py:105,114-116 and src/models/encoder_decoder.py:110-111. */
int ssecg_ce_hard_fwd_bwd(const float *logits, const int64_t *target, const float *conf, float thresh,
                          int N, int num_classes, int L, float grad_scale,
                          float *dlogits, float *partial, void *stream);
/* Soft-label cross-entropy: loss_i = -sum_c p_ic * log_softmax(x_i)_c,
 * dlogits = (softmax(x_i)*sum_c p_ic - p_i) * grad_scale   (src/algorithms/mean_teacher.py:115) */
int ssecg_ce_soft_fwd_bwd(const float *logits, const float *prob, int N, int num_classes, int L,
                          float grad_scale, float *dlogits, float *partial, void *stream);
/* Per-record confusion counts for the segmentation metrics: counts[n][t][p] = #{l: target[n,l]==t and pred[n,l]==p},
 * int32 (N, K, K); positions whose label or prediction is outside [0,K) are skipped.  Everything the reference's
 * validation metric needs (torchmetrics 1.5.2 segmentation.MeanIoU fed one-hot argmax predictions,
 * src/algorithms/base.py:206-216, src/utils/perf_metrics.py:9-47: per-record intersection = diag, union = row + col - diag)
 * and ST++'s checkpoint-agreement score (src/algorithms/stpp.py:32-42,63-80) without moving (B,K,L) one-hots to the host. */
int ssecg_seg_confusion(const int64_t *pred, const int64_t *target, int N, int num_classes, int L,
                        int32_t *counts, void *stream);
/* out[k] = scale * sum_part partial[part][k], k<width (double accumulation, fixed order) */
int ssecg_sum_partials(const float *partial, int parts, int width, float scale, float *out, void *stream);
/* Tail of the two-term losses (src/algorithms/fixmatch.py:116-118 loss = (loss_x + loss_u_s) / 2; mean_teacher.py:117) in one launch:
 * out5 = { loss, loss, loss_x, loss_u, weight } from the {sum loss, sum weight} partial rows of the two cross-entropy launches
 * (px: nx rows, pu: nu rows), loss_x = sx_scale * sum px[:,0], loss_u = su_scale * sum pu[:,0], weight = su_scale * sum pu[:,1];
 * the same fp64 sums as ssecg_sum_partials and the same fp32 combination as the torch expression it replaces (ABI 8). */
int ssecg_loss_pair_finish(const float *px, int nx, const float *pu, int nu, float sx_scale, float su_scale, float *out5, void *stream);

/* ------------------------------------------------------------------------
 * Multi-tensor AdamW and EMA (src/utils/optimizer.py:27-37,
 * src/algorithms/mean_teacher.py:138-149).
 * `table` is a device array of int64 words, `words` per tensor:
 *   AdamW: { param*, grad*, exp_avg*, exp_avg_sq*, numel }            (5 words)
 *   EMA  : { teacher*, student*, numel, student_is_int64 }           (4 words)
 * Hyper-parameters are doubles: the derived scalars (1 - lr*wd, lr/bias_correction1, ...)
 * are formed in double as torch.optim.AdamW's Python side does, then rounded once to fp32.
 * --------------------------------------------------------------------- */
/* skip_flag (device float, may be NULL): a non-zero value turns the launch into a no-op - GradScaler.step()'s
 * "skip the update when the gradients hold an inf/NaN" (src/utils/misc.py:252-253) without a host round trip.
 * step = the caller's 1-based count of LAUNCHES (torch's per-parameter `step` after this call if nothing was ever skipped);
 * skipped_count (device float, may be NULL; owned by one optimizer): incremented by a skipped launch, and subtracted from
 * `step` inside the kernel for the bias corrections 1 - beta^t - so the updates after a skipped step are the ones
 * GradScaler + torch.optim.AdamW produce (a skipped step is no optimizer step there). */
/* coef_dev (optional, device double[5] = ssecg_adamw_coefficients(lr, beta1, beta2, weight_decay, step)): read instead of
 * the by-value lr / step - a launch captured in a HIP graph takes this step's scalars from device memory. */
int ssecg_adamw_coefficients(double lr, double beta1, double beta2, double weight_decay, int step, double *out5);
int ssecg_adamw_multi(const int64_t *table, int ntensors, int64_t max_numel,
                      double lr, double beta1, double beta2, double eps, double weight_decay,
                      int step, const float *skip_flag, float *skipped_count, const double *coef_dev, void *stream);
/* torch.optim.SGD (src/utils/optimizer.py:15-26; dampening 0, no nesterov): table rows { param*, grad*, momentum_buffer*
 * or 0, numel } (4 words); first_step != 0: the buffer is initialised with the (decayed) gradient. */
int ssecg_sgd_multi(const int64_t *table, int ntensors, int64_t max_numel, double lr, double momentum,
                    double weight_decay, int first_step, const float *skip_flag, const double *lr_dev, void *stream);
/* get_grad_norm_ (src/utils/misc.py:265-278, norm_type 2) over the gradients of a pointer table (rows of `words` int64,
 * gradient pointer in column grad_col, element count in column numel_col - the AdamW / SGD tables qualify):
 * out = { ||g||_2, found_inf } (found_inf = 1 if any gradient element is inf/NaN).  With scaler_state != NULL the same
 * launch applies torch.cuda.amp.GradScaler.update() (src/utils/misc.py:254) to the device-resident
 * { scale, growth_tracker, skipped_steps } floats: found_inf -> scale *= backoff_factor, tracker = 0, skipped += 1;
 * otherwise tracker += 1, and when it reaches growth_interval: scale *= growth_factor, tracker = 0.
 * workspace >= ssecg_grad_norm_workspace bytes, caller-owned. */
size_t ssecg_grad_norm_workspace(int ntensors, int64_t max_numel);
int ssecg_grad_norm_multi(const int64_t *table, int ntensors, int words, int grad_col, int numel_col, int64_t max_numel,
                          float *workspace, size_t workspace_bytes, float *out, float *scaler_state,
                          double growth_factor, double backoff_factor, int growth_interval, void *stream);
/* torch.nn.utils.clip_grad_norm_ (src/utils/misc.py:246-248): every gradient *= min(1, max_norm / (norm + 1e-6)) with
 * the norm read from the device (out[0] of ssecg_grad_norm_multi). */
int ssecg_grad_clip_multi(const int64_t *table, int ntensors, int words, int grad_col, int numel_col, int64_t max_numel,
                          const float *norm, double max_norm, void *stream);
int ssecg_ema_multi(const int64_t *table, int ntensors, int64_t max_numel, double decay, void *stream);
/* Gradient staging of the data-parallel step: what torch DDP's reducer does per parameter before the bucket's all-reduce
 * (src/algorithms/fixmatch.py:292-295 wraps the model in DistributedDataParallel; its reducer copies grad / world into the
 * bucket, one launch per parameter) as ONE launch per bucket.  table (device): ntensors rows {src float*, element offset
 * into dst, numel}; dst[offset + i] = src[i] * scale; a null src zero-fills its slot; src may alias dst + offset. */
int ssecg_pack_scaled_multi(const int64_t *table, int ntensors, int64_t max_numel, float *dst, double scale, void *stream);

/* ------------------------------------------------------------------------
 * Winograd F(2,3) form of the 3-tap, stride-1, pad-1, dilation-1 convolutions (14 of the network's 17 k=3 convs,
 * src/models/backbones/resnet.py:31-49 via BasicBlock :55-72, src/models/decode_heads/fcn_head.py:39-47): two outputs
 * from four inputs with 4 instead of 6 multiplications per channel pair, on the same fp32 matrix pipe.  Forward and
 * data gradient are the SAME operation on different operands:
 *   forward      : src = x  (N, Cin, L),  u = wino_weight(w, transposed=0)  -> out (N, Cout, L)
 *   data gradient: src = dy (N, Cout, L), u = wino_weight(w, transposed=1)  -> out = dx (N, Cin, L)
 * with the direct kernels' epilogue: optional per-channel scale/shift (folded BN), residual add (also used to
 * accumulate into an existing gradient: residual == out is allowed), ReLU, and per-channel {sum, sum of squares}
 * partial rows for the train-mode BatchNorm (rows = ssecg_conv1d_wino_parts).  C % 8 == 0 and M % 64 == 0 required
 * (ssecg_conv1d_wino_supported); results agree with ssecg_conv1d_fwd/_dgrad to fp32 rounding (different summation).
 * --------------------------------------------------------------------- */
int ssecg_conv1d_wino_supported(int N, int C, int L, int M);
int ssecg_conv1d_wino_parts(int N, int L, int M);
/* u (4*Cout*Cin floats, 16-byte aligned) from w (Cout, Cin, 3) */
int ssecg_conv1d_wino_weight(const float *w, float *u, int Cout, int Cin, int transposed, void *stream);
/* The same transform for MANY convolutions in one launch (once per optimiser step): device table of ntensors rows
 * {w, u_forward or 0, u_transposed or 0, Cout, Cin} (int64 each); max_elems = max Cout*Cin. */
int ssecg_conv1d_wino_weight_multi(const int64_t *table, int ntensors, int max_elems, void *stream);
/* in_scale/in_shift (both or neither, C <= 512): the input is taken as relu(src*in_scale[c] + in_shift[c]) - the producer's
 * train-mode BatchNorm + ReLU applied while the input is staged, so that activation is never written (padding stays 0). */
int ssecg_conv1d_wino(const float *src, const float *u, float *out, int N, int C, int L, int M,
                      const float *scale, const float *shift, const float *residual, int relu,
                      float *stats_partial, int stats_parts, const float *in_scale, const float *in_shift, void *stream);
/* Weight gradient of the same convolutions in Winograd form (transpose of F(2,3): 4 instead of 6 multiplications per
 * (co, ci, output pair)); dw (Cout, Cin, 3) as ssecg_conv1d_wgrad writes it.  Cin % 128 == 0 and Cout % 128 == 0
 * (ssecg_conv1d_wino_wgrad_supported); workspace >= ssecg_conv1d_wino_wgrad_workspace bytes, caller-owned; slabs are
 * summed in a fixed order (bitwise reproducible, no atomics).  x is the conv's input, dy the gradient of its output. */
/* The same convolutions in Winograd F(4,3) form (four outputs from six inputs: HALF the multiplications of the direct
 * form, 3/4 of F(2,3)'s).  Operand u (since ABI 3): the RAW taps re-laid tap-major, 3*Cout*Cin floats in the order
 * [c/8][tap][(c%8)/4][m][c%4] (m = output channel of the launch, c = its input channel; taps flipped for the data
 * gradient), made by ssecg_conv1d_wino4_weight_multi (table rows {w, u_forward or 0, u_transposed or 0, Cout, Cin}); the
 * kernel forms the six transform planes while staging.  Same epilogue contract as ssecg_conv1d_wino.  C % 16 == 0 and M % 128 == 0 required (ssecg_conv1d_wino4_supported).  fp32 error vs an fp64
 * convolution: relative L2 < 1e-6 (about 2.5x the F(2,3) form). */
int ssecg_conv1d_wino4_supported(int N, int C, int L, int M);
int ssecg_conv1d_wino4_parts(int N, int L, int M);
int ssecg_conv1d_wino4_weight_multi(const int64_t *table, int ntensors, int max_elems, void *stream);
/* split_ws (optional): workspace of ssecg_conv1d_wino4_split(N, C, L, M) * N*M*L floats.  With it, a launch that has fewer tiles
 * than CUs (small batches) contracts its channels in up to 8 K splits side by side and finishes with one pass that sums them,
 * applies scale / shift / residual / ReLU and (ABI 9) emits the BatchNorm sums; the fused input BN is split too; split = 1: not used. */
int ssecg_conv1d_wino4_split(int N, int C, int L, int M);
int ssecg_conv1d_wino4(const float *src, const float *u, float *out, int N, int C, int L, int M,
                       const float *scale, const float *shift, const float *residual, int relu,
                       float *stats_partial, int stats_parts, const float *in_scale, const float *in_shift,
                       float *split_ws, size_t split_ws_bytes, void *stream);
int ssecg_conv1d_wino_wgrad_supported(int N, int Cin, int L, int Cout);
size_t ssecg_conv1d_wino_wgrad_workspace(int N, int Cin, int L, int Cout);
int ssecg_conv1d_wino_wgrad(const float *dy, const float *x, float *dw, int N, int Cin, int L, int Cout,
                            void *workspace, size_t workspace_bytes, const float *x_scale, const float *x_shift,
                            void *stream);   /* x_scale/x_shift: x is taken as relu(x*x_scale[ci] + x_shift[ci]) */
/* The same weight gradient as the transpose of F(4,3) (ABI 10): three taps from four neighbouring output gradients and six
 * inputs, 6 multiplications per (co, ci, output quad) - 3/4 of the F(2,3) form's, half the direct form's.  Cin % 64 == 0 and
 * Cout % 64 == 0 (ssecg_conv1d_wino_wgrad4_supported), same contract otherwise; its own workspace size.  fp32 error vs an fp64
 * gradient: 2e-6 of the tensor's scale (about twice the F(2,3) form's).  Slabs summed in a fixed order: bitwise reproducible. */
int ssecg_conv1d_wino_wgrad4_supported(int N, int Cin, int L, int Cout);
size_t ssecg_conv1d_wino_wgrad4_workspace(int N, int Cin, int L, int Cout);
int ssecg_conv1d_wino_wgrad4(const float *dy, const float *x, float *dw, int N, int Cin, int L, int Cout,
                             void *workspace, size_t workspace_bytes, const float *x_scale, const float *x_shift,
                             void *stream);

/* ------------------------------------------------------------------------
 * On-device record pipeline of the unlabelled loader: strong augmentation + standardisation.
 * --------------------------------------------------------------------- */

/* op ids of the four RandAugment members of configs/base/resnet18/fixmatch.yaml:62-77 */
#define SSECG_AUG_AMPLITUDE_SCALING 0 /* x * N(1, sigma)                         src/utils/transforms.py:340-351 */
#define SSECG_AUG_POWERLINE         1 /* x + (p95-p5)/2 * sin(2 pi f t), t = l/fs  src/utils/transforms.py:480-502 */
#define SSECG_AUG_PARTIAL_WHITE     2 /* x[:, s:s+n] += amplitude * randn[:, :n]   src/utils/transforms.py:518-550,560 */
#define SSECG_AUG_PARTIAL_SINE      3 /* x[:, s:s+n] += amplitude * sin(2 pi t/freq)[:, :n], t = l/L  :504-509,552 */
/* Per-record plan = the random DECISIONS RandAugment (src/utils/transforms.py:647-657), RandomApply (:574-583) and the
 * ops make, int32[SSECG_AUG_PLAN_WIDTH]:
 *   [0..3] op id of layer k (np.random.choice(ops, num_layers, replace=False))   [4] bit k set = layer k fires (rand() < prob)
 *   [5] powerline frequency in Hz (50 | 60)   [6],[7] white-noise count, start   [8],[9] sine-noise count, start
 *   [10] number of layers (<= 4)              [11] reserved                                                         */
#define SSECG_AUG_PLAN_WIDTH 12

/* y[b] = RandAugment(x[b]) for B records of (C, L) fp32, L <= 4096; ops applied in plan order on the running signal, in
 * fp64 like the reference.  scales (B,C,L): AmplitudeScaling factors, white (B,C,L): standard-normal draws of the white
 * noise (its first `count` samples per lead are used); either may be NULL -> drawn in the kernel from `seed`
 * (splitmix64 -> Box-Muller over the element index; stream 1 = scales as 1 + sigma*z, stream 2 = white).
 * sigma / amplitude / sine_freq are the values RandAugment's level leaves in the ops (transforms.py:350,452-455).
 * The output is NOT yet standardised (ssecg_standardize follows, as `transform` does in semi_dataset.py:241-244). */
int ssecg_strong_augment(const float *x, float *y, const int32_t *plan, const float *scales, const float *white,
                         int B, int C, int L, double sigma, double fs, double amplitude, double sine_freq,
                         uint64_t seed, void *stream);
/* y[b] = (x[b] - mean) / std over the n = C*L elements of record b (population std, fp64 two-pass), all zeros when
 * std == 0: Standardize(axis=(-1,-2)) of src/utils/transforms.py:290-310.  In place (y == x) is allowed. */
int ssecg_standardize(const float *x, float *y, int B, int n, void *stream);

/* ------------------------------------------------------------------------
 * Reduced-precision conv path: what `use_amp: true` selects.  Reference: the student forward runs
 * under torch.cuda.amp.autocast (src/algorithms/fixmatch.py:97, mean_teacher.py:98, base.py:122, cps.py:115,
 * stpp.py:159) - nn.Conv1d / BatchNorm1d / ReLU of src/models/backbones/resnet.py:55-72 in 16-bit with fp32
 * accumulation and fp32 batch statistics, the loss in fp32; the teacher / pseudo-label passes of the training steps are outside autocast and stay fp32; evaluate() / test() run
 * INSIDE it (src/algorithms/base.py:202): the same kernels on the running statistics (ssecg_amp_bn_apply_fwd with mean == NULL, ABI 11).
 * Here: bf16 storage, v_mfma_f32_32x32x16_bf16, fp32 master weights.
 *
 * bf16 activations live in HBM in the BLOCKED layout (N, C/8, L, 8): 8 channels of one position = one 16-byte vector
 * (C % 8 == 0).  `void *` tensors below are blocked bf16; weights stay fp32 (Cout, Cin, K) and are turned into bf16
 * MFMA operands once per optimiser step by ssecg_amp_weight_operand_multi.
 * --------------------------------------------------------------------- */
int ssecg_amp_planar_to_blocked(const float *x, void *y, int N, int C, int L, void *stream);   /* (N,C,L) f32 -> blocked bf16 */
int ssecg_amp_blocked_to_planar(const void *x, float *y, int N, int C, int L, void *stream);   /* and back (exact)          */

/* The stem's BatchNorm + ReLU + MaxPool1d(3, 2, 1) (src/models/backbones/resnet.py:245-257, 354-355) writing the pooled activation
 * in the blocked bf16 layout of the reduced-precision student pass (N, C/8, Lin/2, 8): what ssecg_bn_relu_maxpool_fwd +
 * ssecg_amp_planar_to_blocked compute, bit for bit, without the fp32 pooled tensor (ABI 7).  C % 8 == 0, Lin % 4 == 0, 16-byte
 * aligned bases; mean == invstd == NULL: gamma / beta are a folded scale / shift. */
int ssecg_amp_stem_pool_supported(int N, int C, int Lin);
int ssecg_amp_stem_pool_fwd(const void *x, void *yb, int N, int C, int Lin, const float *mean, const float *invstd, const float *gamma,
                            const float *beta, int x16, void *stream);   /* x16 != 0: x is stored as bf16 (planar; ssecg_stem_fwd2 with lp == 2) */
/* table rows (8 x int64): { w*, operand*, Cout, Cin, K, transposed, ntaps, tap0 | tap1 << 8 | tap2 << 16 }:
 *   operand[(cc*ntaps + tt)][h][m][j] = w[m][16cc+8h+j][tap[tt]]  (transposed = 0: forward, m = co)
 *                                     = w[16cc+8h+j][m][tap[tt]]  (transposed = 1: data gradient, m = ci)
 * i.e. (Ck/16)*ntaps*2*M vectors of 8 bf16; max_vectors = the largest such count among the rows. */
int ssecg_amp_weight_operand_multi(const int64_t *table, int ntensors, int max_vectors, void *stream);
/* out[n][m][l*ostride + ooff] = sum_{c,tt} operand[m][c][tt] * src[n][c][l*gmul + tapoff[tt]]  (+ accumulate), rounded to
 * bf16; out rows have Lrow positions.  Forward of nn.Conv1d(k, stride s, pad p): ntaps = k, gmul = s, tapoff[t] = t - p,
 * Lrow = Ldst, ostride 1, ooff 0; data gradients use the transposed operand with tapoff[t] = p - t (stride 1) or the two
 * output-parity phases (stride 2).  stats (optional, forward): per-channel { sum, sum of squares } of the ROUNDED output,
 * [ssecg_amp_conv_parts][M][2] partial rows for ssecg_bn_stats_finalize (stats_parts must equal that count: every row is
 * written).  Csrc % 16 == 0, M % 64 == 0.  Kernels: the weights-stationary kernel (csrc/amp_ws.hip: 3 taps, stride 1, 64 / 128 /
 * 256 source channels, no accumulate), the LDS-DMA ring kernel and the register-staged kernel of csrc/amp.hip otherwise. */
int ssecg_amp_conv_parts(int N, int Csrc, int Lsrc, int M, int Ldst, int ntaps, int gmul, int tapoff0, int tapoff1, int tapoff2,
                         int Lrow, int ostride, int ooff);   /* ABI 5: the row count depends on the kernel that takes the shape */
int ssecg_amp_conv(const void *src, const void *w_operand, void *out, int N, int Csrc, int Lsrc, int M, int Ldst,
                   int ntaps, int gmul, int tapoff0, int tapoff1, int tapoff2, int Lrow, int ostride, int ooff,
                   const void *accumulate, float *stats, int stats_parts, void *stream);
/* y = [relu](x * gamma*invstd + (beta - mean*gamma*invstd) [+ residual]) on blocked bf16, fp32 arithmetic; without a residual
 * one rounding, with one the affine result is rounded before the add and the sum again (autocast's bf16 BatchNorm output and
 * ``out += identity``).  mean == invstd == NULL (ABI 11): eval mode - gamma / beta are the folded scale / shift of the running
 * statistics (ssecg_bn_fold[_multi]): ``evaluate()`` under use_amp runs its eval-mode forward inside autocast
 * (src/algorithms/base.py:202).
 * mask_bytes (optional, ABI 4, needs relu): N*(C/8)*L bytes, one per 16-byte vector of y: bit j = (channel 8*cb + j > 0). */
int ssecg_amp_bn_apply_fwd(const void *x, void *y, int N, int C, int L, const float *mean, const float *invstd,
                           const float *gamma, const float *beta, const void *residual, int relu, unsigned char *mask_bytes,
                           void *stream);
/* ... with the residual given as the raw (blocked bf16) output of the 1x1 downsample convolution and its BatchNorm applied - and
 * rounded to bf16, as the stored identity was - while it is read (ssecg_bn_apply_fwd_resbn; ABI 11). */
int ssecg_amp_bn_apply_fwd_resbn(const void *x, void *y, int N, int C, int L, const float *mean, const float *invstd,
                                 const float *gamma, const float *beta, const void *residual, const float *res_mean,
                                 const float *res_invstd, const float *res_gamma, const float *res_beta, int relu,
                                 unsigned char *mask_bytes, void *stream);
/* BatchNorm backward on blocked bf16.  mode 0: no ReLU; 1: ReLU mask from the saved output y; 2: mask recomputed from
 * x (needs gamma, beta); 3: `y` is the mask_bytes tensor of ssecg_amp_bn_apply_fwd (1/16 of y's bytes).  reduce: partial[ssecg_amp_bn_bwd_parts][C][2] = { sum dz, sum dz*xhat } (-> ssecg_bn_reduce_partials);
 * apply: dx = gamma*invstd*(dz - sums[c][0]/count - xhat*sums[c][1]/count) [, dz] rounded to bf16. */
int ssecg_amp_bn_bwd_parts(int N, int C, int L);
int ssecg_amp_bn_bwd_reduce(const void *dy, const void *y, const void *x, const float *mean, const float *invstd,
                            const float *gamma, const float *beta, int mode, int N, int C, int L, float *partial,
                            void *stream);
int ssecg_amp_bn_bwd_apply(const void *dy, const void *y, const void *x, const float *mean, const float *invstd,
                           const float *gamma, const float *beta, int mode, const double *sums, double count,
                           int N, int C, int L, void *dx, void *dz, void *stream);
/* The pair forms (ABI 11, round 6; ssecg_bn_bwd_reduce_pair): the BatchNorm of a downsample block's 1x1 branch (x2, *2) beside its bn2
 * behind the block's final ReLU mask (mode 1: saved output y, mode 3: the mask_bytes of ssecg_amp_bn_apply_fwd): dy and the mask are
 * read once, partial / partial2 and dx / dx2 are what the two single launches write, bit for bit; no dz is written. */
int ssecg_amp_bn_bwd_reduce_pair(const void *dy, const void *y, int mode, const void *x, const float *mean, const float *invstd,
                                 const void *x2, const float *mean2, const float *invstd2, int N, int C, int L, float *partial,
                                 float *partial2, void *stream);
int ssecg_amp_bn_bwd_apply_pair(const void *dy, const void *y, int mode, const void *x, const float *mean, const float *invstd,
                                const float *gamma, const double *sums, const void *x2, const float *mean2, const float *invstd2,
                                const float *gamma2, const double *sums2, double count, int N, int C, int L, void *dx, void *dx2,
                                void *stream);
/* dw (Cout, Cin, K) fp32 = sum_{n,l} dy[n][co][l] * x[n][ci][l*stride + t - pad] from blocked bf16 operands (fp32
 * accumulation, slabs summed in a fixed order).  K in {1 (pad 0), 3 (pad 1)}, stride 1 or 2, Cin % 64 == Cout % 64 == 0. */
int ssecg_amp_wgrad_supported(int N, int Cin, int Lx, int Cout, int Ldy, int K, int stride, int pad);
size_t ssecg_amp_wgrad_workspace(int N, int Cin, int Lx, int Cout, int Ldy, int K);
int ssecg_amp_wgrad(const void *dy, const void *x, float *dw, int N, int Cin, int Lx, int Cout, int Ldy, int K,
                    int stride, int pad, void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SSECG_H */
