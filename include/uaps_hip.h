/* uaps_hip.h -- C ABI of libuaps_hip.so, the MI355X (gfx950) kernels of the UAPS training step.
 *
 * The reference (djene-mengistu/UAPS) is pure Python on PyTorch and has no FFI of its own
 * (SURVEY.md section 8b); each entry point below replaces a chain of PyTorch ops of the reference
 * step, cited as file:line relative to the reference root.  The host side (the uaps_amd Python package) binds
 * these with ctypes; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all memory;
 *   - tensors are dense fp32 NCHW exactly as the reference's decoders produce them; labels and
 *     pseudo-labels are int64 [B,H,W] as torch.argmax / the reference's data loader produce them;
 *   - `stream` is a hipStream_t (0 = the null stream); calls only enqueue work, never synchronise, never allocate, read no
 *     environment variable and are safe under hipGraph capture.  The library keeps exactly this state, all of it set by the
 *     caller through the entry points named here and none of it stream-ordered (change it between steps):
 *       process-wide   the convolution arithmetic (uaps_conv_set_mode) and planner switches (uaps_conv_set_tuning), the pointer
 *                      to the device-resident step state (uaps_set_step_state: NULL outside a state-mode step; two trainers in
 *                      one process each bracket their steps with set / clear, see uaps_amd/graph.py) and, per device, the pointer
 *                      to that device's sticky error word (uaps_set_error_word);
 *       per thread     LEGACY: the one-shot side arguments of the NEXT call (uaps_next_call_hints), consumed and cleared by that call.
 *                      Every entry point that reads them also exists as `<name>_h(const uaps_call_hints* hints, same arguments)`
 *                      (round 6, ABI 3): the hints of THAT call as its first argument (NULL = none), nothing thread-local read or
 *                      written -- safe from any thread, e.g. PyTorch's autograd thread.  The Python package calls only the *_h forms
 *                      and uaps_conv_ex.  (uaps_next_launch_events, a measurement aid of bench.py, stays per thread.)
 *   - return value: 0 on success, a negative UAPS_E* code for bad arguments, a positive hipError_t
 *     if a launch failed.  Nothing throws.
 *   - number of heads D in [1,8] (main + auxiliary decoders), classes C in [2,8].
 */
#ifndef UAPS_HIP_H
#define UAPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* uaps_stream_t; /* hipStream_t */

#define UAPS_MAX_HEADS 8
#define UAPS_MAX_CLASSES 8

#define UAPS_OK 0
#define UAPS_EINVAL (-1)     /* null pointer, non-positive dimension                       */
#define UAPS_ERANGE (-2)     /* D, C, kernel size, dilation or alignment outside what is built */
#define UAPS_EWORKSPACE (-3) /* workspace smaller than uaps_loss_workspace_bytes() reports */
#define UAPS_ENOFORM (-4)    /* a hint that changes what the call computes (uaps_call_hints::dyt_*) has no form in the kernel this
                              * layer runs on: nothing was launched, nothing is wrong with the arguments -- do the work the hint
                              * stood for yourself (uaps_bn_act_bwd_apply) and call again without it.  (Until round 5 this case
                              * shared UAPS_ERANGE with real range errors.) */

int uaps_abi_version(void);
const char* uaps_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * Loss block of the step: UAPS_train.py:186-282 (+ utilities/pytorch_losses.py:54-89 dice_loss,
 * torch.nn.CrossEntropyLoss / KLDivLoss / LogSoftmax objects created at UAPS_train.py:73-75).
 * ------------------------------------------------------------------------------------------- */

/* Layout of the `scalars` output of the two forward calls (float, device).  Offsets in floats. */
/* unsupervised: */
#define UAPS_U_CE(D, C) 0                                   /* ce[D]    CE_k vs pseudo-label (mean)       :259 */
#define UAPS_U_DICE(D, C) (D)                               /* dice[D]  dice_loss_k vs pseudo-label       :259 */
#define UAPS_U_S(D, C) (2 * (D))                            /* s[D]     0.5 (ce+dice)                      :259 */
#define UAPS_U_E(D, C) (3 * (D))                            /* E[D]     mean_pix exp(-var_k)               :265 */
#define UAPS_U_PS(D, C) (4 * (D))                           /* ps_loss                                     :277 */
#define UAPS_U_LUN(D, C) (4 * (D) + 1)                      /* l_uncert                                    :243 */
#define UAPS_U_LOSS(D, C) (4 * (D) + 2)                     /* cw1*ps_loss + cw2*l_uncert                  :282 */
#define UAPS_U_TOTAL(D, C) (4 * (D) + 3)                    /* uaps_pairloss_fwd / _finalize_sums only (round 6): supervised_loss + the
                                                             * slot above = the step's loss (:282), so that the host needs no add launch */
#define UAPS_U_A1(D, C) (4 * (D) + 4)                       /* a1[D*C] = -(2/C)/(card+eps)   (for backward) */
#define UAPS_U_A2(D, C) (4 * (D) + 4 + (D) * (C))           /* a2[D*C] = (2/C) I/(card+eps)^2               */
#define UAPS_U_I(D, C) (4 * (D) + 4 + 2 * (D) * (C))        /* I[D*C]    sum p_kc [y=c]    pytorch_losses.py:85 */
#define UAPS_U_CARD(D, C) (4 * (D) + 4 + 3 * (D) * (C))     /* card[D*C] sum p_kc + [y=c]  pytorch_losses.py:86 */
#define UAPS_U_CNT(D, C) (4 * (D) + 4 + 4 * (D) * (C))      /* cnt[C]    pixels per pseudo-label class      */
#define UAPS_U_NSCALARS(D, C) (4 * (D) + 4 + 4 * (D) * (C) + (C))
/* supervised: */
#define UAPS_S_CE(D, C) 0                                   /* ce[D]                              :194-197 */
#define UAPS_S_DICE(D, C) (D)                               /* dice[D]                            :201-204 */
#define UAPS_S_SUP(D, C) (2 * (D))                          /* sum_k ce_coef*ce_k + dice_coef*dice_k  :218 */
#define UAPS_S_BAD(D, C) (2 * (D) + 1)                      /* number of labels outside [0,C) (must be 0)  */
#define UAPS_S_A1(D, C) (2 * (D) + 2)
#define UAPS_S_A2(D, C) (2 * (D) + 2 + (D) * (C))
#define UAPS_S_I(D, C) (2 * (D) + 2 + 2 * (D) * (C))
#define UAPS_S_CARD(D, C) (2 * (D) + 2 + 3 * (D) * (C))
#define UAPS_S_CNT(D, C) (2 * (D) + 2 + 4 * (D) * (C))
#define UAPS_S_NSCALARS(D, C) (2 * (D) + 2 + 4 * (D) * (C) + (C))

/* Bytes of scratch the forward calls need (block partial sums); same query serves both. */
int uaps_loss_workspace_bytes(int D, int B, int C, int H, int W, size_t* out_host);

/* Unsupervised branch, forward.  Replaces UAPS_train.py:186-189 (softmax), 223 (mean prediction),
 * 226-236 (KL "variance" maps, exp(-var)), 241-243 (l_uncert), 251-255 (Dirichlet-mixed arg-max
 * pseudo-label; w_host are the np.random.dirichlet weights, applied as fp32 scalars left to right),
 * 259-277 (CE + Dice pseudo-supervision weighted by mean(exp(-var))) and the cw-weighted sum of 282.
 *   logits_host : host array of D device pointers, each fp32 [B,C,H,W] contiguous
 *   pseudo      : out int64 [B,H,W]
 *   var         : out fp32 [D,B,H,W] (var_k = sum_c KL(mean || p_k)); may be NULL to skip the store
 *   scalars     : out fp32 [UAPS_U_NSCALARS(D,C)]
 */
int uaps_unsup_fwd(const float* const* logits_host, const double* w_host, int D, int B, int C, int H, int W,
                   float cw1, float cw2, float eps, int64_t* pseudo, float* var, float* scalars,
                   void* workspace, size_t workspace_bytes, uaps_stream_t stream);

/* Unsupervised branch, backward: d(gscale * (cw1*ps_loss + cw2*l_uncert)) / d logits_k, the closed
 * form of what autograd derives from UAPS_train.py:223-282.
 *   gscale  : device pointer to the upstream gradient (one float), or NULL for 1
 *   dlogits_host : host array of D device pointers, each fp32 [B,C,H,W], overwritten
 */
int uaps_unsup_bwd(const float* const* logits_host, const int64_t* pseudo, const float* scalars, float cw1,
                   float cw2, const float* gscale, int D, int B, int C, int H, int W,
                   float* const* dlogits_host, uaps_stream_t stream);

/* Supervised branch: UAPS_train.py:194-218.  loss = sum_k (ce_coef * CrossEntropy_k + dice_coef * dice_loss_k);
 * the reference's supervised_loss (mean over heads of 0.5 (CE + Dice)) is ce_coef = dice_coef = 0.5 / D.
 * With D = 1: (1, 0) is nn.CrossEntropyLoss()(logits, labels) (UAPS_train.py:75) and (0, 1) is
 * dice_loss(labels.unsqueeze(1), logits, eps) (utilities/pytorch_losses.py:54-89). */
int uaps_sup_fwd(const float* const* logits_host, const int64_t* labels, int D, int B, int C, int H, int W,
                 float ce_coef, float dice_coef, float eps, float* scalars, void* workspace,
                 size_t workspace_bytes, uaps_stream_t stream);
int uaps_sup_bwd(const float* const* logits_host, const int64_t* labels, const float* scalars, float ce_coef,
                 float dice_coef, const float* gscale, int D, int B, int C, int H, int W,
                 float* const* dlogits_host, uaps_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Step state for captured steps (hipGraph replay freezes kernel arguments).  A 64-byte device buffer the host refreshes
 * before every step: words 0-1 a uint64 added to every Philox key (new draws per step), words 2-9 the mixing weights
 * w[8] as float (UAPS_train.py:251), word 10 / 11 cw1 / cw2 (:279-280), word 12 / 13 Adam's lr / (1 - beta1^t) and
 * 1 / sqrt(1 - beta2^t).  With a state set, uaps_pairloss_fwd accepts w_host = NULL, uaps_pairloss_* read cw1 / cw2 from it
 * when passed NaN, uaps_adam_step reads its two scalars from it when step < 1, the perturbation / dropout kernels add the
 * key increment, and uaps_fanout_perturbed draws a FeatureDropout threshold factor on the device for every u[g] < 0.
 * Process-wide (not per stream); NULL (the default) restores the by-value behaviour.
 * ------------------------------------------------------------------------------------------- */
int uaps_set_step_state(const void* device_ptr);
const void* uaps_get_step_state(void);

/* ---------------------------------------------------------------------------------------------
 * General strided convolutions and the stem max-pool of the reference's ResNet (utilities/resnet.py:120 conv 7x7 / 2 pad 3,
 * :124 max-pool 3x3 / 2 pad 1, :8-14 + :147 layer2's 3x3 / 2 and 1x1 / 2): odd kernel sizes <= 7, stride 1 or 2, bias-free,
 * fp32 NCHW, on the exact-f32 matrix instruction (csrc/conv_strided.hip).  They replace aten::convolution(_backward) and
 * aten::max_pool2d_with_indices(_backward) for those layers.  Packed weights: wf [ks*ks][Cin4][Cout16], wb [ks*ks][Cout4][Cin16]
 * (sizes in floats from uaps_convs_pack_floats).  Output size: OH = (H + 2 pad - ks) / stride + 1.
 * ------------------------------------------------------------------------------------------- */
int uaps_convs_pack_floats(int Cout, int Cin, int ks, size_t* fwd_floats_host, size_t* bwd_floats_host);
int uaps_convs_pack_weights(const float* w, int Cout, int Cin, int ks, float* wf, float* wb, uaps_stream_t stream);
int uaps_convs_out_size(int H, int W, int ks, int stride, int pad, int* OH_host, int* OW_host);
int uaps_convs_fwd(const float* x, const float* wf, float* y, int B, int Cin, int Cout, int H, int W, int ks, int stride, int pad,
                   uaps_stream_t stream);
int uaps_convs_bwd_data(const float* dy, const float* wb, float* dx, int B, int Cin, int Cout, int H, int W, int ks, int stride,
                        int pad, uaps_stream_t stream);
int uaps_convs_wrw_workspace_bytes(int B, int Cin, int Cout, int H, int W, int ks, int stride, int pad, size_t* bytes_host);
int uaps_convs_bwd_weight(const float* dy, const float* x, float* dw, int B, int Cin, int Cout, int H, int W, int ks, int stride,
                          int pad, void* workspace, size_t workspace_bytes, uaps_stream_t stream);
/* planes = B*C; y [planes, OH, OW] with OH = (H - 1) / 2 + 1; idx: uint8 [planes, OH, OW], the window position of each arg-max
 * (first maximum wins, NaN propagates: torch.nn.MaxPool2d(3, 2, 1)) */
int uaps_maxpool3x3s2_fwd(const float* x, float* y, void* idx, long planes, int H, int W, uaps_stream_t stream);
int uaps_maxpool3x3s2_bwd(const float* dy, const void* idx, float* dx, long planes, int H, int W, uaps_stream_t stream);
/* y [planes, OH, OW] = x[:, ::2, ::2] (OH = (H - 1) / 2 + 1) and its adjoint (dx [planes, H, W]: dy at the even positions, zero
 * elsewhere): the sampling of a 1x1 / stride 2 convolution (utilities/resnet.py:13-14, 157-161), which then runs as a
 * stride-1 uaps_conv_fwd on the sampled tensor */
/* inverse == 0: xs [B, 4, C, H/2, W/2] = the four stride-2 sampling phases of x [B, C, H, W] (xs[b][2 py + px][c][i][j] =
 * x[b][c][2 i + py][2 j + px]); inverse != 0: the first pointer is xs, the second receives x.  H % 2 == 0, W % 8 == 0, 16-byte
 * aligned pointers (UAPS_ERANGE otherwise).  A 3x3 / stride 2 / padding 1 convolution (utilities/resnet.py:8-10 with stride 2)
 * is a stride-1 uaps_conv_fwd over xs with re-arranged weights */
int uaps_space_to_depth2(const float* x, float* xs, int B, int C, int H, int W, int inverse, uaps_stream_t stream);
int uaps_subsample2_fwd(const float* x, float* y, long planes, int H, int W, uaps_stream_t stream);
int uaps_subsample2_bwd(const float* dy, float* dx, long planes, int H, int W, uaps_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The whole loss block of one training step -- UAPS_train.py:194-218 on the labelled logits and :186-189, 223-282 on the
 * unlabelled logits -- as ONE forward launch (+ a one-block finalize) and ONE backward launch: the first blocks of the
 * grid run the supervised branch, the rest the unsupervised branch; results are bit-identical to uaps_sup_* +
 * uaps_unsup_* (same per-branch block partitioning and fixed-order reductions).  lab_logits / un_logits: D host pointers
 * each, [B,C,H,W] fp32; ce_coef = dice_coef = 0.5 / D (UAPS_train.py:208-218).  `cfg` = 0, or a cap on the blocks per branch.
 *
 * Gathered-batch semantics.  The reference's nn.DataParallel (UAPS_model.py:13) gathers the logits of all GPUs and computes
 * every CE mean, Dice sum and uncertainty mean over that global batch (UAPS_train.py:194-277).  One process per GPU
 * reproduces it with a 1-step exchange: pass `sums_out` (device, uaps_pairloss_num_sums doubles) to uaps_pairloss_fwd -- it
 * then writes this rank's raw sums and no scalars --, sum the buffers over the ranks (one tiny all-reduce), call
 * uaps_pairloss_finalize_sums with the global pixel count, and give that count to uaps_pairloss_bwd as `n_pixels_loss`
 * (0 = the local B*H*W).  The parameter gradients of the ranks then ADD up to the gradient of the global loss.
 * ------------------------------------------------------------------------------------------- */
int uaps_pairloss_workspace_bytes(int D, int C, size_t* bytes_host);
int uaps_pairloss_num_sums(int D, int C, int* count_host);
int uaps_pairloss_fwd(const float* const* lab_logits_host, const float* const* un_logits_host, const int64_t* labels,
                      const double* w_host, int D, int B, int C, int H, int W, float cw1, float cw2, float eps,
                      int64_t* pseudo, float* var /* [D,B,H,W] or NULL */, float* sup_scalars, float* unsup_scalars,
                      double* sums_out /* NULL: finalise locally */, void* workspace, size_t workspace_bytes, int cfg,
                      uaps_stream_t stream);
int uaps_pairloss_finalize_sums(const double* sums, int D, int C, long n_pixels, float cw1, float cw2, float eps,
                                float* sup_scalars, float* unsup_scalars, uaps_stream_t stream);
int uaps_pairloss_bwd(const float* const* lab_logits_host, const float* const* un_logits_host, const int64_t* labels,
                      const int64_t* pseudo, const float* sup_scalars, const float* unsup_scalars, float cw1, float cw2,
                      const float* gscale, int D, int B, int C, int H, int W, long n_pixels_loss,
                      float* const* dlab_host, float* const* dun_host, int cfg, uaps_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Feature perturbations of the auxiliary decoders: utilities/UAPS_unet.py:156-185, applied to all
 * five encoder scales at UAPS_unet.py:227-231.  Random numbers: Philox4x32-10 keyed by `seed`,
 * counter = element index + `offset` (the reference draws from the unseeded CPU generators, so
 * bit parity of the draws is impossible; the *_apply forms take recorded draws for parity tests).
 * ------------------------------------------------------------------------------------------- */

/* FeatureNoise (UAPS_unet.py:172-185): y = x*n + x, n ~ U(-range, range) of shape [C,H,W] shared by
 * the batch.  Backward = the same call on dy with the same seed.  noise_out (fp32 [C,H,W]) may be NULL. */
int uaps_feat_noise(const float* x, float* y, int B, int C, int H, int W, uint64_t seed, uint64_t offset,
                    float range, float* noise_out, uaps_stream_t stream);
/* Same with a given noise tensor (fp32 [C*H*W]); chw = C*H*W. */
int uaps_feat_noise_apply(const float* x, const float* noise, float* y, int B, long chw, uaps_stream_t stream);

/* Dropout(x, p) = F.dropout(x, p, training=True) (UAPS_unet.py:156-158): y = x * keep / (1-p).
 * Backward = the same call on dy with the same seed.  keep_out (uint8 [n]) may be NULL. */
int uaps_feat_bernoulli(const float* x, float* y, long n, uint64_t seed, uint64_t offset, float p,
                        uint8_t* keep_out, uaps_stream_t stream);
int uaps_feat_mask_apply(const float* x, const uint8_t* keep, float scale, float* y, long n, uaps_stream_t stream);

/* FeatureDropout (UAPS_unet.py:161-169): y = x * (mean_c x < u * max_hw mean_c x), u = the
 * np.random.uniform(0.7, 0.9) draw.  keep [B,H,W] uint8 is written for the backward (dx = dy * keep).
 * workspace: uaps_feat_dropout_workspace_bytes(). */
int uaps_feat_dropout_workspace_bytes(int B, int C, int H, int W, size_t* out_host);
int uaps_feat_dropout_fwd(const float* x, float* y, int B, int C, int H, int W, float u, uint8_t* keep,
                          void* workspace, size_t workspace_bytes, uaps_stream_t stream);
int uaps_feat_dropout_bwd(const float* dy, const uint8_t* keep, float* dx, int B, int C, int H, int W,
                          uaps_stream_t stream);

/* Gradient fan-in of an encoder feature map that feeds several decoders (UAPS_unet.py:226-232 use each
 * feature list once per decoder): out = in[0] + ... + in[n-1], left to right, n in [1,4], `count` floats. */
int uaps_sum_tensors(const float* const* in_host, int n, float* out, long count, uaps_stream_t stream);
/* The same fan-in with the backward of the perturbations folded in: out = sum_k P_k(g_k), where P_k re-applies the
 * perturbation in front of decoder k to its incoming gradient (all three are diagonal, so backward = forward on the
 * gradient): mode 0 identity (main decoder), 1 FeatureNoise (Philox offsets per statistics group, `range`),
 * 2 Dropout (one offset, `p`), 3 FeatureDropout (keep mask uint8 [B,H,W] from the forward), 4 the MaxPool2d(2) that
 * feeds the next encoder level (g is [B,C,H/2,W/2], keep = the arg-max positions of uaps_maxpool2x2_fwd).  Host arrays of n <= 8
 * entries; offsets is [n][groups]; groups <= 4; needs H*W % 4 == 0 and 16-byte aligned tensors. */
int uaps_fanin_perturbed(const float* const* g_host, const int* mode_host, const uint8_t* const* keep_host,
                         const uint64_t* offsets_host, int n, int groups, uint64_t seed, float range, float p, int B,
                         int C, int H, int W, float* out, uaps_stream_t stream);

/* The forward counterpart: the n perturbed copies of a feature map (UAPS_unet.py:227-231) written by one pass over f
 * instead of one kernel (and one read of f) per perturbation, with the stand-alone kernels' arithmetic and Philox
 * indexing.  mode 1 FeatureNoise, 2 Dropout, 3 FeatureDropout: run uaps_feat_dropout_stats(f, ...) first (channel-mean
 * attention map + per-image maximum into the uaps_feat_dropout_workspace_bytes(B, ...) workspace) and pass that workspace
 * and the threshold factors u [groups]; keep_host[k] receives the uint8 [B,H,W] mask of a mode-3 output.
 * Host arrays of n <= 8 entries, offsets [n][groups], groups <= 4, H*W % 4 == 0, 16-byte aligned tensors. */
int uaps_feat_dropout_stats(const float* x, int B, int C, int H, int W, void* workspace, size_t workspace_bytes,
                            uaps_stream_t stream);
int uaps_fanout_perturbed(const float* f, float* const* out_host, const int* mode_host, uint8_t* const* keep_host,
                          const uint64_t* offsets_host, const float* u_host, const void* fdrop_workspace, int n, int groups,
                          uint64_t seed, float range, float p, int B, int C, int H, int W, uaps_stream_t stream);

/* nn.MaxPool2d(2) of DownBlock (UAPS_unet.py:55-58): out [B,C,H/2,W/2] plus the arg-max position (dy*2+dx) per output
 * as uint8 (first maximum wins, NaN propagates).  Needs H even, W % 8 == 0, 16-byte aligned x/out.  Its backward is
 * mode 4 of uaps_fanin_perturbed (no zero fill, no scatter, no separate accumulation). */
int uaps_maxpool2x2_fwd(const float* x, int B, int C, int H, int W, float* out, uint8_t* idx, uaps_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * ConvBlock / UpBlock glue between the convolutions (utilities/UAPS_unet.py:36-44, 81-86).
 * ------------------------------------------------------------------------------------------- */

/* nn.BatchNorm2d (training statistics) -> nn.LeakyReLU(slope) -> nn.Dropout(drop_p) on a conv
 * output y [B,C,H,W], fused (UAPS_unet.py:38-40, 42-43).  conv_bias (may be NULL) is the bias of the
 * preceding conv when the caller ran the conv without it: train-mode BN cancels it exactly, it only
 * enters running_mean.  Updates running_mean/running_var (momentum form of nn.BatchNorm2d, unbiased
 * variance) and increments *num_batches_tracked when those pointers are non-NULL.  Writes
 * save_mean/save_invstd [C] for the backward.  Dropout keep-masks come from Philox(seed, offset)
 * and are regenerated by the backward; drop_p = 0 disables it. */
int uaps_bn_workspace_bytes(int B, int C, int H, int W, size_t* out_host);
int uaps_bn_act_fwd_train(const float* y, const float* conv_bias, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                          float eps, float slope, float drop_p, uint64_t seed, uint64_t offset, int B, int C,
                          int H, int W, float* out, float* save_mean, float* save_invstd, void* workspace,
                          size_t workspace_bytes, uaps_stream_t stream);
/* The same with `groups` statistics groups: the batch is `groups` consecutive blocks of B/groups images, each
 * normalised with its own batch statistics (save_mean / save_invstd are [groups][C]) and folded into the running
 * statistics one after the other, exactly as `groups` separate calls in order would (the reference runs the
 * labelled and the unlabelled batch as two forwards, UAPS_train.py:177,185); num_batches_tracked += groups.
 * groups in [1,8], B % groups == 0. */
int uaps_bn_act_fwd_train_grouped(const float* y, const float* conv_bias, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                  float eps, float slope, float drop_p, uint64_t seed, uint64_t offset, int B, int C,
                                  int H, int W, int groups, float* out, float* save_mean, float* save_invstd,
                                  void* workspace, size_t workspace_bytes, uaps_stream_t stream);
/* The same, with the statistics pass already done by the producer of y: `partials` is float2
 * [C][B][parts_per_image] of per-image partial (sum, sum of squares), as uaps_conv_fwd_stats writes them. */
int uaps_bn_act_fwd_train_partials(const void* partials, int parts_per_image, const float* y, const float* conv_bias,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, float momentum, float eps, float slope, float drop_p,
                                   uint64_t seed, uint64_t offset, int B, int C, int H, int W, int groups, float* out,
                                   float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes,
                                   uaps_stream_t stream);
int uaps_bn_act_bwd_grouped(const float* dout, const float* y, const float* gamma, const float* beta,
                            const float* save_mean, const float* save_invstd, float slope, float drop_p, uint64_t seed,
                            uint64_t offset, int B, int C, int H, int W, int groups, float* dy, float* dgamma,
                            float* dbeta, void* workspace, size_t workspace_bytes, uaps_stream_t stream);
/* The same, also writing the gradient of the conv bias in front of the BatchNorm (identically zero: the bias cancels
 * in train-mode normalisation) into dconv_bias [C], which saves the host a fill launch per layer. */
int uaps_bn_act_bwd_grouped_bias(const float* dout, const float* y, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_invstd, float slope, float drop_p,
                                 uint64_t seed, uint64_t offset, int B, int C, int H, int W, int groups, float* dy,
                                 float* dgamma, float* dbeta, float* dconv_bias, void* workspace, size_t workspace_bytes,
                                 uaps_stream_t stream);
/* eval(): running statistics, no dropout.  save_mean receives running_mean - conv_bias (for the backward). */
int uaps_bn_act_fwd_eval(const float* y, const float* conv_bias, const float* gamma, const float* beta,
                         const float* running_mean, const float* running_var, float eps, float slope, int B,
                         int C, int H, int W, float* out, float* save_mean, void* workspace,
                         size_t workspace_bytes, uaps_stream_t stream);
/* Backward of the train-mode op: dy [B,C,H,W], dgamma [C], dbeta [C] from dout and the saved y. */
int uaps_bn_act_bwd(const float* dout, const float* y, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_invstd, float slope, float drop_p, uint64_t seed,
                    uint64_t offset, int B, int C, int H, int W, float* dy, float* dgamma, float* dbeta,
                    void* workspace, size_t workspace_bytes, uaps_stream_t stream);
int uaps_bn_act_bwd_eval(const float* dout, const float* y, const float* gamma, const float* beta,
                         const float* mean_eff, const float* running_var, float eps, float slope, int B, int C,
                         int H, int W, float* dy, void* workspace, size_t workspace_bytes, uaps_stream_t stream);

/* UpBlock.forward lines 83-85: out[:, :Cs] = skip ; out[:, Cs:] = bilinear x2 (align_corners=True) of
 * low [B,Cl,h,w]; out is [B,Cs+Cl,2h,2w].  Backward: dskip = dout[:, :Cs] (may be NULL), dlow = the
 * transposed interpolation of dout[:, Cs:] (gather form, deterministic). */
int uaps_up_cat_fwd(const float* skip, const float* low, float* out, int B, int Cs, int Cl, int h, int w,
                    uaps_stream_t stream);
int uaps_up_cat_bwd(const float* dout, float* dskip, float* dlow, int B, int Cs, int Cl, int h, int w,
                    uaps_stream_t stream);

/* Bound i (out + i * UAPS_BOUND_FLOATS, see uaps_next_call_hints) = max_c(|gamma_i[c]| + |beta_i[c]|) for n BatchNorm layers
 * (host arrays of n device pointers / channel counts; out holds n * UAPS_BOUND_FLOATS floats), one launch: times sqrt(elements per channel and statistics group) this bounds every output of the train-mode
 * BatchNorm (+ LeakyReLU) of utilities/UAPS_unet.py:38-39 (|x_hat| <= sqrt(n - 1) for batch statistics), the operand bound
 * of the convolutions behind it (uaps_next_call_hints). */
int uaps_bn_param_bounds(const float* const* gamma_host, const float* const* beta_host, const int* C_host, int n, float* out,
                         uaps_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Convolutions of the U-Net: every nn.Conv2d of utilities/UAPS_unet.py (ConvBlock 3x3 :36-44,
 * UpBlock.conv1x1 :73, Decoder.out_conv :138-139; stride 1, padding ks/2, ks in {1,3}), fp32 NCHW,
 * computed as an implicit GEMM on the matrix cores in one of three arithmetic modes (uaps_conv_set_mode):
 *   2 (default)  as 1, and where the caller supplies magnitude bounds of the tensor operands (uaps_next_call_hints):
 *                both fp32 operands scaled by a power of two and split into two fp16 pieces (22 significant bits),
 *                three partial products per multiply on v_mfma_f32_*_f16 with fp32 accumulation -- an error below that of
 *                the fp32 matrix instruction's accumulation chain (tests/test_gpu_conv.py) at twice the rate of mode 1,
 *   1            exact three-way bf16 split of both fp32 operands, six partial products per multiply on
 *                v_mfma_f32_16x16x32_bf16 with fp32 accumulation: the error of an fp32 fma chain at 2.67x the
 *                fp32 matrix rate (csrc/conv_split.hpp); used for 16-byte-aligned rows without dilation,
 *   0            the exact-f32 matrix instruction v_mfma_f32_16x16x4_f32 for everything (also: UAPS_CONV_MODE=0,
 *                or bit 28 of `cfg` for one call).
 * They replace the three aten::convolution / convolution_backward calls PyTorch makes per layer.
 *
 * Weights are used in a packed, zero-padded layout produced by uaps_conv_pack_weights from the
 * nn.Conv2d.weight tensor [Cout][Cin][ks][ks]:
 *   wf [ks*ks][Kpad(Cin)][Npad(Cout)]   for the forward,
 *   wb [ks*ks][Kpad(Cout)][Npad(Cin)]   (transposed, taps flipped) for the input gradient,
 * sizes (in floats) from uaps_conv_pack_floats.  `cfg` = 0 selects the tiling automatically; other
 * values are tuning overrides used by tools/bench_conv.py (low byte: output channels per workgroup,
 * 16/32/64; bits 8-9: 1 = 8x32 pixel tile, 2 = 16x16; for bwd_weight the low 24 bits: number of pixel splits).
 * Bits 24-27 of `cfg` are functional: the dilation of a 3x3 kernel, 0/1 (none), 2 or 4, with padding = dilation
 * (the dilated stages of utilities/resnet.py:8-10, 201-203); pass the same value to all three directions.
 * ------------------------------------------------------------------------------------------- */
int uaps_conv_set_mode(int mode);      /* 0 / 1 / 2 as above (default 2); process-wide, not stream-ordered: set it between steps */
/* Planner switches for ablation runs and diagnosis (tools/ablation.sh, tools/diag); 0 = the shipped plan. */
#define UAPS_TUNE_NO_SPLIT_FWD 1u      /* forward / input gradient on the fp32 matrix instruction whatever the mode */
#define UAPS_TUNE_NO_SPLIT_WRW 2u      /* weight gradient likewise */
#define UAPS_TUNE_NO_SMALL 4u          /* no exact-N class kernels (csrc/conv_small.hpp) */
#define UAPS_TUNE_NO_HP16 8u           /* no persistent 16-output-channel kernels (csrc/conv_split_n16.hpp) */
#define UAPS_TUNE_WRW_ROW_MAJOR 16u    /* split weight gradient: row-major instead of column-strip tile order */
#define UAPS_TUNE_WRW_SHORT_TILES 32u  /* split weight gradient: 4-row tiles also on the 16-output-channel layers */
#define UAPS_TUNE_NO_TALL_FWD 64u     /* forward / input gradient of 32-channel blocks: 8-row tiles instead of 16-row ones */
#define UAPS_TUNE_NO_ROW16 128u       /* no full-width-row kernels (csrc/conv_split_row16.hpp): the 8 x 32-tile persistent kernels instead */
#define UAPS_TUNE_NO_ROW_WRW 256u     /* no full-width-row weight-gradient kernels (csrc/conv_split_wrw_row.hpp) */
#define UAPS_TUNE_DEEP_ROWS 512u      /* diagnostic: the full-width-row kernels with two row sets in flight (round 5: measured slower) */
#define UAPS_TUNE_G1_NARROW 1024u    /* 1x1 GEMM kernels: 128 output channels per workgroup also where 256 divide the layer's width */
#define UAPS_TUNE_NO_G 2048u         /* no whole-layer-width tile kernels (csrc/conv_split_g.hpp, round 6): the 8 x 32-tile kernels instead */
#define UAPS_TUNE_G_DEEP 4096u       /* diagnostic: those kernels with all nine taps' weight fragments in registers, loaded around the fetch */
int uaps_conv_set_tuning(unsigned flags);
unsigned uaps_conv_get_tuning(void);

/* Sticky device error word.  The fp16-split convolutions (mode 2) trust the magnitude bounds they are handed: a bound that
 * is too small by more than the 2x headroom lets a scaled operand overflow fp16, and every output it touches comes back
 * NaN.  Those kernels check what they store and OR a UAPS_ERR_* bit into *device_word (4-byte aligned device memory owned
 * and zeroed by the caller; NULL = no reporting) when a stored value is not finite -- so a violated bound, or non-finite data
 * entering a convolution, is reported instead of training on.  Read the word whenever the host synchronises anyway
 * (UAPSTrainer.epoch_metrics / validate / check_errors do).  One word per DEVICE: the call binds (or, with NULL, unbinds) the
 * word of the current device, and a launch reports to the word of the device it runs on.  The word must outlive every
 * launch made while it is bound (the Python package keeps one never-freed word per device, uaps_amd._lib.error_word). */
#define UAPS_ERR_CONV_NONFINITE 1u     /* an fp16-split forward / input-gradient convolution stored a non-finite value */
#define UAPS_ERR_WRW_NONFINITE 2u      /* an fp16-split weight-gradient convolution produced a non-finite partial sum */
int uaps_set_error_word(unsigned* device_word);

/* One-shot side arguments for the NEXT kernel entry point called on this thread; that call consumes and clears them
 * (every convolution entry point, uaps_bn_act_fwd_train_partials, uaps_bn_finalize_train, uaps_bn_act_bwd*,
 * uaps_pairloss_bwd, uaps_up_cat_fwd and uaps_add_relu do; NULL clears).
 *   bound[i], mul[i]  device bound b and host factor m > 0 with |operand i| <= value(b) * m for every element; NULL = unknown.
 *                     A bound is UAPS_BOUND_FLOATS floats (16-byte aligned): value(b) = max over the UAPS_BOUND_SLOTS
 *                     floats b[k * UAPS_BOUND_STRIDE], the rest is padding -- the producing kernels raise the slots with
 *                     atomics, and thousands of workgroups on ONE address would serialise at the memory side.
 *                     Operands: conv forward: 0 = x (for the *_bn entry points: the activation after the fused
 *                     BatchNorm + LeakyReLU), 1 = x2 of *_cat;  conv bwd_data: 0 = dy;  conv bwd_weight: 0 = dy, 1 = x, 2 = x2.
 *                     In mode 2 a 3x3 convolution whose tensor operands all carry a bound runs in the two-piece fp16 form;
 *                     without bounds it runs as in mode 1.  A bound that is too small makes the result wrong (fp16
 *                     overflow), one that is too large by up to 2^10 costs no accuracy.
 *   stats_mean, stats_bias   per-channel device arrays [Cout] (either may be NULL = 0): a convolution that writes BatchNorm
 *                     partial sums (uaps_conv_fwd_stats, the stats output of uaps_conv_fwd_bn / _cat) forms them about the
 *                     shift s_c = stats_mean[c] - stats_bias[c], i.e. sum(v - s), sum((v - s)^2) -- pass the BatchNorm's
 *                     running_mean and the bias of the convolution in front of it, and the variance E[d^2] - E[d]^2 no
 *                     longer cancels for channels with |mean| >> std (fp32 partial sums).  The call that CONSUMES those
 *                     partials (uaps_bn_act_fwd_train_partials, uaps_bn_finalize_train) must be given the same two
 *                     pointers in its own hints; it reads them before it updates running_mean.
 *   out_amax          a bound (UAPS_BOUND_FLOATS floats, zero-initialised by the caller WITH uaps_zero_bounds) whose slots the
 *                     producing kernel raises atomically so that value(out_amax) = maximum |element| of its output tensor: the
 *                     bound of a later convolution's operand.
 * Bound storage is read by the kernels with agent-scope loads and raised by memory-side atomics; whoever else writes it must
 * write past the per-XCD L2s too (uaps_zero_bounds, uaps_bn_param_bounds do; a plain fill or copy of the same kernel stream is
 * coherent only after that kernel has ended -- under a hipGraph replay not even then). */
#define UAPS_BOUND_SLOTS 16
#define UAPS_BOUND_STRIDE 64
#define UAPS_BOUND_FLOATS (UAPS_BOUND_SLOTS * UAPS_BOUND_STRIDE)
typedef struct uaps_call_hints {
    unsigned struct_size;        /* sizeof(uaps_call_hints) of the header the CALLER was built against (round 5).  A shorter struct of
                                  * an older client is accepted and its missing tail reads as zero; 0 or a size beyond this library's
                                  * struct is UAPS_EINVAL -- fields are only ever appended */
    const float* bound[3];
    float mul[3];
    float* out_amax;
    const float* stats_mean;     /* see below */
    const float* stats_bias;
    const float* residual;       /* uaps_bn_act_fwd_train_*: out = relu(bn(y) + residual) -- a residual join (utilities/resnet.py:47-50,
                                  * 88-91) in the BatchNorm's apply pass; slope is ignored, drop_p must be 0; out_amax is honoured */
    /* uaps_conv_bwd_weight_partial*: the `dy` argument is the gradient BEHIND the BatchNorm + LeakyReLU that follows the
     * convolution (d(activation)); the kernel forms dy = BatchNorm backward of it while staging -- from dyt_y (the convolution's
     * raw output), dyt_coef (uaps_bn_act_bwd_prepare) -- and writes it to dyt_out for the input-gradient call.  bound[0] must
     * then be the bound uaps_bn_act_bwd_prepare raised.  UAPS_ENOFORM (nothing launched) when the layer's kernel has no such
     * form: run uaps_bn_act_bwd_apply and call again without these. */
    const float* dyt_y;
    const float* dyt_coef;
    float* dyt_out;
    float dyt_slope;
    int dyt_groups;
    /* uaps_conv_bwd_data (round 5): the input gradient this call produces IS d(activation) of a train-mode BatchNorm + LeakyReLU
     * (the layer in front of the convolution); its kernel also forms that BatchNorm's backward sums -- what the first pass of
     * uaps_bn_act_bwd_prepare computes from (gradient, y) -- in its epilogue: bsum_y the BatchNorm's raw input [B, Cin, H, W],
     * bsum_mean / bsum_invstd [groups][Cin] and bsum_gamma / bsum_beta [Cin] its saved statistics and parameters, bsum_partials
     * float2 [Cin][B][H / 16] (sum d, sum d x_hat per 16-row run of the kernel),
     * bsum_max two zeroed bounds (2 * UAPS_BOUND_FLOATS floats: max|d|, max|x_hat|).  uaps_bn_act_bwd_finalize then replaces
     * uaps_bn_act_bwd_prepare.  UAPS_ENOFORM (nothing launched) where the layer's kernel has no such form (built: the
     * full-width-row kernel, 16 -> 16 channels on a 256-wide map). */
    const float* bsum_y;
    const float* bsum_mean;
    const float* bsum_invstd;
    const float* bsum_gamma;
    const float* bsum_beta;
    void* bsum_partials;
    float* bsum_max;
    float bsum_slope;
    int bsum_groups;
} uaps_call_hints;
int uaps_next_call_hints(const uaps_call_hints* hints);
/* Measurement aid (bench.py): `start` / `stop` are two hipEvent_t created with timing enabled.  The calling thread's next MAIN
 * kernel launch -- the convolution kernel of uaps_conv_fwd* / uaps_conv_bwd_data* / uaps_conv_bwd_weight* / uaps_convs_*, the
 * forward / backward kernel of uaps_pairloss_fwd / _bwd; not their packing, reduce or finalize launches -- attaches them to its
 * dispatch (hipExtLaunchKernel), so that hipEventElapsedTime(start, stop) is that kernel's execution time as a profiler's kernel
 * trace reports it.  One-shot; (NULL, NULL) disarms.  Returns 1 when the previously armed pair was consumed by a launch since
 * the last call, else 0. */
int uaps_next_launch_events(void* start, void* stop);
/* Measurement aid (bench.py: roofline.step_algorithmic_bytes).  uaps_account(1) zeroes a process-wide tally and switches it on:
 * from then on every kernel entry point of this library adds the ALGORITHMIC bytes of the launch it enqueues -- every operand
 * tensor read once and every result written once, fp32 / int64 as the reference holds them (SURVEY.md 8d's per-unit figures);
 * workspaces, partial sums, halo and packed-weight re-reads are not counted.  uaps_account(0) stops counting;
 * uaps_accounted_bytes() reads the tally.  Host-side bookkeeping only: nothing is launched, nothing synchronises. */
int uaps_account(int enable);
double uaps_accounted_bytes(void);
/* Zero n floats of bound storage (a multiple of UAPS_BOUND_FLOATS) with agent-scope stores -- the way bounds handed to
 * uaps_call_hints::out_amax must be cleared (a plain fill may be written back over the atomically raised value). */
int uaps_zero_bounds(float* bounds, long n, uaps_stream_t stream);
int uaps_conv_get_mode(void);
/* both packed buffers hold the fp32 layout followed by the bf16-split and the fp16-split layouts; they must be 16-byte aligned */
int uaps_conv_pack_floats(int Cout, int Cin, int ks, size_t* fwd_floats_host, size_t* bwd_floats_host);
int uaps_conv_pack_weights(const float* w, int Cout, int Cin, int ks, float* wf, float* wb, uaps_stream_t stream);
/* n convolutions packed by one launch (host arrays with n entries; wf[i] or wb[i] may be NULL). */
int uaps_conv_pack_weights_batch(const float* const* w_host, float* const* wf_host, float* const* wb_host,
                                 const int* Cout_host, const int* Cin_host, const int* ks_host, int n,
                                 uaps_stream_t stream);
/* y [B,Cout,H,W] = conv2d(x [B,Cin,H,W], w) (+ bias[Cout] if bias != NULL) */
int uaps_conv_fwd(const float* x, const float* wf, const float* bias, float* y, int B, int Cin, int Cout, int H,
                  int W, int ks, int cfg, uaps_stream_t stream);
/* uaps_conv_fwd that also writes the first pass of the BatchNorm that follows (UAPS_unet.py:37-38, 41-42):
 * stats = float2 [Cout][B][parts_per_image] per-tile (sum, sum of squares) of y, parts_per_image from
 * uaps_conv_fwd_stats_parts; consumed by uaps_bn_act_fwd_train_partials. */
int uaps_conv_fwd_stats(const float* x, const float* wf, const float* bias, float* y, void* stats, int B, int Cin,
                        int Cout, int H, int W, int ks, int cfg, uaps_stream_t stream);
int uaps_conv_fwd_stats_parts(int B, int Cin, int Cout, int H, int W, int ks, int cfg, int* parts_per_image_host);
/* dx [B,Cin,H,W] = conv_transpose2d(dy [B,Cout,H,W], w) */
int uaps_conv_bwd_data(const float* dy, const float* wb, float* dx, int B, int Cin, int Cout, int H, int W,
                       int ks, int cfg, uaps_stream_t stream);
/* dw [Cout][Cin][ks][ks] = sum over batch and pixels of dy (x) x ; dbias [Cout] = sum dy (may be NULL).
 * Partial sums per pixel split go to the workspace and are reduced in a fixed order (deterministic). */
int uaps_conv_wrw_workspace_bytes(int B, int Cin, int Cout, int H, int W, int ks, int cfg, size_t* out_host);
/* The two launches of uaps_conv_bwd_weight as separate calls (same workspace, same dims and cfg):
 * the MFMA kernel that writes the per-split partials, then the fixed-order reduction. */
int uaps_conv_bwd_weight_partial(const float* dy, const float* x, int want_bias, int B, int Cin, int Cout, int H, int W,
                                 int ks, int cfg, void* workspace, size_t workspace_bytes, uaps_stream_t stream);
int uaps_conv_bwd_weight_reduce(const void* workspace, float* dw, float* dbias, int B, int Cin, int Cout, int H, int W,
                                int ks, int cfg, uaps_stream_t stream);
/* The reductions of SEVERAL weight gradients by one launch (per 28 items): items[i] names what uaps_conv_bwd_weight_reduce would
 * have been called with for gradient i (the workspace its *_partial call wrote, dims and cfg of that call); results are
 * bit-identical to the single calls.  The workspaces must be distinct buffers, untouched between the partial call and this one.
 * A training step has ~60 convolutions; their reductions are 5-6 us launches of latency each (uaps_amd/conv.py: deferred_reduces). */
typedef struct uaps_wrw_reduce_item {
    const void* workspace; float* dw; float* dbias;      /* dbias NULL: the partial call ran with want_bias = 0 */
    int B, Cin, Cout, H, W, ks, cfg;
} uaps_wrw_reduce_item;
int uaps_conv_bwd_weight_reduce_batch(const uaps_wrw_reduce_item* items, int n, uaps_stream_t stream);
/* Convolutions over a channel concatenation that is never materialised: the first ConvBlock conv of an UpBlock
 * reads torch.cat([skip, upsampled], dim=1) (UAPS_unet.py:84-85).  x1 [B,C1,H,W], x2 [B,C2,H,W], weights packed for
 * Cin = C1 + C2 as usual; C1 must be a multiple of 16.  The input gradient comes back as two tensors, and the
 * weight gradient reads the two inputs; uaps_conv_bwd_weight_reduce(.., Cin = C1 + C2, ..) finishes it. */
/* cfg bit UAPS_CONV_X2_UP2 on uaps_conv_fwd_cat / uaps_conv_bwd_weight_partial_cat (round 5): x2 is the LOW-resolution tensor
 * [B, C2, H/2, W/2] of an UpBlock and is bilinearly up-sampled x2 (align_corners = True, ATen's arithmetic: bit-identical to
 * uaps_up_cat_fwd's output) while the kernel stages it -- `self.up(x1)` of UAPS_unet.py:74-75, 83 is never materialised.  bound[1]
 * (forward) / bound[2] (weight gradient) is the bound of the low tensor (interpolation is convex).  Built for the full-width-row
 * kernels of up4's first convolution (16 + 16 -> 16 channels, W == 256, H % 16 == 0, fp16-split arithmetic with every operand
 * bounded); anything else returns UAPS_ENOFORM with nothing launched: call uaps_up_cat_fwd and the plain entry point.  The input
 * gradient (uaps_conv_bwd_data_cat) still yields the gradient of the up-sampled tensor; uaps_up_cat_bwd folds it to [B,C2,H/2,W/2]. */
#define UAPS_CONV_X2_UP2 (1 << 10)
/* cfg bit UAPS_CONV_BOUNDED on uaps_conv_fwd_stats_parts and the forward entry points (round 6): the caller PROMISES that the call's
 * hints carry a magnitude bound for every tensor operand (mode 2).  It lets the planner pick kernels that exist in the fp16-split
 * arithmetic only and write their BatchNorm partial sums in a layout of their own (csrc/conv_split_g.hpp: one part per 4-row band on
 * 32-wide maps): uaps_conv_fwd_stats_parts reports that layout for the same dimensions + bit, and a call that sets the bit without
 * the bounds is UAPS_EINVAL, and so is uaps_conv_fwd_bn with the bit on such a shape (the staging-time BatchNorm form keeps the tile
 * kernels and their layout).  Without the bit nothing changes (calls without statistics choose the kernel from the hints alone). */
#define UAPS_CONV_BOUNDED (1 << 11)
int uaps_conv_fwd_cat(const float* x1, int C1, const float* x2, int C2, const float* wf, const float* bias, float* y,
                      void* stats_or_null, int B, int Cout, int H, int W, int ks, int cfg, uaps_stream_t stream);
int uaps_conv_bwd_data_cat(const float* dy, const float* wb, float* dx1, int C1, float* dx2, int C2, int B, int Cout,
                           int H, int W, int ks, int cfg, uaps_stream_t stream);
int uaps_conv_bwd_weight_partial_cat(const float* dy, const float* x1, int C1, const float* x2, int C2, int want_bias,
                                     int B, int Cout, int H, int W, int ks, int cfg, void* workspace,
                                     size_t workspace_bytes, uaps_stream_t stream);
/* conv -> BatchNorm(train) -> LeakyReLU -> conv without the activated tensor in between (the two convs of a ConvBlock,
 * UAPS_unet.py:37-41; an UpBlock's conv1x1 / Decoder.out_conv after a ConvBlock, :73, :138).  uaps_bn_finalize_train turns
 * the first conv's epilogue partials into batch statistics (running statistics updated as nn.BatchNorm2d does) and
 * xf [groups][C] float2 = (scale, shift) = (gamma*invstd, beta - mean*scale); the second conv reads the RAW output y of the
 * first and applies leaky_relu(fma(y, scale, shift)) (0 <= slope <= 1) while staging, in the forward (uaps_conv_fwd_bn) and in its weight gradient
 * (uaps_conv_bwd_weight_partial_bn, finished by uaps_conv_bwd_weight_reduce).  Its input gradient is uaps_conv_bwd_data,
 * followed by uaps_bn_act_bwd_grouped(drop_p = 0) on (that gradient, y).  W % 4 == 0, 16-byte aligned tensors, no
 * dilation, Cin > 4; otherwise UAPS_ERANGE (use the unfused calls). */
int uaps_bn_finalize_train(const void* partials, int parts_per_image, const float* conv_bias, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           float momentum, float eps, int B, int C, int H, int W, int groups, float* save_mean,
                           float* save_invstd, void* xf, uaps_stream_t stream);
int uaps_conv_fwd_bn(const float* x_raw, const void* xf, float slope, int groups, const float* wf, const float* bias,
                     float* y, void* stats_or_null, int B, int Cin, int Cout, int H, int W, int ks, int cfg,
                     uaps_stream_t stream);
int uaps_conv_bwd_weight_partial_bn(const float* dy, const float* x_raw, const void* xf, float slope, int groups,
                                    int want_bias, int B, int Cin, int Cout, int H, int W, int ks, int cfg,
                                    void* workspace, size_t workspace_bytes, uaps_stream_t stream);
/* ---- Explicit form of the convolution calls (round 5): ONE size-versioned argument struct per call, no side channel -------------
 * The entry points above take their optional operands (magnitude bounds, statistics shift, the pending BatchNorm transform of
 * dy) from uaps_next_call_hints, a one-shot thread-local record: a binding has to know which call consumes it.  uaps_conv_ex
 * carries everything in the struct; nothing set before the call is looked at (pending hints of the thread are dropped) and
 * nothing survives it, so it is re-entrant per call and a foreign-language binding sees every operand in one place.  Same
 * kernels, same results, same return codes (UAPS_ENOFORM for a dyt_* request the layer's kernel cannot honour).
 *   op = UAPS_CONV_FWD            y (and y's BatchNorm partials into `stats` when non-NULL) = conv(x [, x2]); with `xf` non-NULL the
 *                                 input is a raw conv output and leaky_relu(fma(x, scale, shift)) is applied while staging
 *        UAPS_CONV_BWD_DATA       dx (= y [, y2 for a two-tensor input]) = conv_transpose(dy = x)
 *        UAPS_CONV_BWD_WEIGHT     per-split partials into `workspace` from (dy = y_grad, x [, x2]); finish with uaps_conv_bwd_weight_reduce
 *   Fields that an op does not use must be zero.  struct_size = sizeof(uaps_conv_call) of the caller's header; fields are only
 *   appended, a shorter struct of an older client reads as zero beyond its size. */
#define UAPS_CONV_FWD 0
#define UAPS_CONV_BWD_DATA 1
#define UAPS_CONV_BWD_WEIGHT 2
typedef struct uaps_conv_call {
    unsigned struct_size;
    int op;
    int B, Cin, Cout, H, W, ks, cfg;     /* Cin = C1 + C2 for a two-tensor input */
    const float* x;  int C1;             /* first input tensor [B,C1,H,W] (C1 = Cin without x2); BWD_DATA: dy [B,Cout,H,W] */
    const float* x2;                     /* second input tensor [B,Cin-C1,H,W] or NULL */
    const float* w_packed;               /* uaps_conv_pack_weights: wf (FWD), wb (BWD_DATA); unused by BWD_WEIGHT */
    const float* bias;                   /* FWD: [Cout] or NULL */
    float* y;                            /* FWD: output; BWD_DATA: dx (first tensor) */
    float* y2;                           /* BWD_DATA of a two-tensor input: dx of the second tensor */
    const float* y_grad;                 /* BWD_WEIGHT: dy [B,Cout,H,W] */
    void* stats;                         /* FWD: float2 [Cout][B][parts] or NULL (uaps_conv_fwd_stats_parts) */
    const void* xf; float xf_slope; int xf_groups;      /* staging-time BatchNorm + LeakyReLU of x (uaps_bn_finalize_train) or NULL */
    int want_bias;                       /* BWD_WEIGHT: also the bias gradient's partials */
    void* workspace; size_t workspace_bytes;            /* BWD_WEIGHT (uaps_conv_wrw_workspace_bytes) */
    uaps_stream_t stream;                /* ABI 3: in FRONT of the hints -- the record below is the part of this struct that grows, and a
                                          * client built against a shorter uaps_call_hints must still have its stream where the library
                                          * reads it (ABI 2 had it behind the hints) */
    uaps_call_hints hints;               /* the operands' bounds, statistics shift, dyt_* ...; hints.struct_size may be 0 = none, and
                                          * must not exceed what struct_size leaves for it */
} uaps_conv_call;
int uaps_conv_ex(const uaps_conv_call* call);

/* Name (as rocprofv3 prints it, without namespace) of the kernel instantiation the calls above launch
 * for these dimensions; buf_host needs >= 64 bytes.  For uaps_conv_bwd_data pass Cin and Cout swapped
 * to uaps_conv_fwd_variant.  Used by bench.py to group its per-launch HIP-event timings. */
int uaps_conv_fwd_variant(int B, int Cin, int Cout, int H, int W, int ks, int cfg, char* buf_host, size_t buflen);
int uaps_conv_wrw_variant(int B, int Cin, int Cout, int H, int W, int ks, int cfg, char* buf_host, size_t buflen);
int uaps_conv_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, int B, int Cin, int Cout, int H,
                         int W, int ks, int cfg, void* workspace, size_t workspace_bytes, uaps_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Sibling consistency terms on two [B,C,H,W] tensors (C in [2,8]): the formulations the reference keeps beside
 * the UAPS block (utilities/losses_1.py, losses_2.py; star-imported at UAPS_train.py:21-22) and its evaluation
 * notebook uses (UAPS-Testing.ipynb cell 24).  a = input logits (receives the gradient), b = target logits.
 *   mse_map [B,C,H,W] = (softmax(a) - softmax(b))^2                      losses_1.py:9-26  softmax_mse_loss
 *   kl_map  [B,H,W]   = sum_c KLDivLoss('none')(log_softmax(a), softmax(b))   notebook cell 24 uncertainty map
 *   kl_mean [1]       = F.kl_div(log_softmax(a), softmax(b), reduction='mean')  losses_1.py:29-48 softmax_kl_loss
 * Any of the three outputs may be NULL.  With probs != 0 the inputs are probabilities, not logits, and kl_mean is
 * F.kl_div(log(a), b, reduction='mean') = kl_loss(pr=a, gt=b) of losses_2.py:201-213.
 * workspace: uaps_pair_workspace_bytes() (needed for kl_mean only).
 * ------------------------------------------------------------------------------------------- */
int uaps_pair_workspace_bytes(size_t* out_host);
int uaps_softmax_pair_fwd(const float* a, const float* b, int probs, int B, int C, int H, int W, float* mse_map,
                          float* kl_map, float* kl_mean, void* workspace, size_t workspace_bytes, uaps_stream_t stream);
/* d(sum(grad_map * mse_map)) / da   and   d(g * kl_mean) / da  (g = *gscalar, 1 if NULL); b gets no gradient. */
int uaps_softmax_mse_bwd(const float* a, const float* b, const float* grad_map, int B, int C, int H, int W, float* da,
                         uaps_stream_t stream);
int uaps_softmax_kl_bwd(const float* a, const float* b, const float* gscalar, int B, int C, int H, int W, float* da,
                        uaps_stream_t stream);
/* d(sum(grad_map * kl_map)) / da and / db (either may be NULL): the per-pixel KL map differentiated w.r.t. both heads,
 * as UCC weights its pseudo-supervision (UCC/UCC_train.py:213-217, 231-232; neither softmax is detached there). */
int uaps_softmax_klmap_bwd(const float* a, const float* b, const float* grad_map, int B, int C, int H, int W, float* da,
                           float* db, uaps_stream_t stream);
/* entropy_map(p) = -sum_c p log(p + 1e-6) as [B,1,H,W] and/or its mean (entropy_minmization), losses_1.py:139-149. */
int uaps_entropy_map(const float* p, int B, int C, int H, int W, float* ent_map, float* ent_mean, void* workspace,
                     size_t workspace_bytes, uaps_stream_t stream);

/* Residual join of the ResNet blocks (utilities/resnet.py:47-50, 88-91): out = relu(a + b) over n floats; the
 * backward of both addends is dx = dout * (out > 0). */
int uaps_add_relu(const float* a, const float* b, float* out, long n, uaps_stream_t stream);
int uaps_relu_bwd(const float* dout, const float* out, float* dx, long n, uaps_stream_t stream);
/* BatchNorm(train) + LeakyReLU backward in two halves (no dropout).  `prepare`: the reductions -- coef [groups][C][8] floats =
 * (mean, invstd, gamma invstd, beta, mean(d), mean(d x_hat), 0, 0), dgamma, dbeta, the zero gradient of a conv bias in front
 * (dconv_bias may be NULL) -- and the zeroed bound dy_bound raised to an upper bound of |dy|.  dy itself is then formed by the
 * weight-gradient kernel of the convolution in front (uaps_call_hints::dyt_*) or by `apply` (uaps_call_hints::out_amax
 * honoured); both give what uaps_bn_act_bwd_grouped writes.  Workspace: uaps_bn_workspace_bytes. */
int uaps_bn_act_bwd_prepare(const float* dout, const float* y, const float* gamma, const float* beta, const float* save_mean,
                            const float* save_invstd, float slope, int B, int C, int H, int W, int groups, float* coef,
                            float* dgamma, float* dbeta, float* dconv_bias, float* dy_bound, void* workspace, size_t workspace_bytes,
                            uaps_stream_t stream);
int uaps_bn_act_bwd_finalize(const void* partials, int parts_per_image, const float* maxes, const float* gamma, const float* beta,
                             const float* save_mean, const float* save_invstd, int B, int C, int H, int W, int groups, float* coef,
                             float* dgamma, float* dbeta, float* dconv_bias, float* dy_bound, uaps_stream_t stream);
int uaps_bn_act_bwd_apply(const float* dout, const float* y, const float* coef, float slope, int B, int C, int H, int W, int groups,
                          float* dy, uaps_stream_t stream);

/* The same when the join's output had k <= 4 consumers: dx = (dout[0] + ... + dout[k-1]) * (out > 0), the k gradients (host
 * array of device pointers) summed in order by this pass instead of by k - 1 accumulation passes in front of it. */
int uaps_relu_bwd_sum(const float* const* dout_host, int k, const float* out, float* dx, long n, uaps_stream_t stream);

/* The two batches of a training step as one tensor (the reference runs model(labelled) and model(unlabelled) one after the other,
 * UAPS_train.py:177 and :185; this build runs them as one pass over both): out [2n] = a [n] followed by b [n].  With
 * uaps_call_hints::out_amax the bound is raised to max|out| on the way. */
int uaps_cat2(const float* a, const float* b, float* out, long n, uaps_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Input pipeline: the per-sample work of the reference's training loader (utilities/dataloaders.py:60-119: albumentations
 * Resize(nearest) / flips / RandomBrightnessContrast / Blur / RandomRotate90 / GaussNoise, then ToTensor + Normalize; the
 * mask follows the geometry and becomes int64) on a decoded uint8 batch in device memory.
 *   images_hwc [B][Hs][Ws][3] uint8 (RGB, as cv2.cvtColor leaves it), masks [B][Hs][Ws] uint8 or NULL
 *   params_i   [B][8] int32 : hflip, vflip, rot_k (np.rot90 count), blur_k (0 / 3 / 5 / 7), noise_on, reserved x3
 *   params_f   [B][4] float : alpha (contrast factor), beta (brightness shift / 255), sigma (noise std, grey levels), reserved
 *   noise      [B][3][Ho][Wo] float already scaled by sigma, or NULL: drawn in the kernel (Philox, Box-Muller) from `seed`
 *   out        [B][3][Ho][Wo] float normalised with mean3_host / std3_host; out_mask [B][Ho][Wo] int64 (NULL with masks)
 * rot_k odd needs Ho == Wo.  Parity unpinned (cv2 / albumentations are not in the build image): see csrc/augment.hip.
 * ------------------------------------------------------------------------------------------- */
int uaps_augment_batch(const uint8_t* images_hwc, const uint8_t* masks, const int* params_i, const float* params_f,
                       const float* noise, uint64_t seed, int B, int Hs, int Ws, int Ho, int Wo, const float* mean3_host,
                       const float* std3_host, float* out, int64_t* out_mask, uaps_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer: torch.optim.Adam(model.parameters(), lr) of UAPS_train.py:112, stepped at :292 -- one multi-tensor
 * launch per 48 tensors instead of PyTorch's per-chunk foreach kernels.  Host arrays of n device pointers / sizes;
 * `step` is the 1-based step count (bias corrections 1 - beta^step are computed on the host in double).
 * weight_decay is the L2 form of torch.optim.Adam (added to the gradient); amsgrad / maximize are not supported.
 * ------------------------------------------------------------------------------------------- */
int uaps_adam_step(float* const* params_host, const float* const* grads_host, float* const* exp_avg_host,
                   float* const* exp_avg_sq_host, const long* numel_host, int n, double lr, double beta1, double beta2,
                   double eps, double weight_decay, long step, uaps_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Metrics: utilities/metrics.py:8-61 (pixel_accuracy, mIoU, mDice) need only the C x C confusion
 * matrix of arg-max(logits) against the labels: counts[t*C + p], int64, overwritten.
 * ------------------------------------------------------------------------------------------- */
int uaps_seg_confusion(const float* logits, const int64_t* labels, int B, int C, int H, int W,
                       int64_t* counts, uaps_stream_t stream);

/* ---- Explicit-hints forms (round 6, ABI 3) -------------------------------------------------------------------------------------
 * `<name>_h(hints, args...)` == `<name>(args...)` with `hints` (NULL or struct_size 0 = none) instead of the thread's pending
 * uaps_next_call_hints record, which is neither read nor cleared.  Same kernels, same results, same return codes; the record is
 * copied size-versioned like uaps_next_call_hints does.  These are what uaps_amd/ (ctypes) binds. */
int uaps_conv_fwd_h(const uaps_call_hints* hints, const float* x, const float* wf, const float* bias, float* y, int B, int Cin, int Cout,
                    int H, int W, int ks, int cfg, uaps_stream_t stream);
int uaps_conv_fwd_stats_h(const uaps_call_hints* hints, const float* x, const float* wf, const float* bias, float* y, void* stats, int B,
                          int Cin, int Cout, int H, int W, int ks, int cfg, uaps_stream_t stream);
int uaps_conv_fwd_bn_h(const uaps_call_hints* hints, const float* x_raw, const void* xf, float slope, int groups, const float* wf,
                       const float* bias, float* y, void* stats, int B, int Cin, int Cout, int H, int W, int ks, int cfg,
                       uaps_stream_t stream);
int uaps_conv_fwd_cat_h(const uaps_call_hints* hints, const float* x1, int C1, const float* x2, int C2, const float* wf, const float* bias,
                        float* y, void* stats, int B, int Cout, int H, int W, int ks, int cfg, uaps_stream_t stream);
int uaps_conv_bwd_data_h(const uaps_call_hints* hints, const float* dy, const float* wb, float* dx, int B, int Cin, int Cout, int H, int W,
                         int ks, int cfg, uaps_stream_t stream);
int uaps_conv_bwd_data_cat_h(const uaps_call_hints* hints, const float* dy, const float* wb, float* dx1, int C1, float* dx2, int C2, int B,
                             int Cout, int H, int W, int ks, int cfg, uaps_stream_t stream);
int uaps_conv_bwd_weight_partial_h(const uaps_call_hints* hints, const float* dy, const float* x, int want_bias, int B, int Cin, int Cout,
                                   int H, int W, int ks, int cfg, void* workspace, size_t workspace_bytes, uaps_stream_t stream);
int uaps_conv_bwd_weight_partial_bn_h(const uaps_call_hints* hints, const float* dy, const float* x_raw, const void* xf, float slope,
                                      int groups, int want_bias, int B, int Cin, int Cout, int H, int W, int ks, int cfg,
                                      void* workspace, size_t workspace_bytes, uaps_stream_t stream);
int uaps_conv_bwd_weight_partial_cat_h(const uaps_call_hints* hints, const float* dy, const float* x1, int C1, const float* x2, int C2,
                                       int want_bias, int B, int Cout, int H, int W, int ks, int cfg, void* workspace,
                                       size_t workspace_bytes, uaps_stream_t stream);
int uaps_bn_act_fwd_train_partials_h(const uaps_call_hints* hints, const void* partials, int parts_per_image, const float* y,
                                     const float* conv_bias, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, int64_t* num_batches_tracked, float momentum, float eps, float slope,
                                     float drop_p, uint64_t seed, uint64_t offset, int B, int C, int H, int W, int groups, float* out,
                                     float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes, uaps_stream_t stream);
int uaps_bn_finalize_train_h(const uaps_call_hints* hints, const void* partials, int parts_per_image, const float* conv_bias,
                             const float* gamma, const float* beta, float* running_mean, float* running_var,
                             int64_t* num_batches_tracked, float momentum, float eps, int B, int C, int H, int W, int groups,
                             float* save_mean, float* save_invstd, void* xf, uaps_stream_t stream);
int uaps_bn_act_bwd_grouped_h(const uaps_call_hints* hints, const float* dout, const float* y, const float* gamma, const float* beta,
                              const float* save_mean, const float* save_invstd, float slope, float drop_p, uint64_t seed,
                              uint64_t offset, int B, int C, int H, int W, int groups, float* dy, float* dgamma, float* dbeta,
                              void* workspace, size_t workspace_bytes, uaps_stream_t stream);
int uaps_bn_act_bwd_grouped_bias_h(const uaps_call_hints* hints, const float* dout, const float* y, const float* gamma, const float* beta,
                                   const float* save_mean, const float* save_invstd, float slope, float drop_p, uint64_t seed,
                                   uint64_t offset, int B, int C, int H, int W, int groups, float* dy, float* dgamma, float* dbeta,
                                   float* dconv_bias, void* workspace, size_t workspace_bytes, uaps_stream_t stream);
int uaps_bn_act_bwd_apply_h(const uaps_call_hints* hints, const float* dout, const float* y, const float* coef, float slope, int B, int C,
                            int H, int W, int groups, float* dy, uaps_stream_t stream);
int uaps_up_cat_fwd_h(const uaps_call_hints* hints, const float* skip, const float* low, float* out, int B, int Cs, int Cl, int h, int w,
                      uaps_stream_t stream);
int uaps_cat2_h(const uaps_call_hints* hints, const float* a, const float* b, float* out, long n, uaps_stream_t stream);
int uaps_add_relu_h(const uaps_call_hints* hints, const float* a, const float* b, float* out, long n, uaps_stream_t stream);
int uaps_pairloss_bwd_h(const uaps_call_hints* hints, const float* const* lab_logits_host, const float* const* un_logits_host,
                        const int64_t* labels, const int64_t* pseudo, const float* sup_scalars, const float* unsup_scalars, float cw1,
                        float cw2, const float* grad_scale, int D, int B, int C, int H, int W, long n_pixels_loss,
                        float* const* dlab_host, float* const* dun_host, int cfg, uaps_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UAPS_HIP_H */
