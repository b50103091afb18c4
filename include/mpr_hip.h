/* mpr_hip.h -- C ABI of libmpr_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * train_multi hot path of imveikka/multimodal_plankton_recognition.
 *
 * The reference has no FFI: its hot path bottoms out in torch / timm / cuDNN calls.  Each entry
 * point below names the reference call it stands in for (paths relative to the reference root).
 * Conventions:
 *   - every pointer is a DEVICE pointer unless stated otherwise; the library never allocates,
 *     frees or synchronises -- the caller (PyTorch's caching allocator in this repo) owns all memory;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing else;
 *   - activations are channels-last bf16 ([B,H,W,C]; a 1-D sequence is H == 1), statistics,
 *     embeddings, losses and parameters / gradients are fp32 (parameters in torch's own layouts);
 *   - return value: 0 = ok, 1 = invalid argument, 2 = HIP runtime error; mpr_last_error() gives the
 *     message for the calling thread.  Nothing aborts.
 *   - "stat rows": train-mode BatchNorm statistics are partial sums [rows][2][C] (sum, sum of squares).  By default
 *     the producers ADD into a small fixed number of zeroed slice rows (fp32 atomics, mpr_conv_set_stat_slices: 4) that
 *     the consuming kernel finalizes itself (mpr_bn_apply_fin, mpr_bn_bwd_apply_fin, mpr_stemf_pool); with slices off they
 *     write one row per workgroup (bitwise reproducible; the *_stat_rows / mpr_bn_reduce_rows functions give the count)
 *     for mpr_bn_reduce_partials / mpr_bn_finalize_stats.
 *   - tuning knobs and timing-experiment hooks (tile variants, kernel on/off switches, in-kernel time stamps, operand
 *     dropping) are NOT part of this header: include/mpr_hip_debug.h.
 */
#ifndef MPR_HIP_H
#define MPR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime ------------------------------------------------------------------------------- */
int mpr_abi_version(void);
const char* mpr_target_arch(void);                 /* "gfx950" */
const char* mpr_last_error(void);
void mpr_set_error(const char* fmt, ...);
int mpr_device_check(char* name, int name_len);    /* 0 iff device 0 is gfx950; name: HOST buffer */
/* opt-in hipEvent profiler around the conv kernels (kinds: 0/1 LDS-DMA igemm fwd/dgrad, 2 LDS-DMA (gather) wgrad,
 * 3/4 register-staged igemm fwd/dgrad, 5 register-staged wgrad, 6/7 shifted-window conv fwd/dgrad, 8 sliding-window
 * wgrad);
 * collect sums elapsed ms / algorithmic FLOPs / launches into HOST variables (kind -1: all). */
int mpr_prof_enable(int on);
int mpr_prof_reset(void);
int mpr_prof_collect(int kind, double* total_ms, double* total_work, int* launches);
/* algorithmic HBM bytes of the recorded launches of `kind`: operands read once, results written once */
int mpr_prof_collect_bytes(int kind, double* total_bytes);

/* ---- convolution as implicit GEMM (bf16 MFMA, fp32 accumulate) ------------------------------
 * Stand in for nn.Conv2d inside timm's ResNet (src/image_encoder.py:24) and nn.Conv1d in
 * ProfileCNN / _BasicBlock (src/profile_encoder.py:125,128,167,187-190), forward and backward. */
int mpr_conv_packed_sizes(int K, int C, int R, int S, long long* fwd_elems, long long* dgrad_elems);
int mpr_conv_pack_weights(const float* w_oihw, void* w_fwd, void* w_dgrad /* may be NULL */, int K, int C, int R,
                          int S, void* stream);
/* the same for a filter of logical shape [K,C,R,S] and element strides (sk, sc, sr, ss): e.g. the [K][R][S][C]
 * memory of a channels-last weight */
int mpr_conv_pack_weights_strided(const float* w, long long sk, long long sc, long long sr, long long ss, void* w_fwd,
                                  void* w_dgrad /* may be NULL */, int K, int C, int R, int S, void* stream);
/* every filter of a model in ONE launch (after the optimizer step): `table` = n device rows of 12 int64
 * {w, w_fwd, w_dgrad (0: none), K, C, R, S, sk, sc, sr, ss, 0} */
int mpr_conv_pack_weights_multi(const void* table, int n, void* stream);
int mpr_conv_fwd_stat_rows(int B, int P, int Q, int K, int C, int R, int S, int sh, int sw, int ph, int pw);
/* BatchNorm partial sums of mpr_conv_fwd: n > 0 (default 8) = every tile adds its sums (fp32 atomics) into one of n slice
 * rows zeroed by the call, so the consumer finalizes from n rows and no pre-reduction launch sits between a convolution
 * and its BatchNorm; 0 = one row per tile (bitwise reproducible); n < 0 = query only.  Returns the previous setting. */
int mpr_conv_set_stat_slices(int n);
/* one-shot: the slice rows given to the NEXT mpr_conv_fwd are already zero (skip its memset) */
int mpr_conv_stats_prezeroed(int on);
/* lend `floats` floats of device scratch to the NEXT mpr_conv_wgrad call (one-shot): the sliding-window kernel then
 * writes each pixel split's partial tile with plain stores and sums the slices in a second kernel on the same stream,
 * instead of fp32 atomics (75 MB per launch at the chip's ~1.3 TB/s atomic rate); too small / NULL: atomics */
int mpr_conv_set_wgrad_scratch(void* buf, long long floats);
int mpr_conv_fwd(const void* x, const void* w_fwd, void* y, float* stats /* may be NULL */, int B, int H, int W,
                 int C, int K, int R, int S, int sh, int sw, int ph, int pw, void* stream);
int mpr_conv_dgrad(const void* dy, const void* w_dgrad, void* dx, const void* add /* may be NULL */, int B, int H,
                   int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, void* stream);
/* stride-2 data gradient + the data gradient of a parallel 1x1 / stride-2 / pad-0 shortcut convolution given on the
 * half-resolution grid, add_even [B, H/2, W/2, C] (it only reaches the even pixels): the block-input gradient of a ResNet
 * downsampling block (timm BasicBlock with downsample, behind src/image_encoder.py:24) without the 3/4-zero shortcut map.
 * Geometries: mpr_conv_dgrad_add_even_supported != 0 */
int mpr_conv_dgrad_add_even_supported(int B, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw);
int mpr_conv_dgrad_s2(const void* dy, const void* w_dgrad, void* dx, const void* add_even, int B, int H, int W, int C,
                      int K, int R, int S, int sh, int sw, int ph, int pw, void* stream);
/* ... + the BatchNorm backward of the block output it differentiates (relu(bn2(x2) + identity) of the PREVIOUS block):
 * dz = gradient masked by mask_y > 0, slices [nslices][2][C] += sum dz, sum dz * xhat(bn_x); add_even may be NULL */
int mpr_conv_dgrad_s2_bn(const void* dy, const void* w_dgrad, void* dz, const void* add_even, const void* mask_y,
                         const void* bn_x, const float* mean, const float* invstd, float* slices, int nslices,
                         int prezeroed, int B, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw,
                         void* stream);
/* dw_oihw != NULL: workspace is zeroed, filled as [K][R][S][C] and permuted into (accumulate: added to) dw_oihw.
 * dw_oihw == NULL: the gradient stays in `workspace` as [K][R][S][C] -- the memory of a channels-last weight's
 * gradient -- zeroed first unless `accumulate` (then the split-K atomics add into what is there) */
int mpr_conv_wgrad(const void* x, const void* dy, float* workspace /* K*R*S*C floats */, float* dw_oihw /* may be NULL */,
                   int accumulate, int B, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph,
                   int pw, void* stream);
/* The same with every per-launch choice as an ARGUMENT instead of a setter armed "for the next call" (round 3; the Python
 * layer uses this form): `scratch` (scratch_floats floats, may be NULL) is lent to THIS call for the window kernel's partial
 * tiles; target_wgs <= 0: the library default; kernel: -1 / 1 = sliding-window kernel where the geometry allows, 0 = never */
int mpr_conv_wgrad_ex(const void* x, const void* dy, float* workspace /* K*R*S*C floats */, float* dw_oihw /* may be NULL */,
                      int accumulate, int B, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw,
                      void* scratch /* may be NULL */, long long scratch_floats, int target_wgs, int kernel, void* stream);

/* Data gradient of a 3x3 / stride 1 / pad 1 convolution whose INPUT was the output of BatchNorm (+ ReLU) -- timm
 * BasicBlock / _BasicBlock (src/profile_encoder.py:132-148: conv -> BN -> ReLU -> conv -> BN -> add -> ReLU): the epilogue
 * applies the ReLU mask and accumulates that BatchNorm's backward sums, dz = mask * (conv_transpose(dy, w) (+ add)),
 * slices[nslices][2][C] += (sum dz, sum dz * (bn_x - mean) * invstd) -- the separate reduction pass over (dy, y, x) of
 * mpr_bn_bwd_reduce disappears (then: mpr_bn_bwd_apply_fin with mask_mode 0 on dz).
 * mask_mode 1: mask = mask_y > 0 (the block output);  2: mask = bf16(bn_x * scale + shift) > 0 (recomputed). */
int mpr_conv_dgrad_bn_supported(int B, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw);
int mpr_conv_dgrad_bn(const void* dy, const void* w_dgrad, void* dz, const void* add /* may be NULL */, int mask_mode,
                      const void* mask_y, const void* bn_x, const float* mean, const float* invstd, const float* scale,
                      const float* shift, float* slices, int nslices, int prezeroed, int B, int H, int W, int C, int K,
                      int R, int S, int sh, int sw, int ph, int pw, void* stream);

/* ---- stem convolutions (few input channels, fp32 input, direct) -----------------------------
 * timm ResNet conv1 (1->64, 7x7/2) and ProfileCNN.conv1 (src/profile_encoder.py:167). */
/* ResNet stem as space-to-depth (7x7/2 on 1 channel == 4x4/1 on the 4 (+4 zero) phase channels, halo
 * materialised): the conv itself then runs on mpr_conv_fwd / mpr_conv_wgrad with C=8, R=S=4, pad 0 */
int mpr_stem_s2d(const float* x, void* xs /* [B][H/2+3][W/2+3][8] bf16 */, int B, int H, int W, void* stream);
int mpr_stem_w_s2d(const float* w /* [K][1][7][7] */, float* w2 /* [K][8][4][4] */, int K, void* stream);
int mpr_stem_dw_gather(const float* dw2, float* dw, int K, int accumulate, void* stream);
int mpr_stem_fwd_stat_rows(int B, int P, int Q, int K);
int mpr_stem_fwd(const float* x, const float* w_oihw, void* y, float* stats /* may be NULL */, int B, int H, int W,
                 int Cin, int K, int R, int S, int sh, int sw, int ph, int pw, void* stream);
int mpr_stem_wgrad(const float* x, const void* dy, float* dw_oihw, int accumulate, int B, int H, int W, int Cin,
                   int K, int R, int S, int sh, int sw, int ph, int pw, void* stream);

/* ---- ResNet stem, fused and recomputed (stem_fused.hip): conv 7x7/2 (1 -> 64) + BatchNorm + ReLU + MaxPool 3x3/2 of
 * timm's ResNet (conv1 / bn1 / act1 / maxpool behind src/image_encoder.py:16,24) without ever storing the
 * full-resolution conv output (822 MB at batch 512): the forward recomputes the cheap convolution in both of its passes,
 * the backward needs no convolution at all (conv linearity: it accumulates dz^T * patches and the patch Gram matrix). */
int mpr_stemf_supported(int H, int W, int K);      /* 1 iff this geometry is served (K == 64, H % 4 == 0, W % 32 == 0) */
/* x fp32 [B,H,W] (one channel) -> xb bf16 [B][H+6][W+8], zero padded; w fp32 [64][1][7][7] -> wp bf16 [64][64] */
int mpr_stemf_prep(const float* x, const float* w, void* xb, void* wp, int B, int H, int W, void* stream);
/* pass A: per-channel sum / sum of squares of the bf16-rounded conv output, ADDED into stats[nslices][2][64] (zeroed here
 * unless `prezeroed`) */
int mpr_stemf_stats(const void* xb, const void* wp, float* stats, int nslices, int prezeroed, int B, int H, int W,
                    void* stream);
/* pass B: pooled [B,H/4,W/4,64] bf16 = maxpool(relu(bn(conv))).  slices != NULL (train): the statistics are finalized
 * here -- scale, shift, mean, invstd are OUTPUTS, the running statistics are updated; slices == NULL (eval): scale and
 * shift are inputs.  idx (may be NULL): 1-byte arg-max code kh*3+kw per pooled element, first maximum in torch's scan
 * order, 15 where the pooled activation is 0 (the winner's ReLU derivative is 0: no gradient) -- all mpr_stemf_bwd needs. */
int mpr_stemf_pool(const void* xb, const void* wp, const float* slices, int nsl, long long count, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* scale,
                   float* shift, float* mean, float* invstd, void* pooled, void* idx, int B, int H, int W, void* stream);
int mpr_stemf_bwd_parts(int B, int H, int W);      /* partial blocks (7168 floats each) mpr_stemf_bwd writes */
int mpr_stemf_bwd(const void* xb, const void* dpooled, const void* idx, float* partial, int B, int H, int W,
                  void* stream);
/* partial blocks -> dW [64][1][7][7], dgamma, dbeta (accumulated into what is there when the flag is set).
 * scratch: 7168 doubles.  eval != 0: BatchNorm ran on the running statistics (mean = running_mean, invstd_or_var =
 * running_var): dx = scale * dz */
int mpr_stemf_bwd_finalize(const float* partial, int nparts, void* scratch, const void* wp, long long count,
                           const float* gamma, const float* mean, const float* invstd_or_var, float eps, int eval,
                           float* dw, int accumulate_dw, float* dgamma, float* dbeta, int accumulate_bn, void* stream);

/* ---- BatchNorm, train and eval (nn.BatchNorm1d/2d defaults: src/profile_encoder.py:126,129,168) */
int mpr_bn_reduce_rows(long long rows, int C);
int mpr_bn_stats(const void* x, float* partials, long long rows, int C, void* stream);
int mpr_bn_reduce_partials(const float* partials, int nparts, float* out /* [nsplit][2][C] */, int nsplit, int C,
                           void* stream);
int mpr_bn_finalize_stats(const float* partials, int nparts, long long count, const float* gamma, const float* beta,
                          float* running_mean /* may be NULL */, float* running_var, float momentum, float eps,
                          float* scale, float* shift, float* mean, float* invstd, int C, void* stream);
int mpr_bn_eval_coefs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, float* scale, float* shift, int C, void* stream);
/* invstd = 1 / sqrt(running_var + eps): with mean = running_mean and count = 0 the backward entry points below treat
 * the layer as the affine map it is in eval mode (dx = gamma * invstd * dz; dgamma, dbeta as usual) */
int mpr_bn_eval_invstd(const float* running_var, float eps, float* invstd, int C, void* stream);
int mpr_bn_apply(const void* x, const float* scale, const float* shift, const void* residual /* may be NULL */,
                 int relu, void* y, long long rows, int C, void* stream);
/* mask_mode: 0 = dz = dy; 1 = dz = dy * (y > 0); 2 = dz = dy * (x*scale+shift > 0); 3 = dz = dy * silu'(x*scale+shift)
 * (mpr_bn_apply's `relu` argument: 0 none, 1 ReLU, 2 SiLU) */
int mpr_bn_bwd_reduce(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                      const float* scale, const float* shift, int mask_mode, float* partials, long long rows, int C,
                      void* stream);
/* the same reduction with every workgroup adding into one of `nslices` rows of slices[nslices][2][C] (zeroed by the call unless `prezeroed`):
 * the consumer (mpr_bn_bwd_apply_fin) finalizes from them directly, no pre-reduction launch in between */
int mpr_bn_bwd_reduce_slices(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                             const float* scale, const float* shift, int mask_mode, float* slices, int nslices,
                             int prezeroed /* the rows are already zero: skip the memset */, long long rows, int C,
                             void* stream);
int mpr_bn_bwd_finalize(const float* partials, int nparts, long long count, const float* gamma, const float* mean,
                        const float* invstd, float* dgamma, float* dbeta, int accumulate, float* coef /* [3][C] */,
                        int C, void* stream);
/* mpr_bn_reduce_partials + mpr_bn_finalize_stats / mpr_bn_bwd_finalize in ONE launch: every workgroup writes its
 * slice row of `slices` [nsplit][2][C], the last one (device-side ticket) sums the slices and finalizes.
 * `ticket`: one int of device memory, zero on entry; the kernel leaves it zero again. */
int mpr_bn_reduce_finalize_stats(const float* partials, int nparts, float* slices, int nsplit, int* ticket,
                                 long long count, const float* gamma, const float* beta,
                                 float* running_mean /* may be NULL */, float* running_var, float momentum, float eps,
                                 float* scale, float* shift, float* mean, float* invstd, int C, void* stream);
int mpr_bn_reduce_bwd_finalize(const float* partials, int nparts, float* slices, int nsplit, int* ticket,
                               long long count, const float* gamma, const float* mean, const float* invstd,
                               float* dgamma, float* dbeta, int accumulate, float* coef /* [3][C] */, int C,
                               void* stream);
/* the finalize steps folded into their consumers (no launch of their own in the dependent chain): every workgroup of
 * the apply kernel sums the `nsl` pre-reduced slice rows [nsl][2][C] itself; workgroup 0 writes what is needed later
 * (scale / shift / mean / invstd + running statistics; dgamma / dbeta).  C <= 512. */
int mpr_bn_apply_fin(const void* x, const float* slices, int nsl, long long count, const float* gamma, const float* beta,
                     float* running_mean /* may be NULL */, float* running_var, float momentum, float eps, float* scale,
                     float* shift, float* mean, float* invstd, const void* residual /* may be NULL */, int relu, void* y,
                     long long rows, int C, void* stream);
/* the output of a residual block with a projection shortcut in ONE pass: y = act(BN(x) + bf16(BN_r(xr))), x and xr the raw
 * outputs of conv2 and of the 1x1 shortcut conv (timm BasicBlock.forward: `shortcut = self.downsample(shortcut); x +=
 * shortcut; x = self.act2(x)`; src/profile_encoder.py:139-147) -- the normalised shortcut map is never stored; bit-identical
 * to mpr_bn_apply_fin(xr) followed by mpr_bn_apply_fin(x, residual).  Either BatchNorm may be finalized already
 * (slices == NULL: scale / shift are inputs) or pending (slices [nsl][2][C], as in mpr_bn_apply_fin).  C <= 512. */
int mpr_bn_apply_dual(const void* x, const float* slices, int nsl, long long count, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                      float* mean, float* invstd, const void* xr, const float* slices_r, int nsl_r, long long count_r,
                      const float* gamma_r, const float* beta_r, float* running_mean_r, float* running_var_r,
                      float momentum_r, float eps_r, float* scale_r, float* shift_r, float* mean_r, float* invstd_r,
                      int relu, void* y, long long rows, int C, void* stream);
int mpr_bn_bwd_apply_fin(const void* dy, const void* y, const void* x, const float* slices, int nsl, long long count,
                         const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                         int accumulate, const float* scale, const float* shift, int mask_mode, void* dx,
                         void* dz_out /* may be NULL */, long long rows, int C, void* stream);
int mpr_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* coef, const float* scale,
                     const float* shift, int mask_mode, void* dx, void* dz_out /* may be NULL */, long long rows,
                     int C, void* stream);

/* ---- pooling (nn.MaxPool1d(3,2,1) src/profile_encoder.py:170, timm maxpool 3x3/2; AdaptiveMaxPool1d(1)
 *      src/profile_encoder.py:177; timm global average pool) */
int mpr_bn_relu_maxpool_fwd(const void* x, const float* scale /* NULL: plain max-pool */, const float* shift, void* y,
                            void* idx /* 1 byte / output element */, int B, int H, int W, int C, int RH, int RW,
                            int SH, int SW, int PH, int PW, void* stream);
int mpr_maxpool_bwd(const void* dy, const void* idx, void* dx, int B, int H, int W, int C, int RH, int RW, int SH,
                    int SW, int PH, int PW, void* stream);
/* stem backward, fused: max-pool gradient gathered on the fly inside both BatchNorm-backward passes
 * (pass 0: partial sums -> mpr_bn_bwd_finalize -> pass 1: dx); the full-resolution gradient never exists */
int mpr_pool_bn_bwd_rows(int B, int H, int W, int C);
int mpr_pool_bn_bwd(int pass, const void* dy_pooled, const void* idx, const void* x, const float* scale,
                    const float* shift, const float* mean, const float* invstd, const float* coef, float* partials,
                    void* dx, int B, int H, int W, int C, int RH, int RW, int SH, int SW, int PH, int PW, void* stream);
int mpr_global_avgpool_fwd(const void* x, float* y, int B, int L, int C, void* stream);
int mpr_global_avgpool_bwd(const float* dy, void* dx, int B, int L, int C, void* stream);
int mpr_global_maxpool_fwd(const void* x, float* y, int* idx, int B, int L, int C, void* stream);
int mpr_global_maxpool_bwd(const float* dy, const int* idx, void* dx, int B, int L, int C, void* stream);

/* ---- fp32 PARITY path of the conv stacks (conv_f32.hip; trainer / card `precision: 32`, as Lightning's flag in
 *      scripts/train_multi.py:99-104 selects fp32 in the reference): fp32 channels-last feature maps, fp32 filters addressed
 *      through element strides (sk, sc, sr, ss) of their logical [K,C,R,S] shape, every product on the exact-fp32 MFMA,
 *      statistics in double, no atomics (bitwise reproducible).  Same reference call sites as the bf16 entry points above:
 *      nn.Conv2d / nn.Conv1d, nn.BatchNorm, nn.MaxPool, global pooling of timm's ResNet (src/image_encoder.py:16,24) and of
 *      ProfileCNN / _BasicBlock (src/profile_encoder.py:111-240). */
int mpr_f32_conv_fwd(const float* x, const float* w, long long sk, long long sc, long long sr, long long ss, float* y, int B,
                     int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, void* stream);
int mpr_f32_conv_dgrad(const float* dy, const float* w, long long sk, long long sc, long long sr, long long ss, float* dx,
                       const float* add /* may be NULL */, int B, int H, int W, int C, int K, int R, int S, int sh, int sw,
                       int ph, int pw, void* stream);
/* floats of scratch mpr_f32_conv_wgrad needs for this geometry (partial tiles of the split over pixels; 0: none) */
long long mpr_f32_conv_wgrad_scratch_floats(int B, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw);
/* dw (strides as the filter's) = or += the weight gradient */
int mpr_f32_conv_wgrad(const float* x, const float* dy, float* dw, long long sk, long long sc, long long sr, long long ss,
                       int accumulate, float* scratch, long long scratch_floats, int B, int H, int W, int C, int K, int R,
                       int S, int sh, int sw, int ph, int pw, void* stream);
/* train-mode BatchNorm on fp32 maps [rows][C]: partial rows of 2*C DOUBLES (sum, sum of squares | sum dz, sum dz*xhat) */
int mpr_f32_bn_parts(long long rows, int C);
int mpr_f32_bn_stats(const float* x, void* partial, long long rows, int C, void* stream);
int mpr_f32_bn_finalize(const void* partial, int nparts, long long count, const float* gamma, const float* beta,
                        float* running_mean /* may be NULL */, float* running_var, float momentum, float eps, float* scale,
                        float* shift, float* mean, float* invstd, int C, void* stream);
int mpr_f32_bn_apply(const float* x, const float* scale, const float* shift, const float* residual /* may be NULL */,
                     int relu, float* y, long long rows, int C, void* stream);
/* dz = dy * (mask_y > 0) (mask_y NULL: dz = dy) */
int mpr_f32_bn_bwd_reduce(const float* dy, const float* mask_y, const float* x, const float* mean, const float* invstd,
                          void* partial, long long rows, int C, void* stream);
/* count == 0: the layer ran on its running statistics (eval): dx = gamma * invstd * dz */
int mpr_f32_bn_bwd_finalize(const void* partial, int nparts, long long count, const float* gamma, const float* mean,
                            const float* invstd, float* dgamma, float* dbeta, int accumulate, float* coef /* [3][C] */,
                            int C, void* stream);
int mpr_f32_bn_bwd_apply(const float* dy, const float* mask_y, const float* x, const float* coef, float* dx,
                         float* dz_out /* may be NULL */, long long rows, int C, void* stream);
/* idx: int32 linear input position ih*W+iw of the first maximum in torch's scan order */
int mpr_f32_maxpool_fwd(const float* x, float* y, int* idx, int B, int H, int W, int C, int RH, int RW, int SH, int SW, int PH,
                        int PW, void* stream);
int mpr_f32_maxpool_bwd(const float* dy, const int* idx, float* dx, int B, int H, int W, int C, int RH, int RW, int SH, int SW,
                        int PH, int PW, void* stream);
/* [B][L][C] -> [B][C]; mode 0 = mean, 1 = max (+ idx: first arg-max position) */
int mpr_f32_global_pool_fwd(const float* x, float* y, int* idx /* mode 1 */, int B, int L, int C, int mode, void* stream);
int mpr_f32_global_pool_bwd(const float* dy, const int* idx, float* dx, int B, int L, int C, int mode, void* stream);

/* ---- exact-fp32 batched GEMM: C = alpha*op(A)*op(B) (+bias) + beta*C --------------------------
 * nn.Linear(bias=False) projections (src/model.py:31-32,40-41,80-82), classifier heads
 * (src/model.py:164,316), similarity matrix and its gradient products (src/coordination.py:38,89). */
int mpr_gemm_f32(const float* A, const float* B, float* C, const float* bias /* [N] or NULL */, int M, int N, int K,
                 int lda, int ldb, int ldc, int transA, int transB, float alpha, float beta, int batch,
                 long long strideA, long long strideB, long long strideC, void* stream);

int mpr_gemm_f32_b2(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int transA,
                    int transB, float alpha, float beta, int outer, int inner, long long sAo, long long sAi,
                    long long sBo, long long sBi, long long sCo, long long sCi, void* stream);

/* ---- transformer encoder pieces, fp32 (ProfileTransformer: nn.TransformerEncoderLayer at
 *      src/profile_encoder.py:22-30,57-68; timm VisionTransformer blocks behind src/image_encoder.py:16,24).
 *      The GEMMs run on mpr_gemm_f32 / mpr_gemm_f32_b2. */
int mpr_add_layernorm_fwd(const float* x, const float* residual /* may be NULL */, const float* gamma,
                          const float* beta, float eps, float* y, float* sum_out /* x + residual, may be NULL */,
                          float* mean, float* rstd, int rows, int D, void* stream);
int mpr_layernorm_bwd_workspace_floats(int rows, int D);
int mpr_layernorm_bwd(const float* dy, const float* s, const float* gamma, const float* mean, const float* rstd,
                      const float* dskip /* may be NULL */, float* ds, float* dgamma, float* dbeta, float* workspace,
                      int accumulate, int rows, int D, void* stream);
int mpr_masked_softmax_fwd(float* S /* [batch*heads*Tq][T], in place */, const void* key_padding_mask /* [batch][T] bytes or NULL */,
                           float scale, int batch, int heads, int Tq, int T, void* stream);
int mpr_softmax_bwd(float* dP /* in place -> dS */, const float* P, float scale, int rows, int T, void* stream);
/* act: 0 none, 1 exact GELU, 2 ReLU, 3 SiLU, 4 sigmoid; optional inverted dropout (mask: 1 byte / element) */
int mpr_bias_act_fwd(const float* x, const float* bias /* [D] or NULL */, int act, float p_drop, unsigned seed, float* y,
                     void* mask, long long n, int D, void* stream);
int mpr_bias_act_bwd(const float* dy, const float* x, const float* bias, int act, float p_drop, const void* mask,
                     float* dx, long long n, int D, void* stream);
int mpr_embedding_add_fwd(const float* x, const float* table, const long long* index, float* y, int rows, int D,
                          void* stream);
int mpr_embedding_bwd(const float* dy, const long long* index, float* dtable, int table_rows, int rows, int D,
                      long long padding_idx, void* stream);
int mpr_add_f32(const float* a, const float* b, float* y, long long n, void* stream);

/* ---- transformer encoder pieces, mixed precision (trainer precision 'bf16-mixed' / '16-mixed': fp32 residual
 *      stream, bf16 GEMM operands; same reference call sites as the fp32 block above).  The linears of a block run on
 *      mpr_conv_fwd / mpr_conv_dgrad / mpr_conv_wgrad as 1x1 convolutions; these entry points are what sits between
 *      them.  Dropout masks are not stored: element i is kept iff hash(seed, i) >= p (regenerated in backward). */
/* s = x + drop(r + rbias) [-> s_out];  y = LayerNorm(s) * gamma + beta -> y32 and / or y16 (gamma NULL: add only).
 * x, s_out, y32 fp32 [rows][D]; r, y16 bf16; D % 4 == 0, D <= 2048 */
int mpr_tf_add_ln_fwd(const float* x /* NULL: zeros */, const void* r /* may be NULL */, const float* rbias /* may be NULL */, float p_drop,
                      unsigned seed, const float* gamma /* may be NULL */, const float* beta, float eps,
                      float* s_out /* may be NULL */, float* y32 /* may be NULL */, void* y16 /* may be NULL */,
                      float* mean, float* rstd, int rows, int D, void* stream);
int mpr_tf_ln_bwd_workspace_floats(int rows, int D);
int mpr_tf_set_ln_bwd_rows(int rows);   /* rows per workgroup of mpr_tf_ln_bwd (tuning knob, default 32); returns the previous value */
/* ds = dLN/ds of the gradient (dy16 bf16 [+ dy32 fp32]) (+ dskip); dgamma / dbeta assigned or accumulated; D <= 1024 */
int mpr_tf_ln_bwd(const void* dy16 /* may be NULL */, const float* dy32 /* may be NULL */, const float* s, const float* gamma,
                  const float* mean, const float* rstd, const float* dskip /* may be NULL */, float* ds, float* dgamma,
                  float* dbeta, float* workspace, int accumulate, int rows, int D, void* stream);
/* y = drop(act(x + bias)), bf16 -> bf16 [rows][D], D % 8 == 0; act: 0 none, 1 exact GELU, 2 ReLU, 3 SiLU, 4 sigmoid */
int mpr_tf_bias_act_fwd(const void* x, const float* bias /* may be NULL */, int act, float p_drop, unsigned seed, void* y,
                        long long rows, int D, void* stream);
/* backward elementwise passes fused with the column sums of a bias gradient (added into dbias):
 * mode 0: dbias += colsum(dy bf16);  mode 1: dx = dy * dropmask/(1-p) * act'(x + bias) (bf16), dbias += colsum(dx);
 * mode 2: dx = bf16(dy fp32 * dropmask/(1-p)), dbias += colsum(dx).  dbias may be NULL in modes 1 and 2. */
int mpr_tf_ew_bwd_workspace_floats(int rows, int D);
int mpr_tf_ew_bwd(int mode, const void* dy, const void* x /* mode 1 */, const float* bias, int act, float p_drop,
                  unsigned seed, void* dx, float* dbias, float* workspace /* needed iff dbias */, int rows, int D,
                  void* stream);
/* fused multi-head self-attention on a packed bf16 qkv [B][T][3*heads*head_dim] (torch MHA / timm layout) whose
 * in-projection bias is added while the operands are loaded; key-padding mask [B][T] bytes; head_dim 64 with T <= 256
 * or head_dim 32 with T <= 288 (mpr_attn_supported).
 * out bf16 [B][T][heads*head_dim]; lse fp32 [B*heads][T] is kept for backward; delta: [B*heads][T] scratch */
int mpr_attn_supported(int T, int head_dim);
int mpr_attn_fwd(const void* qkv, const float* bias /* may be NULL */, const void* key_padding_mask /* may be NULL */,
                 void* out, float* lse, int B, int T, int heads, int head_dim, float scale, float p_drop, unsigned seed,
                 void* stream);
int mpr_attn_bwd(const void* qkv, const float* bias, const void* key_padding_mask, const void* out, const void* dout,
                 const float* lse, float* delta, void* dqkv, int B, int T, int heads, int head_dim, float scale,
                 float p_drop, unsigned seed, void* stream);
/* fp32 -> bf16 (to_bf16 != 0) or bf16 -> fp32; n % 4 == 0 */
int mpr_tf_cast(const void* x, void* y, long long n, int to_bf16, void* stream);

/* ---- coordination losses (src/coordination.py:17-112) ----------------------------------------- */
int mpr_loss_workspace_floats(void);
int mpr_l2norm_fwd(const float* x, float* u, float* inv_norm, int rows, int D, void* stream);
int mpr_l2norm_bwd(const float* du, const float* u, const float* inv_norm, const float* x /* may be NULL */,
                   const float* other, float mse_coef, const float* gout /* [1] or NULL */, float* dx, int rows,
                   int D, void* stream);
int mpr_clip_fwd(const float* S, const float* logit_scale, float* row_lse, float* col_lse, float* diag, float* loss,
                 int buckets, int n, void* stream);
int mpr_clip_bwd(float* S, const float* logit_scale, const float* row_lse, const float* col_lse, const float* gout,
                 float* d_logit_scale, float* workspace, int buckets, int n, void* stream);
/* data-parallel row block of the CLIP loss (the path's one exchange step: the embeddings and two LSE
 * vectors are all-gathered over RCCL by the caller; extension of src/coordination.py:26-47 to a batch
 * sharded over ranks): S_blk [rows][ncols], positives at column diag_off + i */
int mpr_clip_block_fwd(const float* S, const float* logit_scale, float* row_lse, float* diag, float* sum_out,
                       float* workspace, int rows, int ncols, int diag_off, void* stream);
int mpr_clip_block_bwd(float* S, const float* logit_scale, const float* lse_own, const float* lse_other,
                       const float* gout /* [1] or NULL */, float coef, float* d_logit_scale_part, float* workspace,
                       int rows, int ncols, int diag_off, void* stream);
/* data-parallel row block of the SigLIP loss: unnormalised sum of -logsigmoid over the block; gradient scaled by coef */
int mpr_siglip_block_fwd(const float* S, const float* logit_scale, const float* bias, float* sum_out, float* workspace,
                         int rows, int ncols, int diag_off, void* stream);
int mpr_siglip_block_bwd(float* S, const float* logit_scale, const float* bias, float coef,
                         float* d_logit_scale_part /* may be NULL */, float* d_bias_part /* may be NULL */,
                         float* workspace, int rows, int ncols, int diag_off, void* stream);
int mpr_sqdiff_sum(const float* a, const float* b, float* out /* [1] = sum (a-b)^2 */, float* workspace, long long total,
                   void* stream);
int mpr_siglip_fwd(const float* S, const float* logit_scale, const float* bias, float* loss, float* workspace,
                   int buckets, int n, void* stream);
int mpr_siglip_bwd(float* S, const float* logit_scale, const float* bias, const float* gout, float* d_logit_scale,
                   float* d_bias, float* workspace, int buckets, int n, void* stream);
int mpr_mse_add(const float* a, const float* b, float beta, float* loss, float* workspace, long long total,
                void* stream);
/* CLIP / SigLIP without the similarity matrix in memory (csrc/loss_fused.hip; replaces src/coordination.py:33-47 and
 * :81-95 -- normalise, all-pairs logits, both softmax axes, their gradients -- for one process AND for the data-parallel
 * row block).  gathered: normalised embeddings of every rank [world][2][b][D] (world == 1: [2][buckets * b][D], the
 * output of mpr_clipf_norm); bias == NULL selects CLIP.  Shapes and the meaning of lse / out / coef: loss_fused.hip. */
long long mpr_clipf_workspace_floats(int world, int b, int D, int buckets);
int mpr_clipf_norm(const float* image_emb, const float* profile_emb, float* uv /* [2][rows][D] */,
                   float* inv_norm /* [2][rows] */, int rows, int D, void* stream);
int mpr_clipf_fwd(const float* gathered, const float* logit_scale, const float* bias /* NULL: CLIP */,
                  float* lse /* [2][buckets * b]; SigLIP: may be NULL */, float* out /* [1] */, float mul, float* workspace,
                  int world, int rank, int b, int D, int buckets, void* stream);
int mpr_clipf_bwd(const float* gathered, const float* logit_scale, const float* bias, const float* lse_own,
                  const float* lse_other /* [world][2][b] */, float coef, const float* uv, const float* inv_norm,
                  const float* image_emb /* may be NULL */, const float* profile_emb, float mse_coef,
                  const float* gout /* [1] or NULL */, float* d_image, float* d_profile,
                  float* d_logit_scale /* may be NULL */, float* d_bias /* may be NULL */, float* workspace, int world,
                  int rank, int b, int D, int buckets, void* stream);
/* RankLoss (src/coordination.py:115-135) on the raw cosine matrix S [n][n] (diagonal negated on the fly):
 * row / column sums, loss = (mean relu(margin + colsum) + mean relu(margin + rowsum)) / 2, and dL/dS */
int mpr_rank_fwd(const float* S, float margin, float* row_sum, float* col_sum, float* loss, int n, void* stream);
int mpr_rank_bwd(float* G /* [n][n] out */, const float* row_sum, const float* col_sum, float margin,
                 const float* gout /* [1] or NULL */, int n, void* stream);

/* ---- ProfileLSTM (src/profile_encoder.py:71-108): pointwise cell of nn.LSTM, gate order i, f, g, o; the input and
 *      recurrent projections run on mpr_gemm_f32 (time-major: step t is a contiguous [B][4d] slice) */
int mpr_lstm_cell_fwd(const float* G /* [B][4d] x W_ih^T + b_ih + h_prev W_hh^T */, const float* b_hh,
                      const float* c_prev /* NULL: zeros */, float* act /* [B][4d] gate activations, kept for backward */,
                      float* c, float* h, int B, int d, void* stream);
int mpr_lstm_cell_bwd(const float* act, const float* c_prev /* NULL: zeros */, const float* c, const float* dh_a,
                      const float* dh_b /* may be NULL */, float* dc /* in: d c_t from step t+1, out: d c_{t-1} */,
                      float* dG /* [B][4d] */, int B, int d, void* stream);

/* ---- EfficientNet pieces (timm efficientnet_b0 behind src/image_encoder.py:16,24; model_cards/example_multi.yaml:9):
 *      depthwise convolution (nn.Conv2d(groups=C)) on channels-last bf16 with the torch [C][1][R][S] fp32 filter, and the
 *      squeeze-excite gate; 1x1 convs run on mpr_conv_*, BatchNorm on mpr_bn_*, SiLU on mpr_tf_bias_act_fwd / mpr_tf_ew_bwd */
/* wt: C*R*S floats of scratch (the filter is re-laid tap-major there first) */
int mpr_dwconv_fwd(const void* x, const float* w, float* wt, void* y, int B, int H, int W, int C, int R, int S, int sh, int sw,
                   int ph, int pw, void* stream);
int mpr_dwconv_dgrad(const void* dy, const float* w, float* wt, void* dx, int B, int H, int W, int C, int R, int S, int sh,
                     int sw, int ph, int pw, void* stream);
long long mpr_dwconv_wgrad_workspace_floats(int B, int P, int Q, int C, int R, int S);
int mpr_dwconv_wgrad(const void* x, const void* dy, float* dw /* [C][1][R][S] */, float* workspace, int accumulate, int B,
                     int H, int W, int C, int R, int S, int sh, int sw, int ph, int pw, void* stream);
int mpr_se_scale(const void* x /* [B][L][C] bf16 */, const float* gate /* [B][C] */, void* y, int B, int L, int C,
                 void* stream);
int mpr_se_dgate(const void* x, const void* dy, float* dgate /* [B][C] = sum_l dy * x */, int B, int L, int C, void* stream);
int mpr_se_pool(const void* x, float* pooled /* [B][C] = mean_l x */, int B, int L, int C, void* stream);
/* the SE bottleneck on the pooled map in one launch forward / two backward (se_mlp.hip; rd <= mpr_se_mlp_max_rd()):
 *   z1 = pooled W1^T + b1, r = silu(z1), gate = sigmoid(r W2^T + b2)      w1: [rd][C], w2: [C][rd] (the 1x1 conv filters)
 * backward from dgate = d loss / d gate: dz2, dz1 (scratch, [B][C] / [B][rd]), dpooled = dpooled_scale * dz1 W1, and the
 * parameter gradients ACCUMULATED into dw1 / db1 / dw2 / db2 (fp32 atomics: zero them first) */
int mpr_se_mlp_max_rd(void);
int mpr_se_mlp_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* z1, float* r,
                   float* gate, int B, int C, int rd, void* stream);
int mpr_se_mlp_bwd(const float* dgate, const float* gate, const float* z1, const float* r, const float* pooled, const float* w1,
                   const float* w2, float* dz2, float* dz1, float* dpooled, float* dw1, float* db1, float* dw2, float* db2,
                   float dpooled_scale, int B, int C, int rd, void* stream);
/* y = x * gate[b][c] + add[b][c] (the gated map's data gradient with the pooled path's per-image constant folded in) */
int mpr_se_scale_add(const void* x, const float* gate, const float* add, void* y, int B, int L, int C, void* stream);

/* ---- input pipeline, the per-step random part on pre-decoded batches (src/data.py:73-91,124-141,198-204): the random
 *      decisions (crop offsets, flips) are INPUTS (int32 / byte vectors of length B) */
int mpr_aug_image(const void* src_u8 /* [B][S][S] gray bytes, S = ceil(1.05 T) */, const int* top, const int* left,
                  const void* vflip, const void* hflip, float* dst /* [B][1][T][T] in [-1, 1] */, int B, int S, int T,
                  void* stream);
int mpr_aug_profile(const float* raw /* [B][Lmax][C] raw counts, zero-padded */, const int* length, const int* left,
                    const void* reverse, const float* ceiling /* [C] log-scale ceilings */, float* dst /* [B][T][C] */,
                    int B, int Lmax, int C, int S, int T, float sigma, unsigned seed, void* stream);

/* ---- few-shot evaluation: exact k-nearest neighbours + weighted vote (src/ann.py:6-34 as driven by
 *      scripts/benchmark_cross.py:24-96; the reference's approximate NN-descent index is replaced by exact search) */
int mpr_knn_sqnorm(const float* X, float* out /* [rows] */, int rows, int D, void* stream);
/* dots [nq][ng] = X G^T (mpr_gemm_f32; overwritten); metric 0 euclidean, 1 cosine; idx int64 / dist fp32 [nq][k],
 * ascending (distance, index); reported distances are recomputed directly from X and G (exact zeros) */
int mpr_knn_select(float* dots, const float* q_sqnorm, const float* g_sqnorm, const float* X, const float* G, int metric,
                   int nq, int ng, int k, int D, long long* idx, float* dist, void* stream);
/* weights 1/dist (rows containing a zero distance: indicator of the zeros, src/ann.py:28-34), class of largest summed
 * weight, ties to the smallest class id (sklearn weighted_mode) */
int mpr_knn_vote(const long long* idx, const float* dist, const long long* labels /* [gallery] */, int nq, int m,
                 long long* pred, void* stream);

/* ---- optimiser, encoder tail, classifier loss --------------------------------------------------
 * optim.SGD over all parameters (src/model.py:147-148); metadata concat + dropout
 * (src/image_encoder.py:25-29, src/profile_encoder.py:234-240); CrossEntropyLoss + argmax
 * (src/model.py:167,227). */
int mpr_sgd_multi(const void* table /* device {float* p; const float* g; float* m; int64 n}[ntensors] */,
                  int ntensors, long long max_numel, float lr, float momentum, float dampening, float weight_decay,
                  int nesterov, int first_step, void* stream);
int mpr_tail_fwd(const float* feat, const long long* meta, float inv_denom, float p_drop, unsigned seed, float* out,
                 void* mask /* 1 byte / element, needed iff p_drop > 0 */, int B, int F, int Mm, void* stream);
int mpr_tail_bwd(const float* dout, const void* mask, float p_drop, float* dfeat, int B, int F, int Mm, void* stream);
int mpr_softmax_ce(const float* logits, const long long* labels /* may be NULL: argmax only */, float* row_loss,
                   float* loss, long long* argmax, float* dlogits /* may be NULL */, int rows, int C, void* stream);
int mpr_scale_by_scalar(const float* x, const float* s, float* y, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MPR_HIP_H */
