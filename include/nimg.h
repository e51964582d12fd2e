/*
 * nimg.h - C ABI of libnimg.so: MI355X (gfx950) HIP kernels for the neural-imaging channel hot path
 *          RAW -> UNet ISP -> manipulations -> differentiable JPEG -> FAN classifier, forward + backward.
 *
 * The reference (pkorus/neural-imaging, TF 2.1, pure Python) has NO FFI: its seam is the Python duck type
 * TFModel (models/tfmodel.py:86-294).  This header is the native boundary underneath the build's Python mirror of
 * that surface (neural-imaging_amd/models/ *.py): every entry point replaces one TensorFlow op call site of the
 * reference, cited per function.  INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain C types only; every buffer is a CALLER-ALLOCATED DEVICE pointer (no hidden hipMalloc)
 *   - tensors are NHWC float32 unless stated; weights in the Keras layouts the reference's checkpoints use:
 *     Conv2D (kh,kw,Cin,Cout), Conv2DTranspose (kh,kw,Cout,Cin), Dense (in,out)
 *   - asynchronous on `stream` (a hipStream_t passed as void*), no internal synchronisation, capturable in a hipGraph
 *   - returns NIMG_OK (0) or a negative NIMG_ERR_* code; never throws, never exits
 *   - re-entrant: no mutable global state (Q tables etc. are passed per call)
 */
#ifndef NIMG_H
#define NIMG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NIMG_OK 0
#define NIMG_ERR_ARG (-1)     /* invalid shape / null pointer / unsupported configuration */
#define NIMG_ERR_LAUNCH (-2)  /* HIP launch failure (hipGetLastError) */
#define NIMG_ERR_WORKSPACE (-3) /* workspace too small */

/* tensor-storage flags of the *_ex throughput-mode entry points: the named tensor holds bf16 instead of float32 */
#define NIMG_BF16_IN 1    /* in1 AND in2 (convolutions: c1 % 8 == 0, c2 % 8 == 0) / dp (un-pool) */
#define NIMG_BF16_OUT 2   /* out1 AND out2 (o1 % 4 == 0, o2 % 4 == 0) / pool_out / dz of the un-pool */
#define NIMG_BF16_MASK 4  /* act_mask */
#define NIMG_BF16_DZ 8    /* dz of a weight gradient (cout % 8 == 0) */
#define NIMG_D2S_OUT 16   /* 3x3 stride-1 convolutions (o1 % 16 == 0, no out2): out1 - and the act_mask / residual / bf16 copy that
                           * share its indexing - is the depth_to_space(2) image (n, 2 hout, 2 wout, o1 / 4) of the result, the
                           * tf.nn.depth_to_space layout (block 2 dy + dx of pixel (y, x) -> pixel (2y + dy, 2x + dx)):
                           * models/compression.py:233,245,249 forward, and the input gradient of a stride-2 layer computed
                           * over its space-to-depth image (nimg_s2d_conv_weights) */
#define NIMG_MASK_CONV 128 /* with NIMG_D2S_OUT: act_mask keeps the convolution's own (n, hout, wout, o1) layout (the activation it masks
                            * by was itself stored as a space-to-depth image) instead of the depth-to-space layout of out1 */
#define NIMG_COPY_LRELU 64 /* nimg_conv2d_fwd_bf16_res: the bf16 copy holds LeakyReLU(alpha) of the (activation-free) result */
#define NIMG_S2D_OUT 32   /* 3x3 stride-1 convolutions (o1 % 4 == 0, even hout / wout, no out2): out1 (and the bf16 copy) is the
                           * space_to_depth(2) image (n, hout / 2, wout / 2, 4 o1) of the result - the gradient of a depth_to_space
                           * layer, written by the input-gradient pass that produces it (models/compression.py:233,245 backward);
                           * act_mask and residual keep the convolution's own (n, hout, wout, o1) layout */
#define NIMG_UNPOOL_OUT 512 /* internal to nimg_conv2d_dgrad_unpool_out_bf16: the result is routed through a 2x2 max-pool's backward */
#define NIMG_D2S_CONVT 1024 /* internal to nimg_convt2x2_fwd_bf16_ex: the transposed convolution as ONE 1x1 product with 4 cout columns, phase
                              3 - b = channel block b written to pixel (2 y + dy, 2 x + dx); the bias repeats per block */
#define NIMG_POOL_ALSO 256 /* internal to nimg_conv2d_fwd_pool_also_bf16: the pooled tensor is written next to out1, not instead of it */

/* library / ABI version, bumped on any change of an existing entry point's signature or data layout (3: nimg_conv2d_fwd_bf16_res
 * gained out_bf16_copy and stride; 5: the arg-max of nimg_conv1_pool_fwd_c4 / nimg_conv1_wgrad_c4 / nimg_conv1_dgrad_pooled is 2 bits
 * per value).  A binding compares nimg_abi_version() with the NIMG_ABI_VERSION it was written against. */
#define NIMG_ABI_VERSION 5
int nimg_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------------
 * Differentiable JPEG - replaces DifferentiableJPEG.call, models/jpeg.py:91-159, and the Quantization layer it uses,
 * models/layers.py:118-134.  One fused kernel per direction (the reference runs ~25 TF ops).
 * rounding modes = Quantization.rounding (models/layers.py:97): */
#define NIMG_ROUND_ROUND 0    /* tf.round, no gradient            (layers.py:122-123) */
#define NIMG_ROUND_SOFT 1     /* fwd round, bwd 1-cos(2 pi x)     (layers.py:126-128)  <- dJPEG default 'soft' */
#define NIMG_ROUND_SIN 2      /* x - sin(2 pi x)/(2 pi)           (layers.py:125)     */
#define NIMG_ROUND_HARMONIC 3 /* x - sin(2 pi x)/pi (taylor_terms=1; layers.py:130-134) */
#define NIMG_ROUND_IDENTITY 4 /* layers.py:136-137 */

/* x,y: (n,h,w,3) in [0,1]; h,w multiples of 8.   qtab: (3,8,8) float32 tables for Y,Cb,Cr (device).
 * mask (optional, may be NULL): (n,h,w) uint8, bit c set <=> channel c was NOT clipped (needed by the backward).
 * idx  (optional): (n,3,h/8,w/8,8,8) int16  rint(X/Q) quantisation indices - the bit-exact sub-contract.
 * xdq  (optional): same shape float32, dequantised coefficients = 2nd output of the reference model (jpeg.py:159). */
int nimg_djpeg_fwd(const float* x, float* y, const float* qtab, uint8_t* mask, int16_t* idx, float* xdq,
                   int n, int h, int w, int rounding, void* stream);
/* gx = d loss / d x given gy = d loss / d y.  Recomputes the forward DCT from x (no saved coefficients). */
int nimg_djpeg_bwd(const float* x, const float* gy, const uint8_t* mask, const float* qtab, float* gx,
                   int n, int h, int w, int rounding, void* stream);
/* The same backward pass plus the gradient of TRAINABLE quantisation tables (DifferentiableJPEG(trainable=True),
 * models/jpeg.py:57-62 add_weight('Q_mtx_luma' / 'Q_mtx_chroma'); gradient through X / Q ... * Q at :129-131):
 * dq (2,8,8) float32 = [d loss / d Q_luma, d loss / d Q_chroma (Cb + Cr)], (+)= when accumulate.  Deterministic: per-wave
 * partial sums in `workspace` (nimg_djpeg_dq_workspace_bytes), fixed-order reduction. */
size_t nimg_djpeg_dq_workspace_bytes(int n, int h, int w);
int nimg_djpeg_bwd_dq(const float* x, const float* gy, const uint8_t* mask, const float* qtab, float* gx, float* dq,
                      int n, int h, int w, int rounding, int accumulate, void* workspace, size_t workspace_bytes,
                      void* stream);
/* IJG quality scaling - replaces jpeg_qtable, compression/jpeg_helpers.py:264-305.  HOST function: out64 is a host
 * pointer to 64 floats (row-major 8x8).  channel 0 = luma, >0 = chroma. */
int nimg_jpeg_qtable(int quality, int channel, float* out64);

/* ------------------------------------------------------------------------------------------------------------------
 * Convolutions on the matrix cores (float32 MFMA, exact f32).  Replace tf.keras.layers.Conv2D / tf.nn.conv2d call sites:
 *   UNet  models/pipelines.py:191-192,207-208,216 ; FAN models/forensics.py:69,76 ; ConstrainedConv2D models/layers.py:56-57
 *   TwitterDCN models/compression.py:221-237,247-265.
 * in1/in2: the input may be split over two NHWC tensors with c1 + c2 channels (concat-free skip connections,
 *          pipelines.py:206,211); in2 = NULL, c2 = 0 otherwise.   w: (ks,ks,c1+c2,o1+o2) HWIO.   bias: (o1+o2) or NULL.
 * out1/out2: the output may be split the same way (o2 = 0 normally; used by the dgrad of a concat input).
 * act_mask (optional): tensor shaped like out1; out1 is multiplied by LeakyReLU'(act_mask) - fuses the previous
 *          layer's activation derivative into an input-gradient pass.
 * pad_mode: 0 zeros, 1 SYMMETRIC, 2 REFLECT (tf.pad modes folded into the tile load).  act: 0 none, 1 LeakyReLU(alpha).
 * Supported (ks,stride): (1,1) (3,1) (5,1) (2,2) (5,2).  hout/wout/pad_t/pad_l are explicit so TF's asymmetric SAME
 * padding is the caller's choice.  Input gradients = this same entry point on nimg_conv_flip_weights output. */
int nimg_conv2d_fwd(const float* in1, int c1, const float* in2, int c2, const float* w, const float* bias,
                    float* out1, int o1, float* out2, int o2, const float* act_mask, int n, int h, int wd, int ks,
                    int stride, int pad_t, int pad_l, int pad_mode, int hout, int wout, int act, float alpha,
                    void* stream);
/* wt[ks*ks-1-t][co][ci] = w[t][ci][co]: spatially flipped, channel-transposed weights for the input-gradient pass */
int nimg_conv_flip_weights(const float* w, float* wt, int ks_h, int ks_w, int cin, int cout, void* stream);
/* dw (ks,ks,c1+c2,cout) (+)= sum over pixels of in (x) dz - the weight half of tape.gradient (pipelines.py:84-88,
 * forensics.py:118-124, workflows/manipulation_classification.py:280).  Deterministic split-K through `workspace`.
 * db (optional, may be NULL): (cout) bias gradient = column sums of dz, fused into the same pass. */
size_t nimg_conv2d_wgrad_workspace_bytes(int cin, int cout, int ks_h, int ks_w, int n, int hout, int wout);
int nimg_conv2d_wgrad(const float* in1, int c1, const float* in2, int c2, const float* dz, int cout, float* dw,
                      float* db, int n, int h, int wd, int ks, int stride, int pad_t, int pad_l, int pad_mode, int hout,
                      int wout, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* db (cout) (+)= sum over npix pixels of dz (npix, cout) */
size_t nimg_bias_grad_workspace_bytes(long npix, int cout);
int nimg_bias_grad(const float* dz, float* db, long npix, int cout, int accumulate, void* workspace,
                   size_t workspace_bytes, void* stream);
/* flags: NIMG_BF16_DZ = dz is stored as bf16 (cout % 4 == 0, cout / 4 divides 256); sums in float32 */
int nimg_bias_grad_ex(const float* dz, float* db, long npix, int cout, int accumulate, void* workspace,
                      size_t workspace_bytes, int flags, void* stream);
/* Conv2DTranspose(k=2,s=2,SAME), kernel (2,2,cout,cin) - models/pipelines.py:205.  Its input gradient is
 * nimg_conv2d_fwd(ks=2,stride=2) on the same kernel, its weight gradient nimg_conv2d_wgrad(ks=2,stride=2) with the
 * roles of input and output gradient swapped (see neural-imaging_amd/models/pipelines.py). */
int nimg_convt2x2_fwd(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int cin,
                      int cout, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Pooling / layout / element-wise */
/* MaxPool2D 2x2 (pipelines.py:197 SAME, forensics.py:70 VALID; identical on even sizes).  Odd sizes follow VALID: y is
 * (n, h/2, w/2, c) with the last row / column dropped, and the backward pass leaves it a zero gradient. */
int nimg_maxpool2_fwd(const float* x, float* y, int n, int h, int w, int c, void* stream);
/* dz = (route dp to the FIRST arg-max of each window) [+ add] [* LeakyReLU'(yact)]; add may alias dz */
int nimg_maxpool2_bwd(const float* dp, const float* yact, const float* add, float* dz, int n, int h, int w, int c,
                      int apply_lrelu_mask, float alpha, void* stream);
/* The same two passes on bf16-stored tensors (UNet activations / gradients in throughput mode: pipelines.py:197 and its
 * gradient); h, w even, c % 8 == 0; every tensor argument is bf16.  Rounding is monotonic, so pooling the rounded tensor
 * equals rounding the pooled one. */
int nimg_maxpool2_fwd_bf16(const void* x, void* y, int n, int h, int w, int c, void* stream);
int nimg_maxpool2_bwd_bf16(const void* dp, const void* yact, const void* add, void* dz, int n, int h, int w, int c,
                           int apply_lrelu_mask, float alpha, void* stream);
/* Fused Conv2D(SAME, stride 1) -> [LeakyReLU] -> MaxPool2D(2) of the FAN feature extractor (models/forensics.py:69-77:
 * every conv{i} is followed by its pool; the full-resolution activation is consumed by nothing else), float32 MFMA.
 * pool_out (n,h/2,w/2,cout) receives the pooled activation, pool_idx (same shape, bytes; may be NULL for inference)
 * the position 0..3 of the first maximum inside each window (row-major).  h, w even; cout % 4 == 0; ks 3|5. */
int nimg_conv2d_pool_fwd(const float* in, int cin, const float* w, const float* bias, float* pool_out,
                         unsigned char* pool_idx, int cout, int n, int h, int wd, int ks, int act, float alpha,
                         void* stream);
/* The same pass with bf16 MFMA operands (throughput mode): w = the f32 kernel (read when cin <= 4), wb = its
 * nimg_conv_weights_bf16(mode 0) image (read otherwise; cin % 8 == 0). */
int nimg_conv2d_pool_fwd_bf16(const float* in, int cin, const float* w, const void* wb, const float* bias,
                              float* pool_out, unsigned char* pool_idx, int cout, int n, int h, int wd, int ks, int act,
                              float alpha, void* stream);
/* Backward of that epilogue: dz (n,2ho,2wo,c) = dp routed to the stored arg-max [* LeakyReLU'(pooled)], zero elsewhere
 * (sign(window max) == sign(pooled), so the un-stored activation is not needed).  c % 4 == 0. */
int nimg_maxpool2_unpool(const float* dp, const unsigned char* idx, const float* pooled, float* dz, int n, int ho, int wo,
                         int c, int apply_lrelu_mask, float alpha, void* stream);
/* y = [clip01](scale * depth_to_space_DCR(x, 2) + shift); x (n,h,w,4*cout) -> y (n,2h,2w,cout).
 * pipelines.py:218-223 (scale 1, shift 0, clip 1), compression.py:248,263,266-271.  The clip is straight-through:
 * the backward is dx = scale * space_to_depth(dy). */
int nimg_d2s_clip_fwd(const float* x, float* y, int n, int h, int w, int cout, float scale, float shift, int clip,
                      void* stream);
int nimg_d2s_clip_bwd(const float* dy, float* dx, int n, int h, int w, int cout, float scale, void* stream);
int nimg_lrelu_bwd(const float* dy, const float* yact, float* dz, long count, float alpha, void* stream);
int nimg_add(const float* a, const float* b, float* out, long count, void* stream);
/* out = sum of 2..6 float32 tensors of `count` elements (count % 4 == 0; `inputs` is a HOST array of device pointers; out may
 * alias an input): the per-manipulation gradients of workflows/manipulation_classification.py:199-208 summed in one pass. */
int nimg_add_n(const float* const* inputs, int n_inputs, float* out, long count, void* stream);
/* avg_pool downsampling of the channel, workflows/manipulation_classification.py:235 */
int nimg_avgpool_fwd(const float* x, float* y, int n, int h, int w, int c, int factor, void* stream);
int nimg_avgpool_bwd(const float* dy, float* dx, int n, int h, int w, int c, int factor, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Losses, classifier head, optimiser */
/* helpers/tf_helpers.py:31-32  loss = mean((255a - 255b)^2); grad_a (+)= grad_scale * dloss/da (grad_a may be NULL) */
size_t nimg_mse255_workspace_bytes(void);
int nimg_mse255(const float* a, const float* b, float* loss, float* grad_a, long count, float grad_scale,
                int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* helpers/tf_helpers.py:35-36  loss = mean |255a - 255b|; grad_a (+)= grad_scale * 255 sign(a - b) / count (NIP --loss L1) */
size_t nimg_mae255_workspace_bytes(void);
int nimg_mae255(const float* a, const float* b, float* loss, float* grad_a, long count, float grad_scale,
                int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* helpers/tf_helpers.py:39-40  loss = mean_n 255 (1 - tf.image.ssim(y, t, max_val)_n)  (NIP --loss SSIM); y, t (n,h,w,c),
 * h, w >= 11; gauss_win = the 121 window weights (device); grad_y (may be NULL) (+)= grad_scale * dloss/dy. */
size_t nimg_ssim_loss_workspace_bytes(int n, int h, int w, int c, int with_grad);
int nimg_ssim_loss(const float* y, const float* t, float* loss, float* grad_y, int n, int h, int w, int c,
                   float max_val, const float* gauss_win, float grad_scale, int accumulate, void* workspace,
                   size_t workspace_bytes, void* stream);
/* SSIM per image (mean over channels and VALID window positions), a/b (n,h,w,c) in [0,max_val], out (n).
 *   mode 0 = skimage.metrics.structural_similarity(multichannel=True, data_range=max_val) as helpers/metrics.py:9-25 uses
 *            it (7x7 uniform window, sample covariance);   mode 1 = tf.image.ssim(max_val) as models/compression.py:89
 *            uses it (11x11 Gaussian window sigma 1.5 passed by the caller in gauss_win[121], population moments). */
/* The workflow's hand-over to the UNet backward in one pass (workflows/manipulation_classification.py:265-283 under the tape:
 * d loss / d Y = the manipulations' input gradients + lambda_nip * d mse255(Y, target) / d Y, then the gradient of
 * depth_to_space + clip, models/pipelines.py:203-205): dz (n,h,w,12) = space_to_depth of [parts[0] + ... + parts[n_parts-1] +
 * grad_scale * d mse255(y, target) / d y] with parts / y / target (n,2h,2w,3), 1 <= n_parts <= 6; loss (1) = mse255(y, target).
 * Same additions in the same order as nimg_add_n -> nimg_mse255(accumulate) -> nimg_d2s_clip_bwd(scale 1): same bits.
 * workspace: nimg_mse255_workspace_bytes(). */
int nimg_mse255_sum_s2d3(const float* const* parts, int n_parts, const float* y, const float* target, float* loss, float* dz,
                         int n, int h, int w, float grad_scale, void* workspace, size_t workspace_bytes, void* stream);
size_t nimg_ssim_workspace_bytes(int n);
int nimg_ssim(const float* a, const float* b, float* out, int n, int h, int w, int c, int mode, float max_val,
              const float* gauss_win, void* workspace, size_t workspace_bytes, void* stream);
/* FAN head, models/forensics.py:80-94: GAP -> Dense(k, softmax) -> SparseCategoricalCrossentropy on probabilities
 * (Keras eager semantics: clip to [1e-7, 1-1e-7], renormalise).  act (n,hw,c) is the 1x1-conv output AFTER LeakyReLU.
 * labels may be NULL (inference: only gap/probs are written).  loss_scale = 1/batch (mean reduction).  k <= 256 classes (the
 * reference's own bound, models/forensics.py:37): one lane sums up to 16 classes, from 17 on the wave's lanes own the classes. */
int nimg_fan_head_fwd(const float* act, const float* w, const float* b, const int* labels, float* gap, float* probs,
                      float* loss_per, float* dlogits, int n, int hw, int c, int k, float loss_scale, void* stream);
/* dact = gradient wrt the 1x1 conv PRE-activation (LeakyReLU' fused), dw (c,k), db (k), loss (1) */
int nimg_fan_head_bwd(const float* act, const float* gap, const float* w, const float* dlogits,
                      const float* loss_per, float* dact, float* dw, float* db, float* loss, int n, int hw, int c,
                      int k, float loss_scale, float alpha, void* stream);
/* tf.keras.optimizers.Adam over a flat buffer: theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps); g is pre-scaled
 * by grad_scale (e.g. 1/world_size after a sum all-reduce).  step is the 1-based iteration count.  skip_flag (optional
 * device int): when non-zero the update is skipped - the device-side form of the NaN guard at workflows/...:281-283. */
int nimg_adam_step(float* params, const float* grads, float* m, float* v, long count, float lr, float beta1,
                   float beta2, float eps, int step, float grad_scale, const int* skip_flag, void* stream);
/* The same update with the bias-corrected rate lr*sqrt(1-b2^t)/(1-b1^t) read from DEVICE memory (one float): the form a
 * captured hipGraph replays - the host refreshes *lr_t before every replay, the launch itself never changes. */
int nimg_adam_step_dev(float* params, const float* grads, float* m, float* v, long count, const float* lr_t, float beta1,
                       float beta2, float eps, float grad_scale, const int* skip_flag, void* stream);
/* flag[0] |= 1 if any gradient is NaN (device-side version of workflows/manipulation_classification.py:281-282) */
int nimg_nan_flag(const float* g, long count, int* flag, void* stream);
/* flag / scalar bookkeeping of a training step on the library's own kernels: mode 0 dst[0 .. n) = value, mode 1 dst[i] =
 * max(dst[i], src[i]) (the "NaN seen since the last check" word);  nimg_float_fill: dst[0 .. n) = value (a captured step's
 * device-resident learning rate) */
int nimg_int_words(int* dst, const int* src, long n, int value, int mode, void* stream);
int nimg_float_fill(float* dst, long n, float value, void* stream);

/* In-kernel finish of the split-K weight-gradient sums (throughput-mode kernels: nimg_conv2d_wgrad_bf16*, nimg_conv1_wgrad_*,
 * nimg_conv_wgrad_c3k5 ...; replaces the separate fixed-order reduction launch behind tape.gradient(loss, weights),
 * workflows/manipulation_classification.py:280).  The workgroups that contribute to one tile of dw share arrival counters and
 * the last to arrive sums the tile's slabs in a fixed order (deterministic).  The counters are the CALLER's memory: bind `bytes`
 * (a multiple of 4, >= NIMG_TICKET_BYTES recommended) of device memory that is ZERO to a stream; every launch on that stream
 * may use them and leaves them zero.  One buffer per stream that launches weight gradients concurrently; a stream without a
 * binding (or buf = NULL: unbind) keeps the separate reduction launch - results differ from the ticketed ones only in the order
 * of the additions.  The binding table holds NIMG_TICKET_STREAMS streams.  Host-side call, no device work. */
#define NIMG_TICKET_BYTES (64 * 1024)
#define NIMG_TICKET_STREAMS 16
int nimg_bind_tickets(void* stream, void* buf, size_t bytes);

/* A HIP stream whose kernels run on `n_cus` compute units only (hipExtStreamCreateWithCUMask, mask = the low n_cus bits: the
 * driver deals mask bits round-robin over the 8 XCDs, so any n is balanced to within one CU per XCD).  For partitioning the chip
 * between kernel classes that only evict each other when interleaved workgroup by workgroup - the matrix-core-bound weight
 * gradients of the FAN beside the byte-bound backward pass of the image chain (workflows/manipulation_classification.py:260-285
 * under the tape).  *stream is a hipStream_t the caller owns (nimg_stream_destroy). */
int nimg_stream_create_cu_mask(int n_cus, void** stream);
int nimg_stream_destroy(void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * The FAN's head in throughput mode - Conv2D(nf, 1x1, leaky_relu) -> GlobalAveragePooling2D -> Dense(softmax),
 * models/forensics.py:76-94 - with the pooling fused into the 1x1 layer (csrc/head.hip).  hw = pixels per image (64 | 128 | 256),
 * c = channels in = channels out (64 | 128 | 256): nimg_head_fused_ok.  x (n * hw, c) bf16; wimg / wimg_t the 1x1 kernel's bf16
 * images (nimg_conv_weights_bf16 mode 0 / mode 1).
 *   nimg_head_fwd:   gap (n, c) = mean over the image of LeakyReLU(x W + b); mask (n * hw, c / 32) words, bit j of word f = the
 *                    activation of channel 32 f + j is > 0 (NULL at inference) - the activation tensor itself is not written.
 *   nimg_fan_dense_fwd / _bwd: the Dense softmax / sparse-CE classifier on a pooled feature (the two halves of nimg_fan_head_fwd /
 *                    _bwd that do not touch the activation tensor).
 *   nimg_head_dgrad: dx (n * hw, c) bf16 = ((g (x) (bit ? 1 : alpha)) W^T) * LeakyReLU'(in_mask) with g[n] = wdense dlogits[n] / hw:
 *                    the gradient at the INPUT of the 1x1 layer straight from the classifier's dlogits (n, k); in_mask bf16 or NULL.
 *   nimg_head_dact:  the gradient at the 1x1 layer's pre-activation as bf16 (n * hw, c), for its weight gradient. */
int nimg_head_fused_ok(int hw, int c);
int nimg_head_fwd(const void* x, const void* wimg, const float* bias, unsigned* mask, unsigned* mask_p, float* gap, int n, int hw,
                  int c, float alpha, void* stream);
/*   nimg_head_wgrad: weight + bias gradient of the 1x1 layer from its bf16 input x and the PIXEL-MAJOR sign words mask_p
 *                    (n, hw / 32, c): bit j of word [n][b][co] = the activation of channel co at pixel 32 b + j is > 0 (written by
 *                    nimg_head_fwd next to mask) - the gradient at the pre-activation is built inside the kernel.  dw (c, c) in the
 *                    Keras (1, 1, cin, cout) layout, db (c) or NULL; split over the images, fixed-order reduction through
 *                    `workspace` (nimg_head_wgrad_workspace_bytes). */
size_t nimg_head_wgrad_workspace_bytes(int n, int c);
int nimg_head_wgrad(const void* x, const unsigned* mask_p, const float* dlogits, const float* wdense, int k, float* dw, float* db,
                    int n, int hw, int c, float alpha, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
int nimg_head_dgrad(const unsigned* mask, const float* dlogits, const float* wdense, int k, const void* wimg_t,
                    const void* in_mask, void* dx, int n, int hw, int c, float alpha, void* stream);
int nimg_head_dact(const unsigned* mask, const float* dlogits, const float* wdense, int k, void* dact, int n, int hw, int c,
                   float alpha, void* stream);
int nimg_fan_dense_fwd(const float* gap, const float* w, const float* b, const int* labels, float* probs, float* loss_per,
                       float* dlogits, int n, int c, int k, float loss_scale, void* stream);
int nimg_fan_dense_bwd(const float* gap, const float* dlogits, const float* loss_per, float* dw, float* db, float* loss, int n,
                       int c, int k, float loss_scale, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * ConstrainedConv2D kernel re-normalisation, models/layers.py:45-53 (ks=5, channels=3, strength=100) */
int nimg_constrained_kernel_fwd(const float* kernel, float* nf, int ks, int channels, float strength, void* stream);
int nimg_constrained_kernel_bwd(const float* kernel, const float* dnf, float* dkernel, int ks, int channels,
                                float strength, void* stream);
/* backward of tf.pad(SYMMETRIC|REFLECT): fold a gradient on the padded domain (n,h+2p,w+2p,c) onto (n,h,w,c) */
int nimg_fold_pad(const float* dpad, float* dx, int n, int h, int w, int c, int pad, int pad_mode, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Photo manipulations, helpers/tf_helpers.py.  Images (n,h,w,3).  mask: (n,h,w) uint8 clip mask written by the forward
 * (bit c set <=> channel c not clipped), consumed by the backward. */
/* manipulation_gaussian :113-125 - gk25 = 5x5 normalised kernel (device), REFLECT pad */
int nimg_gaussian_fwd(const float* x, float* y, uint8_t* mask, const float* gk25, int n, int h, int w, int clip,
                      void* stream);
int nimg_gaussian_bwd(const float* dy, const uint8_t* mask, float* dx, const float* gk25, int n, int h, int w,
                      void* stream);
/* any odd k x k (k <= 31) per-channel filter with a mirrored border (pad_mode 1 SYMMETRIC, 2 REFLECT) and an optional hard
 * clip: manipulation_gaussian with kernel != 5 (:113-125, REFLECT), manipulation_sharpen(hsv=False) (:156-184, SYMMETRIC),
 * residual(hsv=False) (:127-154, REFLECT, clip 0).  taps = k*k floats (device), mask as nimg_gaussian_fwd (may be NULL). */
int nimg_dwfilter_fwd(const float* x, float* y, uint8_t* mask, const float* taps, int k, int pad_mode, int n, int h, int w,
                      int clip, void* stream);
int nimg_dwfilter_bwd(const float* dy, const uint8_t* mask, float* dx, const float* taps, int k, int pad_mode, int n, int h,
                      int w, void* stream);
/* manipulation_sharpen(hsv=True) :156-184 - gk9 = 3x3 H/V filter (device); aux_hsv (n,h,w,3) scratch kept between
 * forward and backward (filtered HSV; overwritten by the backward) */
int nimg_sharpen_fwd(const float* x, float* y, float* aux_hsv, uint8_t* mask, const float* gk9, int n, int h, int w,
                     void* stream);
int nimg_sharpen_bwd(const float* x, const float* dy, float* aux_hsv, const uint8_t* mask, float* dx,
                     const float* gk9, int n, int h, int w, void* stream);
/* banded linear operator along one spatial axis (CSR rows = output index): manipulation_resample :68-76 and
 * tf.image.resize bilinear down-sampling are two such passes; their backward uses the transposed CSR. */
int nimg_sparse_axis_apply(const float* in, float* out, const int* rowptr, const int* col, const float* val, int n,
                           int hin, int win, int c, int axis, int out_size, void* stream);

/* Building blocks of the MS-SSIM loss (helpers/tf_helpers.py:43-44 -> tf.image.ssim_multiscale, 5 scales, weights 0.0448,
 * 0.2856, 0.3001, 0.2363, 0.1333; the 2x2 average pooling between the scales is nimg_avgpool_fwd / _bwd):
 *  ssim_planes: per (image, channel) means of the SSIM map and of its contrast-structure factor (either may be NULL) over
 *               the VALID positions of the 11x11 Gaussian window, and - which_maps 1 (SSIM) | 2 (cs) - the unscaled
 *               derivative maps w.r.t. the window moments, 3 x n x (h-10) x (w-10) x c floats;
 *  msssim_combine: values (scales, n*c) = cs means of scales 0..S-2 and the SSIM mean of the last one; items (scales) = window
 *               positions per plane; loss = 255 (1 - mean prod relu(v)^w); coef (scales, n*c) = d loss / d v / items;
 *  ssim_maps_grad: grad_y (+)= grad_scale * coef[n, c] * (transposed-window gather of the maps). */
size_t nimg_ssim_planes_workspace_bytes(int n, int c);
int nimg_ssim_planes(const float* y, const float* t, int n, int h, int w, int c, float max_val, const float* gauss_win,
                     float* mean_ssim, float* mean_cs, float* maps, int which_maps, void* workspace,
                     size_t workspace_bytes, void* stream);
int nimg_msssim_combine(const float* values, const float* items, int scales, int planes, float* loss, float* coef,
                        void* stream);
int nimg_ssim_maps_grad(const float* y, const float* t, const float* maps, const float* coef, float* grad_y, int n,
                        int h, int w, int c, const float* gauss_win, float grad_scale, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * GPU-resident training-data feed (helpers/dataset.py:89-131 `Dataset.next_training_batch`, helpers/loading.py:132-211
 * `sample_patch`).  rgb: (n_images, h, w, 3) uint8, raw: (n_images, h/2, w/2, 4) uint16, both resident in HBM.
 * image_idx (b) picks the image of every batch entry; cand_xy (b, attempts, 2) are candidate (xx, yy) patch corners (even
 * numbers, the Bayer alignment of loading.py:160-161); patch = RGB patch size (even).
 *  stats:  var / mean (b, attempts) of every candidate patch / 255 (np.var, np.mean over all three channels; :166-168).
 *  select: the discard policy over each image's candidates -> chosen_xy (b, 2); mode 0 none, 1 'flat' (uniforms (b,
 *          attempts) feed its coin flip, :177-178), 2 'flat-aggressive', 3 'dark-n-textured'; max_attempts = the panic
 *          counter (:156); attempts_used (b, may be NULL) = candidates consumed.
 *  gather: x_out (b, patch/2, patch/2, 4) = raw crop / 65535, y_out (b, patch, patch, 3) = rgb crop / 255 (either may be
 *          NULL), float32 of the float64 quotient like dataset.py:124-126. */
int nimg_patch_stats(const uint8_t* rgb, int n_images, int h, int w, const int* image_idx, const int* cand_xy, int b,
                     int attempts, int patch, double* var_out, double* mean_out, void* stream);
int nimg_patch_select(const int* cand_xy, const float* uniforms, const double* var, const double* mean, int b,
                      int attempts, int max_attempts, int mode, int* chosen_xy, int* attempts_used, void* stream);
int nimg_patch_gather(const uint16_t* raw, const uint8_t* rgb, int n_images, int h, int w, const int* image_idx,
                      const int* xy, int b, int patch, float* x_out, float* y_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Learned codec (TwitterDCN, models/compression.py:197-279) specific pieces */
int nimg_affine(const float* x, float* y, long count, float a, float b, void* stream);     /* y = a*x + b (:219,:268) */
int nimg_lrelu_fwd(const float* x, float* y, long count, float alpha, void* stream);       /* tf.nn.leaky_relu (:224) */
/* Element-wise pieces of the INet / DNet pipelines (models/pipelines.py:238-345): tanh of the gamma MLP (:283) and its
 * derivative through the stored output, the straight-through clip (:287, :341; gradient = identity), and
 * tf.pad(x, [[0,0],[P,P],[P,P],[0,0]], CONSTANT | SYMMETRIC | REFLECT) (:272, :320, :338; backward = nimg_fold_pad). */
int nimg_tanh_fwd(const float* x, float* y, long count, void* stream);
int nimg_tanh_bwd(const float* dy, const float* y, float* dx, long count, void* stream);
int nimg_clip01(const float* x, float* y, long count, void* stream);
int nimg_pad2d(const float* x, float* y, int n, int h, int w, int c, int pad, int pad_mode, void* stream);
/* Element-wise pieces of ClassicISP (models/pipelines.py:416-453 `_ClassicISP.call`, models/layers.py:206-258
 * `DemosaicingLayer`).
 *  residual: y = [clip01](x - alpha[0] * f)  (layers.py:252-255; f NULL = the c_filters=() case, y = [clip01](x)); the clip
 *            is straight-through, so the backward pass is df = -alpha dy, dalpha (+)= -sum(dy f) (fixed-order sum;
 *            workspace = nimg_isp_residual_workspace_bytes() bytes of device scratch), dx = dy.
 *  sigmoid:  activation of the non-residual head (layers.py:235), backward through the stored output.
 *  gamma_ste: y = pow(clip(x, lo, hi), exponent) with a straight-through clip (pipelines.py:449-451: lo 1/255, hi 1,
 *            exponent 1/2.2): dx = dy * exponent * pow(clip(x), exponent - 1). */
int nimg_isp_residual_fwd(const float* x, const float* f, const float* alpha, float* y, long count, int clip,
                          void* stream);
long nimg_isp_residual_workspace_bytes(void);
int nimg_isp_residual_bwd(const float* dy, const float* f, const float* alpha, float* df, float* dalpha,
                          float* workspace, long count, int accumulate, void* stream);
/* Dropout of the FAN's hidden Dense layers at training time (models/forensics.py:88), forward and backward alike:
 * y = keep[i] ? x[i] * scale : 0, keep = the Bernoulli(1 - rate) mask bytes, scale = 1 / (1 - rate). */
int nimg_mask_scale(const float* x, const uint8_t* keep, float* y, long count, float scale, void* stream);
/* Input gradient of a fused 5x5 conv + pool layer (stride 1, SAME) from the POOLED gradient g (n, h/2, wd/2, cout) bf16 + arg-max
 * bytes on the 2:4 structured-sparsity matrix instruction (csrc/dgrad5s.hip): the same result as nimg_conv2d_fwd_bf16_unpool to
 * summation order.  `image` = nimg_conv5_dgrad_sparse_weights(w (5,5,cin,cout) float32) of nimg_conv5_dgrad_sparse_image_bytes
 * bytes, rebuilt whenever w changes.  out (n, h, wd, cin) float32 | bf16 (NIMG_BF16_OUT), act_mask optional (float32 | bf16 with
 * NIMG_BF16_MASK): out *= act_mask > 0 ? 1 : alpha.  cin % 32 == 0, cout % 8 == 0, even h / wd (else NIMG_ERR_ARG). */
size_t nimg_conv5_dgrad_sparse_image_bytes(int cin, int cout);
int nimg_conv5_dgrad_sparse_weights(const float* w, void* image, int cin, int cout, void* stream);
int nimg_conv5_dgrad_sparse(const void* g, const unsigned char* idx, int cout, const void* image, void* out, int cin,
                            const void* act_mask, int n, int h, int wd, float alpha, int flags, void* stream);
/* nimg_conv2d_fwd_bf16_ex for the layers of a residual block (models/compression.py:224-227, 240-243): `residual` (float32, the
 * shape of out1, optional) is added after bias, activation and mask (net + conv(a) forward, d_net + mask * dgrad backward), and
 * `out_bf16_copy` (optional) receives the same result rounded to bf16 next to the float32 out1 (the exact float32 stream for the
 * skip sum, the bf16 copy for the convolutions / weight gradients that read it; with NIMG_COPY_LRELU the copy holds
 * LeakyReLU(alpha) of the result, models/compression.py:224).  At least one of the two; float32 output with o1 % 4 == 0; the
 * residual with 3x3 / stride 1 layers only. */
int nimg_conv2d_fwd_bf16_res(const float* in1, int c1, const void* wb, const float* bias, float* out1, int o1,
                             const float* act_mask, const float* residual, void* out_bf16_copy, int n, int h, int wd, int ks,
                             int stride, int pad_t, int pad_l, int pad_mode, int hout, int wout, int act, float alpha, int flags,
                             void* stream);
/* Backward of the FAN's fused conv + LeakyReLU + MaxPool2D layers conv2..4 (models/forensics.py:73-77) straight from the POOLED
 * gradient g (bf16, already x LeakyReLU') and the arg-max bytes: the MaxPool2D routing is applied while the kernels stage their
 * tiles, so the full-resolution gradient (4x the bytes, 3/4 zeros) is never written nor re-read.  5x5, stride 1, SAME.
 *   _fwd_bf16_unpool:   input gradient = convolution of the un-pooled g with wb = nimg_conv_weights_bf16(w, mode 1); arguments as
 *                       nimg_conv2d_fwd_bf16_ex with (h, wd) the FULL resolution; flags may add NIMG_BF16_OUT / NIMG_BF16_MASK
 *   _wgrad_bf16_unpool: dw (5,5,cin,cout) / db (cout) from the layer input `in` (n,h,wd,cin) bf16 and (g, idx) */
int nimg_conv2d_fwd_bf16_unpool(const void* in_pooled, const unsigned char* in_idx, int c1, const void* wb, const float* bias,
                                float* out1, int o1, const float* act_mask, int n, int h, int w_, int ks, int pad_t, int pad_l,
                                int hout, int wout, int act, float alpha, int flags, void* stream);
int nimg_conv2d_wgrad_bf16_unpool(const void* in, int cin, const void* g, const unsigned char* idx, int cout, float* dw, float* db,
                                  int n, int h, int w_, int ks, int accumulate, void* workspace, size_t workspace_bytes,
                                  void* stream);
/* DEFERRED slab reduction (round 5): nimg_conv2d_wgrad_bf16_deferred = nimg_conv2d_wgrad_bf16_ex (idx == null) or
 * nimg_conv2d_wgrad_bf16_unpool (idx != null) with accumulate = 0, except that the reduction of the split-K partial sums is not
 * launched but described in *entry (nimg_reduce_entry_bytes() bytes of HOST memory); the workspace must stay untouched until
 * nimg_reduce_slabs_batch(entries, n, stream) has been issued on the same stream.  One batch launch runs the reductions of up to
 * nimg_reduce_batch_max() weight gradients (the entries travel as kernel arguments: nothing is copied to the device, the launch
 * can be captured); every sum is bit-identical to the per-layer reduction of the non-deferred entry points. */
int nimg_conv2d_wgrad_bf16_deferred(const void* in1, int c1, const void* in2, int c2, const void* dz, const unsigned char* idx,
                                    int cout, float* dw, float* db, int n, int h, int wd, int ks, int stride, int pad_t, int pad_l,
                                    int pad_mode, int hout, int wout, void* workspace, size_t workspace_bytes, int flags,
                                    void* entry, void* stream);
/* Chained form: the same call, which ALSO runs the reduction owed by a previous deferred / chained call on this stream (pre_entry:
 * the entry that call filled, or NULL) - in the prologue of this call's own kernel where that kernel can (the 3x3 all-taps and the
 * generic bf16 weight-gradient kernels), as a separate launch in front of it otherwise.  The previous call's workspace must stay
 * untouched until this call has been issued; the last entry of a chain goes to nimg_reduce_slabs_batch.  Sums bit-identical to the
 * per-layer reduction.  entry may alias pre_entry. */
int nimg_conv2d_wgrad_bf16_chained(const void* in1, int c1, const void* in2, int c2, const void* dz, const unsigned char* idx,
                                   int cout, float* dw, float* db, int n, int h, int wd, int ks, int stride, int pad_t, int pad_l,
                                   int pad_mode, int hout, int wout, void* workspace, size_t workspace_bytes, int flags,
                                   const void* pre_entry, void* entry, void* stream);
size_t nimg_reduce_entry_bytes(void);
int nimg_reduce_batch_max(void);
int nimg_reduce_slabs_batch(const void* entries, int n, void* stream);

/* The same 5x5, 3 -> 3 convolution with ZERO padding and bf16 matrix-core operands (float32 accumulation, float32 output): the
 * main term of the ConstrainedConv2D input gradient in throughput mode (models/layers.py:56-57 backward; w = the flipped /
 * transposed filter, nimg_conv_flip_weights).  wd % 64 == 0 (else NIMG_ERR_ARG: use nimg_cconv3). */
int nimg_conv5c3_bf16(const float* in, const float* w, float* out, int n, int h, int wd, void* stream);
/* ------------------------------------------------------------------------------------------------------------------
 * FAN front end (csrc/frontend.hip): ConstrainedConv2D, models/layers.py:56-57 (tf.pad SYMMETRIC + VALID conv2d) and the
 * first FAN convolution, models/forensics.py:69-70, as row-band kernels (float32 VALU stencil / bf16 MFMA).
 *
 * nimg_cconv3: 5x5 convolution 3 -> 3, same-size output, no bias: out[y][x][o] = sum in_pad[y+ky-2][x+kx-2][i] w[ky][kx][i][o]
 * with pad_mode 0 (zeros) | 1 (SYMMETRIC).  w = the 225 floats of a (5,5,3,3) HWIO kernel in device memory.
 *   forward:        w = the normalised filter (nimg_constrained_kernel_fwd), pad_mode 1
 *   input gradient: w = nimg_conv_flip_weights(normalised filter), pad_mode 0, then nimg_cconv3_dgrad_border adds what the
 *                   SYMMETRIC pad folds back onto the two outermost rows / columns (needs h, w >= 4).
 * out_f32 (n,h,w,3) float32 and / or out_c4 (n,h,w,4) bf16 = {o0, o1, o2, 1.0} (the 8-byte pixel of the throughput-mode
 * conv1 kernels; the constant channel carries the bias gradient through their weight-gradient GEMM); at least one. */
int nimg_cconv3(const float* in, const float* w, float* out_f32, void* out_c4, int n, int h, int w_, int pad_mode,
                void* stream);
int nimg_cconv3_dgrad_border(const float* dc, const float* nf, float* dx, int n, int h, int w_, void* stream);
/* First FAN convolution in throughput mode, models/forensics.py:69-70: Conv2D(32, 5x5, SAME) + bias + LeakyReLU(alpha) +
 * MaxPool2D(2) in one pass over the bf16 {c0,c1,c2,1} pixels of nimg_cconv3 (bf16 MFMA operands, float32 accumulation).
 * c4 (n,h,w,4) bf16; w (5,5,3,32) float32 HWIO; bias (32) or NULL; pooled (n,h/2,w/2,32) bf16 (out_bf16 = 1) or float32;
 * pool_idx (n,h/2,w/2,8) uint8 or NULL: the arg-max position inside the window (row-major, first maximum wins) as 2 bits per
 * channel - channel c in byte c >> 2, bits 2 (c & 3) .. 2 (c & 3) + 1 (the layout nimg_conv1_wgrad_c4 / nimg_conv1_dgrad_pooled read).
 * h, w even; 0 < alpha <= 1 (1 = no activation). */
int nimg_conv1_pool_fwd_c4(const void* c4, const float* w, const float* bias, void* pooled, unsigned char* pool_idx, int n,
                           int h, int w_, float alpha, int out_bf16, void* stream);
/* Weight and bias gradient of that convolution from the POOLED gradient (backward of nimg_conv1_pool_fwd_c4; the tape of
 * forensics.py:118-124 for this layer): dw (5,5,3,32) (+)= sum c (x) dz, db (32) (+)= sum dz with dz = the MaxPool2D routing of g
 * by the arg-max bytes.  c4 as above; g (n,h/2,w/2,32) bf16 (g_bf16 = 1) or float32, already multiplied by LeakyReLU';
 * db may be NULL.  workspace: nimg_conv1_wgrad_c4_workspace_bytes() bytes (one slab per workgroup, reduced in a fixed order). */
size_t nimg_conv1_wgrad_c4_workspace_bytes(void);
int nimg_conv1_wgrad_c4(const void* c4, const void* g, const unsigned char* pool_idx, float* dw, float* db, int n, int h, int w_,
                        int g_bf16, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* Input gradient of that convolution from the pooled gradient: dc (n,h,w,3) float32 = sum_{ky,kx,o} dz[y+2-ky][x+2-kx][o]
 * w[ky][kx][i][o] (bf16 MFMA operands, float32 accumulation); w = the (5,5,3,32) float32 forward kernel as stored. */
int nimg_conv1_dgrad_pooled(const void* g, const unsigned char* pool_idx, const float* w, float* dc, int n, int h, int w_,
                            int g_bf16, void* stream);
/* Classifier decisions + confusion matrix on the device - replaces the host loop of validate_fan,
 * training/validation.py:163-202 (np.argmax per batch, conf[c, c_] += sum((y == c) * (pred == c_)), one D2H per batch).
 * probs (n,k) float32; labels (n) int32 or NULL; pred (n) int32 or NULL: first arg-max of each row (numpy.argmax);
 * conf (k,k) uint64 or NULL (needs labels): conf[label][pred] += 1, accumulated over calls - the caller zeroes it once per
 * validation pass and reads it back once. */
int nimg_confusion_accumulate(const float* probs, const int* labels, int* pred, unsigned long long* conf, int n, int k,
                              void* stream);
int nimg_sigmoid_fwd(const float* x, float* y, long count, void* stream);
int nimg_sigmoid_bwd(const float* dy, const float* y, float* dx, long count, void* stream);
/* The activations a UNet / FAN / TwitterDCN may be built with besides LeakyReLU(0.2) - helpers/tf_helpers.py:22-28
 * `activation_mapping`, selected by the models' `activation=` hyper-parameter (models/pipelines.py:179, forensics.py:55,
 * compression.py:202): kind 0 leaky_relu(alpha), 1 relu, 2 tanh, 3 sigmoid, 4 softsign x / (1 + |x|).  Element-wise on float32,
 * y may alias x; the backward pass takes the derivative from the stored OUTPUT y (y > 0 ? 1 : alpha | y > 0 | 1 - y^2 |
 * y (1 - y) | (1 - |y|)^2), dx may alias dy.  NIMG_ERR_ARG for another kind. */
int nimg_activation_fwd(const float* x, float* y, long count, int kind, float alpha, void* stream);
int nimg_activation_bwd(const float* dy, const float* y, float* dx, long count, int kind, float alpha, void* stream);
int nimg_gamma_ste_fwd(const float* x, float* y, long count, float lo, float hi, float exponent, void* stream);
int nimg_gamma_ste_bwd(const float* x, const float* dy, float* dx, long count, float lo, float hi, float exponent,
                       void* stream);
/* out (n,2h,2w,c) = in with zeros inserted (stride-2 transposed convolution = zero insertion + stride-1 conv) */
/* 5x5 stride-2 TF-SAME convolution (Conv2D(.., 5, strides=2) of TwitterDCN, models/compression.py:217-229) as a 3x3 stride-1
 * convolution over the space-to-depth image (even h, w; cp >= 4 c block channels, the rest zero):
 *   s2d2_affine_bf16       y (n,h/2,w/2,cp) bf16: y[by][bx][(2 pr + pc) c + ci] = a x[2by + pr][2bx + pc][ci] + b
 *   s2d_conv_weights       w5 (5,5,cin,cout) -> w3 (3,3,cp,cout), tap ky = 2 dy + pr - 1;  _bwd gathers dw3 back into dw5
 *   d2s2_scale             x (n,h,w,c) = scale * depth-to-space of xs (n,h/2,w/2,cp): the input gradient of the layer */
int nimg_s2d2_affine_bf16(const float* x, void* y, int n, int h, int w, int c, int cp, float a, float b, void* stream);
int nimg_s2d_conv_weights(const float* w5, float* w3, int cin, int cp, int cout, void* stream);
int nimg_s2d_conv_weights_bwd(const float* dw3, float* dw5, int cin, int cp, int cout, int accumulate, void* stream);
int nimg_d2s2_scale(const float* xs, float* x, int n, int h, int w, int c, int cp, float scale, void* stream);
int nimg_zero_insert2(const float* in, float* out, int n, int h, int w, int c, void* stream);
/* DiscreteLatent (models/layers.py:183-203): latent = Quantization('soft-codebook' | identity)(scale * z) evaluated in
 * float64 like the reference (layers.py:141), plus the batch-global differentiable entropy of the latent
 * (helpers/tf_helpers.py:290-333).  scale: device scalar (may be NULL = 1).  count_global: number of latent values over
 * ALL ranks (0 = count).  finalize = 0 leaves the K float64 histogram sums at workspace + 1024*K doubles for an
 * all-reduce; call nimg_latent_entropy_finalize afterwards. The workspace must stay untouched until nimg_latent_bwd.
 * soft_codebook: bit 0 = Quantization('soft-codebook') (else identity); bit 1 = the caller's promise that the codebook is
 * unit-spaced (codebook[k] = codebook[0] + k, the reference's consecutive integers, models/layers.py:160-170): with gamma >= 25
 * and v = 50 the kernels then evaluate, for a value inside the codebook's range, only the five centres around the nearest one -
 * the weight of any other centre is below 2e-33 of the nearest one's, under the last bit of every float64 sum it would enter. */
size_t nimg_latent_workspace_bytes(int codebook_size);
int nimg_latent_fwd(const float* z, const float* scale, const float* codebook, int codebook_size, float v,
                    float gamma, int soft_codebook, float* latent, float* entropy, long count, long count_global,
                    void* workspace, size_t workspace_bytes, int finalize, void* stream);
int nimg_latent_entropy_finalize(int codebook_size, long count_global, float* entropy, void* workspace, void* stream);
/* dz = d loss / d z given dlatent (may be NULL) and the entropy term's coefficient d loss / d H; dscale (optional) */
int nimg_latent_bwd(const float* z, const float* scale, const float* latent, const float* dlatent,
                    float entropy_coef, const float* codebook, int codebook_size, float v, float gamma,
                    int soft_codebook, float* dz, float* dscale, int accumulate_dscale, long count, void* workspace,
                    size_t workspace_bytes, void* stream);
/* tf.nn.l2_loss(target - y) = sum((target-y)^2)/2 and grad_y (+)= grad_scale * (y - target)  (compression.py:92-93) */
size_t nimg_l2_loss_workspace_bytes(void);
int nimg_l2_loss(const float* target, const float* y, float* loss, float* grad_y, long count, float grad_scale,
                 int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Throughput mode of the convolutions: bf16 operands on the matrix cores (v_mfma_f32_32x32x16_bf16), float32
 * accumulation, float32 tensors in HBM.  Same semantics / arguments as nimg_conv2d_fwd / nimg_conv2d_wgrad; channel
 * counts must be multiples of 8 (forward) / 4 (weight gradient).  Judged on PSNR / accuracy parity (BASELINE.json),
 * not on the 1e-4 contract.  Weights are laid out once per step by nimg_conv_weights_bf16:
 *   mode 0 (forward)        wb[ci/16][tap][co][ci%16]        = w[tap][ci][co]   (16-channel chunk major, zero padded)
 *   mode 1 (input gradient) wb[co/16][taps-1-tap][ci][co%16] = w[tap][ci][co]   (then call with cin/cout swapped) */
size_t nimg_conv_weights_bf16_bytes(int ks_h, int ks_w, int cin, int cout, int mode);
int nimg_conv_weights_bf16(const float* w, void* wb, int ks_h, int ks_w, int cin, int cout, int mode, void* stream);
/* Backward of a fused conv+pool layer with few input channels (FAN conv1, forensics.py:69-70) without materialising
 * the sparse full-resolution gradient: g (n,h/2,wd/2,cout) = upstream gradient x LeakyReLU'(pooled), idx = arg-max
 * bytes of nimg_conv2d_pool_fwd[_bf16]; the 2x2 un-pooling happens while the tiles are staged.
 *   wgrad: cin 3|4, ks 3|5, SAME, stride 1;  dgrad: ci 3, cz 32, ks 5 (w = the forward kernel as stored). */
int nimg_conv2d_wgrad_pooled_bf16(const float* in, int cin, const float* g, const unsigned char* idx, int cout,
                                  float* dw, float* db, int n, int h, int wd, int ks, int accumulate, void* workspace,
                                  size_t workspace_bytes, void* stream);
int nimg_conv2d_dgrad_fewin_pooled_bf16(const float* g, const unsigned char* idx, const float* w, float* out, int ci,
                                        int cz, int n, int h, int wd, int ks, void* stream);
/* bf16 STORAGE of the FAN-internal tensors in throughput mode.  Every consumer of the pooled activations, of the
 * un-pooled gradients and of the pooled gradients converts them to bf16 MFMA operands anyway, so keeping them in bf16
 * in HBM changes no result bit - it only halves the bytes the kernels stage.  Same arguments as the entry points
 * without _ex, plus `flags` (NIMG_BF16_*) saying which tensors hold bf16. */
int nimg_conv2d_fwd_bf16_ex(const float* in1, int c1, const float* in2, int c2, const void* wb, const float* bias,
                            float* out1, int o1, float* out2, int o2, const float* act_mask, int n, int h, int wd,
                            int ks, int stride, int pad_t, int pad_l, int pad_mode, int hout, int wout, int act,
                            float alpha, int flags, void* stream);
int nimg_conv2d_wgrad_bf16_ex(const float* in1, int c1, const float* in2, int c2, const float* dz, int cout, float* dw,
                              float* db, int n, int h, int wd, int ks, int stride, int pad_t, int pad_l, int pad_mode,
                              int hout, int wout, int accumulate, void* workspace, size_t workspace_bytes, int flags,
                              void* stream);
int nimg_conv2d_pool_fwd_bf16_ex(const float* in, int cin, const float* w, const void* wb, const float* bias,
                                 float* pool_out, unsigned char* pool_idx, int cout, int n, int h, int wd, int ks,
                                 int act, float alpha, int flags, void* stream);
/* 3x3 SAME stride-1 convolution + bias + activation storing BOTH the full output and its 2x2 max-pooled tensor (bf16 in and out;
 * cin % 8 == 0, cout % 8 == 0, even h / wd > 8; pool_idx - the arg-max bytes - optional).  The second convolution of a UNet
 * encoder level + its max_pool2d (models/pipelines.py:160-173): the full tensor is the skip connection. */
int nimg_conv2d_fwd_pool_also_bf16(const float* in, int cin, const void* wb, const float* bias, float* out, float* pool_out,
                                   unsigned char* pool_idx, int cout, int n, int h, int wd, int act, float alpha, void* stream);
/* Row-band STREAMING form of the 3x3 / stride 1 / SAME convolution over bf16-stored activations with 32 output channels
 * (csrc/conv3_rows.hip; the UNet's level-1 layers, models/pipelines.py:190-216, throughput mode): in1 (n, h, wd, c1) [+ in2
 * (n, h, wd, c2)], wb = nimg_conv_weights_bf16 image (mode 0 forward, mode 1 input gradient), bias (32) or null, mask = optional
 * (n, h, wd, 32) bf16 activation whose LeakyReLU' multiplies the result, out (n, h, wd, 32) bf16, pool_out = optional
 * (n, h / 2, wd / 2, 32) bf16 2x2 max-pool of out.  Results are bit-identical to nimg_conv2d_fwd_bf16_ex (+ nimg_maxpool2_fwd_bf16).
 * Shapes: wd == 128, h % 4 == 0, cout == 32, (c1, c2) in {(32, 0), (64, 0), (32, 32)}; or cout == 64 from c1 == 32 as TWO 32-channel
 * tensors out / out2 (a decoder layer's input gradient, no bias / mask / pool).  flags: NIMG_ROWS_F32_OUT = out holds float32.
 * Second level of the UNet: wd == 64, (c1, c2, cout) in {(32, 0, 64), (64, 0, 64)}, out (n, h, wd, 64) bf16 (mask likewise), no pool.
 * Anything else NIMG_ERR_ARG (use the tile kernels). */
#define NIMG_ROWS_F32_OUT 1
int nimg_conv3_rows_bf16(const void* in1, int c1, const void* in2, int c2, const void* wb, const float* bias, const void* mask,
                         void* out, void* out2, void* pool_out, int n, int h, int wd, int cout, int act, float alpha, int flags,
                         void* stream);
/* The UNet's last layer in the same form: 3x3 convolution 32 -> 12 channels + bias, written as clip(depth_to_space(., 2), 0, 1)
 * (models/pipelines.py:216-223; the clip is straight-through, this is its forward value): in (n, h, wd, 32) bf16, wb the mode-0
 * weight image of the (3, 3, 32, 12) kernel, bias (12) or null, y (n, 2 h, 2 wd, 3) float32.  Bit-identical to
 * nimg_conv2d_fwd_bf16_ex + nimg_d2s_clip_fwd(scale 1, shift 0, clip).  wd == 128, h % 4 == 0. */
int nimg_conv3_rows_d2s_bf16(const void* in, int c1, const void* wb, const float* bias, float* y, int n, int h, int wd, void* stream);
/* The UNet's first layer in the same form: Conv2D(32, 3x3, SAME) + bias [+ LeakyReLU(alpha)] on the float32 4-plane RAW stack
 * (models/pipelines.py:190): in (n, h, wd, 4) float32, w (3, 3, 4, 32) float32 HWIO, out (n, h, wd, 32) bf16.  Bit-identical to
 * nimg_conv2d_fwd_smallc_bf16_ex.  wd == 128, h % 4 == 0. */
int nimg_conv3_rows_c4_bf16(const float* in, const float* w, const float* bias, void* out, int n, int h, int wd, int act, float alpha,
                            void* stream);

/* Input gradient of a 3x3 SAME stride-1 convolution whose input was a 2x2 max-pool, written THROUGH that pool (the first
 * convolution of a UNet encoder level, models/pipelines.py:160-173 under the tape), all tensors bf16: dz (n,h,wd,c1) = the
 * convolution's output gradient, wb = its weights in input-gradient form (nimg_conv_weights_bf16 mode 1), act (n,2h,2wd,cout) =
 * the activation the pool read, skip (same shape, optional, may be `out`) = the gradient arriving over the level's skip
 * connection, out (n,2h,2wd,cout) = [route to the first maximum of each window + skip] x LeakyReLU'(act) (apply_mask) - the
 * arithmetic of nimg_conv2d_fwd_bf16_ex(bf16 out) followed by nimg_maxpool2_bwd_bf16, same bits.  c1 % 8 == 0, cout % 8 == 0. */
int nimg_conv2d_dgrad_unpool_out_bf16(const float* dz, int c1, const void* wb, const float* act, const float* skip, float* out,
                                      int cout, int n, int h, int wd, int apply_mask, float alpha, void* stream);
int nimg_conv2d_wgrad_pooled_bf16_ex(const float* in, int cin, const float* g, const unsigned char* idx, int cout,
                                     float* dw, float* db, int n, int h, int wd, int ks, int accumulate, void* workspace,
                                     size_t workspace_bytes, int flags, void* stream);     /* NIMG_BF16_DZ: g is bf16 */
int nimg_conv2d_dgrad_fewin_pooled_bf16_ex(const float* g, const unsigned char* idx, const float* w, float* out, int ci,
                                           int cz, int n, int h, int wd, int ks, int flags, void* stream);
int nimg_maxpool2_unpool_ex(const float* dp, const unsigned char* idx, const float* pooled, float* dz, int n, int ho,
                            int wo, int c, int apply_lrelu_mask, float alpha, int flags, void* stream);
/* Conv2DTranspose(cout, [2,2], [2,2]) forward (pipelines.py:205) in throughput mode: four 1x1 products (one per output
 * phase) on the matrix core in one launch.  wb = nimg_conv_weights_bf16(w, 2, 2, cout, cin, mode 1) of the Keras
 * (2,2,Cout,Cin) kernel; x (n,h,wd,cin) -> y (n,2h,2wd,cout); cin % 8 == 0. */
int nimg_convt2x2_fwd_bf16(const float* x, const void* wb, const float* bias, float* y, int n, int h, int wd, int cin,
                           int cout, void* stream);
/* flags: NIMG_BF16_IN = x is stored as bf16, NIMG_BF16_OUT = y is stored as bf16 (cout % 4 == 0) */
int nimg_convt2x2_fwd_bf16_ex(const float* x, const void* wb, const float* bias, float* y, int n, int h, int wd, int cin,
                              int cout, int flags, void* stream);
/* every bf16 weight image of a model in one launch: `table` = n_entries x 4 int64 on the device,
 * {w pointer, wb pointer, (kh*kw << 32) | mode, (cin << 32) | cout}, each entry as nimg_conv_weights_bf16 would build it */
int nimg_conv_weights_bf16_batch(const void* table, int n_entries, void* stream);
int nimg_conv2d_fwd_bf16(const float* in1, int c1, const float* in2, int c2, const void* wb, const float* bias,
                         float* out1, int o1, float* out2, int o2, const float* act_mask, int n, int h, int wd,
                         int ks, int stride, int pad_t, int pad_l, int pad_mode, int hout, int wout, int act,
                         float alpha, void* stream);
size_t nimg_conv2d_wgrad_bf16_workspace_bytes(int cin, int cout, int ks_h, int ks_w, int n, int hout, int wout);
int nimg_conv2d_wgrad_bf16(const float* in1, int c1, const float* in2, int c2, const float* dz, int cout, float* dw,
                           float* db, int n, int h, int wd, int ks, int stride, int pad_t, int pad_l, int pad_mode,
                           int hout, int wout, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* FAN front end in throughput mode (models/forensics.py:69, the 3-channel side of the first convolution):
 * few INPUT channels (cin 3 or 4; float32 HWIO weights are converted in-kernel), K = (tap, ci) packed; and the input
 * gradient of a (ks,ks,3,32) SAME stride-1 convolution with the kx loop folded into the MFMA N dimension (w = the
 * FORWARD kernel as stored).  nimg_conv2d_wgrad_bf16 picks the (tap, ci)-packed weight-gradient kernel by itself. */
int nimg_conv2d_fwd_smallc_bf16(const float* in, int cin, const float* w, const float* bias, float* out, int cout,
                                int n, int h, int wd, int ks, int pad_mode, int act, float alpha, void* stream);
/* flags: NIMG_BF16_OUT = out is stored as bf16 (cout % 4 == 0): the UNet's first convolution (pipelines.py:191, 4 -> 32) */
int nimg_conv2d_fwd_smallc_bf16_ex(const float* in, int cin, const float* w, const float* bias, float* out, int cout,
                                   int n, int h, int wd, int ks, int pad_mode, int act, float alpha, int flags, void* stream);
int nimg_conv2d_dgrad_fewin_bf16(const float* dz, const float* w, float* out, int ci, int cz, int n, int h, int wd,
                                 int ks, void* stream);

/* manipulation_awgn / _gamma / _median, helpers/tf_helpers.py:79-110.  awgn takes the noise tensor from the caller
 * (tf.random.normal is not reproducible); its mask is one byte per ELEMENT.  median: kernel odd <= 9,
 * REFLECT pad, sel (n,h,w,3) uint8 = window index of the selected element; nimg_median_bwd scatters with atomic adds into
 * a ZEROED dx. */
int nimg_awgn_fwd(const float* x, const float* noise, float* y, uint8_t* mask, long count, float strength,
                  void* stream);
int nimg_awgn_bwd(const float* x, const float* noise, const float* dy, const uint8_t* mask, float* dx, long count,
                  float strength, void* stream);
int nimg_gamma_fwd(const float* x, float* y, long count, float gamma, void* stream);
int nimg_gamma_bwd(const float* x, const float* dy, float* dx, long count, float gamma, void* stream);
int nimg_median_fwd(const float* x, float* y, uint8_t* sel, int n, int h, int w, int kernel, void* stream);
int nimg_median_bwd(const float* dy, const uint8_t* sel, float* dx, int n, int h, int w, int kernel, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NIMG_H */
