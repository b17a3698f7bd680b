/*
 * rcgan_hip.h -- flat C ABI of librcgan_hip.so: the MI355X (gfx950) kernels behind the RCGAN
 * generator/discriminator training step.
 *
 * The reference (tkkiran/Robust-Conditional-GAN) has no FFI / custom-op seam: every FLOP is a stock
 * TensorFlow-1.5 kernel reached through its L1 Python op wrappers.  This ABI sits directly under an
 * L1-compatible Python shim (robust-conditional-gan_amd/ops_mnist.py, ops_cifar.py); each entry point
 * names the reference call site whose TF kernel it replaces (paths relative to the reference root).
 *
 * Conventions:
 *   - every entry returns int: 0 = ok, <0 = RCGAN_E*; rcgan_last_error(ctx) gives the message.
 *     No exceptions or aborts cross the ABI.
 *   - all buffers are caller-owned DEVICE pointers; the library never frees or retains them past the
 *     call.  Scratch is a caller-provided workspace (see the *_workspace_bytes helpers).
 *   - one hipStream_t per ctx; every call is asynchronous on that stream; a ctx is not thread-safe;
 *     distinct ctxs are independent (one ctx per rank/process).
 *   - activations are NHWC; conv filters HWIO ([kh][kw][Cin][Cout]); transposed-conv filters
 *     [kh][kw][Cout][Cin] (mnist/ops.py:74).  Parameters, their gradients and optimiser state are
 *     always fp32; activations / activation gradients are fp32 or bf16 (desc.dtype).
 */
#ifndef RCGAN_HIP_H
#define RCGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCGAN_OK 0
#define RCGAN_EINVALID_ARG (-1)
#define RCGAN_EUNSUPPORTED_SHAPE (-2)
#define RCGAN_EWORKSPACE_TOO_SMALL (-3)
#define RCGAN_EHIP (-4)
#define RCGAN_ERCCL (-5)

#define RCGAN_F32 0
#define RCGAN_BF16 1
#define RCGAN_F16 2   /* IEEE half: only in the fp16 build of the library (librcgan_hip_f16.so), which has no bf16 */

/* activation codes shared by several entry points */
#define RCGAN_ACT_NONE 0
#define RCGAN_ACT_RELU 1
#define RCGAN_ACT_LRELU 2   /* tf.maximum(x, 0.2x): mnist/ops.py:94-95 */
#define RCGAN_ACT_TANH 3
#define RCGAN_ACT_SIGMOID 4

typedef struct rcgan_ctx rcgan_ctx;

/* ---- context ------------------------------------------------------------------------------- */
/* stream: a hipStream_t owned by the caller (e.g. torch.cuda.current_stream().cuda_stream) or NULL
 * for the null stream.  Replaces: tf.Session(config) at cifar10/gan_resnet.py:494-496, mnist/main.py:100. */
int rcgan_create(rcgan_ctx** out, int device, void* stream);
int rcgan_destroy(rcgan_ctx* ctx);
const char* rcgan_last_error(rcgan_ctx* ctx);
const char* rcgan_version(void);
/* The 16-bit activation dtype this build of the library computes in: RCGAN_BF16 (librcgan_hip.so) or RCGAN_F16
 * (librcgan_hip_f16.so, the same sources compiled with -DRCGAN_HALF_FP16=1: v_mfma_f32_16x16x32_f16, v_cvt_f16_f32).
 * The other 16-bit dtype is rejected with RCGAN_EINVALID_ARG.  (The reference computes in fp32; BASELINE configs 3 / 5.) */
int rcgan_half_dtype(void);
/* CRC-32C (Castagnoli, reflected 0x82F63B78), host-side, no GPU needed: crc of `n` bytes continuing from `crc`
 * (pass 0 to start).  The checksum TensorFlow's V2 checkpoint bundles carry per tensor and per table block
 * (tf.train.Saver at cifar10/gan_resnet.py:906, mnist/model.py:265); used by the bundle reader/writer of the host layer. */
unsigned rcgan_crc32c(unsigned crc, const void* data, size_t n);
int rcgan_set_stream(rcgan_ctx* ctx, void* stream);
/* Fork/join onto the context's second stream: launches between side_begin and side_end run on it, ordered after
 * everything issued before the fork; side_join makes the main stream wait for them.  Used to run a layer's filter
 * gradient next to its data gradient (two sess.run-internal independent TF ops: conv2d_backprop_filter /
 * conv2d_backprop_input).  Works under rcgan_graph_begin/end (becomes a fork/join in the captured graph). */
int rcgan_side_begin(rcgan_ctx* ctx);
int rcgan_side_end(rcgan_ctx* ctx);
int rcgan_side_join(rcgan_ctx* ctx);
int rcgan_stream_sync(rcgan_ctx* ctx);
/* HIP-event timing on the ctx stream (bench.py roofline leg): slot in [0,64). */
int rcgan_event_record(rcgan_ctx* ctx, int slot);
int rcgan_event_elapsed_ms(rcgan_ctx* ctx, int slot_start, int slot_end, float* ms);
/* Per-kernel profiling: while armed, every launch of the chosen kernel is bracketed by HIP events on the ctx
 * stream (eager launches only, not graph replays).  rcgan_prof_end returns the number of launches, their
 * summed duration and their summed ALGORITHMIC flops (2*M*K*Cout per launch). */
#define RCGAN_PROF_CONV_MFMA_128 1   /* conv_mfma_kernel<128,128> (fwd + dgrad) */
#define RCGAN_PROF_CONV_MFMA_64 2    /* conv_mfma_kernel<64,64> */
#define RCGAN_PROF_WGRAD_MFMA 3      /* conv_mfma_wgrad_kernel */
#define RCGAN_PROF_CONV_P8 4         /* conv_mfma_p8_kernel: 256 x 256 tile, 8 wavefronts (fwd + dgrad of the 256-channel layers) */
#define RCGAN_PROF_CONV_P8N 5        /* conv_mfma_p8n_kernel: 256 x 128 tile */
#define RCGAN_PROF_GATHER_F32 7      /* gemm_gather_kernel (conv_direct.hip): the fp32 path's gather GEMM on the fp32 matrix cores */
#define RCGAN_PROF_ALLREDUCE 6       /* every all-reduce group of comm.hip (eager launches only); "flops" = bytes exchanged per rank */
int rcgan_prof_begin(rcgan_ctx* ctx, int which);
int rcgan_prof_end(rcgan_ctx* ctx, int* launches, double* total_ms, double* total_flops);
/* Flops the kernels of the last rcgan_prof_begin .. rcgan_prof_end section EXECUTED: equal to the algorithmic count except for
 * the sub-pixel form of the upsample-3x3 convolutions (four 2x2 convolutions with summed filters: 4/9 of the multiply-adds). */
int rcgan_prof_executed_flops(rcgan_ctx* ctx, double* executed_flops);
/* How many launches of that section also applied the batch norm in front of the convolution to their staged input
 * (rcgan_conv2d_fwd_bn_residual on a halo-patch kernel): their time includes that work, their FLOP count does not. */
int rcgan_prof_bn_in_launches(rcgan_ctx* ctx, int* launches);
/* Diagnostics: while `stamps` is non-null, every workgroup of the 256 x 256 convolution kernel writes 8 x uint64 to
 * stamps[workgroup * 8 ..]: s_memtime at {start, tap table built, first K-tile landed, K loop done, stores issued,
 * stores complete}, HW_ID, XCC_ID (scripts/exp_p8_timeline.py).  Pass null to switch it off. */
int rcgan_debug_stamps(rcgan_ctx* ctx, void* stamps);
/* hipGraph capture of everything launched on the ctx stream between begin/end; replay with launch.
 * Replaces the per-step sess.run dispatch (gan_resnet.py:931,938; mnist/model.py:347-372). */
int rcgan_graph_begin(rcgan_ctx* ctx);
int rcgan_graph_end(rcgan_ctx* ctx, int* graph_id);
/* Capture contract of the entry points that take NO workspace argument.  The fp32 path of rcgan_conv2d_fwd, rcgan_deconv2d_bwd_data[_cols],
 * rcgan_linear_bwd_data and the narrow (<= 2 output channels) data gradient keep partial tiles / per-source-pixel products in two scratch
 * buffers hidden in the context that grow on demand with hipMalloc.  An allocation cannot be captured: inside rcgan_graph_begin .. end a call
 * that would have to grow a buffer returns RCGAN_EHIP (hipErrorStreamCaptureUnsupported) and the capture has to be dropped with
 * rcgan_graph_abort.  So before capturing either run every shape of the captured body once eagerly, or reserve the scratch up front:
 * rcgan_reserve_scratch grows BOTH buffers to at least `bytes` each (never inside a capture; buffers only grow; an outgrown buffer stays
 * alive until rcgan_destroy because earlier graphs address it).  Upper bounds: split reduction 8 x 4 bytes x the output elements of the
 * largest GEMM with < 128 output tiles of 64 x 64 (<= 16 MiB); narrow data gradient 4 bytes x n x oh x ow x kh x kw x cin of the layer.
 * rcgan_scratch_bytes reports the current sizes (either pointer may be null). */
int rcgan_reserve_scratch(rcgan_ctx* ctx, size_t bytes);
int rcgan_scratch_bytes(rcgan_ctx* ctx, size_t* split_reduction_bytes, size_t* narrow_bytes);
/* Leave a capture that a failed launch has made impossible to finish; what was recorded is dropped (nothing of it ran). */
int rcgan_graph_abort(rcgan_ctx* ctx);
int rcgan_graph_launch(rcgan_ctx* ctx, int graph_id);
int rcgan_graph_destroy(rcgan_ctx* ctx, int graph_id);

/* ---- convolution ---------------------------------------------------------------------------- */
#define RCGAN_CONV_IN_UPSAMPLE2X 1  /* conv reads nearest-2x-upsampled input (gan_resnet.py:263-264 folded in) */
#define RCGAN_CONV_IN_RELU 2        /* relu applied to the input on load (gan_resnet.py:305,317,347) */
#define RCGAN_CONV_ACCUMULATE 4     /* out += result (residual add, gan_resnet.py:328,353) */
#define RCGAN_CONV_FORCE_DIRECT 8   /* testing: bypass the MFMA path */
#define RCGAN_CONV_OUT_MEANPOOL2 16 /* ConvMeanPool (gan_resnet.py:241-247): the 2x2 mean pool folded into the convolution -- y [n,h/2,w/2,cout] =
                                      * meanpool2(conv(x)) + bias, computed as ONE 4x4 stride-2 convolution with summed filters (4/9 of the
                                      * multiply-adds), only where rcgan_conv_fused_pool_ok says so.  In rcgan_conv2d_bwd_weight[_group] the
                                      * flag means "dy is the POOLED gradient [n,h/2,w/2,cout]" (sub-pixel filter gradient, 4/9 of the
                                      * multiply-adds, no spread dy), only where rcgan_conv_wgrad_pool_ok says so */

#define RCGAN_CONV_RESID_UPSAMPLE2X 32 /* rcgan_conv2d_fwd_residual: the residual is [n,h/2,w/2,cout] and is added nearest-upsampled -- the
                                      * shortcut of an up block evaluated BEFORE the upsample (a 1x1 convolution commutes with it:
                                      * gan_resnet.py:258-272, 295-328), a quarter of its multiply-adds and of its bytes; only where
                                      * rcgan_conv_resid_up_ok says so */

typedef struct rcgan_conv_desc {
  int n, h, w, cin;   /* logical conv input: after the 2x upsample when IN_UPSAMPLE2X is set */
  int cout, kh, kw, stride;
  int dtype;          /* RCGAN_F32 or the build's 16-bit dtype (rcgan_half_dtype()): activations and activation gradients */
  int flags;
} rcgan_conv_desc;

/* Filter preparation: master fp32 HWIO filter (optionally divided by *sigma, the spectral norm) ->
 * the layouts the conv kernels consume.  `prepared` must hold rcgan_conv_prepared_bytes(desc).
 * Replaces the W_bar = W / sigma reshape (mnist/sn.py:55-62) + the filter argument of tf.nn.conv2d. */
size_t rcgan_conv_prepared_bytes(const rcgan_conv_desc* d);
int rcgan_conv_prepare(rcgan_ctx* ctx, const rcgan_conv_desc* d, const float* w_hwio,
                       const float* sigma /* device scalar or NULL */, void* prepared);

/* The same for many filters in ONE launch (every conv of a network at the start of a step).  items: HOST array. */
typedef struct rcgan_prepare_item {
  rcgan_conv_desc desc;      /* only kh, kw, cin, cout, stride, dtype, flags matter */
  const float* w;
  const float* sigma;        /* device scalar or NULL */
  void* prepared;            /* rcgan_conv_prepared_bytes(&desc) */
} rcgan_prepare_item;
int rcgan_conv_prepare_batch(rcgan_ctx* ctx, const rcgan_prepare_item* items, int n_items);
/* The same launch with the projection head's label embeddings riding in it as extra workgroups:
 * E[l][j] = (sum_k table[l][k] * w_e[k][j]) / sigma_e + b_e[j]  (embedding.py:29-51 + D.Embedding_y, gan_resnet.py:414-421).
 * E depends on parameters only; computed here it leaves the step's dependency chain (rcgan_head_desc::E_pre).  e may be NULL. */
typedef struct rcgan_embed_desc {
  int v, e_dim, d;
  const float *table, *w_e, *sigma_e /* device scalar or NULL */, *b_e /* or NULL */;
  float* E;               /* [v][d] */
} rcgan_embed_desc;
int rcgan_conv_prepare_batch_embed(rcgan_ctx* ctx, const rcgan_prepare_item* items, int n_items, const rcgan_embed_desc* e);
/* ... and with the critic step's INPUT work riding too (it depends on the step's inputs, not on its parameters, and would otherwise
 * be five launches at the head of the step's dependency chain): dequantisation noise noise[i] ~ U[noise_lo, noise_hi) drawn from the
 * device stream exactly as rcgan_rng_fill(kind 0) over n*3072 floats would (the stream then advances by as much), the preprocessing
 * of rcgan_preprocess_cifar into rows [0, n) of x [2n,32,32,3] (gan_resnet.py:548-551), the 2x2 mean pool of all 2n images
 * (rows [n, 2n) = the fakes: already in x, or copied there from a slice of `fakes`; pooled [2n,16,16,3] or NULL: D.Block.1's shortcut input, :239-240, :346) and a
 * zero-fill of fill_count floats (the gradient slab; or NULL).  Same bits as the separate entry points.  e / si may be NULL. */
typedef struct rcgan_step_inputs_desc {
  int n, dtype;
  const int32_t* images;   /* [n][3][32][32] */
  void* x; void* pooled;
  float noise_lo, noise_hi;
  uint64_t seed; void* rng_state;
  float* fill; size_t fill_count;
  /* optional: the step's fakes are slice *fake_slice (a device counter, moved on mod n_slices by the launch) of
   * fakes [n_slices][n][32][32][3]: copied into rows [n, 2n) of x by the same launch.  NULL: they are in x already. */
  const void* fakes; void* fake_slice; int n_slices;
} rcgan_step_inputs_desc;
int rcgan_conv_prepare_batch_riders(rcgan_ctx* ctx, const rcgan_prepare_item* items, int n_items, const rcgan_embed_desc* e,
                                    const rcgan_step_inputs_desc* si);
/* (round 6) ... and the fragment-major filter copies the fused 8x8 stage and the register-filter kernel read (rcgan_dtrunk*,
 * rcgan_conv2d_rf; the layout of rcgan_fragments_prepare), written by the SAME launch straight from the fp32 weights instead of by a
 * launch of its own behind it: frags[i] names items[frags[i].item] (a 16-bit 3x3 stride-1 filter, channels % 64 == 0) and the two
 * destinations -- forward rows / rotated data-gradient rows, 9*cin*cout elements each -- with ctn channel tiles of 16 rows per block and
 * ss K-steps of 32 per slice (the 8x8 stage: 2, 36; the register-filter kernel: 4, 18).  Bit-identical to rcgan_conv_prepare +
 * rcgan_fragments_prepare (the same fp32 product W / sigma rounded once).  At most 12 per call; e / si may be NULL. */
typedef struct rcgan_frag_item { int item; int ctn, ss; void* fwd; void* bwd; } rcgan_frag_item;
int rcgan_conv_prepare_batch_frags(rcgan_ctx* ctx, const rcgan_prepare_item* items, int n_items, const rcgan_embed_desc* e,
                                   const rcgan_step_inputs_desc* si, const rcgan_frag_item* frags, int n_frags);

size_t rcgan_conv_workspace_bytes(const rcgan_conv_desc* d);
/* y = conv2d_SAME(x, w) (+bias).  Replaces tf.nn.conv2d + bias_add: mnist/ops.py:62-65,
 * cifar10/common/ops/conv2d.py:181-216.  x: [n, h(/2), w(/2), cin]; y: [n, oh, ow, cout]. */
/* 1 if the matrix-core kernels take d (3x3, stride 1, 16-bit, power-of-two image, channels % 64 == 0) with RCGAN_CONV_OUT_MEANPOOL2 */
int rcgan_conv_fused_pool_ok(const rcgan_conv_desc* d);
/* 1 if rcgan_conv2d_fwd_residual takes d with RCGAN_CONV_RESID_UPSAMPLE2X */
int rcgan_conv_resid_up_ok(const rcgan_conv_desc* d);
int rcgan_conv2d_fwd(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared,
                     const float* bias /* or NULL */, void* y);
/* y = conv2d_SAME(x, w) (+bias) + residual: the pre-activation residual sum `shortcut + output` of
 * gan_resnet.py:328 folded into the convolution's epilogue.  residual: [n, oh, ow, cout] or NULL, must not alias y. */
/* y = conv2d_SAME(act(batch_norm(x))) (+bias) with the (conditional) batch norm + activation of cond_batchnorm / nonlinearity
 * (normalization.py:27-59, gan_resnet.py:350-352) applied to the convolution's staged input instead of being written out and read back:
 * for a forward-only pass (the critic steps' generator forwards) the normalised tensor never exists.  x: the batch norm's INPUT;
 * mean / rstd [segments][cin] from rcgan_bn_fwd_segments(..., y = NULL) (statistics only) or rcgan_bn_stats; gamma / beta [n_labels][cin];
 * labels [n] or NULL.  Same values as rcgan_bn_apply_* followed by rcgan_conv2d_fwd (the affine is evaluated in the same fp32
 * sequence and rounded to 16 bits at the same point).  rcgan_conv_bn_in_ok: the small-output image-end layers (G.Output: 256 -> 3) and
 * (round 5) the 3x3 / upsample-3x3 layers the halo-patch kernels take (conv_mfma8h.hip: 16- / 32-wide (low-resolution) images, enough
 * 256-pixel tiles to fill the chip -- G.Block.2.Conv2, G.Block.3.Conv1 / Conv2 at the bench batches); d->flags may carry
 * RCGAN_CONV_IN_UPSAMPLE2X (the norm sits in front of the upsample, as in UpsampleConv) and, for the _residual form,
 * RCGAN_CONV_RESID_UPSAMPLE2X; not IN_RELU (the norm's own activation is `act`: RCGAN_ACT_NONE or RCGAN_ACT_RELU), ACCUMULATE, OUT_MEANPOOL2. */
int rcgan_conv_bn_in_ok(const rcgan_conv_desc* d);
int rcgan_conv2d_fwd_bn(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias, void* y,
                        int segments, const int32_t* labels, const float* gamma, const float* beta, const float* mean, const float* rstd,
                        int act);
/* ... + residual in the epilogue (rcgan_conv2d_fwd_residual's forms; halo-patch kernels only when residual != NULL). */
int rcgan_conv2d_fwd_bn_residual(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias,
                                 const void* residual, void* y, int segments, const int32_t* labels, const float* gamma, const float* beta,
                                 const float* mean, const float* rstd, int act);
int rcgan_conv2d_fwd_residual(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared,
                              const float* bias /* or NULL */, const void* residual /* or NULL */, void* y);
/* dx = d(conv)/dx.  With IN_RELU, dx is masked by x>0 (x = the pre-activation input).  With
 * IN_UPSAMPLE2X dx is the gradient w.r.t. the low-resolution input.  ACCUMULATE: dx += ...
 * Replaces tf.nn.conv2d_backprop_input (autodiff of the above).  ws: rcgan_conv_workspace_bytes. */
int rcgan_conv2d_bwd_data(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* dy, const void* prepared,
                          const void* x /* needed with IN_RELU */, void* dx, void* ws, size_t ws_bytes);
/* dx = conv2d_backprop_input(dy) (masked by x > 0 under RCGAN_CONV_IN_RELU) + residual, written out of place: the
 * accumulate of a second gradient contribution without touching the buffer that holds the first one (needed when that
 * buffer is still to be read, e.g. by a deferred filter gradient).  residual: [n, h, w, cin], must not alias dx. */
int rcgan_conv2d_bwd_data_residual(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* dy, const void* prepared,
                                   const void* x /* IN_RELU mask or NULL */, const void* residual, void* dx, void* ws, size_t ws_bytes);
/* dw (fp32 HWIO) = d(conv)/dw (dw = or += by accumulate), dbias = sum dy (if non-NULL).
 * Replaces tf.nn.conv2d_backprop_filter + BiasAddGrad.  Upsample-3x3 (IN_UPSAMPLE2X: x stored at [n,h/2,w/2,cin]) and
 * ConvMeanPool (OUT_MEANPOOL2: dy pooled) layers on the matrix cores are computed in their sub-pixel form: 16 Cin x Cout
 * products over the low-resolution grid folded into the 9 taps by the slab reduction, instead of 9 over the full one. */
/* 1 if rcgan_conv2d_bwd_weight takes d with RCGAN_CONV_OUT_MEANPOOL2 (dy = the pooled gradient) */
int rcgan_conv_wgrad_pool_ok(const rcgan_conv_desc* d);
int rcgan_conv2d_bwd_weight(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* dy,
                            float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes);
/* The filter gradients of `n` layers in one call (all x / dy available: the end of a backward pass).  Same results as n
 * rcgan_conv2d_bwd_weight calls; the layers the three-tap matrix-core kernel takes share ONE launch (+ one slab reduction):
 * the discriminator's 8x8 / 16x16 layers are launch-latency-bound one by one.  dbiases[i] may be NULL.
 * ws: at least twice the largest rcgan_conv2d_workspace_bytes of the layers plus the sum of their slab sizes. */
int rcgan_conv2d_bwd_weight_group(rcgan_ctx* ctx, int n, const rcgan_conv_desc* descs, const void* const* xs, const void* const* dys,
                                  float* const* dws, float* const* dbiases, int accumulate, void* ws, size_t ws_bytes);

/* Transposed convolution, filter [kh][kw][Cout][Cin] used as-is (no preparation, fp32).
 * d describes the *forward conv* it is the gradient of: n,h,w,cin = deconv OUTPUT (n,H,W,Cout_deconv),
 * d.cout = deconv input channels.  Replaces tf.nn.conv2d_transpose at mnist/ops.py:78-79. */
int rcgan_deconv2d_fwd(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const float* w,
                       const float* bias, void* y);
int rcgan_deconv2d_bwd_data(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* dy, const float* w, void* dx);
/* ... only the first n_cols input channels of it, dense ([n, h', w', n_cols]; RCGAN_CONV_ACCUMULATE in d.flags adds onto dx): the input of
 * the generator's transposed convolutions is conv_cond_concat(x, y) (mnist/ops.py:46-51, model.py:716-726) and the label channels need no
 * gradient -- the x part lands straight in x's gradient, a third of the column tiles (138 -> 128 channels) and the split kernel go away.
 * n_cols = 0: all of them. */
int rcgan_deconv2d_bwd_data_cols(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* dy, const float* w, void* dx, int n_cols);
int rcgan_deconv2d_bwd_weight(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* dy,
                              float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes);
/* The same filter gradient for x = conv_cond_concat(t, yb) (mnist/ops.py:46-51; the generator's g_h2 / g_h3, model.py:722-731): the first
 * n_cols channels of x are real, the other d.cout - n_cols (<= 16) are yb[n][:] (fp32 [n][d.cout - n_cols]) on every pixel.  The gather
 * GEMM runs over the n_cols real columns only (for 128 + 10 channels: two 64-wide column tiles instead of three); the label columns are
 * dW[kh][kw][c][n_cols + l] = sum_n yb[n][l] * (sum of dy[n] over the sub-grid of output pixels tap (kh, kw) reaches): one pass over dy, a
 * small product, and one reduction that writes all d.cout columns.  Same values as rcgan_deconv2d_bwd_weight up to fp32 summation
 * order.  Workspace: rcgan_deconv2d_bwd_weight_concat_bytes(d, n_cols). */
size_t rcgan_deconv2d_bwd_weight_concat_bytes(const rcgan_conv_desc* d, int n_cols);
int rcgan_deconv2d_bwd_weight_concat(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* dy, int n_cols, const float* yb,
                                     float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes);

/* ---- dense ----------------------------------------------------------------------------------- */
/* y[m,n] = x[m,k] @ w[k,n] (+bias).  w fp32 row-major, optionally divided by *sigma.
 * Replaces tf.matmul + bias: mnist/ops.py:114-116, cifar10/common/ops/linear.py:161-180. */
int rcgan_linear_fwd(rcgan_ctx* ctx, int m, int k, int n, int dtype, const void* x, const float* w,
                     const float* sigma, const float* bias, void* y);
int rcgan_linear_bwd_data(rcgan_ctx* ctx, int m, int k, int n, int dtype, const void* dy, const float* w,
                          const float* sigma, void* dx, int accumulate);
/* dw is the gradient w.r.t. the matrix actually multiplied (W or W_bar). */
int rcgan_linear_bwd_weight(rcgan_ctx* ctx, int m, int k, int n, int dtype, const void* x, const void* dy,
                            float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes);
size_t rcgan_linear_workspace_bytes(int m, int k, int n);

/* ---- normalisation ----------------------------------------------------------------------------- */
/* Batch statistics over rows of x[rows][c]: mean, biased var -> rstd = rsqrt(var+eps).
 * moving_mean / moving_var (may be NULL) get the TF fused-batch-norm update
 * m -= (m - batch)*(1-decay) with the UNBIASED variance (mnist/ops.py:38-44).
 * ws: rcgan_bn_workspace_bytes(rows, c). */
size_t rcgan_bn_workspace_bytes(int rows, int c);
int rcgan_bn_stats(rcgan_ctx* ctx, int rows, int c, int dtype, const void* x, float eps,
                   float* mean, float* rstd, float* moving_mean, float* moving_var, float decay,
                   void* ws, size_t ws_bytes);
/* y = act( gamma[l]*(x-mean)*rstd + beta[l] ), l = labels[row / rows_per_sample] (labels NULL -> row 0
 * of a [1][c] table = plain batch norm).  Replaces tf.nn.batch_normalization + embedding_lookup
 * (cifar10/common/ops/normalization.py:47-57) and tf.contrib.layers.batch_norm (mnist/ops.py:38-44)
 * fused with the following relu / lrelu (gan_resnet.py:305,317,366; mnist/model.py:661-719). */
int rcgan_bn_apply_fwd(rcgan_ctx* ctx, int n, int rows_per_sample, int c, int n_labels, int dtype, const void* x,
                       const int32_t* labels, const float* gamma, const float* beta,
                       const float* mean, const float* rstd, int act, void* y, void* ws, size_t ws_bytes);
/* rcgan_bn_stats + rcgan_bn_apply_fwd for `nseg` independent segments of n_per_seg samples stored back to back: each
 * segment is normalised with its OWN batch statistics (= nseg separate Generator() calls of the reference, e.g. the
 * N_CRITIC generator forwards of one iteration, gan_resnet.py:540-546,928-947, evaluated as one batch).  Forward only.
 * mean / rstd: [nseg][c] outputs; ws: nseg * rcgan_bn_workspace_bytes(n_per_seg*rows_per_sample, c). */
int rcgan_bn_fwd_segments(rcgan_ctx* ctx, int nseg, int n_per_seg, int rows_per_sample, int c, int n_labels, int dtype,
                          const void* x, const int32_t* labels, const float* gamma, const float* beta, float eps, int act,
                          float* mean, float* rstd, void* y, void* ws, size_t ws_bytes);
/* Backward of stats+apply (gradient flows through the batch statistics).  dgamma/dbeta: [n_labels][c]
 * (= or += by accumulate); dx = or += by accumulate_dx.  y is the forward output (activation mask). */
int rcgan_bn_bwd(rcgan_ctx* ctx, int n, int rows_per_sample, int c, int n_labels, int dtype,
                 const void* x, const void* y, const void* dy, const int32_t* labels,
                 const float* gamma, const float* mean, const float* rstd, int act,
                 void* dx, int accumulate_dx, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes);
/* rcgan_bn_bwd with the forward's beta (offset) table: for ReLU / leaky-ReLU the activation mask is then recomputed from x with
 * the forward's exact arithmetic (the fused power-of-two-channel kernels), so y is not read: 2 instead of 3 tensor reads per pass. */
int rcgan_bn_bwd2(rcgan_ctx* ctx, int n, int rows_per_sample, int c, int n_labels, int dtype, const void* x, const void* y,
                  const void* dy, const int32_t* labels, const float* gamma, const float* beta, const float* mean, const float* rstd,
                  int act, void* dx, int accumulate_dx, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes);
/* Inference-mode batch norm with moving statistics (gen_sampler, mnist/model.py:745-754). */
int rcgan_bn_infer(rcgan_ctx* ctx, int rows, int c, int dtype, const void* x, const float* gamma,
                   const float* beta, const float* moving_mean, const float* moving_var, float eps,
                   int act, void* y);
/* Its adjoint w.r.t. x: dx (= or +=) dy * act'(y) * gamma / sqrt(moving_var + eps).  Used by recover_labels
 * (mnist/model.py:494-640), which differentiates the frozen sampler w.r.t. its latent input. */
int rcgan_bn_infer_bwd(rcgan_ctx* ctx, int rows, int c, int dtype, const void* y, const void* dy, const float* gamma,
                       const float* moving_var, float eps, int act, void* dx, int accumulate);

/* ---- spectral normalisation (mnist/sn.py:31-75 == cifar10/common/ops/sn.py:31-75) --------------- */
typedef struct rcgan_sn_item {
  const float* w;   /* [k][c] (any filter reshaped to [-1, c]) */
  float* u;         /* [c] persistent; overwritten with u' iff update != 0 */
  float* sigma;     /* [1] out */
  float* save;      /* scratch kept for the backward: rcgan_sn_save_floats(k,c) floats */
  int k, c, update;
} rcgan_sn_item;
size_t rcgan_sn_save_floats(int k, int c);
/* One power iteration for every item in one launch (items: HOST array, copied into the launch). */
int rcgan_sn_power_iter(rcgan_ctx* ctx, const rcgan_sn_item* items, int n_items);
typedef struct rcgan_sn_bwd_item {
  const float* w; const float* dwbar; float* dw; float* save /* also scratch */; int k, c, accumulate;
} rcgan_sn_bwd_item;
/* dW from dW_bar, differentiating THROUGH the power iteration (no stop_gradient in the reference). */
int rcgan_sn_bwd(rcgan_ctx* ctx, const rcgan_sn_bwd_item* items, int n_items);
/* The same two launches with the optimiser of a SINGLE-RANK step in the second (round 6): after dW is formed for a chunk of
 * rows, TF ApplyAdam (tf.train.AdamOptimizer, gan_resnet.py:802-808; the arithmetic of rcgan_adam_tf) updates those rows of the
 * optimiser group's slabs w / m / v in place; rider workgroups do the same for the slab's other parameters (`ranges`: host array
 * of n_ranges {lo, hi} float offsets).  Items and ranges must tile [0, count) exactly, every item's w / dw must point into w / g at
 * the same offset.  hyper: DEVICE {lr, t} with t = the number of updates applied so far: the call advances it by one (in its first
 * launch) and uses the advanced value for the bias correction -- a captured step replays with a fresh count, the host rewrites the
 * pair only when lr changes.  dW is still written to g.  Not for data-parallel steps (the all-reduce sits between the gradient
 * and the update) nor for a dynamic loss scale (the overflow verdict does). */
typedef struct rcgan_sn_adam {
  float* w; float* g; float* m; float* v; size_t count;
  float* hyper;
  float beta1, beta2, eps, clip, grad_scale;
  int n_ranges; const size_t* ranges;
} rcgan_sn_adam;
int rcgan_sn_bwd_adam(rcgan_ctx* ctx, const rcgan_sn_bwd_item* items, int n_items, const rcgan_sn_adam* opt);

/* ---- elementwise / resampling -------------------------------------------------------------------- */
int rcgan_act_fwd(rcgan_ctx* ctx, size_t count, int dtype, int act, const void* x, void* y);
/* dx = dy * act'(.)  (mask/derivative taken from y for tanh/sigmoid, from x for relu/lrelu). */
int rcgan_act_bwd(rcgan_ctx* ctx, size_t count, int dtype, int act, const void* x_or_y, const void* dy,
                  void* dx, int accumulate);
int rcgan_add(rcgan_ctx* ctx, size_t count, int dtype, const void* a, const void* b, void* y);
int rcgan_axpby(rcgan_ctx* ctx, size_t count, int dtype, float alpha, const void* a, float beta, void* y);
int rcgan_cast(rcgan_ctx* ctx, size_t count, int src_dtype, const void* src, int dst_dtype, void* dst);
/* 2x2 mean pool (gan_resnet.py:239-240) and its adjoint; nearest 2x upsample (:263-264) and its adjoint. */
int rcgan_meanpool2_fwd(rcgan_ctx* ctx, int n, int h, int w, int c, int dtype, const void* x, void* y);
int rcgan_meanpool2_bwd(rcgan_ctx* ctx, int n, int h, int w, int c, int dtype, const void* dy, void* dx, int accumulate);
int rcgan_upsample2_fwd(rcgan_ctx* ctx, int n, int h, int w, int c, int dtype, const void* x, void* y);
int rcgan_upsample2_bwd(rcgan_ctx* ctx, int n, int h, int w, int c, int dtype, const void* dy, void* dx, int accumulate);
/* y[n,h,w,:c1] = x ; y[n,h,w,c1:] = onehot rows yb[n,:c2]  (conv_cond_concat, mnist/ops.py:46-51);
 * the adjoint copies the first c1 channels back. */
/* Zero-pad the channel axis: y[rows][before + c + after] (tf.pad; the option-A shortcut of the generated-label-accuracy
   classifier, cifar10/resnet-110/graph_optimized.pb used by gan_resnet.py:424-455).  Inference only. */
int rcgan_pad_channels(rcgan_ctx* ctx, size_t rows, int c, int before, int after, int dtype, const void* x, void* y);
int rcgan_concat_channels_fwd(rcgan_ctx* ctx, int n, int hw, int c1, int c2, int dtype, const void* x,
                              const float* yb, void* y);
int rcgan_concat_channels_bwd(rcgan_ctx* ctx, int n, int hw, int c1, int c2, int dtype, const void* dy, void* dx);
/* The batch repeated `reps` times back to back, y[r][i] = x[i] (count elements per copy), and its adjoint dx[i] (+)= sum_r dy[r][i]:
 * the reference's ten discriminator() calls on the same images, one per label (mnist/model.py:152-163 `unbiased`, :187-197
 * `estimate_confuse`), as ONE pass over 10 x batch samples for the discriminators whose convolutions see the label
 * (disc_type=vanilla, --concat_y).  rcgan_transpose_f32: y[c][r] (+)= x[r][c] -- that pass's [labels][samples] logits as the
 * [samples][labels] matrix of tf.concat(D_logits_all, 1) (:165,199), and the adjoint. */
int rcgan_tile_rows_fwd(rcgan_ctx* ctx, size_t count, int reps, int dtype, const void* x, void* y);
int rcgan_tile_rows_bwd(rcgan_ctx* ctx, size_t count, int reps, int dtype, const void* dy, void* dx, int accumulate);
int rcgan_transpose_f32(rcgan_ctx* ctx, int rows, int cols, const float* x, float* y, int accumulate);
/* uint8-as-int32 CHW [n][3][32][32] + dequantisation noise (fp32, CHW order) -> NHWC in [-1,1):
 * 2*(x/256-.5)+noise (gan_resnet.py:548-551). */
int rcgan_preprocess_cifar(rcgan_ctx* ctx, int n, const int32_t* images_chw, const float* noise_chw,
                           int dtype, void* y_nhwc);

/* Counter-based device RNG (Philox4x32-10).  kind 0: uniform [lo,hi); kind 1: normal(mean=lo, std=hi).
 * state: DEVICE uint64[1] stream offset, advanced on the stream after the draw (so a replayed graph
 * draws fresh numbers), or NULL for offset 0.  Replaces tf.random_normal (gan_resnet.py:359) and the
 * tf.random_uniform dequantisation noise (gan_resnet.py:549); the TF Philox stream itself is not
 * reproducible, parity tests feed explicit noise instead. */
int rcgan_rng_fill(rcgan_ctx* ctx, size_t count, int dtype, int kind, float lo, float hi, uint64_t seed,
                   void* state, void* y);

/* ---- discriminator head + losses -------------------------------------------------------------------- */
/* feat[n][c] = mean_hw(relu(x)) (gan_resnet.py:405-407; act = RCGAN_ACT_NONE gives the plain spatial
 * mean of mnist/model.py:679) and its adjoint. */
int rcgan_act_meanhw_fwd(rcgan_ctx* ctx, int n, int hw, int c, int dtype, int act, const void* x, float* feat);
int rcgan_act_meanhw_bwd(rcgan_ctx* ctx, int n, int hw, int c, int dtype, int act, const void* x,
                         const float* dfeat, void* dx);
/* rows gathered from a [v][d] table / scatter-added back (tf.nn.embedding_lookup, embedding.py:51). */
int rcgan_gather_rows(rcgan_ctx* ctx, int n, int d, const float* table, const int32_t* idx, float* out);
int rcgan_scatter_add_rows(rcgan_ctx* ctx, int n, int d, int v, const float* src, const int32_t* idx,
                           float* table_grad, int accumulate);
/* logit[n] = psi[n] + <feat[n,:], emb[n,:]>  (gan_resnet.py:588,763; mnist/model.py:685) + adjoint. */
int rcgan_proj_logit_fwd(rcgan_ctx* ctx, int n, int d, const float* feat, const float* psi, const float* emb,
                         float* logit);
int rcgan_proj_logit_bwd(rcgan_ctx* ctx, int n, int d, const float* feat, const float* emb, const float* dlogit,
                         float* dfeat, float* dpsi, float* demb, int acc_feat);
/* logits[n][v] = psi[n] + <feat[n,:], E[v,:]> for every label v (rcgan-u: gan_resnet.py:654-660, 737-740;
 * the reference re-runs the projection for all 10 labels -- features are computed once here) + adjoint. */
int rcgan_proj_logit_all_fwd(rcgan_ctx* ctx, int n, int d, int v, const float* feat, const float* psi,
                             const float* E, float* logits);
int rcgan_proj_logit_all_bwd(rcgan_ctx* ctx, int n, int d, int v, const float* feat, const float* E,
                             const float* dlogits, float* dfeat, float* dpsi, float* dE, int acc_feat);
/* Loss terms.  Each adds weight*L to *loss_acc (device scalar) and WRITES dlogit = weight * dL/dlogit.
 *   kind HINGE_REAL: mean relu(1-x)   HINGE_FAKE: mean relu(1+x)   NEG_MEAN: -mean x  (gan_resnet.py:604-605,773)
 *   CE_ONES / CE_ZEROS: mean sigmoid_ce(x, 1|0) (mnist/model.py:139-145)
 * x: [rows][cols]; optional row weights wts[rows][cols] turn the inner mean into
 * mean_rows( sum_cols( term * wts ) )  (unbiased :647, rcgan-u :684,759; mnist/model.py:201-204);
 * dwts (may be NULL) receives weight * dL/dwts (needed for the learned confusion matrix). */
#define RCGAN_LOSS_HINGE_REAL 0
#define RCGAN_LOSS_HINGE_FAKE 1
#define RCGAN_LOSS_NEG_MEAN 2
#define RCGAN_LOSS_CE_ONES 3
#define RCGAN_LOSS_CE_ZEROS 4
int rcgan_loss_fwd_bwd(rcgan_ctx* ctx, int kind, int rows, int cols, const float* x, const float* wts,
                       float weight, float* loss_acc, float* dlogit, float* dwts);
/* weight * mean over all rows*cols of sigmoid_cross_entropy_with_logits(x, onehot(labels))
 * (perm regulariser: gan_resnet.py:693-694,782-783; mnist/model.py:218-221). */
/* The 8x8 stage of the CIFAR discriminator -- D.Block.3 .. D.Block.6, four identity-shortcut residual blocks of two 3x3
 * convolutions 128 -> 128 each (gan_resnet.py:275-328, 398-404) -- as ONE launch: a workgroup carries one image through all
 * eight layers with the activations in LDS.  16-bit activations only; x0 / outs / masks are [n][8][8][128].
 *   forward  (backward = 0): layer 2b = h_b = conv1(relu(x_b)) + bias, layer 2b+1 = x_{b+1} = x_b + conv2(relu(h_b)) + bias;
 *            outs[i] receives layer i's output (outs[7] is the stage's output, the others are what the backward pass needs).
 *   backward (backward = 1): x0 = gradient of the stage's output; layers run last to first: masks[0] belongs to
 *            block 6's conv2 (mask = h_6), masks[1] to its conv1 (mask = x_6), ...; outs[2j] = dh (gradient at
 *            conv1's output), outs[2j+1] = dx (gradient at the block's input; outs[7] is the gradient of the stage's input).
 * rcgan_dtrunk_prepare: prepared[i] = rcgan_conv_prepare layout of layer i's 3x3 128 -> 128 filter in FORWARD order (forward rows,
 * then data-gradient rows) -> frag (rcgan_dtrunk_fragment_bytes()): both directions' filters re-laid fragment-major -- the 16 bytes a
 * lane feeds one MFMA with are contiguous per wavefront and K-step, so the stage's filter loads are whole KiB blocks; one launch per
 * set of weights, shared by the forward and the backward pass. */
size_t rcgan_dtrunk_fragment_bytes(void);
int rcgan_dtrunk_prepare(rcgan_ctx* ctx, const void* const* prepared, void* frag);
int rcgan_dtrunk(rcgan_ctx* ctx, int n, int backward, const void* x0, const void* frag, const float* const* bias,
                 const void* const* masks, void* const* outs);
/* The same stage with the discriminator's relu + mean over the 8 x 8 pixels (gan_resnet.py:405-407) at its boundary.  Forward: feat
 * (optional) receives the pooled features [n][128] fp32 of the stored 16-bit output.  Backward: with feat = the gradient of the pooled
 * features and xlast = outs[7] of the forward call, the incoming gradient dy[p][c] = xlast[p][c] > 0 ? feat[c] / 64 : 0 is formed
 * inside the launch and written to dy_out [n][8][8][128] for the last layer's filter gradient (x0 may be NULL).  The projection head
 * then runs on [n][128] features without touching the activations. */
int rcgan_dtrunk_pooled(rcgan_ctx* ctx, int n, int backward, const void* x0, const void* frag, const float* const* bias,
                        const void* const* masks, void* const* outs, float* feat, const void* xlast, void* dy_out);

/* ---- register-filter convolution (csrc/conv_rf.hip) -----------------------------------------------------------------------------
 * One 3x3 stride-1 SAME layer on a small image grid (8x8, 16x16; 128 -> 128 channels), forward or data gradient, with the workgroup's
 * filter slice in registers and its input patch resident in LDS: the same values as rcgan_conv2d_fwd(_residual) / rcgan_conv2d_bwd_data
 * up to fp32 summation order (tf.nn.conv2d + bias_add, conv2d_backprop_input: cifar10/common/ops/conv2d.py:181-216 for D.Block.2.Conv1).
 * rcgan_conv_rf_ok: 1 if the kernel takes d (flags within IN_RELU | ACCUMULATE).  rcgan_conv_rf_prepare: ONE launch re-lays n prepared
 * filters (rcgan_conv_prepare layout) fragment-major into frags[i] (rcgan_conv_rf_fragment_bytes each; both directions).
 * rcgan_conv2d_rf: backward = 0: y = conv(x) (+bias) (+residual), input ReLU under IN_RELU; backward = 1: x = dy, y = dx, masked by
 * mask_x > 0 under IN_RELU, += under ACCUMULATE, + residual. */
int rcgan_conv_rf_ok(const rcgan_conv_desc* d);
size_t rcgan_conv_rf_fragment_bytes(const rcgan_conv_desc* d);
int rcgan_conv_rf_prepare(rcgan_ctx* ctx, int n, const rcgan_conv_desc* descs, const void* const* prepared, void* const* frags);
/* rcgan_dtrunk_prepare (trunk_prepared = the stage's eight prepared filters in forward order, or NULL) and rcgan_conv_rf_prepare (n <= 12)
 * as ONE launch: every fragment-major copy a critic step needs. */
int rcgan_fragments_prepare(rcgan_ctx* ctx, const void* const* trunk_prepared, void* trunk_frag, int n, const rcgan_conv_desc* descs,
                            const void* const* prepared, void* const* frags);
int rcgan_conv2d_rf(rcgan_ctx* ctx, const rcgan_conv_desc* d, int backward, const void* x, const void* frag, const float* bias,
                    const void* mask_x, const void* residual, void* y);
/* Fused projection head: pooled features -> psi (D.Output, SN linear d -> 1), label embeddings E = table @ W_e / sigma_e + b_e
 * (embedding.py:29-51 + D.Embedding_y, gan_resnet.py:414-421), logits psi + <feat, E[l]> (:588, :654-660), loss terms and ALL
 * gradients in one launch.  Rows [0, rows_a) form part a, rows [rows_a, n) part b (real | fake of the critic step, :604-606);
 * each part has its loss kind and EITHER int32 labels [rows] (one-hot weights) OR a weight matrix [rows, v] (confusion-matrix
 * rows :682-684, C^-1 rows :647) with optional d/d(weights).  ws: (v*d + n*(v+1) + (v+1)*d + 256) floats of scratch.  loss_acc += weight * (mean over the part's rows);
 * dfeat is written, the five parameter gradients are accumulated (+=); any output may be null. */
typedef struct {
  int n, d, v, e_dim;
  int rows_a, kind_a, kind_b;
  float weight;
  const int32_t* labels_a; const float* wts_a; float* dwts_a;
  const int32_t* labels_b; const float* wts_b; float* dwts_b;
  /* Optional (all zero = off): pool the features inside the head -- feat[s][j] = mean over hw of act(x[s][p][j]) (the relu +
   * reduce_mean of gan_resnet.py:405-407) from the trunk's output x [n][hw][d] (x_dtype; d % 128 == 0), and write the gradient
   * straight to dx [n][hw][d] (or NULL).  The `feat` argument is then an OUTPUT buffer [n][d]. */
  const void* x; void* dx;
  int x_dtype, hw, act;
  /* Optional: E [v][d], the label embeddings table @ W_e / sigma_e + b_e computed earlier in the step by
   * rcgan_conv_prepare_batch_embed (they depend on parameters only): the head then starts with its logit kernel. */
  const float* E_pre;
  /* Optional: DEFERRED parameter gradients.  Nothing in the backward pass waits for dw_out / db_out / dtable / dw_e / db_e, so
   * with defer_ws set (>= (n*(v+1) + (v+1)*d) floats that stay untouched until the gradients have been written) the call launches
   * the logit kernel only and leaves the two parameter-gradient launches pending in the context: the next rcgan_dtrunk backward
   * launch carries the first as extra workgroups, the next rcgan_conv2d_bwd_weight_group launch the second, and rcgan_head_flush
   * launches on the spot whatever is still pending (call it before anything reads those gradients).  Same values. */
  void* defer_ws;
  size_t defer_ws_bytes;
} rcgan_head_desc;
int rcgan_proj_head_fwd_bwd(rcgan_ctx* ctx, const rcgan_head_desc* hd, const float* feat, const float* w_out, const float* sigma_out,
                            const float* b_out, const float* table, const float* w_e, const float* sigma_e, const float* b_e,
                            float* loss_acc, float* logits, float* dfeat, float* dw_out, float* db_out, float* dtable, float* dw_e,
                            float* db_e, void* ws, size_t ws_bytes);
int rcgan_head_flush(rcgan_ctx* ctx);
int rcgan_bce_onehot_fwd_bwd(rcgan_ctx* ctx, int rows, int cols, const float* x, const int32_t* labels,
                             float weight, float* loss_acc, float* dx);
/* recover_labels objective (mnist/model.py:533-537): gen [r*ydim, pix] = one generated image per (real sample r, label y),
 * actual [r, pix], yrec [r, ydim] = softmax of the recovered label logits.
 *   loss = mean_r sum_y yrec[r,y] * mean_pix (actual[r] - gen[r,y])^2
 * Writes loss [1], dgen (same shape as gen, or NULL) and dyrec [r, ydim] (or NULL).  ws: r*ydim floats. */
int rcgan_recover_mse_fwd_bwd(rcgan_ctx* ctx, int r, int ydim, int pix, int dtype, const void* gen, const void* actual,
                              const float* yrec, float* loss, void* dgen, float* dyrec, void* ws, size_t ws_bytes);
/* C = softmax(logits) row-wise and its adjoint (gan_resnet.py:522, mnist/model.py:106). */
int rcgan_softmax_rows_fwd(rcgan_ctx* ctx, int rows, int cols, const float* logits, float* p);
int rcgan_softmax_rows_bwd(rcgan_ctx* ctx, int rows, int cols, const float* p, const float* dp, float* dlogits,
                           int accumulate);

/* ---- optimiser ---------------------------------------------------------------------------------------- */
/* tf.train.AdamOptimizer on a flat fp32 range (model.py:250-262, gan_resnet.py:802-817):
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; w -= lr_t*m/(sqrt(v)+eps); optional clip to [-clip,clip]
 * (variable constraint of mnist/ops.py:102-111; clip<=0 disables).  hyper: DEVICE float[2] = {lr, t}
 * so a captured graph can be replayed with new values.  grad_scale multiplies g first (1/world_size). */
int rcgan_adam_tf(rcgan_ctx* ctx, size_t count, float* w, const float* g, float* m, float* v,
                  const float* hyper, float beta1, float beta2, float eps, float clip, float grad_scale);
/* The same with {lr, t} as launch arguments: an eagerly launched step (every optimiser step of both training loops: the
 * all-reduce sits between the captured graph and Adam) needs no host-to-device copy of two floats in front of it. */
int rcgan_adam_tf_host(rcgan_ctx* ctx, size_t count, float* w, const float* g, float* m, float* v,
                       float lr, float t, float beta1, float beta2, float eps, float clip, float grad_scale);
int rcgan_fill_f32(rcgan_ctx* ctx, size_t count, float* p, float value);
/* dst[i] = src[i] for `count` 4-byte words (both DEVICE-accessible), as a kernel on the context's stream: the hand-over of a step's
 * packed input batch (the feed_dict of gan_resnet.py:931,938) into the static input slab the captured step reads. */
int rcgan_copy_words(rcgan_ctx* ctx, size_t count, const void* src, void* dst);

/* ---- batch statistics out of the producing convolution (tf.nn.moments of normalization.py:47 fused into conv2d.py:181-216) ----------------
 * For the convolutions the 256 x 256 eight-wave kernel takes (rcgan_conv_stats_ok: 16-bit activations, Cout = 256, whole 256-pixel tiles,
 * the 32 x 32 generator block at n >= 50), rcgan_conv2d_fwd_stats is rcgan_conv2d_fwd_residual that also leaves, per pixel tile, the column
 * sums of the STORED output (bias, residual and the 16-bit rounding included) and of its squares in tile_sums (rcgan_conv_stats_bytes);
 * rcgan_bn_stats_from_tiles turns them into mean / rstd [nseg][Cout] of nseg equal runs of samples (biased variance, as rcgan_bn_stats):
 * the statistics pass over the activation tensor (one full read) disappears.  rcgan_bn_apply_segments: the apply half of
 * rcgan_bn_fwd_segments for statistics obtained that way. */
int rcgan_conv_stats_ok(const rcgan_conv_desc* d);
size_t rcgan_conv_stats_bytes(const rcgan_conv_desc* d);
int rcgan_conv2d_fwd_stats(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias,
                           const void* residual, void* y, float* tile_sums);
int rcgan_bn_stats_from_tiles(rcgan_ctx* ctx, const rcgan_conv_desc* d, int nseg, float eps, const float* tile_sums, float* mean, float* rstd);
int rcgan_bn_apply_segments(rcgan_ctx* ctx, int nseg, int n_per_seg, int rows_per_sample, int c, int n_labels, int dtype, const void* x,
                            const int32_t* labels, const float* gamma, const float* beta, const float* mean, const float* rstd, int act,
                            void* y, void* ws, size_t ws_bytes);

/* ---- gradient (loss) scaling for 16-bit activations ------------------------------------------------------------------------
 * The reference trains in fp32 (no counterpart in gan_resnet.py); BASELINE config 5 asks for fp16 activations, whose 5 exponent
 * bits need the loss -- hence every activation gradient -- scaled up.  rcgan_set_grad_scale: from now on every GRADIENT the loss
 * kernels emit (rcgan_loss_fwd_bwd, rcgan_bce_onehot_fwd_bwd, rcgan_proj_head_fwd_bwd) is multiplied by
 * host_scale * (*dev_scale if dev_scale != NULL); the loss VALUES they accumulate stay unscaled.  dev_scale is DEVICE memory read at
 * execution time, so a captured step follows a scale that changes between replays.  Default: 1, NULL. */
int rcgan_set_grad_scale(rcgan_ctx* ctx, float host_scale, const float* dev_scale);
/* Dynamic loss scaling.  ls_state: DEVICE float[4] = {scale, applied steps since the last change, non-finite flag, skipped steps}.
 *   rcgan_grad_finite_check : raises the flag if g[0,count) holds an inf / nan (after the all-reduce: every rank sees the same sum);
 *   rcgan_adam_tf_dyn       : rcgan_adam_tf_host that (a) does nothing when the flag is raised (the step is skipped), (b) divides the
 *                             gradient by the current scale on top of grad_scale, (c) takes t = *t_dev + 1 (t_dev: DEVICE float[1],
 *                             the number of APPLIED updates of this group -- a skipped step does not advance the bias correction);
 *   rcgan_loss_scale_update : after the optimiser launches of a step: flag raised -> scale = max(scale/2, min_scale), flag cleared,
 *                             skipped++; else t_dev0 / t_dev1 (either may be NULL) += 1 and after growth_interval applied steps in
 *                             a row scale = min(2*scale, max_scale). */
int rcgan_grad_finite_check(rcgan_ctx* ctx, size_t count, const float* g, float* ls_state);
int rcgan_adam_tf_dyn(rcgan_ctx* ctx, size_t count, float* w, const float* g, float* m, float* v, float lr, const float* t_dev,
                      float beta1, float beta2, float eps, float clip, float grad_scale, const float* ls_state);
int rcgan_loss_scale_update(rcgan_ctx* ctx, float* ls_state, float* t_dev0, float* t_dev1, float growth_interval, float min_scale,
                            float max_scale);

/* ---- data-parallel gradient exchange (RCCL over xGMI) ------------------------------------------------------------------------
 * Replaces the reference's in-graph towers (cifar10/gan_resnet.py:529-546 tf.split over DEVICES, :697,786 tf.add_n(costs) / len(DEVICES)):
 * one process per GPU holds one tower; the gradients of the MEAN cost are the all-reduce SUM of the ranks' gradient buckets times
 * 1/world (applied by the optimiser kernel's grad_scale).  Every call is asynchronous on a stream and may be captured into the step's
 * hipGraph; RCGAN_ERCCL on any RCCL failure (rcgan_last_error carries ncclGetErrorString).
 *   rcgan_comm_unique_id       rank 0: 128 opaque bytes (ncclUniqueId) to hand to every rank (the host side moves them: file, TCP store, ...)
 *   rcgan_comm_init            collective over the `world` ranks: ncclCommInitRank on the context's device
 *   rcgan_comm_init_stub       single-process test double: "every rank holds what this rank holds", i.e. sum = world * x -- lets one GPU run
 *                              the whole world > 1 step schedule (buckets, side stream, in-graph optimiser) and compare it with world = 1
 *   rcgan_allreduce_sum        in place, fp32, on the context's stream
 *   rcgan_allreduce_sum_buckets  the same for several buckets as ONE RCCL group
 *   rcgan_allreduce_sum_async  on the communication stream, ordered after everything queued on the context's stream so far: the bucket must
 *                              be final; later launches of the backward pass run beside it (xGMI transfers under compute)
 *   rcgan_allreduce_join       the context's stream waits for the asynchronous buckets (before the optimiser reads them)
 *   rcgan_comm_count           the number of ranks the COMMUNICATOR reports (ncclCommCount; the test double: its world)
 *   rcgan_allreduce_sum_bf16_buckets  the same sum with the buckets travelling as bfloat16 (half the bytes over xGMI): every bucket is
 *                              rounded to bf16 (nearest even) into scratch16, summed there (ncclBfloat16) and widened back over the fp32 bucket,
 *                              three launches + one RCCL group whatever n.  scratch16: device memory, 2 bytes per float of all buckets together
 *                              (each bucket's part 256-byte aligned: rcgan_allreduce_bf16_scratch_bytes).  The sum of N bf16 values carries
 *                              8 mantissa bits: a gradient-precision trade the caller opts into
 *   rcgan_comm_stub_model      cost model of the test double: every all-reduce group then occupies its stream for
 *                              latency_us + 2 (world - 1) / world * bytes / bus_gbps (a one-thread kernel that watches the wall clock), so a
 *                              single GPU reports what a world-size-N step would take under a stated link model; 0, 0 = free */
#define RCGAN_COMM_ID_BYTES 128
int rcgan_comm_unique_id(void* id_out);
int rcgan_comm_init(rcgan_ctx* ctx, const void* id, int world, int rank);
int rcgan_comm_init_stub(rcgan_ctx* ctx, int world);
int rcgan_comm_destroy(rcgan_ctx* ctx);
int rcgan_comm_world(rcgan_ctx* ctx);
int rcgan_allreduce_sum(rcgan_ctx* ctx, float* buf, size_t count);
int rcgan_allreduce_sum_buckets(rcgan_ctx* ctx, int n, float* const* bufs, const size_t* counts);
int rcgan_allreduce_sum_async(rcgan_ctx* ctx, float* buf, size_t count);
int rcgan_allreduce_join(rcgan_ctx* ctx);
int rcgan_comm_count(rcgan_ctx* ctx, int* ranks);
size_t rcgan_allreduce_bf16_scratch_bytes(int n, const size_t* counts);
int rcgan_allreduce_sum_bf16_buckets(rcgan_ctx* ctx, int n, float* const* bufs, const size_t* counts, void* scratch16, size_t scratch_bytes);
int rcgan_comm_stub_model(rcgan_ctx* ctx, double bus_gbps, double latency_us);
/* why librccl could not be bound ("" when it was, or when nothing has tried yet) */
const char* rcgan_comm_load_error(void);
/* {a, b} -> p[0..1] by a one-thread launch on the stream: sets the DEVICE {lr, t} of rcgan_adam_tf in front of a replayed graph without a
 * host-to-device copy. */
int rcgan_set2_f32(rcgan_ctx* ctx, float* p, float a, float b);

/* ---- self test ---------------------------------------------------------------------------------------- */
/* Runs the MFMA / ds_read_b64_tr_b16 fragment-layout probes on the device (call once, outside graph
 * capture).  Fails iff the MFMA operand layout this library assumes does not hold; a failed
 * transpose-read probe only switches the filter-gradient kernel to its slower scalar LDS reads. */
int rcgan_selftest(rcgan_ctx* ctx);
#define RCGAN_QUERY_TR_READ 0   /* 1 = ds_read_b64_tr_b16 path in use, 0 = scalar fallback, -1 = not probed */
int rcgan_query(rcgan_ctx* ctx, int what);

#ifdef __cplusplus
}
#endif
#endif /* RCGAN_HIP_H */
