// Argument blocks shared by conv_mfma.hip (kernels) and api.hip (dispatch).
#pragma once
#include "common.h"

struct ImgWArgs;    // conv_image.h
struct ImgWGroup;

struct MfmaConvArgs {
  const bf16_t* in;       // [N][H(/2)][W(/2)][Cin]
  const bf16_t* wt;       // [Cout][T*Cin]
  const float* bias;      // [Cout] or null
  const bf16_t* mask;     // [M][Cout] or null: output zeroed where mask <= 0 (ReLU backward)
  const bf16_t* resid;    // [M][Cout] or null: added to the output (residual connection, gan_resnet.py:328)
  int resid_up;           // the residual lives on the HALF-resolution grid [N][H/2][W/2][Cout] and is added nearest-upsampled (lw, lh >= 1)
  bf16_t* out;            // [M][Cout]
  const bf16_t* zero;     // >= 16 zero bytes (halo source of the direct-to-LDS loader)
  int N, H, W, Cin, Cout, KH, KW, PT, PL;
  int up, relu_in, accumulate;
  int cm;                 // eight-wave kernels: channel-major K order (all taps of a 64-channel chunk, then the next chunk)
  int lw, lh;             // log2(W), log2(H) when both are powers of two (pixel decode by shifts), else -1
  long M;
  unsigned long long* stamps;   // diagnostics (rcgan_debug_stamps), normally null
  // Sub-pixel form of a 3x3 convolution behind the nearest 2x upsample (eight-wave kernels): output pixel (2i + ph, 2j + pw)
  // only sees the 2x2 source pixels (i + a - 1 + ph, j + b - 1 + pw), a, b in {0, 1}, each with the SUM of the filter taps that
  // fall on it -- four 2x2 convolutions over the low-resolution grid, 4/9 of the multiply-adds.  wph: the four summed filters
  // [phase = ph*2 + pw][Cout][(a*2 + b)*Cin + ci] (conv_prepare_phase_kernel), phase = 1: use them.  phase = 2: the data
  // gradient of that form, wph = [Cin][(u*4 + v)*Cout + co] (16 taps at source stride 2 over the full-resolution dy).
  const bf16_t* wph;
  int phase;
  // eight-wave 256 x 256 kernel only: per-tile column sums of the stored output and of its squares, [pixel tile][Cout][2] fp32 --
  // the batch-norm statistics of the layer behind this convolution come out of its epilogue (bn.hip: bn_tile_stats_finish_kernel)
  float* stats;
  // halo-patch kernels (conv_mfma8h.hip) only: the input is act(cond_batch_norm(in)) -- the affine + ReLU of the batch norm IN FRONT of
  // this convolution applied to the staged patch in LDS, once per staged element, the normalised tensor never written (forward-only
  // passes: rcgan_conv2d_fwd_bn_residual).  mean / rstd [segments][Cin], gamma / beta [labels][Cin], labels [N] or null, bn_seg_samples
  // images per segment; bn_act RCGAN_ACT_NONE or RCGAN_ACT_RELU.  bn_mean == nullptr: off.
  const float* bn_mean = nullptr;
  const float* bn_rstd = nullptr;
  const float* bn_gamma = nullptr;
  const float* bn_beta = nullptr;
  const int32_t* bn_labels = nullptr;
  int bn_seg_samples = 1, bn_act = 0;
};

struct MfmaWgradArgs {
  const bf16_t* x;        // [N][H(/2)][W(/2)][Cin]
  const bf16_t* dy;       // [M][Cout]
  float* slab;            // [nz][slab_stride]: T*Cin*Cout filter-gradient partials (+ Cout bias-gradient partials)
  long slab_stride;       // floats between the slabs of two pixel chunks
  int want_bias;          // also emit column sums of dy (bias gradient) at slab[T*Cin*Cout ..]
  const bf16_t* zero;     // >= 16 zero bytes (halo / tail source of the direct-to-LDS loader)
  int N, H, W, Cin, Cout, KH, KW, PT, PL;
  int up, relu_in, use_tr;
  int lw, lh;
  long M, m_chunk;
  // Sub-pixel form of the filter gradient of an upsample-3x3 (sub = 1) or ConvMeanPool (sub = 2) layer, three-tap kernel only.
  // The reduction index runs over the LOW-resolution grid (N, H, W, lw, lh, M describe it; up = 0) and one operand is gathered
  // with stride 2 at a parity (pa, pb):
  //   sub 1: G[pa][pb][s][d] = sum_m x[n, i + dh, j + dw] * dy[n, 2i + pa, 2j + pb],   dh = s - 1 + pa, dw = d - 1 + pb
  //          (x = the stored low-resolution input, dy on the full-resolution grid)
  //   sub 2: G[pa][pb][s][d] = sum_m x[n, 2(i + dh) + pa, 2(j + dw) + pb] * dy[n, i, j],   dh = s - pa, dw = d - pb
  //          (x on the full-resolution grid, dy = the POOLED gradient; dW carries the pool's 1/4)
  // 16 matrices of Cin x Cout over M pixels instead of 9 over 4M: 4/9 of the multiply-adds.  The slab holds the 16 cells
  // [(pa*2 + pb)*4 + s*2 + d]; the slab reduction folds them into the 9 taps: dW[kh][kw] = scale * sum over the four parities of
  // G[pa][pb][S(pa, kh)][S(pb, kw)], S(sub 1) = {0: 0,1,1; 1: 0,0,1}, S(sub 2) = {0: 0,0,1; 1: 0,1,1}.
  //   sub 3: a small 1x1 filter (a down block's shortcut) on the two-tap body -- x and dy as they are, one tile row, the second
  //          tap idle: its filter gradient runs inside the grouped launch of the pass instead of a 16 us launch of its own
  int sub;
  int cells;              // filter cells per slab in front of the bias tail: KH*KW, or 16 in the sub-pixel form
};
// tile rows of the three-tap kernel's grid / multiply-adds per pixel and channel pair: reference formulation and executed
static inline int wgrad3_rows(const MfmaWgradArgs& a) { return a.sub == 3 ? 1 : (a.sub ? 8 : a.KH); }
static inline int wgrad3_alg_taps(const MfmaWgradArgs& a) { return a.sub == 3 ? 1 : (a.sub ? 36 : a.KH * a.KW); }
static inline int wgrad3_exec_taps(const MfmaWgradArgs& a) { return a.sub == 3 ? 1 : (a.sub ? 16 : a.KH * a.KW); }

static inline int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

bool mfma_eligible(const rcgan_conv_desc* d);
bool mfma_wgrad_eligible(const rcgan_conv_desc* d);
bool mfma_phase_filters(const rcgan_conv_desc* d);
bool mfma_phase_dgrad_ok(const rcgan_conv_desc* d);
int conv_prepare_phase_launch(rcgan_ctx* ctx, int n, const float* const* ws, const float* const* sigmas, bf16_t* const* outs, const int* cins, const int* couts,
                              const int* kinds);
bool mfma_pool_ok(const rcgan_conv_desc* d);
int mfma_conv_launch(rcgan_ctx* ctx, const MfmaConvArgs& a);
int bn_tile_stats_finish_launch(rcgan_ctx* ctx, const float* part, int c, int nseg, int tiles_per_seg, int ngroups, long group_stride,
                                double count, float eps, float* mean, float* rstd);      // bn.hip
bool mfma_conv_is_p8(const MfmaConvArgs& a);              // routed to the 256 x 256 eight-wave kernel
bool mfma_conv8_phase_form(const MfmaConvArgs& a);        // ... which evaluates it in the sub-pixel (phase-major tile) form
int mfma_wgrad_splits(const rcgan_conv_desc* d, long M);
bool mfma_wgrad3_plan(MfmaWgradArgs& a, int nz, unsigned* gx, unsigned* gy, long px_per_block);
bool mfma_wgrad3_takes(const MfmaWgradArgs& a);
int mfma_wgrad_sub_kind(const rcgan_conv_desc* d, int use_tr);
int mfma_wgrad_sub_splits(const rcgan_conv_desc* d, long M);
int mfma_wgrad3_group_launch(rcgan_ctx* ctx, int n, const MfmaWgradArgs* args, const unsigned* gx, const unsigned* gy, int family,
                             const ImgWGroup* img, bool carry_head = false);
bool mfma_wgrad_tap_plan(MfmaWgradArgs& a, int nz, unsigned* gx, unsigned* gy);
// conv_wgrad9.hip: all nine taps of a plain 3x3 layer in one workgroup (dy and x staged once for the three filter rows)
#define WGRAD9_GROUP_MAX 12
bool mfma_wgrad9_takes(const MfmaWgradArgs& a);
bool mfma_wgrad9_plan(MfmaWgradArgs& a, int nz, unsigned* gx, unsigned* gy, long px_per_block);
int mfma_wgrad9_group_launch(rcgan_ctx* ctx, int n, const MfmaWgradArgs* args, const unsigned* gx, const unsigned* gy);
int mfma_wgrad_launch(rcgan_ctx* ctx, MfmaWgradArgs& a, int nz, bool* bias_done, size_t ws_bytes);
int mfma_prepare_launch(rcgan_ctx* ctx, const float* w, const float* sigma, bf16_t* wt, bf16_t* wd, int T, int Cin, int Cout);
int direct_prepare_launch(rcgan_ctx* ctx, const float* w, const float* sigma, float* out, long total);
int mfma_selftest(rcgan_ctx* ctx, int* host_result);
// image-end kernels (conv_image.hip): bf16 convs with a <= 3-channel side
#define WGRAD_GROUP_MAX_HOST 12   /* = WGRAD_GROUP_MAX of conv_mfma.hip */
int img_side(const rcgan_conv_desc* d);          // 0: not taken; 1: cin small; 2: cout small
size_t img_extra_offset(const rcgan_conv_desc* d);
size_t img_extra_bytes(const rcgan_conv_desc* d);
size_t img_wgrad_ws_bytes(const rcgan_conv_desc* d);
int img_prepare_launch(rcgan_ctx* ctx, const rcgan_conv_desc* d, const float* w, const float* sigma, void* prepared);
int img_fwd(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias, void* y);
int img_dgrad(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* dy, const void* prepared, void* dx, int accumulate);
bool img_fwd_bn_ok(const rcgan_conv_desc* d);
int img_fwd_bn(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias, void* y,
               const float* mean, const float* rstd, const float* gamma, const float* beta, const int32_t* labels, int segments, int act);
int img_wgrad_plan(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* dy, int target_wgs, ImgWArgs* a, int* cb, int* nwg);
int img_wgrad(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, int accumulate,
              void* ws, size_t ws_bytes);
int mfma_conv8_launch(rcgan_ctx* ctx, const MfmaConvArgs& a, bool wide, bool halo_patch);      // conv_mfma8.hip: 256 x 256 / 256 x 128 tiles, 8 wavefronts (halo_patch: conv_mfma8h.hip)
bool mfma_conv8_halo_takes(const MfmaConvArgs& a);                             // conv_mfma8h.hip: 256 x 256 tile, pixel operand as an LDS patch
int mfma_conv8_halo_launch(rcgan_ctx* ctx, const MfmaConvArgs& a);
bool mfma_conv8n_halo_takes(const MfmaConvArgs& a);                            // ... its 256 x 128-tile sibling (Cout % 128 == 0)
int mfma_conv8n_halo_launch(rcgan_ctx* ctx, const MfmaConvArgs& a);
int mfma_conv_bn_route(const MfmaConvArgs& a);                                 // would mfma_conv_launch run it on a halo-patch kernel?  1: 256 x 256, 2: 256 x 128, 0: no
struct SmallGemmArgs;
struct StepInputsArgs;
int conv_prepare_batch_launch(rcgan_ctx* ctx, const rcgan_prepare_item* items, int n, const SmallGemmArgs* gemm = nullptr,
                              const StepInputsArgs* inputs = nullptr, const rcgan_frag_item* frags = nullptr, int n_frags = 0);
