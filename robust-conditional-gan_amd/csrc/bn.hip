// Batch norm / conditional batch norm (+ fused activation), forward and backward.
//   reference: tf.contrib.layers.batch_norm (mnist/ops.py:38-44) and
//   tf.nn.moments + embedding_lookup + tf.nn.batch_normalization (cifar10/common/ops/normalization.py:47-57).
// HBM-bound: x is read once for the statistics and once for the apply; the backward reads (x, y, dy)
// once for the per-group sums and once for the apply.  Reductions are two-level and deterministic
// (per-row-group partials in fp32, combined in fp64) -- no atomics.
#include "common.h"

#define MAX_LABELS 16

// partial[g][0][c] = sum_rows v1, partial[g][1][c] = sum_rows v2 over row group g (rows_per_group rows)
// MODE 0: (x, x*x)         MODE 1: (dy', dy'*xhat) with dy' = dy*act'(y), xhat = (x-mean)*rstd
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(long rows, int c, long rows_per_group, const T* x, const T* y,
                                                         const T* dy, const float* mean, const float* rstd, int act,
                                                         float* partial) {
  __shared__ float red[2][4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long rb = (long)blockIdx.y * rows_per_group;
  long re = rb + rows_per_group;
  if (re > rows) re = rows;
  float s1 = 0.f, s2 = 0.f;
  if (col < c) {
    float mu = 0.f, rs = 1.f;
    if (MODE == 1) { mu = mean[col]; rs = rstd[col]; }
    for (long r = rb + rl; r < re; r += 4) {
      const long off = r * c + col;
      if (MODE == 0) {
        float v = Elem<T>::ld(x + off);
        s1 += v; s2 += v * v;
      } else {
        float g = Elem<T>::ld(dy + off);
        if (act != RCGAN_ACT_NONE) g *= act_grad(act, Elem<T>::ld(y + off));
        float xh = (Elem<T>::ld(x + off) - mu) * rs;
        s1 += g; s2 += g * xh;
      }
    }
  }
  red[0][rl][threadIdx.x & 63] = s1;
  red[1][rl][threadIdx.x & 63] = s2;
  __syncthreads();
  if (threadIdx.x < 64 && col < c) {
    const int t = threadIdx.x;
    partial[((long)blockIdx.y * 2 + 0) * c + col] = red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t];
    partial[((long)blockIdx.y * 2 + 1) * c + col] = red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t];
  }
}

__global__ void bn_stats_finalize_kernel(int c, int ngroups, long rows, const float* partial, float eps, float* mean,
                                         float* rstd, float* mm, float* mv, float decay) {
  int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= c) return;
  double s1 = 0.0, s2 = 0.0;
  for (int g = 0; g < ngroups; ++g) {
    s1 += (double)partial[((long)g * 2 + 0) * c + col];
    s2 += (double)partial[((long)g * 2 + 1) * c + col];
  }
  double mu = s1 / (double)rows;
  double var = s2 / (double)rows - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[col] = (float)mu;
  rstd[col] = (float)(1.0 / sqrt(var + (double)eps));
  if (mm) {
    // TF fused batch norm: moving -= (moving - batch) * (1 - decay), variance with Bessel's correction
    double uvar = rows > 1 ? var * ((double)rows / (double)(rows - 1)) : var;
    float om = 1.f - decay;
    mm[col] = mm[col] - (mm[col] - (float)mu) * om;
    mv[col] = mv[col] - (mv[col] - (float)uvar) * om;
  }
}

template <typename T>
__global__ void bn_apply_fwd_kernel(long total, int rows_per_sample, int c, const T* x, const int32_t* labels,
                                    const float* gamma, const float* beta, const float* mean, const float* rstd, int act, T* y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int ch = (int)(i % c);
    long row = i / c;
    int l = labels ? labels[row / rows_per_sample] : 0;
    // tf.nn.batch_normalization: inv = rsqrt(var+eps)*scale; y = x*inv + (offset - mean*inv)
    float inv = rstd[ch] * gamma[(long)l * c + ch];
    float v = Elem<T>::ld(x + i) * inv + (beta[(long)l * c + ch] - mean[ch] * inv);
    Elem<T>::st(y + i, act_apply(act, v));
  }
}

// combine the per-group backward partials: dgamma/dbeta per label and the two batch-wide sums
__global__ void bn_bwd_combine_kernel(int c, int ngroups, int n_labels, const int32_t* group_labels, const float* gamma,
                                      const float* partial, float* dgamma, float* dbeta, float* s12, int accumulate) {
  int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= c) return;
  double db[MAX_LABELS], dg[MAX_LABELS];
  for (int l = 0; l < MAX_LABELS; ++l) { db[l] = 0.0; dg[l] = 0.0; }
  for (int g = 0; g < ngroups; ++g) {
    int l = group_labels ? group_labels[g] : 0;
    db[l] += (double)partial[((long)g * 2 + 0) * c + col];
    dg[l] += (double)partial[((long)g * 2 + 1) * c + col];
  }
  double s1 = 0.0, s2 = 0.0;
  for (int l = 0; l < n_labels; ++l) {
    double gm = (double)gamma[(long)l * c + col];
    s1 += gm * db[l];
    s2 += gm * dg[l];
    float og = (float)dg[l], ob = (float)db[l];
    if (accumulate) { og += dgamma[(long)l * c + col]; ob += dbeta[(long)l * c + col]; }
    dgamma[(long)l * c + col] = og;
    dbeta[(long)l * c + col] = ob;
  }
  s12[col] = (float)s1;
  s12[c + col] = (float)s2;
}

template <typename T>
__global__ void bn_bwd_apply_kernel(long total, long rows, int rows_per_sample, int c, const T* x, const T* y, const T* dy,
                                    const int32_t* labels, const float* gamma, const float* mean, const float* rstd,
                                    const float* s12, int act, T* dx, int accumulate_dx) {
  const float invM = 1.f / (float)rows;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int ch = (int)(i % c);
    long row = i / c;
    int l = labels ? labels[row / rows_per_sample] : 0;
    float g = Elem<T>::ld(dy + i);
    if (act != RCGAN_ACT_NONE) g *= act_grad(act, Elem<T>::ld(y + i));
    float rs = rstd[ch];
    float xh = (Elem<T>::ld(x + i) - mean[ch]) * rs;
    float dxh = g * gamma[(long)l * c + ch];
    float v = rs * (dxh - s12[ch] * invM - xh * s12[c + ch] * invM);
    if (accumulate_dx) v += Elem<T>::ld(dx + i);
    Elem<T>::st(dx + i, v);
  }
}

template <typename T>
__global__ void bn_infer_kernel(long total, int c, const T* x, const float* gamma, const float* beta, const float* mm,
                                const float* mv, float eps, int act, T* y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int ch = (int)(i % c);
    float v = (Elem<T>::ld(x + i) - mm[ch]) / sqrtf(mv[ch] + eps) * gamma[ch] + beta[ch];
    Elem<T>::st(y + i, act_apply(act, v));
  }
}

static inline long stats_group_rows(long rows) {
  long g = 512;
  while (rows / g > 2048) g *= 2;
  return g;
}

static inline int ew_grid2(long total) {
  long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" {

size_t rcgan_bn_workspace_bytes(int rows, int c) {
  // stats: ngroups*2*c ; bwd: ngroups*2*c + 2*c, with ngroups <= max(rows/512, n) <= rows
  long ng = (rows + 511) / 512 + 1;
  if (ng < 4096) ng = 4096;   // per-sample grouping (n <= 4096 samples)
  return (size_t)(ng * 2 * (long)c + 2 * (long)c) * sizeof(float) + 256;
}

int rcgan_bn_stats(rcgan_ctx* ctx, int rows, int c, int dtype, const void* x, float eps, float* mean, float* rstd,
                   float* mm, float* mv, float decay, void* ws, size_t ws_bytes) {
  long rpg = stats_group_rows(rows);
  int ng = cdiv(rows, rpg);
  size_t need = (size_t)ng * 2 * c * sizeof(float);
  if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
  float* partial = (float*)ws;
  dim3 grid(cdiv(c, 64), ng);
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL((bn_partial_kernel<T, 0>), grid, dim3(256), 0, ctx->stream, (long)rows, c, rpg,
                                                   (const T*)x, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr,
                                                   (const float*)nullptr, 0, partial));
  RC_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(cdiv(c, 256)), dim3(256), 0, ctx->stream, c, ng, (long)rows,
                     (const float*)partial, eps, mean, rstd, mm, mv, decay);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_bn_apply_fwd(rcgan_ctx* ctx, int n, int rows_per_sample, int c, int dtype, const void* x, const int32_t* labels,
                       const float* gamma, const float* beta, const float* mean, const float* rstd, int act, void* y) {
  long total = (long)n * rows_per_sample * c;
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(bn_apply_fwd_kernel<T>, dim3(ew_grid2(total)), dim3(256), 0, ctx->stream, total,
                                                   rows_per_sample, c, (const T*)x, labels, gamma, beta, mean, rstd, act, (T*)y));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_bn_bwd(rcgan_ctx* ctx, int n, int rows_per_sample, int c, int n_labels, int dtype, const void* x, const void* y,
                 const void* dy, const int32_t* labels, const float* gamma, const float* mean, const float* rstd, int act,
                 void* dx, int accumulate_dx, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes) {
  RC_REQUIRE(ctx, n_labels >= 1 && n_labels <= MAX_LABELS, "n_labels %d", n_labels);
  RC_REQUIRE(ctx, labels != nullptr || n_labels == 1, "labels required for n_labels > 1");
  long rows = (long)n * rows_per_sample;
  long rpg;
  int ng;
  if (labels) { rpg = rows_per_sample; ng = n; }          // one group per sample: group label = sample label
  else { rpg = stats_group_rows(rows); ng = cdiv(rows, rpg); }
  size_t need = ((size_t)ng * 2 * c + 2 * (size_t)c) * sizeof(float);
  if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
  float* partial = (float*)ws;
  float* s12 = partial + (size_t)ng * 2 * c;
  dim3 grid(cdiv(c, 64), ng);
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL((bn_partial_kernel<T, 1>), grid, dim3(256), 0, ctx->stream, rows, c, rpg,
                                                   (const T*)x, (const T*)y, (const T*)dy, mean, rstd, act, partial));
  RC_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(bn_bwd_combine_kernel, dim3(cdiv(c, 128)), dim3(128), 0, ctx->stream, c, ng, n_labels, labels, gamma,
                     (const float*)partial, dgamma, dbeta, s12, accumulate);
  RC_LAUNCH_CHECK(ctx);
  long total = rows * c;
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(ew_grid2(total)), dim3(256), 0, ctx->stream, total, rows,
                                                   rows_per_sample, c, (const T*)x, (const T*)y, (const T*)dy, labels, gamma, mean,
                                                   rstd, (const float*)s12, act, (T*)dx, accumulate_dx));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_bn_infer(rcgan_ctx* ctx, int rows, int c, int dtype, const void* x, const float* gamma, const float* beta,
                   const float* mm, const float* mv, float eps, int act, void* y) {
  long total = (long)rows * c;
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(bn_infer_kernel<T>, dim3(ew_grid2(total)), dim3(256), 0, ctx->stream, total, c,
                                                   (const T*)x, gamma, beta, mm, mv, eps, act, (T*)y));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

}  // extern "C"
