// Discriminator head, projection logit, RCGAN / RCGAN-U loss terms, confusion-matrix softmax.
//   reference: cifar10/gan_resnet.py:405-412 (relu + spatial mean), :588 (projection), :604-606, :647,
//   :654-660, :682-684, :751-760, :773 (losses), :522 (C = softmax(logits)), :692-695/:781-784 (perm BCE);
//   mnist/model.py:135-145, 199-221, 679-685.
// All of this is a few KB of fp32: each op is one small launch built on wavefront reductions.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void act_meanhw_fwd_kernel(int hw, int c, int act, const T* x, float* feat) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long base = (long)blockIdx.y * hw * c;
  float s = 0.f;
  if (col < c)
    for (int r = rl; r < hw; r += 4) s += act_apply(act, Elem<T>::ld(x + base + (long)r * c + col));
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (threadIdx.x < 64 && col < c)
    feat[(long)blockIdx.y * c + col] = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)hw;
}

template <typename T>
__global__ void act_meanhw_bwd_kernel(long total, int hw, int c, int act, const T* x, const float* dfeat, T* dx) {
  const float inv = 1.f / (float)hw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int ch = (int)(i % c);
    long n = i / ((long)hw * c);
    float v = dfeat[n * c + ch] * inv * act_grad(act, Elem<T>::ld(x + i));
    Elem<T>::st(dx + i, v);
  }
}

__global__ void gather_rows_kernel(int n, int d, const float* table, const int32_t* idx, float* out) {
  long total = (long)n * d;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    out[i] = table[(long)idx[i / d] * d + (i % d)];
}

// deterministic scatter-add: one thread per table element loops over the n source rows
__global__ void scatter_add_rows_kernel(int n, int d, int v, const float* src, const int32_t* idx, float* tg, int accumulate) {
  long total = (long)v * d;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int row = (int)(i / d), col = (int)(i % d);
    float s = accumulate ? tg[i] : 0.f;
    for (int r = 0; r < n; ++r)
      if (idx[r] == row) s += src[(long)r * d + col];
    tg[i] = s;
  }
}

// one wavefront per row
__global__ __launch_bounds__(256) void proj_logit_fwd_kernel(int n, int d, const float* feat, const float* psi, const float* emb, float* logit) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  float s = 0.f;
  for (int j = lane; j < d; j += 64) s += feat[(long)row * d + j] * emb[(long)row * d + j];
  s = wave_sum(s);
  if (lane == 0) logit[row] = s + (psi ? psi[row] : 0.f);
}

__global__ void proj_logit_bwd_kernel(int n, int d, const float* feat, const float* emb, const float* dlogit, float* dfeat,
                                      float* dpsi, float* demb, int acc_feat) {
  long total = (long)n * d;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long row = i / d;
    float g = dlogit[row];
    if (dfeat) { float v = g * emb[i]; if (acc_feat) v += dfeat[i]; dfeat[i] = v; }
    if (demb) demb[i] = g * feat[i];
    if (dpsi && (i % d) == 0) dpsi[row] = g;
  }
}

// logits[n][v] = psi[n] + <feat[n,:], E[v,:]>
__global__ __launch_bounds__(256) void proj_all_fwd_kernel(int n, int d, int v, const float* feat, const float* psi, const float* E, float* logits) {
  const int lane = threadIdx.x & 63;
  const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (long)n * v) return;
  const int row = (int)(item / v), lab = (int)(item % v);
  float s = 0.f;
  for (int j = lane; j < d; j += 64) s += feat[(long)row * d + j] * E[(long)lab * d + j];
  s = wave_sum(s);
  if (lane == 0) logits[item] = s + psi[row];
}

__global__ void proj_all_bwd_kernel(int n, int d, int v, const float* feat, const float* E, const float* dl, float* dfeat,
                                    float* dpsi, float* dE, int acc_feat) {
  // three independent index ranges handled by one launch
  const long t1 = (long)n * d, t2 = (long)v * d, t3 = n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < t1 + t2 + t3; i += (long)gridDim.x * blockDim.x) {
    if (i < t1) {
      int row = (int)(i / d), j = (int)(i % d);
      float s = acc_feat ? dfeat[i] : 0.f;
      for (int l = 0; l < v; ++l) s += dl[(long)row * v + l] * E[(long)l * d + j];
      dfeat[i] = s;
    } else if (i < t1 + t2) {
      long q = i - t1;
      int l = (int)(q / d), j = (int)(q % d);
      float s = 0.f;
      for (int r = 0; r < n; ++r) s += dl[(long)r * v + l] * feat[(long)r * d + j];
      dE[q] = s;
    } else {
      long r = i - t1 - t2;
      float s = 0.f;
      for (int l = 0; l < v; ++l) s += dl[r * v + l];
      dpsi[r] = s;
    }
  }
}

__device__ __forceinline__ void loss_term(int kind, float x, float* t, float* d) {
  switch (kind) {
    case RCGAN_LOSS_HINGE_REAL: { float z = 1.f - x; *t = z > 0.f ? z : 0.f; *d = z > 0.f ? -1.f : 0.f; break; }
    case RCGAN_LOSS_HINGE_FAKE: { float z = 1.f + x; *t = z > 0.f ? z : 0.f; *d = z > 0.f ? 1.f : 0.f; break; }
    case RCGAN_LOSS_NEG_MEAN: *t = -x; *d = -1.f; break;
    case RCGAN_LOSS_CE_ONES: { float sp = log1pf(expf(-fabsf(x))); *t = fmaxf(x, 0.f) - x + sp; *d = 1.f / (1.f + expf(-x)) - 1.f; break; }
    default: { float sp = log1pf(expf(-fabsf(x))); *t = fmaxf(x, 0.f) + sp; *d = 1.f / (1.f + expf(-x)); break; }
  }
}

__global__ __launch_bounds__(256) void loss_kernel(int kind, int rows, int cols, const float* x, const float* wts, float weight,
                                                    float* loss_acc, float* dlogit, float* dwts) {
  __shared__ float red[4];
  const long total = (long)rows * cols;
  const float inv_rows = 1.f / (float)rows;
  const float inv_all = 1.f / (float)total;
  float acc = 0.f;
  for (long i = threadIdx.x; i < total; i += 256) {
    float t, d;
    loss_term(kind, x[i], &t, &d);
    float wf = wts ? wts[i] * inv_rows : inv_all;
    acc += t * wf;
    if (dlogit) dlogit[i] = weight * d * wf;
    if (dwts) dwts[i] = weight * t * inv_rows;
  }
  acc = block_sum256(acc, red);
  if (threadIdx.x == 0 && loss_acc) *loss_acc += weight * acc;
}

__global__ __launch_bounds__(256) void bce_onehot_kernel(int rows, int cols, const float* x, const int32_t* labels, float weight,
                                                          float* loss_acc, float* dx) {
  __shared__ float red[4];
  const long total = (long)rows * cols;
  const float inv_all = 1.f / (float)total;
  float acc = 0.f;
  for (long i = threadIdx.x; i < total; i += 256) {
    int r = (int)(i / cols), cidx = (int)(i % cols);
    float z = labels[r] == cidx ? 1.f : 0.f;
    float v = x[i];
    acc += (fmaxf(v, 0.f) - v * z + log1pf(expf(-fabsf(v)))) * inv_all;
    if (dx) dx[i] = weight * (1.f / (1.f + expf(-v)) - z) * inv_all;
  }
  acc = block_sum256(acc, red);
  if (threadIdx.x == 0 && loss_acc) *loss_acc += weight * acc;
}

__global__ void softmax_rows_fwd_kernel(int rows, int cols, const float* l, float* p) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float m = -INFINITY;
  for (int j = 0; j < cols; ++j) m = fmaxf(m, l[r * cols + j]);
  float s = 0.f;
  for (int j = 0; j < cols; ++j) s += expf(l[r * cols + j] - m);
  for (int j = 0; j < cols; ++j) p[r * cols + j] = expf(l[r * cols + j] - m) / s;
}

__global__ void softmax_rows_bwd_kernel(int rows, int cols, const float* p, const float* dp, float* dl, int accumulate) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float dot = 0.f;
  for (int j = 0; j < cols; ++j) dot += dp[r * cols + j] * p[r * cols + j];
  for (int j = 0; j < cols; ++j) {
    float v = p[r * cols + j] * (dp[r * cols + j] - dot);
    if (accumulate) v += dl[r * cols + j];
    dl[r * cols + j] = v;
  }
}

static inline int g1(long total) {
  long b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}


// recover_labels objective (mnist/model.py:533-537): gen holds, for every real sample r, one generated image per
// label y (row r*ydim + y).  sq[r][y] = mean over pixels of (actual[r] - gen[r,y])^2;
// loss = mean_r sum_y sq[r][y] * yrec[r][y].  One workgroup per (r, y): writes its loss term, d loss / d yrec and
// d loss / d gen; the terms are summed in a fixed order by recover_sum_kernel.
template <typename T>
__global__ __launch_bounds__(256) void recover_mse_kernel(int r_count, int ydim, int pix, const T* gen, const T* actual, const float* yrec,
                                                          float* terms, T* dgen, float* dyrec) {
  __shared__ float red[4];
  const int row = blockIdx.x, r = row / ydim;
  const T* g = gen + (long)row * pix;
  const T* a = actual + (long)r * pix;
  const float w = yrec[row], inv_r = 1.f / (float)r_count, inv_p = 1.f / (float)pix;
  float s = 0.f;
  for (int i = threadIdx.x; i < pix; i += 256) {
    const float d = Elem<T>::ld(g + i) - Elem<T>::ld(a + i);
    s += d * d;
    if (dgen) Elem<T>::st(dgen + (long)row * pix + i, 2.f * d * inv_p * w * inv_r);
  }
  s = block_sum256(s, red);
  if (threadIdx.x == 0) {
    const float sq = s * inv_p;
    terms[row] = sq * w * inv_r;
    if (dyrec) dyrec[row] = sq * inv_r;
  }
}

__global__ __launch_bounds__(256) void recover_sum_kernel(int n, const float* terms, float* loss) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += terms[i];
  s = block_sum256(s, red);
  if (threadIdx.x == 0) *loss = s;
}

extern "C" {

int rcgan_act_meanhw_fwd(rcgan_ctx* ctx, int n, int hw, int c, int dtype, int act, const void* x, float* feat) {
  dim3 grid(cdiv(c, 64), n);
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(act_meanhw_fwd_kernel<T>, grid, dim3(256), 0, ctx->stream, hw, c, act, (const T*)x, feat));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_act_meanhw_bwd(rcgan_ctx* ctx, int n, int hw, int c, int dtype, int act, const void* x, const float* dfeat, void* dx) {
  long total = (long)n * hw * c;
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(act_meanhw_bwd_kernel<T>, dim3(g1(total)), dim3(256), 0, ctx->stream, total, hw, c, act, (const T*)x, dfeat, (T*)dx));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_gather_rows(rcgan_ctx* ctx, int n, int d, const float* table, const int32_t* idx, float* out) {
  hipLaunchKernelGGL(gather_rows_kernel, dim3(g1((long)n * d)), dim3(256), 0, ctx->stream, n, d, table, idx, out);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_scatter_add_rows(rcgan_ctx* ctx, int n, int d, int v, const float* src, const int32_t* idx, float* tg, int accumulate) {
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(g1((long)v * d)), dim3(256), 0, ctx->stream, n, d, v, src, idx, tg, accumulate);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_proj_logit_fwd(rcgan_ctx* ctx, int n, int d, const float* feat, const float* psi, const float* emb, float* logit) {
  hipLaunchKernelGGL(proj_logit_fwd_kernel, dim3(cdiv(n, 4)), dim3(256), 0, ctx->stream, n, d, feat, psi, emb, logit);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_proj_logit_bwd(rcgan_ctx* ctx, int n, int d, const float* feat, const float* emb, const float* dlogit, float* dfeat,
                         float* dpsi, float* demb, int acc_feat) {
  hipLaunchKernelGGL(proj_logit_bwd_kernel, dim3(g1((long)n * d)), dim3(256), 0, ctx->stream, n, d, feat, emb, dlogit, dfeat, dpsi, demb, acc_feat);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_proj_logit_all_fwd(rcgan_ctx* ctx, int n, int d, int v, const float* feat, const float* psi, const float* E, float* logits) {
  hipLaunchKernelGGL(proj_all_fwd_kernel, dim3(cdiv((long)n * v, 4)), dim3(256), 0, ctx->stream, n, d, v, feat, psi, E, logits);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_proj_logit_all_bwd(rcgan_ctx* ctx, int n, int d, int v, const float* feat, const float* E, const float* dlogits,
                             float* dfeat, float* dpsi, float* dE, int acc_feat) {
  long total = (long)n * d + (long)v * d + n;
  hipLaunchKernelGGL(proj_all_bwd_kernel, dim3(g1(total)), dim3(256), 0, ctx->stream, n, d, v, feat, E, dlogits, dfeat, dpsi, dE, acc_feat);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_loss_fwd_bwd(rcgan_ctx* ctx, int kind, int rows, int cols, const float* x, const float* wts, float weight,
                       float* loss_acc, float* dlogit, float* dwts) {
  RC_REQUIRE(ctx, kind >= 0 && kind <= RCGAN_LOSS_CE_ZEROS, "kind %d", kind);
  hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, ctx->stream, kind, rows, cols, x, wts, weight, loss_acc, dlogit, dwts);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_bce_onehot_fwd_bwd(rcgan_ctx* ctx, int rows, int cols, const float* x, const int32_t* labels, float weight,
                             float* loss_acc, float* dx) {
  hipLaunchKernelGGL(bce_onehot_kernel, dim3(1), dim3(256), 0, ctx->stream, rows, cols, x, labels, weight, loss_acc, dx);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_softmax_rows_fwd(rcgan_ctx* ctx, int rows, int cols, const float* l, float* p) {
  hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3(cdiv(rows, 64)), dim3(64), 0, ctx->stream, rows, cols, l, p);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_softmax_rows_bwd(rcgan_ctx* ctx, int rows, int cols, const float* p, const float* dp, float* dl, int accumulate) {
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(cdiv(rows, 64)), dim3(64), 0, ctx->stream, rows, cols, p, dp, dl, accumulate);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_recover_mse_fwd_bwd(rcgan_ctx* ctx, int r_count, int ydim, int pix, int dtype, const void* gen, const void* actual,
                              const float* yrec, float* loss, void* dgen, float* dyrec, void* ws, size_t ws_bytes) {
  RC_REQUIRE(ctx, r_count > 0 && ydim > 0 && pix > 0, "bad recover shape [%d,%d,%d]", r_count, ydim, pix);
  const size_t need = (size_t)r_count * ydim * sizeof(float);
  if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
  float* terms = (float*)ws;
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(recover_mse_kernel<T>, dim3(r_count * ydim), dim3(256), 0, ctx->stream, r_count, ydim, pix,
                                                   (const T*)gen, (const T*)actual, yrec, terms, (T*)dgen, dyrec));
  RC_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(recover_sum_kernel, dim3(1), dim3(256), 0, ctx->stream, r_count * ydim, (const float*)terms, loss);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

}  // extern "C"
