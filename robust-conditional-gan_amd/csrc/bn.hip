// Batch norm / conditional batch norm (+ fused activation), forward and backward.
//   reference: tf.contrib.layers.batch_norm (mnist/ops.py:38-44) and
//   tf.nn.moments + embedding_lookup + tf.nn.batch_normalization (cifar10/common/ops/normalization.py:47-57).
// HBM-bound: x is read once for the statistics and once for the apply; the backward reads (x, y, dy)
// once for the per-group sums and once for the apply.  Reductions are two-level and deterministic
// (per-row-group partials in fp32, combined in fp64) -- no atomics.
#include "common.h"
#include "conv_mfma.h"

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

// dx (=|+=) dy * act'(y) * gamma / sqrt(moving_var + eps): the adjoint of bn_infer w.r.t. its input (statistics and
// affine parameters are constants of the frozen sampler, mnist/model.py:494-640 recover_labels)
template <typename T>
__global__ void bn_infer_bwd_kernel(long total, int c, const T* y, const T* dy, const float* gamma, const float* mv, float eps, int act,
                                    T* dx, int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    float g = Elem<T>::ld(dy + i);
    if (act != RCGAN_ACT_NONE) g *= act_grad(act, Elem<T>::ld(y + i));
    float v = g * gamma[ch] / sqrtf(mv[ch] + eps);
    if (accumulate) v += Elem<T>::ld(dx + i);
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


// ------------------------------------------------------------------------------------------------
// 8-wide vectorised variants (c % 8 == 0): one 16-B (bf16) / two 16-B (fp32) accesses per thread per row
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ld8(const float* p, float* v) {
  float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16_t* p, float* v) {
  uint4 a = *(const uint4*)p;
  uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[2 * j] = bf16_to_f32((bf16_t)(w[j] & 0xffff)); v[2 * j + 1] = bf16_to_f32((bf16_t)(w[j] >> 16)); }
}
__device__ __forceinline__ void st8(float* p, const float* v) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
  *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void st8(bf16_t* p, const float* v) {
  uint4 pk;
  pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  pk.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
  pk.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
  *(uint4*)p = pk;
}

// block = 32 channel-chunks (256 channels) x 8 row lanes; grid = (ceil(chunks/32), groups)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_partial_vec_kernel(long rows, int c, long rows_per_group, const T* x, const T* y,
                                                             const T* dy, const float* mean, const float* rstd, int act,
                                                             float* partial) {
  __shared__ float red[2][8][256];
  const int chunk = blockIdx.x * 32 + (threadIdx.x & 31);
  const int rl = threadIdx.x >> 5;
  const bool on = chunk * 8 < c;
  const long rb = (long)blockIdx.y * rows_per_group;
  long re = rb + rows_per_group;
  if (re > rows) re = rows;
  float s1[8], s2[8], mu[8], rs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; mu[j] = 0.f; rs[j] = 1.f; }
  if (on) {
    if (MODE == 1) { ld8(mean + chunk * 8, mu); ld8(rstd + chunk * 8, rs); }
#pragma unroll 2
    for (long r = rb + rl; r < re; r += 8) {
      const long off = r * c + chunk * 8;
      float xv[8];
      ld8(x + off, xv);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += xv[j]; s2[j] += xv[j] * xv[j]; }
      } else {
        float gv[8], yv[8];
        ld8(dy + off, gv);
        if (act != RCGAN_ACT_NONE) {
          ld8(y + off, yv);
#pragma unroll
          for (int j = 0; j < 8; ++j) gv[j] *= act_grad(act, yv[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += gv[j]; s2[j] += gv[j] * (xv[j] - mu[j]) * rs[j]; }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[0][rl][(threadIdx.x & 31) * 8 + j] = s1[j];
    red[1][rl][(threadIdx.x & 31) * 8 + j] = s2[j];
  }
  __syncthreads();
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col < c) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { a += red[0][q][threadIdx.x]; b += red[1][q][threadIdx.x]; }
    partial[((long)blockIdx.y * 2 + 0) * c + col] = a;
    partial[((long)blockIdx.y * 2 + 1) * c + col] = b;
  }
}

// per-(label, channel) affine of the forward: A = rstd*gamma, B = beta - mean*A   (tf.nn.batch_normalization)
__global__ void bn_table_fwd_kernel(int n_labels, int c, const float* gamma, const float* beta, const float* mean,
                                    const float* rstd, float* A, float* B) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_labels * c) return;
  int ch = i % c;
  float a = rstd[ch] * gamma[i];
  A[i] = a;
  B[i] = beta[i] - mean[ch] * a;
}

// backward constants: A = rstd*gamma [labels][c];  dx = A*g + P*x + Q with P = -rstd^2*s2/M, Q = -P*mean - rstd*s1/M
__global__ void bn_table_bwd_kernel(int n_labels, int c, long rows, const float* gamma, const float* mean, const float* rstd,
                                    const float* s12, float* A, float* PQ) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_labels * c) A[i] = rstd[i % c] * gamma[i];
  if (i < c) {
    const float invM = 1.f / (float)rows;
    float rs = rstd[i];
    float p = -rs * rs * s12[c + i] * invM;
    PQ[i] = p;
    PQ[c + i] = -p * mean[i] - rs * s12[i] * invM;
  }
}

template <typename T>
__global__ void bn_apply_vec_kernel(long nchunks, int rows_per_sample, int c, const T* x, const int32_t* labels,
                                    const float* A, const float* B, int act, T* y) {
  const int cpr = c / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr) * 8;
    const long row = i / cpr;
    const int l = labels ? labels[row / rows_per_sample] : 0;
    float xv[8], a[8], b[8];
    ld8(x + row * c + ch, xv);
    ld8(A + (long)l * c + ch, a);
    ld8(B + (long)l * c + ch, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = act_apply(act, xv[j] * a[j] + b[j]);
    st8(y + row * c + ch, xv);
  }
}

template <typename T>
__global__ void bn_bwd_apply_vec_kernel(long nchunks, int rows_per_sample, int c, const T* x, const T* y, const T* dy,
                                        const int32_t* labels, const float* A, const float* PQ, int act, T* dx, int accumulate_dx) {
  const int cpr = c / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr) * 8;
    const long row = i / cpr;
    const int l = labels ? labels[row / rows_per_sample] : 0;
    const long off = row * c + ch;
    float xv[8], gv[8], a[8], p[8], q[8];
    ld8(x + off, xv);
    ld8(dy + off, gv);
    if (act != RCGAN_ACT_NONE) {
      float yv[8];
      ld8(y + off, yv);
#pragma unroll
      for (int j = 0; j < 8; ++j) gv[j] *= act_grad(act, yv[j]);
    }
    ld8(A + (long)l * c + ch, a);
    ld8(PQ + ch, p);
    ld8(PQ + c + ch, q);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = a[j] * gv[j] + p[j] * xv[j] + q[j];
    if (accumulate_dx) {
      float d[8];
      ld8(dx + off, d);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += d[j];
    }
    st8(dx + off, o);
  }
}

// ------------------------------------------------------------------------------------------------
// Fused variants (c % 64 == 0): statistics and their combination in ONE launch, affine tables folded into
// the apply kernels -> 2 launches per batch norm (was 4) in each direction.
//   * grid (c/64, groups); block = 8 channel chunks (64 channels) x 32 row lanes
//   * every workgroup writes its group's partial sums with agent-scope stores, then bumps an arrival counter of
//     its column block; the LAST arrival of a column block combines all groups of these 64 channels in a fixed
//     order (deterministic, fp64) and resets the counter for the next launch / graph replay.  Agent-scope
//     accesses go to the memory side, so no workgroup pays an L2 write-back/invalidate (8 XCDs = 8 L2s).
// ------------------------------------------------------------------------------------------------
// The affine part of batch norm as ONE fixed sequence of fp32 operations -- inv = rstd*gamma, c0 = fma(-mean, inv, beta),
// pre = fma(x, inv, c0) -- shared by the forward apply kernel and the backward kernels: the backward pass can then recompute
// the sign of the pre-activation (the ReLU / leaky-ReLU mask) from x bit-exactly instead of reading y a second and third time
// (y = act(pre) rounded to 16 bits has the sign of pre: bf16 / fp16 rounding keeps it, underflow to zero aside).
__device__ __forceinline__ float bn_pre(float x, float inv, float c0) { return __fmaf_rn(x, inv, c0); }
// ... rounded the way the forward stored act(pre): the recomputed mask is then bit-identical to one read from y, including the
// pre-activations that underflow to zero in fp16 (0 < pre < 2^-25; seen once in 4M elements in the parity tests)
template <typename T> __device__ __forceinline__ float bn_stored(float v);
template <> __device__ __forceinline__ float bn_stored<float>(float v) { return v; }
template <> __device__ __forceinline__ float bn_stored<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }
__device__ __forceinline__ float bn_c0(float mean, float inv, float beta) { return __fmaf_rn(-mean, inv, beta); }

__device__ __forceinline__ float ld_sc1_f1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, soff, 16));
}

struct BnFusedArgs {
  long rows; int c; long rows_per_group; int ngroups;
  int nseg;                             // forward only: blockIdx.z = segment; x, partial, counter, mean, rstd advance per segment
  const void *x, *y, *dy;
  const float *mean_in, *rstd_in;       // backward only
  int act;
  float* partial;                       // [ngroups][2][c]
  unsigned* counter;                    // [c/64]
  // forward finish
  float eps; float *mean, *rstd, *mm, *mv; float decay;
  // backward finish
  int n_labels; int groups_per_sample; const int32_t* labels; int n_samples;
  const float* gamma; float *dgamma, *dbeta; int accumulate; float* PQ;
  const float* beta;                    // backward, optional: with it the activation mask is recomputed from x (no read of y)
  int nsub;                             // conditional backward, samples of < 32 rows: a workgroup takes 32 rows = nsub samples (0, 1: off)
};

template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_fused_reduce_kernel(BnFusedArgs a) {
  // dynamic LDS: the row buffer red[2][32][64] of the streaming phase; the finisher of the conditional backward reuses it as
  // lacc[2][n_labels][4][64] (per-label sums, private per (lane4, column)).  Sharing the storage (16 KiB, 20 KiB with ten labels,
  // instead of 60 KiB of static arrays) is what lets six workgroups instead of two stay on a CU: this kernel is a latency chain
  // (load -> LDS -> write-through store -> arrival), so its time is rounds x chain.
  extern __shared__ float uni[];
  __shared__ double fin[2][4][64];
  __shared__ float lab_s[2][64];
  __shared__ int is_last;
  __shared__ int lab_ids[128];
#define RED(k, q, col) uni[((k) * 32 + (q)) * 64 + (col)]
#define LACC(k, l, ln, col) uni[(((k) * NL + (l)) * 4 + (ln)) * 64 + (col)]
  const int NL = a.n_labels;
  const int nsub = a.nsub > 1 ? a.nsub : 1;     // samples per workgroup when a sample has fewer than 32 rows
  const int t = threadIdx.x;
  const int c = a.c;
  const int c0 = blockIdx.x * 64;
  const int chunk = t & 7, rl = t >> 3;
  const long rb = (long)blockIdx.y * a.rows_per_group;
  long re = rb + a.rows_per_group;
  if (re > a.rows) re = a.rows;
  if (MODE == 0 && a.nseg > 1) {        // independent statistics per segment of a.rows rows (batched generator forward)
    const long sg = blockIdx.z;
    a.x = (const T*)a.x + sg * a.rows * c;
    a.partial += sg * (long)a.ngroups * 2 * c;
    a.counter += sg * (c / 64) * RC_LINE_STRIDE;
    a.mean += sg * c; a.rstd += sg * c;
  }
  const T* x = (const T*)a.x; const T* y = (const T*)a.y; const T* dy = (const T*)a.dy;
  float s1[8], s2[8], mu[8], rs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; mu[j] = 0.f; rs[j] = 1.f; }
  float ainv[8], ac0[8];
  const bool mask_x = MODE == 1 && a.beta != nullptr;
  constexpr int UNR = MODE == 0 ? 4 : 2;
  if (MODE == 1) { ld8(a.mean_in + c0 + chunk * 8, mu); ld8(a.rstd_in + c0 + chunk * 8, rs); }
  if (mask_x) {                           // the group's rows belong to one sample: one label
    const int smp = nsub > 1 ? (int)blockIdx.y * nsub + rl / (32 / nsub) : (int)blockIdx.y / a.groups_per_sample;
    const long lo = (long)(a.labels ? a.labels[smp] : 0) * c + c0 + chunk * 8;
    float gm8[8], bt8[8];
    ld8(a.gamma + lo, gm8); ld8(a.beta + lo, bt8);
#pragma unroll
    for (int j = 0; j < 8; ++j) { ainv[j] = rs[j] * gm8[j]; ac0[j] = bn_c0(mu[j], ainv[j], bt8[j]); }
  }
#pragma unroll UNR
  for (long r = rb + rl; r < re; r += 32) {
    const long off = r * c + c0 + chunk * 8;
    float xv[8];
    ld8(x + off, xv);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += xv[j]; s2[j] += xv[j] * xv[j]; }
    } else {
      float gv[8];
      ld8(dy + off, gv);
      if (mask_x) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] *= act_grad(a.act, bn_stored<T>(bn_pre(xv[j], ainv[j], ac0[j])));
      } else if (a.act != RCGAN_ACT_NONE) {
        float yv[8];
        ld8(y + off, yv);
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] *= act_grad(a.act, yv[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += gv[j]; s2[j] += gv[j] * (xv[j] - mu[j]) * rs[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { RED(0, rl, chunk * 8 + j) = s1[j]; RED(1, rl, chunk * 8 + j) = s2[j]; }
  __syncthreads();
  // wavefront 0 writes both partial rows and then signals: the agent-scope release of thread 0 orders the
  // stores of its OWN wavefront, so no other wavefront has to fence
  if (t < 64) {
    const int qn = 32 / nsub;
    for (int sub = 0; sub < nsub; ++sub) {
      float sa = 0.f, sb = 0.f;
#pragma unroll 8
      for (int q = sub * qn; q < (sub + 1) * qn; ++q) { sa += RED(0, q, t); sb += RED(1, q, t); }
      // agent-scope stores (write through to the memory side: visible to every XCD without an L2 write-back)
      const long pr = (long)blockIdx.y * nsub + sub;
      __hip_atomic_store(a.partial + (pr * 2 + 0) * c + c0 + t, sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.partial + (pr * 2 + 1) * c + c0 + t, sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's stores are performed before it signals
  }
  // ---- arrival: the last workgroup of this column block finishes ------------------------------------------
  if (t == 0) {
    unsigned prev = __hip_atomic_fetch_add(a.counter + blockIdx.x * RC_LINE_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (prev == (unsigned)a.ngroups - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  // the partials are read back with agent-scope loads (served from the memory side, never from a stale cache line)
  const int col = t & 63, lane4 = t >> 6;
  const float* part = a.partial + c0 + col;
  const long gstride = 2L * c;

  if (MODE == 0 || a.labels == nullptr) {
    double d1 = 0.0, d2 = 0.0;
#pragma unroll 8
    for (int g = lane4; g < a.ngroups; g += 4) {
      d1 += (double)__hip_atomic_load(part + g * gstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      d2 += (double)__hip_atomic_load(part + g * gstride + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    fin[0][lane4][col] = d1; fin[1][lane4][col] = d2;
    __syncthreads();
    if (t < 64) {
      d1 = fin[0][0][t] + fin[0][1][t] + fin[0][2][t] + fin[0][3][t];
      d2 = fin[1][0][t] + fin[1][1][t] + fin[1][2][t] + fin[1][3][t];
      const int ch = c0 + t;
      if (MODE == 0) {
        double mean = d1 / (double)a.rows;
        double var = d2 / (double)a.rows - mean * mean;
        if (var < 0.0) var = 0.0;
        a.mean[ch] = (float)mean;
        a.rstd[ch] = (float)(1.0 / sqrt(var + (double)a.eps));
        if (a.mm) {
          // TF fused batch norm: moving -= (moving - batch) * (1 - decay), variance with Bessel's correction
          double uvar = a.rows > 1 ? var * ((double)a.rows / (double)(a.rows - 1)) : var;
          float om = 1.f - a.decay;
          a.mm[ch] = a.mm[ch] - (a.mm[ch] - (float)mean) * om;
          a.mv[ch] = a.mv[ch] - (a.mv[ch] - (float)uvar) * om;
        }
      } else {
        float og = (float)d2, ob = (float)d1;
        if (a.accumulate) { og += a.dgamma[ch]; ob += a.dbeta[ch]; }
        a.dgamma[ch] = og; a.dbeta[ch] = ob;
        lab_s[0][t] = (float)d1; lab_s[1][t] = (float)d2;
      }
    }
  } else {
    // conditional: the groups of one sample are contiguous.  Thread (column, lane4) owns samples lane4*8..+7 of
    // every batch of 32: it sums their groups (independent loads) and adds the result to its PRIVATE per-label
    // slot in LDS; afterwards the four lanes of a (label, column) are summed in a fixed order.
    const int gps = a.groups_per_sample;
    for (int l = 0; l < NL; ++l) { LACC(0, l, lane4, col) = 0.f; LACC(1, l, lane4, col) = 0.f; }
    // FB*32 samples per round: FB*16 independent loads per thread and group; buffer loads (one 32-bit offset per sample, the
    // second row through the scalar offset; samples past the end read zero).  Measured: 64 samples per round 3 us faster than
    // 128 (142 VGPRs), and than 32 with 64-bit addresses
    constexpr int FB = 2;
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void*)a.partial, 0, (int)((long)a.n_samples * gps * gstride * 4), 0x00020000);
    const unsigned sstride = (unsigned)(gps * gstride * 4);
    for (int s0 = 0; s0 < a.n_samples; s0 += FB * 32) {
      __syncthreads();
      int lab = -1;
      if (t < FB * 32 && s0 + t < a.n_samples) lab = a.labels[s0 + t];
      float d1[FB][8], d2[FB][8];
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int k = 0; k < 8; ++k) { d1[b][k] = 0.f; d2[b][k] = 0.f; }
      for (int q = 0; q < gps; ++q) {
        const unsigned o0 = (unsigned)(c0 + col) * 4u + (unsigned)q * (unsigned)(gstride * 4);
#pragma unroll
        for (int b = 0; b < FB; ++b) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const unsigned o = o0 + (unsigned)(s0 + b * 32 + lane4 * 8 + k) * sstride;
            d1[b][k] += ld_sc1_f1(rs_p, o, 0);
            d2[b][k] += ld_sc1_f1(rs_p, o, c * 4);
          }
        }
      }
      if (t < FB * 32) lab_ids[t] = lab;
      __syncthreads();
#pragma unroll
      for (int b = 0; b < FB; ++b) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int l = lab_ids[b * 32 + lane4 * 8 + k];
          if (l >= 0 && l < NL) { LACC(0, l, lane4, col) += d1[b][k]; LACC(1, l, lane4, col) += d2[b][k]; }
        }
      }
    }
    __syncthreads();
    double q1 = 0.0, q2 = 0.0;
#pragma unroll
    for (int slot = 0; slot < MAX_LABELS / 4; ++slot) {
      const int l = lane4 + 4 * slot;
      if (l < NL) {
        const float acc1 = (LACC(0, l, 0, col) + LACC(0, l, 1, col)) + (LACC(0, l, 2, col) + LACC(0, l, 3, col));
        const float acc2 = (LACC(1, l, 0, col) + LACC(1, l, 1, col)) + (LACC(1, l, 2, col) + LACC(1, l, 3, col));
        const long o = (long)l * c + c0 + col;
        const float gm = a.gamma[o];
        float og = acc2, ob = acc1;
        if (a.accumulate) { og += a.dgamma[o]; ob += a.dbeta[o]; }
        a.dgamma[o] = og; a.dbeta[o] = ob;
        q1 += (double)gm * (double)acc1;
        q2 += (double)gm * (double)acc2;
      }
    }
    fin[0][lane4][col] = q1; fin[1][lane4][col] = q2;
  }
  if (MODE == 1) {
    __syncthreads();
    if (t < 64) {
      // dx = A*g + P*x + Q with A = rstd*gamma[label], P = -rstd^2*s2/M, Q = -P*mean - rstd*s1/M,
      // s1 = sum_l gamma_l*dbeta_l, s2 = sum_l gamma_l*dgamma_l
      const int ch = c0 + t;
      double q1, q2;
      if (a.labels == nullptr) {
        const double gmm = (double)a.gamma[ch];
        q1 = gmm * (double)lab_s[0][t]; q2 = gmm * (double)lab_s[1][t];
      } else {
        q1 = fin[0][0][t] + fin[0][1][t] + fin[0][2][t] + fin[0][3][t];
        q2 = fin[1][0][t] + fin[1][1][t] + fin[1][2][t] + fin[1][3][t];
      }
      const float invM = 1.f / (float)a.rows;
      const float r = a.rstd_in[ch];
      const float p = -r * r * (float)q2 * invM;
      a.PQ[ch] = p;
      a.PQ[c + ch] = -p * a.mean_in[ch] - r * (float)q1 * invM;
    }
  }
  if (t == 0) __hip_atomic_store(a.counter + blockIdx.x * RC_LINE_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#undef RED
#undef LACC
}

// dynamic LDS of the kernel above
static inline size_t bn_fused_lds(int mode, int n_labels) {
  size_t red = 2 * 32 * 64 * sizeof(float), lacc = mode == 1 ? (size_t)n_labels * 2 * 4 * 64 * sizeof(float) : 0;
  return red > lacc ? red : lacc;
}

// ------------------------------------------------------------------------------------------------
// Tree variant of the fused reduction for LARGE tensors (>= a few MB).  The kernel above gives every workgroup a 64-channel
// column block: 128 bytes out of every 512-byte (c = 256) row, with the other three quarters of the row read by workgroups
// on other XCDs at other times -- measured 2.5-2.9 TB/s where the contiguous apply kernels reach 5-6.  Here a workgroup
// streams WHOLE rows (rows_per_group x c contiguous bytes), and because every workgroup then holds partial sums of every
// channel, the combination is a two-level arrival tree instead of "last workgroup of the column":
//   level 1: groups are dealt to clusters (16 consecutive groups; backward with labels: the groups of one sample).  The
//            last group of a cluster to arrive adds the cluster's partials in group order -> cpart[cluster][2][c] (fp64).
//   level 2: the last cluster to finish adds the cluster partials in cluster order (per label for the conditional
//            backward) and writes mean / rstd (or dgamma / dbeta / P,Q).
// Every sum has a fixed order: results are bit-reproducible.  Partials travel with agent-scope (write-through) stores and
// loads; counters reset themselves (graph replay).
// ------------------------------------------------------------------------------------------------
struct BnTreeArgs {
  BnFusedArgs f;             // partial = [ngroups][2c] fp32
  float* cpart;              // [nclusters][2c]
  unsigned* counters;        // [nclusters] + [1]   (per segment)
  int cs, nclusters;         // groups per cluster
};

typedef unsigned int v4u32 __attribute__((__vector_size__(16)));
// 16-byte agent-scope (sc1) accesses: served by / written through to the memory side, never a stale or dirty line of an XCD's L2
__device__ __forceinline__ float4 ld_sc1_f4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);
  const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];     // (named scalars: __builtin_bit_cast of a vector ELEMENT reads element 0)
  return make_float4(__builtin_bit_cast(float, e0), __builtin_bit_cast(float, e1), __builtin_bit_cast(float, e2), __builtin_bit_cast(float, e3));
}
__device__ __forceinline__ void st_sc1_f4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float4 f) {
  const v4u32 v = {__builtin_bit_cast(unsigned, f.x), __builtin_bit_cast(unsigned, f.y), __builtin_bit_cast(unsigned, f.z), __builtin_bit_cast(unsigned, f.w)};
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, 16);
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_tree_reduce_kernel(BnTreeArgs ta) {
  extern __shared__ __attribute__((aligned(16))) float tsm[];
  __shared__ int flag;
  BnFusedArgs& a = ta.f;
  const int t = threadIdx.x;
  const int c = a.c, c2 = 2 * c;
  const int cpr = c >> 3, lcpr = __ffs(cpr) - 1;
  const int chunk = t & (cpr - 1), rl = t >> lcpr, RL = 256 >> lcpr;
  const int g = blockIdx.x;
  const long rb = (long)g * a.rows_per_group;
  long re = rb + a.rows_per_group;
  if (re > a.rows) re = a.rows;
  if (MODE == 0 && a.nseg > 1) {
    const long sg = blockIdx.y;
    a.x = (const T*)a.x + sg * a.rows * c;
    a.partial += sg * (long)a.ngroups * c2;
    ta.cpart += sg * (long)ta.nclusters * c2;
    ta.counters += sg * (ta.nclusters + 1) * RC_LINE_STRIDE;
    a.mean += sg * c; a.rstd += sg * c;
  }
  const T* x = (const T*)a.x; const T* y = (const T*)a.y; const T* dy = (const T*)a.dy;
  float s1[8], s2[8], mu[8], rs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; mu[j] = 0.f; rs[j] = 1.f; }
  float ainv[8], ac0[8];
  const bool mask_x = MODE == 1 && a.beta != nullptr;
  if (MODE == 1) { ld8(a.mean_in + chunk * 8, mu); ld8(a.rstd_in + chunk * 8, rs); }
  if (mask_x) {
    const long lo = (long)(a.labels ? a.labels[g / a.groups_per_sample] : 0) * c + chunk * 8;
    float gm8[8], bt8[8];
    ld8(a.gamma + lo, gm8); ld8(a.beta + lo, bt8);
#pragma unroll
    for (int j = 0; j < 8; ++j) { ainv[j] = rs[j] * gm8[j]; ac0[j] = bn_c0(mu[j], ainv[j], bt8[j]); }
  }
#pragma unroll 4
  for (long r = rb + rl; r < re; r += RL) {
    const long off = r * c + chunk * 8;
    float xv[8];
    ld8(x + off, xv);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += xv[j]; s2[j] += xv[j] * xv[j]; }
    } else {
      float gv[8];
      ld8(dy + off, gv);
      if (mask_x) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] *= act_grad(a.act, bn_stored<T>(bn_pre(xv[j], ainv[j], ac0[j])));
      } else if (a.act != RCGAN_ACT_NONE) {
        float yv[8];
        ld8(y + off, yv);
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] *= act_grad(a.act, yv[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += gv[j]; s2[j] += gv[j] * (xv[j] - mu[j]) * rs[j]; }
    }
  }
  // ---- workgroup partial: sum over the row lanes through LDS ([2][RL][c] floats), written through as [2c] ----------
  float* red = tsm;
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[(0 * RL + rl) * c + chunk * 8 + j] = s1[j]; red[(1 * RL + rl) * c + chunk * 8 + j] = s2[j]; }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc(a.partial, 0, (int)((long)a.ngroups * c2 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cp = __builtin_amdgcn_make_buffer_rsrc(ta.cpart, 0, (int)((long)ta.nclusters * c2 * 4), 0x00020000);
  const int Q = c2 >> 2;                                     // 16-byte quads of one partial row
  for (int qd = t; qd < Q; qd += 256) {
    const int q = qd * 4 >= c ? 1 : 0, ch = qd * 4 - q * c;
    float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < RL; ++k) {
      const float4 v = *(const float4*)(red + (q * RL + k) * c + ch);
      sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
    }
    st_sc1_f4(rs_part, (unsigned)(((long)g * c2 + qd * 4) * 4), sm);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wavefront drains its write-through stores ...
  __syncthreads();                                           // ... before one lane signals
  const int cl = g / ta.cs;
  const int cl_groups = min(ta.cs, a.ngroups - cl * ta.cs);
  if (t == 0) {
    const unsigned prev = __hip_atomic_fetch_add(ta.counters + cl * RC_LINE_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag = (prev == (unsigned)cl_groups - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (!flag) return;
  // ---- level 1: this workgroup closes its cluster: cpart[cl] = sum of its groups' partials, in group order ------------
  for (int qd = t; qd < Q; qd += 256) {
    float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned base = (unsigned)(((long)cl * ta.cs * c2 + qd * 4) * 4);
#pragma unroll 16
    for (int k = 0; k < cl_groups; ++k) {
      const float4 v = ld_sc1_f4(rs_part, base + (unsigned)k * (unsigned)c2 * 4u);
      sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
    }
    st_sc1_f4(rs_cp, (unsigned)(((long)cl * c2 + qd * 4) * 4), sm);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    __hip_atomic_store(ta.counters + cl * RC_LINE_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned prev = __hip_atomic_fetch_add(ta.counters + ta.nclusters * RC_LINE_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag = (prev == (unsigned)ta.nclusters - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (!flag) return;
  // ---- level 2: the last cluster to finish closes the reduction --------------------------------------------------------
  const bool cond = MODE == 1 && a.labels != nullptr;
  const int ncl = ta.nclusters;
  if (!cond) {
    double* fin = (double*)tsm;                              // [2c]
    for (int qd = t; qd < Q; qd += 256) {
      double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll 16
      for (int k = 0; k < ncl; ++k) {
        const float4 v = ld_sc1_f4(rs_cp, (unsigned)(((long)k * c2 + qd * 4) * 4));
        d0 += (double)v.x; d1 += (double)v.y; d2 += (double)v.z; d3 += (double)v.w;
      }
      fin[qd * 4 + 0] = d0; fin[qd * 4 + 1] = d1; fin[qd * 4 + 2] = d2; fin[qd * 4 + 3] = d3;
    }
    __syncthreads();
    for (int ch = t; ch < c; ch += 256) {
      const double d1 = fin[ch], d2 = fin[c + ch];
      if (MODE == 0) {
        const double mean = d1 / (double)a.rows;
        double var = d2 / (double)a.rows - mean * mean;
        if (var < 0.0) var = 0.0;
        a.mean[ch] = (float)mean;
        a.rstd[ch] = (float)(1.0 / sqrt(var + (double)a.eps));
        if (a.mm) {
          const double uvar = a.rows > 1 ? var * ((double)a.rows / (double)(a.rows - 1)) : var;
          const float om = 1.f - a.decay;
          a.mm[ch] = a.mm[ch] - (a.mm[ch] - (float)mean) * om;
          a.mv[ch] = a.mv[ch] - (a.mv[ch] - (float)uvar) * om;
        }
      } else {
        float og = (float)d2, ob = (float)d1;
        if (a.accumulate) { og += a.dgamma[ch]; ob += a.dbeta[ch]; }
        a.dgamma[ch] = og; a.dbeta[ch] = ob;
        const double gmm = (double)a.gamma[ch];
        const float invM = 1.f / (float)a.rows;
        const float r = a.rstd_in[ch];
        const float p = -r * r * (float)(gmm * d2) * invM;
        a.PQ[ch] = p;
        a.PQ[c + ch] = -p * a.mean_in[ch] - r * (float)(gmm * d1) * invM;
      }
    }
  } else {
    // conditional backward: cluster = sample.  Thread (quad, sample lane) bins its samples' partial rows by label into
    // lacc[SL][n_labels][2c] (LDS; every (lane, quad) column is private to one thread), then a thread per channel adds the
    // lanes in order.
    const int nl = a.n_labels;
    const int SL = Q >= 256 ? 1 : 256 / Q;
    float* lacc = tsm;
    for (int i = t; i < SL * nl * c2; i += 256) lacc[i] = 0.f;
    __syncthreads();
    const int sl = Q >= 256 ? 0 : t / Q;
    for (int qd = Q >= 256 ? t : t - sl * Q; qd < Q; qd += 256) {
      for (int s0 = sl; s0 < ncl; s0 += SL * 16) {
        float4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int s = s0 + k * SL;
          v[k] = s < ncl ? ld_sc1_f4(rs_cp, (unsigned)(((long)s * c2 + qd * 4) * 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int s = s0 + k * SL;
          if (s < ncl) {
            const int l = a.labels[s];
            if (l >= 0 && l < nl) {
              float4* dst = (float4*)(lacc + ((long)sl * nl + l) * c2 + qd * 4);
              float4 o = *dst;
              o.x += v[k].x; o.y += v[k].y; o.z += v[k].z; o.w += v[k].w;
              *dst = o;
            }
          }
        }
      }
    }
    __syncthreads();
    for (int ch = t; ch < c; ch += 256) {
      double q1 = 0.0, q2 = 0.0;
      for (int l = 0; l < nl; ++l) {
        float a1 = 0.f, a2 = 0.f;
        for (int k = 0; k < SL; ++k) { a1 += lacc[((long)k * nl + l) * c2 + ch]; a2 += lacc[((long)k * nl + l) * c2 + c + ch]; }
        const long o = (long)l * c + ch;
        float og = a2, ob = a1;
        if (a.accumulate) { og += a.dgamma[o]; ob += a.dbeta[o]; }
        a.dgamma[o] = og; a.dbeta[o] = ob;
        const double gm = (double)a.gamma[o];
        q1 += gm * (double)a1;
        q2 += gm * (double)a2;
      }
      const float invM = 1.f / (float)a.rows;
      const float r = a.rstd_in[ch];
      const float p = -r * r * (float)q2 * invM;
      a.PQ[ch] = p;
      a.PQ[c + ch] = -p * a.mean_in[ch] - r * (float)q1 * invM;
    }
  }
  if (t == 0) __hip_atomic_store(ta.counters + ta.nclusters * RC_LINE_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// y = act(x*A + B), A = rstd*gamma[label], B = beta[label] - mean*A, computed in-line: the thread's 8 channels are fixed
// (grid stride is a multiple of the chunks per row), so mean/rstd live in registers.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_fused_kernel(long nchunks, int rows_per_sample, int c, const T* x, const int32_t* labels,
                                                             const float* gamma, const float* beta, const float* mean,
                                                             const float* rstd, int act, T* y, int n_per_seg) {
  if (gridDim.y > 1) {                                // blockIdx.y = segment: nchunks / n_per_seg count ONE segment
    const long sg = blockIdx.y;
    x += sg * nchunks * 8; y += sg * nchunks * 8;
    if (labels) labels += sg * n_per_seg;
    mean += sg * c; rstd += sg * c;
  }
  const unsigned cpr = (unsigned)c / 8u;              // a power of two on this path
  const int lcpr = __ffs((int)cpr) - 1;
  const int lrps = (rows_per_sample & (rows_per_sample - 1)) == 0 ? __ffs(rows_per_sample) - 1 : -1;
  const unsigned i0 = blockIdx.x * 256u + threadIdx.x;
  const unsigned stride = gridDim.x * 256u;
  const int ch = (int)(i0 & (cpr - 1u)) * 8;
  float mu[8], rs[8];
  ld8(mean + ch, mu); ld8(rstd + ch, rs);
  for (unsigned i = i0; i < (unsigned)nchunks; i += stride) {
    const unsigned row = i >> lcpr;
    const int l = labels ? labels[lrps >= 0 ? (row >> lrps) : (row / (unsigned)rows_per_sample)] : 0;
    float xv[8], g[8], b[8];
    ld8(x + (long)row * c + ch, xv);
    ld8(gamma + (long)l * c + ch, g);
    ld8(beta + (long)l * c + ch, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float inv = rs[j] * g[j];
      xv[j] = act_apply(act, bn_pre(xv[j], inv, bn_c0(mu[j], inv, b[j])));
    }
    st8(y + (long)row * c + ch, xv);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(long nchunks, int rows_per_sample, int c, const T* x, const T* y, const T* dy,
                                                                 const int32_t* labels, const float* gamma, const float* rstd,
                                                                 const float* PQ, int act, T* dx, int accumulate_dx,
                                                                 const float* beta, const float* mean) {
  const unsigned cpr = (unsigned)c / 8u;              // a power of two on this path
  const int lcpr = __ffs((int)cpr) - 1;
  const int lrps = (rows_per_sample & (rows_per_sample - 1)) == 0 ? __ffs(rows_per_sample) - 1 : -1;
  const unsigned i0 = blockIdx.x * 256u + threadIdx.x;
  const unsigned stride = gridDim.x * 256u;
  const int ch = (int)(i0 & (cpr - 1u)) * 8;
  float rs[8], p[8], q[8], mu[8];
  ld8(rstd + ch, rs); ld8(PQ + ch, p); ld8(PQ + c + ch, q);
  const bool mask_x = beta != nullptr;      // activation mask from x (bn_pre) instead of a read of y
  if (mask_x) ld8(mean + ch, mu);
  for (unsigned i = i0; i < (unsigned)nchunks; i += stride) {
    const unsigned row = i >> lcpr;
    const int l = labels ? labels[lrps >= 0 ? (row >> lrps) : (row / (unsigned)rows_per_sample)] : 0;
    const long off = (long)row * c + ch;
    float xv[8], gv[8], gm[8];
    ld8(x + off, xv);
    ld8(dy + off, gv);
    ld8(gamma + (long)l * c + ch, gm);
    if (mask_x) {
      float bt[8];
      ld8(beta + (long)l * c + ch, bt);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float inv = rs[j] * gm[j];
        gv[j] *= act_grad(act, bn_stored<T>(bn_pre(xv[j], inv, bn_c0(mu[j], inv, bt[j]))));
      }
    } else if (act != RCGAN_ACT_NONE) {
      float yv[8];
      ld8(y + off, yv);
#pragma unroll
      for (int j = 0; j < 8; ++j) gv[j] *= act_grad(act, yv[j]);
    }
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rs[j] * gm[j] * gv[j] + p[j] * xv[j] + q[j];
    if (accumulate_dx) {
      float d[8];
      ld8(dx + off, d);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += d[j];
    }
    st8(dx + off, o);
  }
}

// fused path: 64-channel column blocks, thread-fixed channel chunks in the apply kernels
static inline bool bn_fused_ok(int c) {
  return c >= 64 && c <= 2048 && (c & (c - 1)) == 0;      // power of two: chunks per row divide (or are) the block size
}

static inline int apply_grid_fused(long nchunks, int c) {
  const int cpr = c / 8;
  long b = (nchunks + 255) / 256;
  if (b > 8192) b = 8192;
  if (cpr > 256) { long m = cpr / 256; b = (b + m - 1) / m * m; }   // stride (b*256) must be a multiple of cpr
  if (b < 1) b = 1;
  return (int)b;
}

static inline long stats_group_rows(long rows, int c = 0, int dtype = -1) {
  long g = 512;
  while (rows / g > 2048) g *= 2;
  // fp32 path (MNIST: 64-channel critic layers of 1024 .. 25088 rows): 512 rows per workgroup are 128 dependent rounds of loads in 2 .. 49
  // workgroups -- a latency chain of 17-20 us for a few hundred KB.  Shorter groups until ~256 workgroups exist (the finisher reads <= 256 partials)
  if (dtype == RCGAN_F32 && c > 0)
    while (g > 32 && (rows / g) * (c / 64 > 0 ? c / 64 : 1) < 256) g /= 2;
  return g;
}

static inline int ew_grid2(long total) {
  long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

// tree path (power-of-two channel counts; full rows per workgroup need c/8 <= 256 chunk lanes): OFF unless RCGAN_BN_TREE_MIN gives
// an element threshold.  Measured (scripts/bench_bn.py, MI355X): the reductions are bound by their serial arrival chain
// (write-through partials, drain, counter, finisher loads), not by the access pattern, and the tree has one hop more than the column
// kernel.  While the column kernel's arrival counters shared one 128-byte line the tree won on the biggest tensor ([320,32,32,256]
// bf16, 168 MB: statistics 37 vs 55 us); with one counter per line (RC_LINE_STRIDE) the column kernel is faster at every size
// (that tensor: 36.2 vs 36.7 us, backward 202 vs 226 us; [128,16,16,256]: 6.6 vs 16 us).  The tests lower the threshold so the
// path stays covered.
static inline bool bn_tree_ok(long rows, int c) {
  static long min_elems = -1;
  if (min_elems < 0) {
    const char* e = getenv("RCGAN_BN_TREE");
    const char* m = getenv("RCGAN_BN_TREE_MIN");
    min_elems = (e && atoi(e) == 0) ? (1L << 62) : (m ? atol(m) : (1L << 62));
  }
  return bn_fused_ok(c) && rows * (long)c >= min_elems;
}

// launch of the tree reduction.  ws layout: partial [nseg][ng][2c] | cpart [nseg][ncl][2c] | (backward) PQ [2c]
template <int MODE>
static int launch_bn_tree(rcgan_ctx* ctx, int dtype, BnFusedArgs& a, int nseg, int cs, void* ws, size_t ws_bytes, float** pq_out) {
  const int c = a.c, ng = a.ngroups;
  const int ncl = cdiv(ng, cs);
  const size_t n_part = (size_t)nseg * ng * 2 * c, n_cp = (size_t)nseg * ncl * 2 * c;
  const size_t need = (n_part + n_cp + 2 * (size_t)c) * sizeof(float);
  if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
  RC_REQUIRE(ctx, (size_t)nseg * (ncl + 1) <= 8192, "too many clusters (%d x %d)", nseg, ncl);
  RC_REQUIRE(ctx, (size_t)ng * 2 * c * 4 < (1ull << 31) && (size_t)ncl * 2 * c * 4 < (1ull << 31), "partials exceed a buffer descriptor");
  BnTreeArgs ta;
  a.partial = (float*)ws;
  a.nseg = nseg;
  ta.cpart = a.partial + n_part;
  if (pq_out) { *pq_out = ta.cpart + n_cp; a.PQ = *pq_out; }
  ta.counters = ctx->tree_line_counters();
  ta.cs = cs; ta.nclusters = ncl;
  ta.f = a;
  const int Q = 2 * c / 4, SL = Q >= 256 ? 1 : 256 / Q;
  size_t lds = 16384;                                             // [2][RL][c] floats = 4096 floats
  if ((size_t)4 * c * sizeof(float) > lds) lds = (size_t)4 * c * sizeof(float);
  if (MODE == 1 && a.labels) { const size_t l2 = (size_t)SL * a.n_labels * 2 * c * sizeof(float); if (l2 > lds) lds = l2; }
  RC_REQUIRE(ctx, lds <= 128 * 1024, "tree reduction needs %zu bytes of LDS", lds);
  if (dtype == RCGAN_F32) {
    static size_t set = 0;
    if (lds > set) { RC_HIP(ctx, hipFuncSetAttribute((const void*)bn_tree_reduce_kernel<float, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); set = lds; }
    hipLaunchKernelGGL((bn_tree_reduce_kernel<float, MODE>), dim3(ng, nseg), dim3(256), lds, ctx->stream, ta);
  } else if (dtype == RCGAN_H16) {
    static size_t set = 0;
    if (lds > set) { RC_HIP(ctx, hipFuncSetAttribute((const void*)bn_tree_reduce_kernel<bf16_t, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); set = lds; }
    hipLaunchKernelGGL((bn_tree_reduce_kernel<bf16_t, MODE>), dim3(ng, nseg), dim3(256), lds, ctx->stream, ta);
  } else {
    RC_FAIL(ctx, RCGAN_EINVALID_ARG, "bad dtype %d", dtype);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// rows per group of the tree path: ~512 workgroups, a multiple of the row lanes, at least 16 rows
static inline long tree_group_rows(long rows, int c) {
  const int RL = 256 / (c / 8) > 0 ? 256 / (c / 8) : 1;
  static long target = -1;
  if (target < 0) { const char* e = getenv("RCGAN_BN_TREE_GROUPS"); target = e ? atol(e) : 512; }
  long rpg = cdiv(rows, target);
  if (rpg < 16) rpg = 16;
  rpg = (rpg + RL - 1) / RL * RL;
  return rpg;
}

// mean / rstd of `nseg` segments from the per-tile column sums a convolution's epilogue left (conv_mfma8.hip): part[tile][c][2] =
// (sum, sum of squares) over the tile's 256 pixels.  Segment s owns the tiles [g * group_stride + s * tps, + tps) of every group g
// (one group, or the four phases of the sub-pixel form).  grid (c / 16, nseg) x 256 threads: 16 channels x 16 tile subsets, every
// thread's loads independent of one another (one memory round trip), all sums in a fixed order in fp64.
__global__ __launch_bounds__(256) void bn_tile_stats_finish_kernel(const float* __restrict__ part, int c, int tps, int ngroups, long group_stride,
                                                                   double inv_count, float eps, float* mean, float* rstd) {
  __shared__ double red[2][16][16];
  const int cl = threadIdx.x & 15, sub = threadIdx.x >> 4, ch = blockIdx.x * 16 + cl, sg = blockIdx.y;
  double s1 = 0.0, s2 = 0.0;
  for (int g = 0; g < ngroups; ++g) {
    const long t0 = (long)g * group_stride + (long)sg * tps;
    for (int tb = sub; tb < tps; tb += 16 * 8) {
      float2 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int t = tb + q * 16;
        v[q] = t < tps ? *(const float2*)(part + ((t0 + t) * c + ch) * 2) : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) { s1 += (double)v[q].x; s2 += (double)v[q].y; }
    }
  }
  red[0][sub][cl] = s1; red[1][sub][cl] = s2;
  __syncthreads();
  if (sub == 0) {
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { a1 += red[0][q][cl]; a2 += red[1][q][cl]; }
    const double mu = a1 * inv_count;
    double var = a2 * inv_count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[(long)sg * c + ch] = (float)mu;
    rstd[(long)sg * c + ch] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

int bn_tile_stats_finish_launch(rcgan_ctx* ctx, const float* part, int c, int nseg, int tiles_per_seg, int ngroups, long group_stride,
                                double count, float eps, float* mean, float* rstd) {
  RC_REQUIRE(ctx, c % 16 == 0 && tiles_per_seg >= 1, "bad tile statistics layout");
  hipLaunchKernelGGL(bn_tile_stats_finish_kernel, dim3(c / 16, nseg), dim3(256), 0, ctx->stream, part, c, tiles_per_seg, ngroups, group_stride,
                     1.0 / count, eps, mean, rstd);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

extern "C" {

size_t rcgan_bn_workspace_bytes(int rows, int c) {
  // stats: ngroups*2*c ; bwd: ngroups*2*c + 2*c, with ngroups <= max(rows/512, n) <= rows; the tree path adds its cluster
  // partials (<= one per group)
  long ng = (rows + 511) / 512 + 1;
  if (ng < 4096) ng = 4096;   // per-sample grouping (n <= 4096 samples)
  ng *= 2;
  return (size_t)(ng * 2 * (long)c + 4 * (long)c + 2 * MAX_LABELS * (long)c) * sizeof(float) + 256;
}

int rcgan_bn_stats(rcgan_ctx* ctx, int rows, int c, int dtype, const void* x, float eps, float* mean, float* rstd,
                   float* mm, float* mv, float decay, void* ws, size_t ws_bytes) {
  long rpg = stats_group_rows(rows, c, dtype);
  int ng = cdiv(rows, rpg);
  size_t need = (size_t)ng * 2 * c * sizeof(float);
  if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
  float* partial = (float*)ws;
  if (bn_tree_ok(rows, c)) {
    BnFusedArgs a = {};
    a.rows = rows; a.c = c; a.rows_per_group = tree_group_rows(rows, c); a.ngroups = (int)cdiv((long)rows, a.rows_per_group); a.x = x;
    a.eps = eps; a.mean = mean; a.rstd = rstd; a.mm = mm; a.mv = mv; a.decay = decay;
    return launch_bn_tree<0>(ctx, dtype, a, 1, 16, ws, ws_bytes, nullptr);
  }
  if (bn_fused_ok(c)) {
    BnFusedArgs a = {};
    a.rows = rows; a.c = c; a.rows_per_group = rpg; a.ngroups = ng; a.x = x; a.partial = partial;
    a.counter = ctx->line_counters() + RC_LCOUNTER_BN * RC_LINE_STRIDE;
    a.eps = eps; a.mean = mean; a.rstd = rstd; a.mm = mm; a.mv = mv; a.decay = decay;
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL((bn_fused_reduce_kernel<T, 0>), dim3(c / 64, ng), dim3(256), bn_fused_lds(0, 0), ctx->stream, a));
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  if (c % 8 == 0) {
    dim3 grid(cdiv(c / 8, 32), ng);
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL((bn_partial_vec_kernel<T, 0>), grid, dim3(256), 0, ctx->stream, (long)rows, c, rpg,
                                                     (const T*)x, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr,
                                                     (const float*)nullptr, 0, partial));
  } else {
    dim3 grid(cdiv(c, 64), ng);
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL((bn_partial_kernel<T, 0>), grid, dim3(256), 0, ctx->stream, (long)rows, c, rpg,
                                                     (const T*)x, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr,
                                                     (const float*)nullptr, 0, partial));
  }
  RC_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(cdiv(c, 256)), dim3(256), 0, ctx->stream, c, ng, (long)rows,
                     (const float*)partial, eps, mean, rstd, mm, mv, decay);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_bn_apply_fwd(rcgan_ctx* ctx, int n, int rows_per_sample, int c, int n_labels, int dtype, const void* x,
                       const int32_t* labels, const float* gamma, const float* beta, const float* mean, const float* rstd,
                       int act, void* y, void* ws, size_t ws_bytes) {
  long total = (long)n * rows_per_sample * c;
  if (bn_fused_ok(c)) {
    long nchunks = total / 8;
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(bn_apply_fused_kernel<T>, dim3(apply_grid_fused(nchunks, c)), dim3(256), 0, ctx->stream,
                                                     nchunks, rows_per_sample, c, (const T*)x, labels, gamma, beta, mean, rstd, act, (T*)y, 0));
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  if (c % 8 == 0 && ws_bytes >= (size_t)2 * n_labels * c * sizeof(float)) {
    float* A = (float*)ws;
    float* B = A + (size_t)n_labels * c;
    hipLaunchKernelGGL(bn_table_fwd_kernel, dim3(cdiv((long)n_labels * c, 256)), dim3(256), 0, ctx->stream, n_labels, c, gamma, beta, mean, rstd, A, B);
    RC_LAUNCH_CHECK(ctx);
    long nchunks = total / 8;
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(bn_apply_vec_kernel<T>, dim3(ew_grid2(nchunks)), dim3(256), 0, ctx->stream, nchunks,
                                                     rows_per_sample, c, (const T*)x, labels, (const float*)A, (const float*)B, act, (T*)y));
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(bn_apply_fwd_kernel<T>, dim3(ew_grid2(total)), dim3(256), 0, ctx->stream, total,
                                                   rows_per_sample, c, (const T*)x, labels, gamma, beta, mean, rstd, act, (T*)y));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// Forward batch norm of `nseg` independent segments of n_per_seg samples each (x = the segments back to back): every
// segment gets its own batch statistics, exactly as nseg separate calls would, in two launches for all of them.
// mean / rstd: [nseg][c] scratch outputs.
int rcgan_bn_fwd_segments(rcgan_ctx* ctx, int nseg, int n_per_seg, int rows_per_sample, int c, int n_labels, int dtype, const void* x,
                          const int32_t* labels, const float* gamma, const float* beta, float eps, int act,
                          float* mean, float* rstd, void* y, void* ws, size_t ws_bytes) {
  RC_REQUIRE(ctx, nseg >= 1 && n_per_seg >= 1, "segments %d x %d", nseg, n_per_seg);
  const long rows = (long)n_per_seg * rows_per_sample;
  if (!bn_fused_ok(c) || nseg * (c / 64) > 256) {       // other channel counts: one segment at a time
    const size_t esz = dtype_size(dtype);
    for (int sg = 0; sg < nseg; ++sg) {
      const char* xs = (const char*)x + (size_t)sg * rows * c * esz;
      char* ys = (char*)y + (size_t)sg * rows * c * esz;
      int rc = rcgan_bn_stats(ctx, (int)rows, c, dtype, xs, eps, mean + (size_t)sg * c, rstd + (size_t)sg * c, nullptr, nullptr, 0.f, ws, ws_bytes);
      if (rc) return rc;
      if (!y) continue;
      rc = rcgan_bn_apply_fwd(ctx, n_per_seg, rows_per_sample, c, n_labels, dtype, xs, labels ? labels + (size_t)sg * n_per_seg : nullptr,
                              gamma, beta, mean + (size_t)sg * c, rstd + (size_t)sg * c, act, ys, ws, ws_bytes);
      if (rc) return rc;
    }
    return RCGAN_OK;
  }
  if (bn_tree_ok(rows, c)) {
    BnFusedArgs a = {};
    a.rows = rows; a.c = c; a.rows_per_group = tree_group_rows(rows, c); a.ngroups = (int)cdiv(rows, a.rows_per_group); a.x = x;
    a.eps = eps; a.mean = mean; a.rstd = rstd; a.mm = nullptr; a.mv = nullptr; a.decay = 0.f;
    int rc = launch_bn_tree<0>(ctx, dtype, a, nseg, 16, ws, ws_bytes, nullptr);
    if (rc) return rc;
  } else {
    const long rpg = stats_group_rows(rows);
    const int ng = cdiv(rows, rpg);
    const size_t need = (size_t)nseg * ng * 2 * c * sizeof(float);
    if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
    BnFusedArgs a = {};
    a.rows = rows; a.c = c; a.rows_per_group = rpg; a.ngroups = ng; a.nseg = nseg; a.x = x; a.partial = (float*)ws;
    a.counter = ctx->line_counters() + RC_LCOUNTER_BNSEG * RC_LINE_STRIDE;
    a.eps = eps; a.mean = mean; a.rstd = rstd; a.mm = nullptr; a.mv = nullptr; a.decay = 0.f;
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL((bn_fused_reduce_kernel<T, 0>), dim3(c / 64, ng, nseg), dim3(256), bn_fused_lds(0, 0), ctx->stream, a));
    RC_LAUNCH_CHECK(ctx);
  }
  if (!y) return RCGAN_OK;             // statistics only: the consumer normalises on load (rcgan_conv2d_fwd_bn)
  const long nchunks = rows * c / 8;
  int gx = apply_grid_fused(nchunks, c);
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(bn_apply_fused_kernel<T>, dim3(gx, nseg), dim3(256), 0, ctx->stream,
                                                   nchunks, rows_per_sample, c, (const T*)x, labels, gamma, beta, (const float*)mean, (const float*)rstd,
                                                   act, (T*)y, n_per_seg));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_bn_apply_segments(rcgan_ctx* ctx, int nseg, int n_per_seg, int rows_per_sample, int c, int n_labels, int dtype, const void* x,
                            const int32_t* labels, const float* gamma, const float* beta, const float* mean, const float* rstd, int act,
                            void* y, void* ws, size_t ws_bytes) {
  RC_REQUIRE(ctx, nseg >= 1 && n_per_seg >= 1, "segments %d x %d", nseg, n_per_seg);
  if (nseg == 1) return rcgan_bn_apply_fwd(ctx, n_per_seg, rows_per_sample, c, n_labels, dtype, x, labels, gamma, beta, mean, rstd, act, y, ws, ws_bytes);
  if (!bn_fused_ok(c)) {           // channel counts the fused kernel does not take: segment by segment, as rcgan_bn_fwd_segments does
    const size_t seg_elems = (size_t)n_per_seg * rows_per_sample * c;
    for (int sg = 0; sg < nseg; ++sg) {
      int rc = rcgan_bn_apply_fwd(ctx, n_per_seg, rows_per_sample, c, n_labels, dtype, (const char*)x + sg * seg_elems * dtype_size(dtype),
                                  labels ? labels + (size_t)sg * n_per_seg : nullptr, gamma, beta, mean + (size_t)sg * c, rstd + (size_t)sg * c,
                                  act, (char*)y + sg * seg_elems * dtype_size(dtype), ws, ws_bytes);
      if (rc != RCGAN_OK) return rc;
    }
    return RCGAN_OK;
  }
  const long rows = (long)n_per_seg * rows_per_sample;
  const long nchunks = rows * c / 8;
  const int gx = apply_grid_fused(nchunks, c);
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(bn_apply_fused_kernel<T>, dim3(gx, nseg), dim3(256), 0, ctx->stream,
                                                   nchunks, rows_per_sample, c, (const T*)x, labels, gamma, beta, mean, rstd, act, (T*)y, n_per_seg));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_bn_bwd(rcgan_ctx* ctx, int n, int rows_per_sample, int c, int n_labels, int dtype, const void* x, const void* y,
                 const void* dy, const int32_t* labels, const float* gamma, const float* mean, const float* rstd, int act,
                 void* dx, int accumulate_dx, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes) {
  return rcgan_bn_bwd2(ctx, n, rows_per_sample, c, n_labels, dtype, x, y, dy, labels, gamma, nullptr, mean, rstd, act, dx, accumulate_dx,
                       dgamma, dbeta, accumulate, ws, ws_bytes);
}

int rcgan_bn_bwd2(rcgan_ctx* ctx, int n, int rows_per_sample, int c, int n_labels, int dtype, const void* x, const void* y,
                  const void* dy, const int32_t* labels, const float* gamma, const float* beta, const float* mean, const float* rstd,
                  int act, void* dx, int accumulate_dx, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes) {
  // with beta, ReLU / leaky ReLU masks are recomputed from x on the fused paths (two instead of three tensor reads per pass)
  const float* beta_m = (beta && (act == RCGAN_ACT_RELU || act == RCGAN_ACT_LRELU)) ? beta : nullptr;
  RC_REQUIRE(ctx, n_labels >= 1 && n_labels <= MAX_LABELS, "n_labels %d", n_labels);
  RC_REQUIRE(ctx, labels != nullptr || n_labels == 1, "labels required for n_labels > 1");
  long rows = (long)n * rows_per_sample;
  long rpg;
  int ng;
  if (bn_tree_ok(rows, c) && !(labels && (size_t)n_labels * 2 * c * sizeof(float) > 96 * 1024) && (!labels || n <= 8000)) {
    // tree path: full rows per workgroup; with labels a cluster is exactly one sample (gps groups)
    BnFusedArgs a = {};
    int cs = 16;
    if (labels) {
      const int RL = 256 / (c / 8) > 0 ? 256 / (c / 8) : 1;
      int gps = 1;
      rpg = rows_per_sample;
      static long target = -1;
      if (target < 0) { const char* e = getenv("RCGAN_BN_TREE_GROUPS"); target = e ? atol(e) : 512; }
      while ((long)n * gps < target && rpg % 2 == 0 && rpg / 2 >= 16 && (rpg / 2) % RL == 0 && gps < 16) { gps *= 2; rpg /= 2; }
      ng = n * gps; cs = gps;
    } else {
      rpg = tree_group_rows(rows, c); ng = (int)cdiv(rows, rpg);
    }
    a.rows = rows; a.c = c; a.rows_per_group = rpg; a.ngroups = ng; a.x = x; a.y = y; a.dy = dy;
    a.mean_in = mean; a.rstd_in = rstd; a.act = act;
    a.n_labels = n_labels; a.groups_per_sample = cs; a.labels = labels; a.n_samples = n;
    a.gamma = gamma; a.dgamma = dgamma; a.dbeta = dbeta; a.accumulate = accumulate; a.beta = beta_m;
    float* PQ = nullptr;
    int rc = launch_bn_tree<1>(ctx, dtype, a, 1, cs, ws, ws_bytes, &PQ);
    if (rc) return rc;
    long nchunks = rows * c / 8;
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(bn_bwd_apply_fused_kernel<T>, dim3(apply_grid_fused(nchunks, c)), dim3(256), 0, ctx->stream,
                                                     nchunks, rows_per_sample, c, (const T*)x, (const T*)y, (const T*)dy, labels, gamma, rstd,
                                                     (const float*)PQ, act, (T*)dx, accumulate_dx, beta_m, mean));
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  if (bn_fused_ok(c)) {
    int gps = 1, nsub = 1;
    if (labels) {
      // groups never straddle samples; split each sample until the grid has enough workgroups (six fit on a CU); samples of
      // fewer than 32 rows (the 4x4 stage) go nsub to a workgroup, which still writes one partial row per sample
      static long target = -1, maxg = -1;
      if (target < 0) {
        // (round 3, same box: 512 / 512 -> 5.827 ms per iteration, 1024 / 1024 -> 5.801, 2048 / 2048 -> 5.815: the 32 x 32 layers gain from more
        // workgroups, the 8 x 8 / 16 x 16 ones lose to the longer finisher)
        const char* e = getenv("RCGAN_BN_BWD_WGS"); target = e ? atol(e) : 1024;
        const char* m = getenv("RCGAN_BN_BWD_MAXG"); maxg = m ? atol(m) : 1024;
      }
      rpg = rows_per_sample;
      if (rows_per_sample < 32 && 32 % rows_per_sample == 0 && n % (32 / rows_per_sample) == 0) { nsub = 32 / rows_per_sample; rpg = 32; }
      else while ((long)n * gps * (c / 64) < target && rpg % 2 == 0 && rpg / 2 >= 32 && (long)n * gps * 2 <= maxg) { gps *= 2; rpg /= 2; }
      ng = n * gps;
    } else {
      rpg = stats_group_rows(rows, c, dtype); ng = cdiv(rows, rpg);
    }
    const int nwg = ng / nsub;
    size_t need = ((size_t)ng * 2 * c + 2 * (size_t)c) * sizeof(float);
    if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
    BnFusedArgs a = {};
    a.rows = rows; a.c = c; a.rows_per_group = rpg; a.ngroups = ng; a.x = x; a.y = y; a.dy = dy;
    a.mean_in = mean; a.rstd_in = rstd; a.act = act; a.partial = (float*)ws; a.counter = ctx->line_counters() + RC_LCOUNTER_BN * RC_LINE_STRIDE;
    a.ngroups = nwg; a.nsub = nsub;
    a.n_labels = n_labels; a.groups_per_sample = gps; a.labels = labels; a.n_samples = n;
    a.gamma = gamma; a.dgamma = dgamma; a.dbeta = dbeta; a.accumulate = accumulate;
    // mask from x needs a group's rows to belong to one sample (conditional grouping) or no labels at all
    a.beta = beta_m;
    a.PQ = a.partial + (size_t)ng * 2 * c;
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL((bn_fused_reduce_kernel<T, 1>), dim3(c / 64, nwg), dim3(256), bn_fused_lds(1, labels ? n_labels : 0), ctx->stream, a));
    RC_LAUNCH_CHECK(ctx);
    long nchunks = rows * c / 8;
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(bn_bwd_apply_fused_kernel<T>, dim3(apply_grid_fused(nchunks, c)), dim3(256), 0, ctx->stream,
                                                     nchunks, rows_per_sample, c, (const T*)x, (const T*)y, (const T*)dy, labels, gamma, rstd,
                                                     (const float*)a.PQ, act, (T*)dx, accumulate_dx, beta_m, mean));
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  if (labels) { rpg = rows_per_sample; ng = n; }          // one group per sample: group label = sample label
  else { rpg = stats_group_rows(rows, c, dtype); ng = cdiv(rows, rpg); }
  size_t need = ((size_t)ng * 2 * c + 2 * (size_t)c) * sizeof(float);
  if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
  float* partial = (float*)ws;
  float* s12 = partial + (size_t)ng * 2 * c;
  const bool vec = c % 8 == 0 && ws_bytes >= need + ((size_t)n_labels * c + 2 * (size_t)c) * sizeof(float);
  if (vec) {
    dim3 grid(cdiv(c / 8, 32), ng);
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL((bn_partial_vec_kernel<T, 1>), grid, dim3(256), 0, ctx->stream, rows, c, rpg,
                                                     (const T*)x, (const T*)y, (const T*)dy, mean, rstd, act, partial));
  } else {
    dim3 grid(cdiv(c, 64), ng);
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL((bn_partial_kernel<T, 1>), grid, dim3(256), 0, ctx->stream, rows, c, rpg,
                                                     (const T*)x, (const T*)y, (const T*)dy, mean, rstd, act, partial));
  }
  RC_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(bn_bwd_combine_kernel, dim3(cdiv(c, 128)), dim3(128), 0, ctx->stream, c, ng, n_labels, labels, gamma,
                     (const float*)partial, dgamma, dbeta, s12, accumulate);
  RC_LAUNCH_CHECK(ctx);
  long total = rows * c;
  if (vec) {
    float* A = s12 + 2 * (size_t)c;
    float* PQ = A + (size_t)n_labels * c;
    hipLaunchKernelGGL(bn_table_bwd_kernel, dim3(cdiv((long)n_labels * c, 256)), dim3(256), 0, ctx->stream, n_labels, c, rows, gamma, mean, rstd,
                       (const float*)s12, A, PQ);
    RC_LAUNCH_CHECK(ctx);
    long nchunks = total / 8;
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(bn_bwd_apply_vec_kernel<T>, dim3(ew_grid2(nchunks)), dim3(256), 0, ctx->stream, nchunks,
                                                     rows_per_sample, c, (const T*)x, (const T*)y, (const T*)dy, labels, (const float*)A,
                                                     (const float*)PQ, act, (T*)dx, accumulate_dx));
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
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

int rcgan_bn_infer_bwd(rcgan_ctx* ctx, int rows, int c, int dtype, const void* y, const void* dy, const float* gamma,
                       const float* mv, float eps, int act, void* dx, int accumulate) {
  RC_REQUIRE(ctx, act == RCGAN_ACT_NONE || act == RCGAN_ACT_RELU || act == RCGAN_ACT_TANH || act == RCGAN_ACT_SIGMOID,
             "activation %d has no output-side derivative", act);
  long total = (long)rows * c;
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(bn_infer_bwd_kernel<T>, dim3(ew_grid2(total)), dim3(256), 0, ctx->stream, total, c,
                                                   (const T*)y, (const T*)dy, gamma, mv, eps, act, (T*)dx, accumulate));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

}  // extern "C"
