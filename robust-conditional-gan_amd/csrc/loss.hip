// Discriminator head, projection logit, RCGAN / RCGAN-U loss terms, confusion-matrix softmax.
//   reference: cifar10/gan_resnet.py:405-412 (relu + spatial mean), :588 (projection), :604-606, :647,
//   :654-660, :682-684, :751-760, :773 (losses), :522 (C = softmax(logits)), :692-695/:781-784 (perm BCE);
//   mnist/model.py:135-145, 199-221, 679-685.
// All of this is a few KB of fp32: each op is one small launch built on wavefront reductions.
#include "common.h"
#include "small_gemm.h"

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

// gs_host * (*gs_dev): the gradient (loss) scale of 16-bit activations -- applied to the gradients only, the loss value stays unscaled
__device__ __forceinline__ float grad_scale_of(float gs_host, const float* gs_dev) { return gs_dev ? gs_host * gs_dev[0] : gs_host; }

__global__ __launch_bounds__(256) void loss_kernel(int kind, int rows, int cols, const float* x, const float* wts, float weight,
                                                    float* loss_acc, float* dlogit, float* dwts, float gs_host, const float* gs_dev) {
  __shared__ float red[4];
  const float gw = weight * grad_scale_of(gs_host, gs_dev);
  const long total = (long)rows * cols;
  const float inv_rows = 1.f / (float)rows;
  const float inv_all = 1.f / (float)total;
  float acc = 0.f;
  for (long i = threadIdx.x; i < total; i += 256) {
    float t, d;
    loss_term(kind, x[i], &t, &d);
    float wf = wts ? wts[i] * inv_rows : inv_all;
    acc += t * wf;
    if (dlogit) dlogit[i] = gw * d * wf;
    if (dwts) dwts[i] = gw * t * inv_rows;
  }
  acc = block_sum256(acc, red);
  if (threadIdx.x == 0 && loss_acc) *loss_acc += weight * acc;
}

__global__ __launch_bounds__(256) void bce_onehot_kernel(int rows, int cols, const float* x, const int32_t* labels, float weight,
                                                          float* loss_acc, float* dx, float gs_host, const float* gs_dev) {
  __shared__ float red[4];
  const float gw = weight * grad_scale_of(gs_host, gs_dev);
  const long total = (long)rows * cols;
  const float inv_all = 1.f / (float)total;
  float acc = 0.f;
  for (long i = threadIdx.x; i < total; i += 256) {
    int r = (int)(i / cols), cidx = (int)(i % cols);
    float z = labels[r] == cidx ? 1.f : 0.f;
    float v = x[i];
    acc += (fmaxf(v, 0.f) - v * z + log1pf(expf(-fabsf(v)))) * inv_all;
    if (dx) dx[i] = gw * (1.f / (1.f + expf(-v)) - z) * inv_all;
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


// ---------------------------------------------------------------------------------------------------------
// Fused projection head: everything between the discriminator's pooled features and the loss, forward AND backward,
// as two (forward + data gradient) or four (+ parameter gradients) short launches -- it replaces ~16 launches of a few
// microseconds each per step: D.Output linear, embedding gather, D.Embedding_y linear, projection logit, the loss terms and
// all their gradient kernels.
//   psi[s]      = <feat[s], w_out> / sigma_out + b_out                       (gan_resnet.py:408-411, Linear SN 128 -> 1)
//   E[l]        = table[l] @ W_e / sigma_e + b_e                             (:414-421 for every label l: only V = 10 distinct rows exist)
//   logit[s,l]  = psi[s] + <feat[s], E[l]>                                   (:588, :654-660)
//   part p of the rows (e.g. real | fake) contributes  weight/rows_p * sum_s sum_l w[s,l] * loss_kind_p(logit[s,l]),
//   w = onehot(labels[s]) or an explicit [rows_p, V] weight matrix (confusion rows :682-684 / C^-1 rows :647).
// Gradients: dfeat written; dw_out / db_out / dtable / dW_e / db_e ACCUMULATED (+=) into the caller's buffers (the SN
// weights' d/dW_bar scratch or the gradient slab, zeroed at the start of the step); dwts written.  Because only V rows of
// E exist, the embedding gradients collapse to V x d problems: dE[l] = sum_s dlogit[s,l] feat[s], dW_e = table^T dE,
// dtable = dE W_e^T / sigma_e.
// ---------------------------------------------------------------------------------------------------------
#include "head_rider.h"

// two adjacent elements as one access (the pooled-feature path of head_logit_kernel: a lane owns a channel pair)
template <typename T> __device__ __forceinline__ void ld2(const T* p, float& a, float& b);
template <typename T> __device__ __forceinline__ void st2(T* p, float a, float b);
template <> __device__ __forceinline__ void ld2<float>(const float* p, float& a, float& b) { const float2 v = *(const float2*)p; a = v.x; b = v.y; }
template <> __device__ __forceinline__ void st2<float>(float* p, float a, float b) { *(float2*)p = make_float2(a, b); }
template <> __device__ __forceinline__ void ld2<bf16_t>(const bf16_t* p, float& a, float& b) {
  const uint32_t v = *(const uint32_t*)p;
  a = bf16_to_f32((bf16_t)(v & 0xFFFFu)); b = bf16_to_f32((bf16_t)(v >> 16));
}
template <> __device__ __forceinline__ void st2<bf16_t>(bf16_t* p, float a, float b) {
  *(uint32_t*)p = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
}
#define HEAD_XC 32          // pixels of a sample requested per memory round trip

// Four short multi-workgroup launches (a single workgroup would be latency-bound: hundreds of dependent L2 round trips):
//   embed : E[l][j]                      small-left GEMM, grid d/16: 16 columns x 16 k-lanes, all rows l
//   logit : per sample psi, logits, loss, dlogit, dfeat       grid n/4           one sample per wavefront; the last
//           workgroup to arrive adds the loss partials in workgroup order
//   dE    : dE[l][j] = sum_s dlogit[s,l] feat[s][j]  (row v: the psi column)     the same small-left GEMM
//   wgrad : dW_e += table^T dE,  dtable += dE W_e^T / sigma_e,  dw_out, db_out, db_e   grid (e_dim*d + ...)/256
// out[l][j] = scale * sum_k A(l,k) * B[k][j] + bias[j] for a FEW rows l (L <= HEAD_MAX_V + 1): the embedding of every label
// (A = table, B = W_e) and dE = dlogit^T feat (A = dlogit transposed, B = feat).  The reduction index is what is long, so a
// workgroup owns 16 columns and splits k over 16 lanes per column; per chunk of SG_KC = 320 reduction indices every thread
// requests its <= 20 B elements AND its share of the A chunk (-> LDS) before it waits: ONE memory round trip per chunk
// (the 64-columns x 4-lanes form these two kernels had walked k in 38 dependent steps: 13 us for 0.4 MFLOP).
// rowsum_out (optional): += the sum over k of row rowsum_row of A (the bias gradient of D.Output).
__global__ __launch_bounds__(256) void head_smallgemm_kernel(SmallGemmArgs g) {
  __shared__ float As[SG_AS_FLOATS];
  __shared__ float red[SG_RED_FLOATS];
  small_gemm_body(g, blockIdx.x, As, red);
}

// T: element type of a.x (ignored when a.x is null)
template <typename T>
__global__ __launch_bounds__(256) void head_logit_kernel(HeadArgs a, const float* Eg, float* dlg, float* losspart, unsigned* counter) {
  extern __shared__ __attribute__((aligned(16))) float hs[];
  const int n = a.n, d = a.d, v = a.v, vp = v + 1;
  float* E = hs;                    // [v][d]
  float* red = E + v * d;           // [4]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // one sample per wavefront; everything the sample needs from memory is requested before the first wait
  const int s = blockIdx.x * 4 + wave;
  const int p = s < a.part[0].rows ? 0 : 1;
  const HeadPart& P = a.part[p];
  const int sr = p ? s - a.part[0].rows : s;
  const float inv_rows = 1.f / (float)P.rows;
  const float gw = a.weight * grad_scale_of(a.gs_host, a.gs_dev);
  const bool xm = a.x != nullptr;
  // channel of slot q of this lane: lane + 64 q, or (pooling from x) the adjacent pair 2 lane, 2 lane + 1 of each 128-channel half
  auto jof = [&](int q) { return xm ? 2 * lane + (q & 1) + 128 * (q >> 1) : lane + q * 64; };
  float f[HEAD_MAX_D / 64], wo[HEAD_MAX_D / 64], df[HEAD_MAX_D / 64];
  int lab_pre = -1;
  float wt_pre = 0.f;
  const T* const xs = (const T*)a.x + (long)(s < n ? s : 0) * a.hw * d;
  const int xcp = (a.hw + HEAD_XC - 1) / HEAD_XC, nxc = xm ? xcp * (d / 128) : 0;
  float xa[HEAD_XC], xb[HEAD_XC];
  // (no branch per load: pixels past the end re-read the last one and are dropped in the sum)
  auto x_load = [&](int c) __attribute__((always_inline)) {
    const int h = c / xcp, p0 = (c - h * xcp) * HEAD_XC;
#pragma unroll
    for (int u = 0; u < HEAD_XC; ++u) ld2<T>(xs + (long)min(p0 + u, a.hw - 1) * d + h * 128 + 2 * lane, xa[u], xb[u]);
  };
  if (s < n) {
#pragma unroll
    for (int q = 0; q < HEAD_MAX_D / 64; ++q) {
      const int j = jof(q);
      f[q] = (!xm && j < d) ? a.feat[(long)s * d + j] : 0.f;
      wo[q] = j < d ? a.w_out[j] : 0.f;
    }
    if (P.labels) lab_pre = P.labels[sr];
    if (P.wts && lane < v) wt_pre = P.wts[(long)sr * v + lane];
  }
  for (int c = 0; c < nxc; ++c) {        // (workgroup-uniform trip count; the activation switch once per chunk)
    const int h = c / xcp, left = a.hw - (c - h * xcp) * HEAD_XC;
    x_load(c);
    float s0 = 0.f, s1 = 0.f;
    if (a.act == RCGAN_ACT_RELU) {
#pragma unroll
      for (int u = 0; u < HEAD_XC; ++u) { s0 += u < left ? fmaxf(xa[u], 0.f) : 0.f; s1 += u < left ? fmaxf(xb[u], 0.f) : 0.f; }
    } else {
#pragma unroll 4
      for (int u = 0; u < HEAD_XC; ++u) { s0 += u < left ? act_apply(a.act, xa[u]) : 0.f; s1 += u < left ? act_apply(a.act, xb[u]) : 0.f; }
    }
#pragma unroll
    for (int hh = 0; hh < HEAD_MAX_D / 128; ++hh)
      if (hh == h) { f[2 * hh] += s0; f[2 * hh + 1] += s1; }
  }
  if (xm) {
    const float ih = 1.f / (float)a.hw;
#pragma unroll
    for (int q = 0; q < HEAD_MAX_D / 64; ++q) f[q] *= ih;
    if (a.feat_out && s < n) {
#pragma unroll
      for (int h = 0; h < HEAD_MAX_D / 128; ++h)
        if (h * 128 < d) *(float2*)(a.feat_out + (long)s * d + h * 128 + 2 * lane) = make_float2(f[2 * h], f[2 * h + 1]);
    }
  }
  const float inv_so = a.sigma_out ? 1.f / a.sigma_out[0] : 1.f;
  const float bo = a.b_out ? a.b_out[0] : 0.f;
  for (int i = t; i < v * d; i += 256) E[i] = Eg[i];
  __syncthreads();
  float lacc = 0.f;
  if (s < n) {
    const int lab = P.labels ? lab_pre : -1;
    float ps = 0.f;
#pragma unroll
    for (int q = 0; q < HEAD_MAX_D / 64; ++q) {
      wo[q] *= inv_so;
      ps += f[q] * wo[q];
      df[q] = 0.f;
    }
    ps = wave_sum(ps) + bo;
    float dsum = 0.f;
    for (int l = 0; l < v; ++l) {
      float g = 0.f;
      if (lab < 0 || lab == l) {                           // wave-uniform
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < HEAD_MAX_D / 64; ++q) { const int j = jof(q); dot += j < d ? f[q] * E[l * d + j] : 0.f; }
        const float x = wave_sum(dot) + ps;
        float tt, dd;
        loss_term(P.kind, x, &tt, &dd);
        const float wf = (P.wts ? __shfl(wt_pre, l) : 1.f) * inv_rows;
        lacc += tt * wf;
        g = gw * dd * wf;
#pragma unroll
        for (int q = 0; q < HEAD_MAX_D / 64; ++q) { const int j = jof(q); df[q] += j < d ? g * E[l * d + j] : 0.f; }
        if (lane == 0) {
          if (P.dwts) P.dwts[(long)sr * v + l] = gw * tt * inv_rows;
          if (a.logits) a.logits[(long)s * v + l] = x;
        }
      } else if (lane == 0 && a.logits) {
        a.logits[(long)s * v + l] = 0.f;
      }
      if (lane == 0) dlg[s * vp + l] = g;
      dsum += g;
    }
    if (lane == 0) dlg[s * vp + v] = dsum;
#pragma unroll
    for (int q = 0; q < HEAD_MAX_D / 64; ++q) df[q] += dsum * wo[q];
    if (a.dfeat) {
#pragma unroll
      for (int q = 0; q < HEAD_MAX_D / 64; ++q) { const int j = jof(q); if (j < d) a.dfeat[(long)s * d + j] = df[q]; }
    }
    if (xm && a.dx) {
      // d mean(act(x)) / dx: the adjoint of the pooling above, x read again (L2), a chunk of pixels per round trip
      T* dxs = (T*)a.dx + (long)s * a.hw * d;
      const float inv = 1.f / (float)a.hw;
      for (int c = 0; c < nxc; ++c) {
        const int h = c / xcp, p0 = (c - h * xcp) * HEAD_XC;
        x_load(c);
        float g0 = 0.f, g1 = 0.f;
#pragma unroll
        for (int hh = 0; hh < HEAD_MAX_D / 128; ++hh)
          if (hh == h) { g0 = df[2 * hh] * inv; g1 = df[2 * hh + 1] * inv; }
        if (a.act == RCGAN_ACT_RELU) {
#pragma unroll
          for (int u = 0; u < HEAD_XC; ++u)
            if (p0 + u < a.hw)
              st2<T>(dxs + (long)(p0 + u) * d + h * 128 + 2 * lane, xa[u] > 0.f ? g0 : 0.f, xb[u] > 0.f ? g1 : 0.f);
        } else {
#pragma unroll 4
          for (int u = 0; u < HEAD_XC; ++u)
            if (p0 + u < a.hw)
              st2<T>(dxs + (long)(p0 + u) * d + h * 128 + 2 * lane, g0 * act_grad(a.act, xa[u]), g1 * act_grad(a.act, xb[u]));
        }
      }
    }
  }
  // ---- loss: per-workgroup partial, summed in workgroup order by the last arrival (deterministic) -------------------
  if (lane == 0) red[wave] = lacc;
  __syncthreads();
  if (t == 0) {
    __hip_atomic_store(losspart + blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1u) {
      float tot = 0.f;
      for (unsigned w = 0; w < gridDim.x; ++w) tot += __hip_atomic_load(losspart + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.loss_acc) *a.loss_acc += a.weight * tot;
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(256) void head_wgrad_kernel(HeadArgs a, const float* dEg) {
  extern __shared__ __attribute__((aligned(16))) float hs[];
  head_wgrad_body(a, dEg, blockIdx.x, hs);
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
  hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, ctx->stream, kind, rows, cols, x, wts, weight, loss_acc, dlogit, dwts, ctx->gscale_host,
                     ctx->gscale_dev);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// dE[l] = sum_s dlogit[s,l] feat[s]  (row v: the psi column -> dw_out; its sum over the samples -> db_out)
static SmallGemmArgs head_de_gemm(const HeadArgs& a, const float* dlg, float* dEg) {
  const int vp = a.v + 1;
  return SmallGemmArgs{vp, a.n, a.d, dlg, 1, vp, a.feat, nullptr, nullptr, dEg, a.db_out, a.v};
}

// dE (small-left GEMM over the samples) and the parameter gradients from the dlogit rows the logit kernel left in the scratch
static int head_param_grads(rcgan_ctx* ctx, const HeadArgs& a, const float* dlg, float* dEg, int from_stage = 1) {
  if (from_stage <= 1) {
    hipLaunchKernelGGL(head_smallgemm_kernel, dim3(cdiv(a.d, 16)), dim3(256), 0, ctx->stream, head_de_gemm(a, dlg, dEg));
    RC_LAUNCH_CHECK(ctx);
  }
  hipLaunchKernelGGL(head_wgrad_kernel, dim3(head_wgrad_blocks(a)), dim3(256), (size_t)(a.v + 1) * a.d * sizeof(float), ctx->stream, a, (const float*)dEg);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// the deferred form (head_rider.h): what the two launches above need, parked in the context until a launch carries it
struct HeadDeferred { HeadArgs a; const float* dlg; float* dEg; };
static_assert(sizeof(HeadDeferred) <= sizeof(((rcgan_ctx*)nullptr)->head_blob), "rcgan_ctx::head_blob too small");

bool head_take_gemm(rcgan_ctx* ctx, SmallGemmArgs* out) {
  if (ctx->head_stage != 1) return false;
  const HeadDeferred* h = (const HeadDeferred*)ctx->head_blob;
  *out = head_de_gemm(h->a, h->dlg, h->dEg);
  ctx->head_stage = 2;
  return true;
}

bool head_take_wgrad(rcgan_ctx* ctx, HeadWgradRider* out) {
  if (ctx->head_stage != 2) return false;
  const HeadDeferred* h = (const HeadDeferred*)ctx->head_blob;
  out->a = h->a; out->dEg = h->dEg; out->blocks = head_wgrad_blocks(h->a);
  ctx->head_stage = 0;
  return true;
}

int rcgan_proj_head_fwd_bwd(rcgan_ctx* ctx, const rcgan_head_desc* hd, const float* feat, const float* w_out, const float* sigma_out,
                            const float* b_out, const float* table, const float* w_e, const float* sigma_e, const float* b_e,
                            float* loss_acc, float* logits, float* dfeat, float* dw_out, float* db_out, float* dtable, float* dw_e,
                            float* db_e, void* ws, size_t ws_bytes) {
  RC_REQUIRE(ctx, hd && feat && w_out && table && w_e, "null argument");
  RC_REQUIRE(ctx, hd->n >= 1 && hd->n <= HEAD_MAX_N && hd->d >= 1 && hd->d <= HEAD_MAX_D && hd->v >= 1 && hd->v <= HEAD_MAX_V && hd->e_dim >= 1,
             "head shape n %d d %d v %d e %d", hd->n, hd->d, hd->v, hd->e_dim);
  RC_REQUIRE(ctx, hd->rows_a >= 1 && hd->rows_a <= hd->n, "rows_a %d of %d", hd->rows_a, hd->n);
  HeadArgs a;
  a.n = hd->n; a.d = hd->d; a.v = hd->v; a.e_dim = hd->e_dim; a.weight = hd->weight;
  a.gs_host = ctx->gscale_host; a.gs_dev = ctx->gscale_dev;
  a.part[0] = {hd->rows_a, hd->kind_a, hd->labels_a, hd->wts_a, hd->dwts_a};
  a.part[1] = {hd->n - hd->rows_a, hd->kind_b, hd->labels_b, hd->wts_b, hd->dwts_b};
  for (int p = 0; p < 2; ++p) {
    if (a.part[p].rows == 0) continue;
    RC_REQUIRE(ctx, a.part[p].kind >= 0 && a.part[p].kind <= RCGAN_LOSS_CE_ZEROS, "kind %d", a.part[p].kind);
    RC_REQUIRE(ctx, (a.part[p].labels != nullptr) != (a.part[p].wts != nullptr), "part %d needs labels or weights (exactly one)", p);
  }
  a.feat = feat; a.w_out = w_out; a.sigma_out = sigma_out; a.b_out = b_out; a.table = table; a.w_e = w_e; a.sigma_e = sigma_e; a.b_e = b_e;
  a.loss_acc = loss_acc; a.logits = logits; a.dfeat = dfeat; a.dw_out = dw_out; a.db_out = db_out; a.dtable = dtable; a.dw_e = dw_e; a.db_e = db_e;
  a.x = hd->x; a.dx = hd->dx; a.hw = hd->hw; a.act = hd->act; a.feat_out = nullptr;
  if (a.x) {
    RC_REQUIRE(ctx, a.d % 128 == 0 && a.hw >= 1, "pooling inside the head needs d %% 128 == 0 (d %d, hw %d)", a.d, a.hw);
    RC_REQUIRE(ctx, hd->x_dtype == RCGAN_F32 || hd->x_dtype == RCGAN_H16, "bad x dtype");
    a.feat_out = (float*)feat;      // the features become an OUTPUT (read by the parameter-gradient kernels)
  }
  const int vp = a.v + 1;
  const size_t need = ((size_t)a.v * a.d + (size_t)a.n * vp + (size_t)vp * a.d + 256) * sizeof(float);
  if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
  float* Eg = (float*)ws;
  float* dlg = Eg + (size_t)a.v * a.d;
  float* dEg = dlg + (size_t)a.n * vp;
  float* losspart = dEg + (size_t)vp * a.d;
  const bool want_params = dw_out || db_out || dtable || dw_e || db_e;
  const bool defer = hd->defer_ws != nullptr && want_params;
  if (defer) {        // dlogit and dE outlive this call: they live in the caller's buffer, not in the shared workspace
    const size_t dneed = ((size_t)a.n * vp + (size_t)vp * a.d) * sizeof(float);
    if (hd->defer_ws_bytes < dneed) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "defer_ws: need %zu have %zu", dneed, hd->defer_ws_bytes);
    dlg = (float*)hd->defer_ws;
    dEg = dlg + (size_t)a.n * vp;
  }
  if (ctx->head_stage) {            // an earlier head's deferred launches were never carried: they go first
    int rc = rcgan_head_flush(ctx);
    if (rc) return rc;
  }
  if (hd->E_pre) {
    Eg = (float*)hd->E_pre;         // the label embeddings were computed earlier in the step (rcgan_conv_prepare_batch_embed)
  } else {
    const int dblk = cdiv(a.d, 16);
    SmallGemmArgs ge = {a.v, a.e_dim, a.d, a.table, a.e_dim, 1, a.w_e, a.sigma_e, a.b_e, Eg, nullptr, 0};
    hipLaunchKernelGGL(head_smallgemm_kernel, dim3(dblk), dim3(256), 0, ctx->stream, ge);
    RC_LAUNCH_CHECK(ctx);
  }
  const int nwg = cdiv(a.n, 4);
  RC_REQUIRE(ctx, nwg <= 256, "too many rows for the loss partials");
  const size_t lds = ((size_t)a.v * a.d + 4) * sizeof(float);
  if (a.x && hd->x_dtype == RCGAN_H16)
    hipLaunchKernelGGL(head_logit_kernel<bf16_t>, dim3(nwg), dim3(256), lds, ctx->stream, a, (const float*)Eg, dlg, losspart, ctx->counters() + RC_COUNTER_HEAD);
  else
    hipLaunchKernelGGL(head_logit_kernel<float>, dim3(nwg), dim3(256), lds, ctx->stream, a, (const float*)Eg, dlg, losspart, ctx->counters() + RC_COUNTER_HEAD);
  RC_LAUNCH_CHECK(ctx);
  if (defer) {
    HeadDeferred* h = (HeadDeferred*)ctx->head_blob;
    h->a = a; h->dlg = dlg; h->dEg = dEg;
    ctx->head_stage = 1;
    return RCGAN_OK;
  }
  if (want_params) return head_param_grads(ctx, a, dlg, dEg);
  return RCGAN_OK;
}

int rcgan_head_flush(rcgan_ctx* ctx) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  if (!ctx->head_stage) return RCGAN_OK;
  const HeadDeferred h = *(const HeadDeferred*)ctx->head_blob;
  const int stage = ctx->head_stage;
  ctx->head_stage = 0;
  return head_param_grads(ctx, h.a, h.dlg, h.dEg, stage);
}

int rcgan_bce_onehot_fwd_bwd(rcgan_ctx* ctx, int rows, int cols, const float* x, const int32_t* labels, float weight,
                             float* loss_acc, float* dx) {
  hipLaunchKernelGGL(bce_onehot_kernel, dim3(1), dim3(256), 0, ctx->stream, rows, cols, x, labels, weight, loss_acc, dx, ctx->gscale_host,
                     ctx->gscale_dev);
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
