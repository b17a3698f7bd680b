// Spectral normalisation: one power iteration per weight, forward and backward, batched.
//   reference: mnist/sn.py:31-75 == cifar10/common/ops/sn.py:31-75
//     v = l2n(u W^T); u' = l2n(v W); sigma = v W u'^T; W_bar = W / sigma; u <- u' (unless NO_OPS)
//   the reference puts no stop_gradient on v / u', so the backward differentiates the iteration.
// In the reference each weight costs ~20 tiny dependent TF ops per D instantiation (pure latency);
// here every SN weight of the discriminator is one workgroup of ONE launch: the 4 mat-vecs run as
// wavefront reductions out of LDS/L2.
#include "common.h"

#define SN_MAX_K 4096
#define SN_MAX_C 1024
#define SN_BATCH 24
#define SN_EPS 1e-12f
#define SN_NT 1024          // threads per workgroup (16 wavefronts): one workgroup per weight, latency-bound
#define SN_NW (SN_NT / 64)

__device__ __forceinline__ float block_sum_nt(float v, float* red /* >= SN_NW floats */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < SN_NW; ++i) r += red[i];
  return r;
}

struct SnBatch { rcgan_sn_item it[SN_BATCH]; };
struct SnBwdBatch { rcgan_sn_bwd_item it[SN_BATCH]; };

// save layout: a[k] v[k] b[c] u2[c] uin[c] {na, nb, sigma, 0}
__global__ __launch_bounds__(SN_NT) void sn_fwd_kernel(SnBatch batch) {
  __shared__ float a_s[SN_MAX_K];
  __shared__ float red[SN_NW];
  __shared__ float part[SN_NT];
  const rcgan_sn_item it = batch.it[blockIdx.x];
  const int k = it.k, c = it.c;
  const float* w = it.w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* sv_a = it.save;
  float* sv_v = sv_a + k;
  float* sv_b = sv_v + k;
  float* sv_u2 = sv_b + c;
  float* sv_uin = sv_u2 + c;
  float* sv_s = sv_uin + c;

  // a = W u  (one wavefront per row)
  float na2 = 0.f;
  for (int r = wave; r < k; r += SN_NW) {
    float s = 0.f;
    for (int j = lane; j < c; j += 64) s += w[(long)r * c + j] * it.u[j];
    s = wave_sum(s);
    if (lane == 0) { a_s[r] = s; na2 += s * s; }
  }
  na2 = block_sum_nt(lane == 0 ? na2 : 0.f, red);
  const float na = sqrtf(na2);
  const float inv_na = 1.f / (na + SN_EPS);
  for (int r = tid; r < k; r += SN_NT) {
    float av = a_s[r];
    sv_a[r] = av;
    float v = av * inv_na;
    sv_v[r] = v;
    a_s[r] = v;            // a_s now holds v
  }
  __syncthreads();
  // b = v W : thread -> (column, k-lane)
  int cpad = 1;
  while (cpad < c && cpad < SN_NT) cpad <<= 1;
  const int klanes = SN_NT / cpad;
  const int kl = tid / cpad;
  float nb2 = 0.f;
  for (int cb = 0; cb < c; cb += cpad) {
    const int col = cb + (tid % cpad);
    float s = 0.f;
    if (col < c) {
#pragma unroll 4
      for (int r = kl; r < k; r += klanes) s += a_s[r] * w[(long)r * c + col];
    }
    part[tid] = s;
    __syncthreads();
    if (tid < cpad && col < c) {
      float t = 0.f;
      for (int q = 0; q < klanes; ++q) t += part[q * cpad + tid];
      sv_b[col] = t;
      nb2 += t * t;
    }
    __syncthreads();
  }
  nb2 = block_sum_nt(nb2, red);
  const float nb = sqrtf(nb2);
  const float inv_nb = 1.f / (nb + SN_EPS);
  // u' = b/(|b|+eps); sigma = b . u'
  float sg = 0.f;
  for (int j = tid; j < c; j += SN_NT) {
    float b = sv_b[j];
    float u2 = b * inv_nb;
    sv_u2[j] = u2;
    sv_uin[j] = it.u[j];
    sg += b * u2;
  }
  sg = block_sum_nt(sg, red);
  if (it.update)
    for (int j = tid; j < c; j += SN_NT) it.u[j] = sv_b[j] * inv_nb;
  if (tid == 0) {
    sv_s[0] = na; sv_s[1] = nb; sv_s[2] = sg; sv_s[3] = 0.f;
    *it.sigma = sg;
  }
}

__global__ __launch_bounds__(SN_NT) void sn_bwd_kernel(SnBwdBatch batch) {
  __shared__ float dv_s[SN_MAX_K];
  __shared__ float db_s[SN_MAX_C];
  __shared__ float red[SN_NW];
  const rcgan_sn_bwd_item it = batch.it[blockIdx.x];
  const int k = it.k, c = it.c;
  const float* w = it.w;
  const float* g = it.dwbar;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* sv_a = it.save;
  const float* sv_v = sv_a + k;
  const float* sv_b = sv_v + k;
  const float* sv_u2 = sv_b + c;
  const float* sv_uin = sv_u2 + c;
  const float* sv_s = sv_uin + c;
  const float na = sv_s[0], nb = sv_s[1], sigma = sv_s[2];
  const long total = (long)k * c;

  float gw = 0.f;
  if ((total & 3) == 0) {
    const float4* g4 = (const float4*)g;
    const float4* w4 = (const float4*)w;
#pragma unroll 4
    for (long i = tid; i < total / 4; i += SN_NT) {
      float4 a = g4[i], b = w4[i];
      gw += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
  } else {
    for (long i = tid; i < total; i += SN_NT) gw += g[i] * w[i];
  }
  gw = block_sum_nt(gw, red);
  const float dsigma = -gw / (sigma * sigma);
  // sigma = b.u2, u2 = b/(nb+eps)
  float dot = 0.f;
  for (int j = tid; j < c; j += SN_NT) { float b = sv_b[j]; dot += dsigma * b * b; }
  dot = block_sum_nt(dot, red);
  const float inv_nb = 1.f / (nb + SN_EPS);
  const float coef_b = dot / (nb * (nb + SN_EPS) * (nb + SN_EPS));
  for (int j = tid; j < c; j += SN_NT) {
    float b = sv_b[j];
    db_s[j] = dsigma * sv_u2[j] + dsigma * b * inv_nb - b * coef_b;
  }
  __syncthreads();
  // dv = W db  (wavefront per row); dva = dv . a
  float dva = 0.f;
  for (int r = wave; r < k; r += SN_NW) {
    float s = 0.f;
    for (int j = lane; j < c; j += 64) s += w[(long)r * c + j] * db_s[j];
    s = wave_sum(s);
    if (lane == 0) { dv_s[r] = s; dva += s * sv_a[r]; }
  }
  dva = block_sum_nt(lane == 0 ? dva : 0.f, red);
  const float inv_na = 1.f / (na + SN_EPS);
  const float coef_a = dva / (na * (na + SN_EPS) * (na + SN_EPS));
  for (int r = tid; r < k; r += SN_NT) dv_s[r] = dv_s[r] * inv_na - sv_a[r] * coef_a;   // da
  __syncthreads();
  const float inv_sigma = 1.f / sigma;
  if (SN_NT % c == 0) {
    // thread -> fixed column, rows strided: no per-element division
    const int j = tid % c, rstep = SN_NT / c;
    const float dbj = db_s[j], uj = sv_uin[j];
#pragma unroll 4
    for (int r = tid / c; r < k; r += rstep) {
      const long i = (long)r * c + j;
      float v = g[i] * inv_sigma + sv_v[r] * dbj + dv_s[r] * uj;
      if (it.accumulate) v += it.dw[i];
      it.dw[i] = v;
    }
  } else {
    for (long i = tid; i < total; i += SN_NT) {
      int r = (int)(i / c), j = (int)(i % c);
      float v = g[i] * inv_sigma + sv_v[r] * db_s[j] + dv_s[r] * sv_uin[j];
      if (it.accumulate) v += it.dw[i];
      it.dw[i] = v;
    }
  }
}

extern "C" {

size_t rcgan_sn_save_floats(int k, int c) { return (size_t)2 * k + 3 * (size_t)c + 4; }

int rcgan_sn_power_iter(rcgan_ctx* ctx, const rcgan_sn_item* items, int n_items) {
  for (int base = 0; base < n_items; base += SN_BATCH) {
    SnBatch b;
    int n = n_items - base < SN_BATCH ? n_items - base : SN_BATCH;
    for (int i = 0; i < n; ++i) {
      b.it[i] = items[base + i];
      if (b.it[i].k > SN_MAX_K || b.it[i].c > SN_MAX_C || b.it[i].k < 1 || b.it[i].c < 1)
        RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "sn weight [%d,%d]", b.it[i].k, b.it[i].c);
    }
    hipLaunchKernelGGL(sn_fwd_kernel, dim3(n), dim3(SN_NT), 0, ctx->stream, b);
    RC_LAUNCH_CHECK(ctx);
  }
  return RCGAN_OK;
}

int rcgan_sn_bwd(rcgan_ctx* ctx, const rcgan_sn_bwd_item* items, int n_items) {
  for (int base = 0; base < n_items; base += SN_BATCH) {
    SnBwdBatch b;
    int n = n_items - base < SN_BATCH ? n_items - base : SN_BATCH;
    for (int i = 0; i < n; ++i) {
      b.it[i] = items[base + i];
      if (b.it[i].k > SN_MAX_K || b.it[i].c > SN_MAX_C || b.it[i].k < 1 || b.it[i].c < 1)
        RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "sn weight [%d,%d]", b.it[i].k, b.it[i].c);
    }
    hipLaunchKernelGGL(sn_bwd_kernel, dim3(n), dim3(SN_NT), 0, ctx->stream, b);
    RC_LAUNCH_CHECK(ctx);
  }
  return RCGAN_OK;
}

}  // extern "C"
