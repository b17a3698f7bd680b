// The small-left GEMM of the projection head (loss.hip) as a device function, so that the label-embedding product can also ride in
// the batched filter-preparation launch (conv_mfma.hip): it depends on parameters only and would otherwise sit, 12-15 us long, in
// the middle of every step's dependency chain.
#pragma once
#include "common.h"

#define HEAD_MAX_V 16
#define SG_KC 320
#define SG_UB (SG_KC / 16)
struct SmallGemmArgs {
  int L, K, d;
  const float* A; int lda_l, lda_k;
  const float* B;
  const float* sigma;      // scale = 1 / sigma[0], or 1 if null
  const float* bias;       // [d] or null
  float* out;              // [L][d]
  float* rowsum_out; int rowsum_row;
};
#define SG_AS_FLOATS ((HEAD_MAX_V + 1) * SG_KC)
#define SG_RED_FLOATS (16 * (HEAD_MAX_V + 1) * 16)
// bx: the workgroup's index among the cdiv(d, 16) workgroups of the product; As / redp: SG_AS_FLOATS / SG_RED_FLOATS floats of LDS
__device__ __forceinline__ void small_gemm_body(const SmallGemmArgs& g, int bx, float* As, float* redp) {
  float (*red)[HEAD_MAX_V + 1][16] = (float (*)[HEAD_MAX_V + 1][16])redp;
  const int t = threadIdx.x, jj = t & 15, kl = t >> 4, j = bx * 16 + jj;
  const int L = g.L, d = g.d;
  float acc[HEAD_MAX_V + 1];
#pragma unroll
  for (int l = 0; l <= HEAD_MAX_V; ++l) acc[l] = 0.f;
  float rs = 0.f;
  const float scale = g.sigma ? 1.f / g.sigma[0] : 1.f;
  const float bj = (g.bias && j < d) ? g.bias[j] : 0.f;
  for (int kc0 = 0; kc0 < g.K; kc0 += SG_KC) {
    const int kc = min(SG_KC, g.K - kc0);
    float w[SG_UB];
#pragma unroll
    for (int u = 0; u < SG_UB; ++u) {
      const int k = kl + 16 * u;
      w[u] = (k < kc && j < d) ? g.B[(long)(kc0 + k) * d + j] : 0.f;
    }
    if (kc0) __syncthreads();                   // the previous chunk of A has been consumed
    for (int i = t; i < L * kc; i += 256) {
      const int l = i / kc, k = i - l * kc;
      As[l * SG_KC + k] = g.A[(long)l * g.lda_l + (long)(kc0 + k) * g.lda_k];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SG_UB; ++u) {
      const int k = kl + 16 * u;
      if (k < kc) {
#pragma unroll
        for (int l = 0; l <= HEAD_MAX_V; ++l)
          if (l < L) acc[l] += As[l * SG_KC + k] * w[u];
      }
    }
    if (g.rowsum_out && bx == 0 && t < 64)
      for (int k = t; k < kc; k += 64) rs += As[g.rowsum_row * SG_KC + k];
  }
#pragma unroll
  for (int l = 0; l <= HEAD_MAX_V; ++l)
    if (l < L) red[kl][l][jj] = acc[l];
  __syncthreads();
  if (t < L * 16) {
    const int l = t >> 4;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; q += 2) { s0 += red[q][l][jj]; s1 += red[q + 1][l][jj]; }
    if (j < d) g.out[l * d + j] = (s0 + s1) * scale + bj;
  }
  if (g.rowsum_out && bx == 0 && t < 64) {
    rs = wave_sum(rs);
    if (t == 0) g.rowsum_out[0] += rs;
  }
}

