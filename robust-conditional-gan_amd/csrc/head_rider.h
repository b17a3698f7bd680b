// The projection head's argument block and the body of its last parameter-gradient kernel, shared with the launches they can ride in
// (loss.hip has the description of the head).  DEFERRED parameter gradients (rcgan_head_desc::defer_ws): nothing in the backward
// pass waits for dE / dW_e / dtable / dw_out / db_*, yet as launches of their own they sit, ~21 us long, in the middle of the
// critic step's dependency chain.  Deferred, the head leaves {arguments, stage} in the context and
//   stage 1: the small-left GEMM dE = dlogit^T feat rides as cdiv(d, 16) extra workgroups in the next rcgan_dtrunk backward launch
//            (128 workgroups of one per CU: half the chip is idle beside them),
//   stage 2: the dW_e / dtable / dw_out / db_e sums ride as trailing workgroups of the pass's grouped filter-gradient launch,
// and rcgan_head_flush launches whatever is still pending on its own (a graph without those launches, or an unusual order).
#pragma once
#include "common.h"
#include "small_gemm.h"

#define HEAD_MAX_D 256
#define HEAD_MAX_N 1024
struct HeadPart { int rows, kind; const int32_t* labels; const float* wts; float* dwts; };
struct HeadArgs {
  int n, d, v, e_dim;
  HeadPart part[2];
  float weight;
  float gs_host; const float* gs_dev;      // gradient scale (rcgan_set_grad_scale)
  const float *feat, *w_out, *sigma_out, *b_out, *table, *w_e, *sigma_e, *b_e;
  float *loss_acc, *logits, *dfeat, *dw_out, *db_out, *dtable, *dw_e, *db_e;
  // optional: the features are pooled HERE from the trunk's output x [n][hw][d] (feat = mean over hw of act(x), written to
  // feat_out for the parameter-gradient kernels) and the gradient goes straight back to dx [n][hw][d] -- the two
  // act_meanhw launches around the head disappear (d % 128 == 0)
  const void* x; void* dx; float* feat_out; int hw, act;
};


struct HeadWgradRider { int blocks; const float* dEg; HeadArgs a; };
static inline int head_wgrad_blocks(const HeadArgs& a) { return cdiv(a.e_dim * a.d, 256) + cdiv(a.e_dim, 4) + 1; }
// host side (loss.hip): hand the pending stage to a launch that carries it; false when nothing is pending
extern "C" bool head_take_gemm(rcgan_ctx* ctx, SmallGemmArgs* out);
extern "C" bool head_take_wgrad(rcgan_ctx* ctx, HeadWgradRider* out);

// items: [0, ed*d) dW_e;  then ed wavefront-items for dtable (one k each);  then d items for dw_out and d for db_e
__device__ __forceinline__ void head_wgrad_body(const HeadArgs& a, const float* dEg, const int b, float* hs) {
  const int d = a.d, v = a.v, ed = a.e_dim;
  float* dE = hs;                   // [v+1][d]
  for (int i = threadIdx.x; i < (v + 1) * d; i += 256) dE[i] = dEg[i];
  __syncthreads();
  const int nb_w = (ed * d + 255) / 256;                  // blocks of the dW_e range
  const int nb_t = (ed + 3) / 4;                          // blocks of the dtable range (4 wavefronts = 4 k per block)
  if (b < nb_w) {
    const int o = b * 256 + threadIdx.x;
    if (o < ed * d && a.dw_e) {
      const int k = o / d, j = o - k * d;
      float acc = 0.f;
      for (int l = 0; l < v; ++l) acc += a.table[l * ed + k] * dE[l * d + j];
      a.dw_e[o] += acc;
    }
  } else if (b < nb_w + nb_t) {
    const int k = (b - nb_w) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k < ed && a.dtable) {
      const float inv_se = a.sigma_e ? 1.f / a.sigma_e[0] : 1.f;
      float w[HEAD_MAX_D / 64];
#pragma unroll
      for (int q = 0; q < HEAD_MAX_D / 64; ++q) { const int j = lane + q * 64; w[q] = j < d ? a.w_e[(long)k * d + j] : 0.f; }
      for (int l = 0; l < v; ++l) {
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < HEAD_MAX_D / 64; ++q) { const int j = lane + q * 64; dot += j < d ? w[q] * dE[l * d + j] : 0.f; }
        dot = wave_sum(dot);
        if (lane == 0) a.dtable[(long)l * ed + k] += dot * inv_se;
      }
    }
  } else {
    for (int j = threadIdx.x; j < d; j += 256) {
      if (a.dw_out) a.dw_out[j] += dE[v * d + j];
      if (a.db_e) { float tot = 0.f; for (int l = 0; l < v; ++l) tot += dE[l * d + j]; a.db_e[j] += tot; }
    }
  }
}

