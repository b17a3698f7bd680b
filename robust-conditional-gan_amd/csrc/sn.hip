// Spectral normalisation: one power iteration per weight, forward and backward, batched.
//   reference: mnist/sn.py:31-75 == cifar10/common/ops/sn.py:31-75
//     v = l2n(u W^T); u' = l2n(v W); sigma = v W u'^T; W_bar = W / sigma; u <- u' (unless NO_OPS)
//   the reference puts no stop_gradient on v / u', so the backward differentiates the iteration.
// In the reference each weight costs ~20 tiny dependent TF ops per D instantiation (pure latency);
// here every SN weight of the discriminator is handled by the same 2 (forward) / 3 (backward) launches.
#include <algorithm>
#include <utility>
#include <vector>

#include "common.h"

#define SN_MAX_K 4096
#define SN_MAX_C 1024
#define SN_BATCH 24
#define SN_EPS 1e-12f
#define SN_RB 32            // rows of W per workgroup
#define SN_NT 256           // threads per workgroup (4 wavefronts)
#define SN_NW (SN_NT / 64)
#define SN_MAX_CHUNKS (SN_MAX_K / SN_RB)

// Each weight is cut into chunks of SN_RB rows; a launch is grid (chunks, weights), so that the
// biggest filters ([1152,128] = 590 KB) are read by 36 workgroups instead of one.  The scalar
// couplings between the phases of the iteration (|a|, |b|, sigma, <G,W>, <dv,a>) are resolved
// between launches through a few partial sums kept in the save buffer:
//   save layout: a[k'] v[k'] dv[k'] | b[c'] u2[c'] uin[c'] db[c'] (k', c' = k, c rounded up to 4) | {na, nb, sigma, 0} | pb[chunks][c] | pgw[chunks] pdva[chunks]
struct SnLayout {
  float *a, *v, *dv, *b, *u2, *uin, *db, *s, *pb, *pgw, *pdva;
  int chunks;
};
__host__ __device__ inline int sn_chunks(int k) { return (k + SN_RB - 1) / SN_RB; }
__device__ __forceinline__ SnLayout sn_layout(float* save, int k, int c) {
  SnLayout L;
  L.chunks = sn_chunks(k);
  const int kp = (k + 3) & ~3, cp = (c + 3) & ~3;      // 16-byte aligned segments (float4 access)
  L.a = save; L.v = L.a + kp; L.dv = L.v + kp;
  L.b = L.dv + kp; L.u2 = L.b + cp; L.uin = L.u2 + cp; L.db = L.uin + cp;
  L.s = L.db + cp;
  L.pb = L.s + 4;
  L.pgw = L.pb + (long)L.chunks * c;
  L.pdva = L.pgw + L.chunks;
  return L;
}

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

struct SnBatch { rcgan_sn_item it[SN_BATCH]; unsigned* arrive; /* one self-resetting arrival counter per weight, a 128-byte line apart; NULL: two launches */ };
struct SnBwdBatch { rcgan_sn_bwd_item it[SN_BATCH]; };

template <bool AGENT> __device__ __forceinline__ void sn_fwd_finish(const rcgan_sn_item& it, float* red);

// forward 1/2: a = W u for the chunk's rows, and the chunk's share of a W (= |a| * b)
__global__ __launch_bounds__(SN_NT) void sn_fwd_rows_kernel(SnBatch batch) {
  __shared__ float a_s[SN_RB];
  const rcgan_sn_item it = batch.it[blockIdx.y];
  const int k = it.k, c = it.c;
  const int r0 = blockIdx.x * SN_RB;
  if (r0 >= k) return;
  const int rows = min(SN_RB, k - r0);
  const SnLayout L = sn_layout(it.save, k, c);
  const float* w = it.w + (long)r0 * c;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (every row of the wavefront requested before the first reduction: one memory round trip instead of SN_RB / SN_NW)
  float part[SN_RB / SN_NW];
#pragma unroll
  for (int i = 0; i < SN_RB / SN_NW; ++i) {
    const int r = wave + i * SN_NW;
    float s = 0.f;
    if (r < rows)
      for (int j = lane; j < c; j += 64) s += w[(long)r * c + j] * it.u[j];
    part[i] = s;
  }
#pragma unroll
  for (int i = 0; i < SN_RB / SN_NW; ++i) {
    const int r = wave + i * SN_NW;
    const float s = wave_sum(part[i]);
    // (agent-scope stores: written through to the memory side, visible to the finishing workgroup on whatever XCD it runs)
    if (lane == 0 && r < rows) { a_s[r] = s; __hip_atomic_store(L.a + r0 + r, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  }
  __syncthreads();
  for (int j = tid; j < c; j += SN_NT) {
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < rows; ++r) s += a_s[r] * w[(long)r * c + j];
    __hip_atomic_store(L.pb + (long)blockIdx.x * c + j, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (batch.arrive == nullptr) return;
  // ---- the LAST workgroup of a weight to arrive finishes it (round 4: sn_fwd_finish_kernel was a launch of its own, 6.4 us of pure
  //      dependency in front of every critic step's filter preparation).  u is only rewritten here, after every workgroup has read it.
  __shared__ int is_last;
  __shared__ float red[SN_NW];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's stores are performed ...
  __syncthreads();                                       // ... and every wavefront's, before the workgroup signals
  if (tid == 0) {
    unsigned* ctr = batch.arrive + blockIdx.y * 32;
    const unsigned prev = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = prev == (unsigned)L.chunks - 1u ? 1 : 0;
    if (is_last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
  }
  __syncthreads();
  if (!is_last) return;
  sn_fwd_finish<true>(it, red);
}

// forward 2/2 (one workgroup per weight): norms, v, b, u', sigma, u update.  AGENT: a and pb were written by OTHER workgroups of the SAME
// launch (the fused form below): read them with agent-scope loads, served from the memory side, never from a stale cache line
template <bool AGENT>
__device__ __forceinline__ void sn_fwd_finish(const rcgan_sn_item& it, float* red) {
  const int k = it.k, c = it.c;
  const SnLayout L = sn_layout(it.save, k, c);
  const int tid = threadIdx.x;
  auto ld = [&](const float* p) __attribute__((always_inline)) -> float {
    return AGENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
  };
  float na2 = 0.f;
  for (int r = tid; r < k; r += SN_NT) { float a = ld(L.a + r); na2 += a * a; }
  na2 = block_sum_nt(na2, red);
  const float na = sqrtf(na2);
  const float inv_na = 1.f / (na + SN_EPS);
  for (int r = tid; r < k; r += SN_NT) L.v[r] = ld(L.a + r) * inv_na;
  float nb2 = 0.f;
  for (int j = tid; j < c; j += SN_NT) {
    float s = 0.f;
#pragma unroll 12
    for (int q = 0; q < L.chunks; ++q) s += ld(L.pb + (long)q * c + j);      // (independent loads, requested a dozen at a time)
    s *= inv_na;
    L.b[j] = s;
    nb2 += s * s;
  }
  nb2 = block_sum_nt(nb2, red);
  const float nb = sqrtf(nb2);
  const float inv_nb = 1.f / (nb + SN_EPS);
  float sg = 0.f;
  for (int j = tid; j < c; j += SN_NT) {
    float b = L.b[j];
    float u2 = b * inv_nb;
    L.u2[j] = u2;
    L.uin[j] = it.u[j];
    if (it.update) it.u[j] = u2;
    sg += b * u2;
  }
  sg = block_sum_nt(sg, red);
  if (tid == 0) {
    L.s[0] = na; L.s[1] = nb; L.s[2] = sg; L.s[3] = 0.f;
    *it.sigma = sg;
  }
}
__global__ __launch_bounds__(SN_NT) void sn_fwd_finish_kernel(SnBatch batch) {
  __shared__ float red[SN_NW];
  sn_fwd_finish<false>(batch.it[blockIdx.x], red);
}

// ---- backward (round 6: two launches instead of three, optionally with the optimiser in the second) ----------------------------
// dW = dW_bar / sigma + v (x) db + da (x) u_in, with  dsigma = -<dW_bar, W> / sigma^2,  db = dsigma * q,
//   q = u2 + b / (nb + eps) - b * |b|^2 / (nb (nb + eps)^2)            (a c-vector of FORWARD quantities only),
//   dv = W db = dsigma * p,  p = W q,   <dv, a> = dsigma * <p, a>,   da = dv / (na + eps) - a <dv, a> / (na (na + eps)^2).
// Rounds 1-5 ran <dW_bar, W> | dv = W db | dW as three launches because db needs dsigma.  But db is dsigma TIMES a vector that does
// not depend on the gradient: p = W q and <p, a> come out of the SAME pass over W that forms <dW_bar, W>, and dsigma multiplies
// them afterwards.  Pass 1: per 32-row chunk <dW_bar, W>, p and <p, a>.  Pass 2: the scalars, da, dW -- and, for a single-rank
// step with a static loss scale (rcgan_sn_bwd_adam), TF-Adam on the rows just produced, with rider workgroups for the
// slab's other parameters: the critic step's optimiser launch and the 8 us in front of it are gone.
#define SN_MAX_RANGES 48
struct SnAdam {
  float *w_base, *m_base, *v_base, *g_base;      // the optimiser group's slabs; an item's w / dw point into w_base / g_base at the same offset
  const float* hyper;                            // device {lr, t}: t was advanced by pass 1 of the same call
  float beta1, beta2, eps, clip, grad_scale;
  int riders;                                    // this launch carries the rider row (the last batch of a call)
  int n_ranges;
  unsigned lo[SN_MAX_RANGES], hi[SN_MAX_RANGES]; // float offsets [lo, hi) of the slab NOT covered by the items
};

__device__ __forceinline__ void sn_adam_elem(float g, float& m, float& v, float& w, float alpha, float omb1, float omb2, float eps, float clip,
                                             float grad_scale) {
  // (the operation sequence of adam_tf_kernel, api.hip)
  const float gi = g * grad_scale;
  m = m + (gi - m) * omb1;
  v = v + (gi * gi - v) * omb2;
  w = w - (m * alpha) / (sqrtf(v) + eps);
  if (clip > 0.f) w = fminf(fmaxf(w, -clip), clip);
}

// pass 1/2: q (every workgroup recomputes the c-vector), and for the chunk's rows <dW_bar, W>, p = W q, <p, a>
__global__ __launch_bounds__(SN_NT) void sn_bwd_p1_kernel(SnBwdBatch batch, float* hyper_inc) {
  __shared__ float q_s[SN_MAX_C];
  __shared__ float red[SN_NW];
  // the optimiser's step count moves on HERE, one launch in front of the one that reads it (rcgan_sn_bwd_adam)
  if (hyper_inc != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) hyper_inc[1] += 1.f;
  const rcgan_sn_bwd_item it = batch.it[blockIdx.y];
  const int k = it.k, c = it.c;
  const int r0 = blockIdx.x * SN_RB;
  if (r0 >= k) return;
  const int rows = min(SN_RB, k - r0);
  const SnLayout L = sn_layout(it.save, k, c);
  const float* w = it.w + (long)r0 * c;
  const float* g = it.dwbar + (long)r0 * c;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // c <= 128 (every weight of these critics): the wavefront's rows of W and dW_bar are requested HERE, in front of the q chain (two
  // block reductions and a barrier), and wait in registers -- one memory round trip less on the launch's critical path
  constexpr int NR = SN_RB / SN_NW;
  const bool small_c = c <= 128;
  float wr[NR][2], gr[NR][2];
  if (small_c) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = wave + i * SN_NW;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = lane + 64 * e;
        const bool ok = r < rows && j < c;
        wr[i][e] = ok ? w[(long)r * c + j] : 0.f;
        gr[i][e] = ok ? g[(long)r * c + j] : 0.f;
      }
    }
  }
  const float nb = L.s[1];
  float s2 = 0.f;
  for (int j = tid; j < c; j += SN_NT) { const float b = L.b[j]; s2 += b * b; }
  s2 = block_sum_nt(s2, red);
  const float inv_nb = 1.f / (nb + SN_EPS);
  const float coef_q = s2 / (nb * (nb + SN_EPS) * (nb + SN_EPS));
  for (int j = tid; j < c; j += SN_NT) {
    const float b = L.b[j];
    const float q = L.u2[j] + b * inv_nb - b * coef_q;
    q_s[j] = q;
    if (blockIdx.x == 0) L.db[j] = q;
  }
  __syncthreads();
  float gw = 0.f, pa = 0.f;
  float part[SN_RB / SN_NW], av[SN_RB / SN_NW];
#pragma unroll
  for (int i = 0; i < SN_RB / SN_NW; ++i) {          // every row of the wavefront requested before the first reduction
    const int r = wave + i * SN_NW;
    float s = 0.f;
    if (small_c) {
      const float q0 = lane < c ? q_s[lane] : 0.f, q1 = lane + 64 < c ? q_s[lane + 64] : 0.f;
      s = wr[i][0] * q0 + wr[i][1] * q1;
      gw += gr[i][0] * wr[i][0] + gr[i][1] * wr[i][1];
    } else if (r < rows) {
      for (int j = lane; j < c; j += 64) {
        const float wv = w[(long)r * c + j];
        s += wv * q_s[j];
        gw += g[(long)r * c + j] * wv;
      }
    }
    part[i] = s;
    av[i] = (lane == 0 && r < rows) ? L.a[r0 + r] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < SN_RB / SN_NW; ++i) {
    const int r = wave + i * SN_NW;
    const float s = wave_sum(part[i]);
    if (lane == 0 && r < rows) { L.dv[r0 + r] = s; pa += s * av[i]; }      // (the dv slot holds p = W q)
  }
  gw = block_sum_nt(gw, red);
  pa = block_sum_nt(lane == 0 ? pa : 0.f, red);
  if (tid == 0) { L.pgw[blockIdx.x] = gw; L.pdva[blockIdx.x] = pa; }
}

// pass 2/2: dsigma, da for the chunk's rows, dW = dW_bar/sigma + (dsigma v) (x) q + da (x) u_in; ADAM: the update of those rows, and
// the rider row (blockIdx.y == number of items) for the parameters between the spectrally normalised ones
template <bool ADAM>
__global__ __launch_bounds__(SN_NT) void sn_bwd_p2_kernel(SnBwdBatch batch, int n_items, SnAdam ad) {
  __shared__ float da_s[SN_RB];
  __shared__ float v_s[SN_RB];
  __shared__ float red[SN_NW];
  const int tid = threadIdx.x;
  float alpha = 0.f, omb1 = 0.f, omb2 = 0.f;
  if (ADAM) {
    const float lr = ad.hyper[0], t = ad.hyper[1];
    alpha = lr * sqrtf(1.f - powf(ad.beta2, t)) / (1.f - powf(ad.beta1, t));
    omb1 = 1.f - ad.beta1;
    omb2 = 1.f - ad.beta2;
    if ((int)blockIdx.y >= n_items) {               // riders: biases, embeddings, the slab's alignment holes (zeros stay zeros)
      for (int rg = 0; rg < ad.n_ranges; ++rg)
        for (unsigned o = ad.lo[rg] + blockIdx.x * SN_NT + tid; o < ad.hi[rg]; o += gridDim.x * SN_NT) {
          float m = ad.m_base[o], v = ad.v_base[o], wv = ad.w_base[o];
          sn_adam_elem(ad.g_base[o], m, v, wv, alpha, omb1, omb2, ad.eps, ad.clip, ad.grad_scale);
          ad.m_base[o] = m; ad.v_base[o] = v; ad.w_base[o] = wv;
        }
      return;
    }
  }
  const rcgan_sn_bwd_item it = batch.it[blockIdx.y];
  const int k = it.k, c = it.c;
  const int r0 = blockIdx.x * SN_RB;
  if (r0 >= k) return;
  const int rows = min(SN_RB, k - r0);
  const SnLayout L = sn_layout(it.save, k, c);
  const float* g = it.dwbar + (long)r0 * c;
  float* dw = it.dw + (long)r0 * c;
  // c <= 128 and a multiple of 4 (every weight of these critics): the chunk's dW_bar, the accumulate target and the optimiser's three
  // slabs are requested HERE, in front of the scalar chain (two block reductions and a barrier) -- up to four 16-byte pieces per
  // thread and tensor wait in registers
  constexpr int PF = SN_RB * 128 / 4 / SN_NT;       // 4
  const bool pf = (c & 3) == 0 && c <= 128;
  const long aoff = ADAM ? (long)(dw - ad.g_base) : 0;       // this chunk's offset inside the group's slabs
  float4 pg[PF], pd[PF], pm[PF], pv[PF], pw[PF];
  if (pf) {
    const int n4 = rows * (c >> 2);
#pragma unroll
    for (int e = 0; e < PF; ++e) {
      const int i = tid + e * SN_NT;
      const bool ok = i < n4;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      pg[e] = ok ? ((const float4*)g)[i] : z4;
      pd[e] = (ok && it.accumulate) ? ((const float4*)dw)[i] : z4;
      if (ADAM) {
        pm[e] = ok ? ((const float4*)(ad.m_base + aoff))[i] : z4;
        pv[e] = ok ? ((const float4*)(ad.v_base + aoff))[i] : z4;
        pw[e] = ok ? ((const float4*)(ad.w_base + aoff))[i] : z4;
      }
    }
  }
  const float na = L.s[0], sigma = L.s[2];
  float gw = 0.f, pa = 0.f;
  for (int q = tid; q < L.chunks; q += SN_NT) { gw += L.pgw[q]; pa += L.pdva[q]; }
  gw = block_sum_nt(gw, red);
  pa = block_sum_nt(pa, red);
  const float dsigma = -gw / (sigma * sigma);
  const float dva = dsigma * pa;
  const float inv_na = 1.f / (na + SN_EPS);
  const float coef_a = dva / (na * (na + SN_EPS) * (na + SN_EPS));
  if (tid < rows) {
    da_s[tid] = dsigma * L.dv[r0 + tid] * inv_na - L.a[r0 + tid] * coef_a;
    v_s[tid] = dsigma * L.v[r0 + tid];
  }
  __syncthreads();
  const float inv_sigma = 1.f / sigma;
  if (pf) {
    const int c4 = c >> 2, n4 = rows * c4;
    float4* dw4 = (float4*)dw;
    const float4* q4 = (const float4*)L.db;
    const float4* u4 = (const float4*)L.uin;
#pragma unroll
    for (int e = 0; e < PF; ++e) {
      const int i = tid + e * SN_NT;
      if (i >= n4) break;
      const int r = i / c4, j = i - r * c4;
      const float4 gv = pg[e], d = q4[j], u = u4[j];
      const float vr = v_s[r], ar = da_s[r];
      float4 o;
      o.x = gv.x * inv_sigma + vr * d.x + ar * u.x;
      o.y = gv.y * inv_sigma + vr * d.y + ar * u.y;
      o.z = gv.z * inv_sigma + vr * d.z + ar * u.z;
      o.w = gv.w * inv_sigma + vr * d.w + ar * u.w;
      if (it.accumulate) { o.x += pd[e].x; o.y += pd[e].y; o.z += pd[e].z; o.w += pd[e].w; }
      dw4[i] = o;
      if (ADAM) {
        float4 m = pm[e], v = pv[e], wv = pw[e];
        sn_adam_elem(o.x, m.x, v.x, wv.x, alpha, omb1, omb2, ad.eps, ad.clip, ad.grad_scale);
        sn_adam_elem(o.y, m.y, v.y, wv.y, alpha, omb1, omb2, ad.eps, ad.clip, ad.grad_scale);
        sn_adam_elem(o.z, m.z, v.z, wv.z, alpha, omb1, omb2, ad.eps, ad.clip, ad.grad_scale);
        sn_adam_elem(o.w, m.w, v.w, wv.w, alpha, omb1, omb2, ad.eps, ad.clip, ad.grad_scale);
        ((float4*)(ad.m_base + aoff))[i] = m;
        ((float4*)(ad.v_base + aoff))[i] = v;
        ((float4*)(ad.w_base + aoff))[i] = wv;
      }
    }
  } else if ((c & 3) == 0) {
    const int c4 = c >> 2;
    const float4* g4 = (const float4*)g;
    float4* dw4 = (float4*)dw;
    const float4* q4 = (const float4*)L.db;
    const float4* u4 = (const float4*)L.uin;
    for (int i = tid; i < rows * c4; i += SN_NT) {
      const int r = i / c4, j = i - r * c4;
      const float4 gv = g4[i], d = q4[j], u = u4[j];
      const float vr = v_s[r], ar = da_s[r];
      float4 o;
      o.x = gv.x * inv_sigma + vr * d.x + ar * u.x;
      o.y = gv.y * inv_sigma + vr * d.y + ar * u.y;
      o.z = gv.z * inv_sigma + vr * d.z + ar * u.z;
      o.w = gv.w * inv_sigma + vr * d.w + ar * u.w;
      if (it.accumulate) { float4 p = dw4[i]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
      dw4[i] = o;
      if (ADAM) {
        float4* m4 = (float4*)(ad.m_base + aoff) + i;
        float4* vv4 = (float4*)(ad.v_base + aoff) + i;
        float4* w4 = (float4*)(ad.w_base + aoff) + i;
        float4 m = *m4, v = *vv4, wv = *w4;
        sn_adam_elem(o.x, m.x, v.x, wv.x, alpha, omb1, omb2, ad.eps, ad.clip, ad.grad_scale);
        sn_adam_elem(o.y, m.y, v.y, wv.y, alpha, omb1, omb2, ad.eps, ad.clip, ad.grad_scale);
        sn_adam_elem(o.z, m.z, v.z, wv.z, alpha, omb1, omb2, ad.eps, ad.clip, ad.grad_scale);
        sn_adam_elem(o.w, m.w, v.w, wv.w, alpha, omb1, omb2, ad.eps, ad.clip, ad.grad_scale);
        *m4 = m; *vv4 = v; *w4 = wv;
      }
    }
  } else {
    for (int i = tid; i < rows * c; i += SN_NT) {
      const int r = i / c, j = i - r * c;
      float o = g[i] * inv_sigma + v_s[r] * L.db[j] + da_s[r] * L.uin[j];
      if (it.accumulate) o += dw[i];
      dw[i] = o;
      if (ADAM) {
        float m = ad.m_base[aoff + i], v = ad.v_base[aoff + i], wv = ad.w_base[aoff + i];
        sn_adam_elem(o, m, v, wv, alpha, omb1, omb2, ad.eps, ad.clip, ad.grad_scale);
        ad.m_base[aoff + i] = m; ad.v_base[aoff + i] = v; ad.w_base[aoff + i] = wv;
      }
    }
  }
}

extern "C" {

size_t rcgan_sn_save_floats(int k, int c) {
  const size_t kp = ((size_t)k + 3) & ~(size_t)3, cp = ((size_t)c + 3) & ~(size_t)3;
  return 3 * kp + 4 * cp + 4 + (size_t)sn_chunks(k) * ((size_t)c + 2);
}

int rcgan_sn_power_iter(rcgan_ctx* ctx, const rcgan_sn_item* items, int n_items) {
  for (int base = 0; base < n_items; base += SN_BATCH) {
    SnBatch b;
    int n = n_items - base < SN_BATCH ? n_items - base : SN_BATCH;
    int maxk = 1;
    for (int i = 0; i < n; ++i) {
      b.it[i] = items[base + i];
      if (b.it[i].k > SN_MAX_K || b.it[i].c > SN_MAX_C || b.it[i].k < 1 || b.it[i].c < 1)
        RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "sn weight [%d,%d]", b.it[i].k, b.it[i].c);
      if (b.it[i].k > maxk) maxk = b.it[i].k;
    }
    // (opt-in: it saves the second launch but not its latency -- 0.01 ms per iteration -- and the two-launch form needs no cross-workgroup
    // visibility argument)
    static const int fused = [] { const char* e = getenv("RCGAN_SN_FUSED_FINISH"); return e ? atoi(e) : 0; }();
    b.arrive = fused ? ctx->tree_counters() : nullptr;       // (SN_BATCH lines of the otherwise unused counter block)
    hipLaunchKernelGGL(sn_fwd_rows_kernel, dim3(sn_chunks(maxk), n), dim3(SN_NT), 0, ctx->stream, b);
    RC_LAUNCH_CHECK(ctx);
    if (!fused) {
      hipLaunchKernelGGL(sn_fwd_finish_kernel, dim3(n), dim3(SN_NT), 0, ctx->stream, b);
      RC_LAUNCH_CHECK(ctx);
    }
  }
  return RCGAN_OK;
}

static int sn_bwd_launch(rcgan_ctx* ctx, const rcgan_sn_bwd_item* items, int n_items, const rcgan_sn_adam* opt) {
  for (int base = 0; base < n_items; base += SN_BATCH) {
    SnBwdBatch b;
    int n = n_items - base < SN_BATCH ? n_items - base : SN_BATCH;
    int maxk = 1;
    for (int i = 0; i < n; ++i) {
      b.it[i] = items[base + i];
      if (b.it[i].k > SN_MAX_K || b.it[i].c > SN_MAX_C || b.it[i].k < 1 || b.it[i].c < 1)
        RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "sn weight [%d,%d]", b.it[i].k, b.it[i].c);
      if (b.it[i].k > maxk) maxk = b.it[i].k;
    }
    const dim3 grid(sn_chunks(maxk), n);
    // (the step count moves on once per call: in the first batch's first launch)
    hipLaunchKernelGGL(sn_bwd_p1_kernel, grid, dim3(SN_NT), 0, ctx->stream, b, (opt && base == 0) ? opt->hyper : (float*)nullptr);
    RC_LAUNCH_CHECK(ctx);
    SnAdam ad = {};
    if (opt) {
      const bool last = base + n >= n_items;
      ad.w_base = opt->w; ad.m_base = opt->m; ad.v_base = opt->v; ad.g_base = opt->g; ad.hyper = opt->hyper;
      ad.beta1 = opt->beta1; ad.beta2 = opt->beta2; ad.eps = opt->eps; ad.clip = opt->clip; ad.grad_scale = opt->grad_scale;
      ad.riders = last ? 1 : 0;
      ad.n_ranges = last ? opt->n_ranges : 0;
      for (int r = 0; r < ad.n_ranges; ++r) { ad.lo[r] = (unsigned)opt->ranges[2 * r]; ad.hi[r] = (unsigned)opt->ranges[2 * r + 1]; }
      hipLaunchKernelGGL(sn_bwd_p2_kernel<true>, dim3(grid.x, n + (last ? 1 : 0)), dim3(SN_NT), 0, ctx->stream, b, n, ad);
    } else {
      hipLaunchKernelGGL(sn_bwd_p2_kernel<false>, grid, dim3(SN_NT), 0, ctx->stream, b, n, ad);
    }
    RC_LAUNCH_CHECK(ctx);
  }
  return RCGAN_OK;
}

int rcgan_sn_bwd(rcgan_ctx* ctx, const rcgan_sn_bwd_item* items, int n_items) { return sn_bwd_launch(ctx, items, n_items, nullptr); }

int rcgan_sn_bwd_adam(rcgan_ctx* ctx, const rcgan_sn_bwd_item* items, int n_items, const rcgan_sn_adam* opt) {
  RC_REQUIRE(ctx, items && n_items >= 1 && opt && opt->w && opt->g && opt->m && opt->v && opt->hyper, "null argument");
  RC_REQUIRE(ctx, opt->count < (1ull << 32) && opt->n_ranges >= 0 && opt->n_ranges <= SN_MAX_RANGES && (opt->n_ranges == 0 || opt->ranges),
             "%zu parameters, %d ranges (at most %d)", opt->count, opt->n_ranges, SN_MAX_RANGES);
  // every item inside the slabs at the same offset in w and g, items and ranges disjoint and together covering [0, count)
  std::vector<std::pair<size_t, size_t>> iv;
  for (int i = 0; i < n_items; ++i) {
    const size_t sz = (size_t)items[i].k * items[i].c;
    RC_REQUIRE(ctx, items[i].w >= opt->w && items[i].w + sz <= opt->w + opt->count, "item %d: w outside the slab", i);
    RC_REQUIRE(ctx, items[i].dw - opt->g == items[i].w - opt->w, "item %d: dw and w at different slab offsets", i);
    iv.push_back({(size_t)(items[i].w - opt->w), (size_t)(items[i].w - opt->w) + sz});
  }
  for (int r = 0; r < opt->n_ranges; ++r) {
    RC_REQUIRE(ctx, opt->ranges[2 * r] <= opt->ranges[2 * r + 1] && opt->ranges[2 * r + 1] <= opt->count, "range %d outside the slab", r);
    iv.push_back({opt->ranges[2 * r], opt->ranges[2 * r + 1]});
  }
  std::sort(iv.begin(), iv.end());
  size_t at = 0;
  for (auto& p : iv) {
    if (p.first == p.second) continue;
    RC_REQUIRE(ctx, p.first == at, "items and ranges must tile the slab: gap or overlap at float offset %zu (next piece starts at %zu)", at, p.first);
    at = p.second;
  }
  RC_REQUIRE(ctx, at == opt->count, "items and ranges cover %zu of %zu parameters", at, opt->count);
  return sn_bwd_launch(ctx, items, n_items, opt);
}

}  // extern "C"
