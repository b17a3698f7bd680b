// Spectral normalisation: one power iteration per weight, forward and backward, batched.
//   reference: mnist/sn.py:31-75 == cifar10/common/ops/sn.py:31-75
//     v = l2n(u W^T); u' = l2n(v W); sigma = v W u'^T; W_bar = W / sigma; u <- u' (unless NO_OPS)
//   the reference puts no stop_gradient on v / u', so the backward differentiates the iteration.
// In the reference each weight costs ~20 tiny dependent TF ops per D instantiation (pure latency);
// here every SN weight of the discriminator is handled by the same 2 (forward) / 3 (backward) launches.
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

// backward 1/3: the chunk's share of <dW_bar, W>
__global__ __launch_bounds__(SN_NT) void sn_bwd_gw_kernel(SnBwdBatch batch) {
  __shared__ float red[SN_NW];
  const rcgan_sn_bwd_item it = batch.it[blockIdx.y];
  const int k = it.k, c = it.c;
  const int r0 = blockIdx.x * SN_RB;
  if (r0 >= k) return;
  const int rows = min(SN_RB, k - r0);
  const SnLayout L = sn_layout(it.save, k, c);
  const float* w = it.w + (long)r0 * c;
  const float* g = it.dwbar + (long)r0 * c;
  const long total = (long)rows * c;
  const int tid = threadIdx.x;
  float gw = 0.f;
  if ((c & 3) == 0) {
    const float4* g4 = (const float4*)g;
    const float4* w4 = (const float4*)w;
    for (long i = tid; i < total / 4; i += SN_NT) {
      float4 a = g4[i], b = w4[i];
      gw += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
  } else {
    for (long i = tid; i < total; i += SN_NT) gw += g[i] * w[i];
  }
  gw = block_sum_nt(gw, red);
  if (tid == 0) L.pgw[blockIdx.x] = gw;
}

// backward 2/3: dsigma -> db (every workgroup recomputes the c-vector), dv = W db for the chunk's rows
__global__ __launch_bounds__(SN_NT) void sn_bwd_dv_kernel(SnBwdBatch batch) {
  __shared__ float db_s[SN_MAX_C];
  __shared__ float red[SN_NW];
  const rcgan_sn_bwd_item it = batch.it[blockIdx.y];
  const int k = it.k, c = it.c;
  const int r0 = blockIdx.x * SN_RB;
  if (r0 >= k) return;
  const int rows = min(SN_RB, k - r0);
  const SnLayout L = sn_layout(it.save, k, c);
  const float* w = it.w + (long)r0 * c;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float na = L.s[0], nb = L.s[1], sigma = L.s[2];
  (void)na;
  float gw = 0.f;
  for (int q = tid; q < L.chunks; q += SN_NT) gw += L.pgw[q];
  gw = block_sum_nt(gw, red);
  const float dsigma = -gw / (sigma * sigma);
  // sigma = b.u2, u2 = b/(nb+eps)
  float dot = 0.f;
  for (int j = tid; j < c; j += SN_NT) { float b = L.b[j]; dot += dsigma * b * b; }
  dot = block_sum_nt(dot, red);
  const float inv_nb = 1.f / (nb + SN_EPS);
  const float coef_b = dot / (nb * (nb + SN_EPS) * (nb + SN_EPS));
  for (int j = tid; j < c; j += SN_NT) {
    float b = L.b[j];
    float d = dsigma * L.u2[j] + dsigma * b * inv_nb - b * coef_b;
    db_s[j] = d;
    if (blockIdx.x == 0) L.db[j] = d;
  }
  __syncthreads();
  float dva = 0.f;
  float part[SN_RB / SN_NW], av[SN_RB / SN_NW];
#pragma unroll
  for (int i = 0; i < SN_RB / SN_NW; ++i) {          // every row of the wavefront requested before the first reduction
    const int r = wave + i * SN_NW;
    float s = 0.f;
    if (r < rows)
      for (int j = lane; j < c; j += 64) s += w[(long)r * c + j] * db_s[j];
    part[i] = s;
    av[i] = (lane == 0 && r < rows) ? L.a[r0 + r] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < SN_RB / SN_NW; ++i) {
    const int r = wave + i * SN_NW;
    const float s = wave_sum(part[i]);
    if (lane == 0 && r < rows) { L.dv[r0 + r] = s; dva += s * av[i]; }
  }
  dva = block_sum_nt(lane == 0 ? dva : 0.f, red);
  if (tid == 0) L.pdva[blockIdx.x] = dva;
}

// backward 3/3: da for the chunk's rows, dW = dW_bar/sigma + v (x) db + da (x) u_in
__global__ __launch_bounds__(SN_NT) void sn_bwd_dw_kernel(SnBwdBatch batch) {
  __shared__ float da_s[SN_RB];
  __shared__ float v_s[SN_RB];
  __shared__ float red[SN_NW];
  const rcgan_sn_bwd_item it = batch.it[blockIdx.y];
  const int k = it.k, c = it.c;
  const int r0 = blockIdx.x * SN_RB;
  if (r0 >= k) return;
  const int rows = min(SN_RB, k - r0);
  const SnLayout L = sn_layout(it.save, k, c);
  const float* g = it.dwbar + (long)r0 * c;
  float* dw = it.dw + (long)r0 * c;
  const int tid = threadIdx.x;
  const float na = L.s[0], sigma = L.s[2];
  float dva = 0.f;
  for (int q = tid; q < L.chunks; q += SN_NT) dva += L.pdva[q];
  dva = block_sum_nt(dva, red);
  const float inv_na = 1.f / (na + SN_EPS);
  const float coef_a = dva / (na * (na + SN_EPS) * (na + SN_EPS));
  if (tid < rows) {
    da_s[tid] = L.dv[r0 + tid] * inv_na - L.a[r0 + tid] * coef_a;
    v_s[tid] = L.v[r0 + tid];
  }
  __syncthreads();
  const float inv_sigma = 1.f / sigma;
  if ((c & 3) == 0) {
    const int c4 = c >> 2;
    const float4* g4 = (const float4*)g;
    float4* dw4 = (float4*)dw;
    const float4* db4 = (const float4*)L.db;
    const float4* u4 = (const float4*)L.uin;
    for (int i = tid; i < rows * c4; i += SN_NT) {
      const int r = i / c4, j = i - r * c4;
      const float4 gv = g4[i], d = db4[j], u = u4[j];
      const float vr = v_s[r], ar = da_s[r];
      float4 o;
      o.x = gv.x * inv_sigma + vr * d.x + ar * u.x;
      o.y = gv.y * inv_sigma + vr * d.y + ar * u.y;
      o.z = gv.z * inv_sigma + vr * d.z + ar * u.z;
      o.w = gv.w * inv_sigma + vr * d.w + ar * u.w;
      if (it.accumulate) { float4 p = dw4[i]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
      dw4[i] = o;
    }
  } else {
    for (int i = tid; i < rows * c; i += SN_NT) {
      const int r = i / c, j = i - r * c;
      float o = g[i] * inv_sigma + v_s[r] * L.db[j] + da_s[r] * L.uin[j];
      if (it.accumulate) o += dw[i];
      dw[i] = o;
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

int rcgan_sn_bwd(rcgan_ctx* ctx, const rcgan_sn_bwd_item* items, int n_items) {
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
    hipLaunchKernelGGL(sn_bwd_gw_kernel, grid, dim3(SN_NT), 0, ctx->stream, b);
    RC_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(sn_bwd_dv_kernel, grid, dim3(SN_NT), 0, ctx->stream, b);
    RC_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(sn_bwd_dw_kernel, grid, dim3(SN_NT), 0, ctx->stream, b);
    RC_LAUNCH_CHECK(ctx);
  }
  return RCGAN_OK;
}

}  // extern "C"
