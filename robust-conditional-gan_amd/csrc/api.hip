// extern "C" entry points: context, conv / dense dispatch (MFMA vs direct), optimiser, graphs.
#include "conv_mfma.h"
#include "small_gemm.h"
#include "step_inputs.h"
#include "conv_image.h"

// ---- provided by the other translation units ------------------------------------------------------

size_t direct_wgrad_ws_bytes(const rcgan_conv_desc* d);
template <typename T> int direct_fwd(rcgan_ctx*, const rcgan_conv_desc*, const T*, const float*, const float*, const float*, T*, int n_cols = 0);
template <typename T> int direct_dgrad(rcgan_ctx*, const rcgan_conv_desc*, const T*, const float*, const float*, const float*, const T*, T*, int);
template <typename T> int direct_wgrad(rcgan_ctx*, const rcgan_conv_desc*, const T*, const T*, float*, float*, int, void*, size_t);
template <typename T> int direct_wgrad_cols(rcgan_ctx*, const rcgan_conv_desc*, const T*, const T*, int, const float*, float*, int, void*, size_t);
size_t direct_wgrad_cols_ws_bytes(const rcgan_conv_desc* d, int c1);
template <typename T> int colsum_launch(rcgan_ctx*, const T*, long, int, float*, int, float*);
template <typename T> int linear_fwd(rcgan_ctx*, long, long, long, const T*, const float*, const float*, const float*, T*);
template <typename T> int linear_dgrad(rcgan_ctx*, long, long, long, const T*, const float*, const float*, T*, int);
template <typename T> int linear_wgrad(rcgan_ctx*, long, long, long, const T*, const T*, float*, float*, int, void*, size_t);
size_t linear_wgrad_ws_bytes(long m, long k, long n);
template <typename T> int sumpool2_masked_launch(rcgan_ctx*, int, int, int, int, const T*, const T*, T*, int);
int small_fwd_kind(const rcgan_conv_desc* d);
int small_dgrad_kind(const rcgan_conv_desc* d);
int small_wgrad_kind(const rcgan_conv_desc* d);
size_t small_wgrad_ws_bytes(const rcgan_conv_desc* d);
template <typename T> int small_fwd(rcgan_ctx*, const rcgan_conv_desc*, int, const T*, const float*, const float*, T*);
template <typename T> int small_dgrad(rcgan_ctx*, const rcgan_conv_desc*, int, const T*, const float*, T*, int);
template <typename T> int small_wgrad(rcgan_ctx*, const rcgan_conv_desc*, int, const T*, const T*, float*, float*, int, void*, size_t);

// out[i] (= or +=) sum_z slab[z][i]; the slabs may carry a bias-gradient tail of `nbias` floats after `count`
__global__ void slab_reduce2_kernel(const float* slab, long stride, float* out, long count, float* bias_out, int nbias, int nz,
                                    int accumulate) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count + nbias) return;
  float s = 0.f;
#pragma unroll 4
  for (int z = 0; z < nz; ++z) s += slab[(long)z * stride + i];
  float* o = i < count ? out + i : bias_out + (i - count);
  if (accumulate) s += *o;
  *o = s;
}

// the same reduction for several filter gradients in one launch (blockIdx.y = problem).  Slabs of the image-end kernels
// (conv_image.h: [32][Cb] + 32 floats per workgroup) are described by bias_off / orient: the bias sums sit at bias_off instead of
// behind the filter elements, and with orient 1 (the small side is dy) slab element k*Cb + n goes to dW[(t*Cb + n)*Cs + c],
// k = t*Cs + c.
#define REDUCE_GROUP_MAX 16
struct SlabReduceGroup {
  struct Item {
    const float* slab; long stride; float* out; long count; float* bias_out; int nbias, nz, accumulate;
    long bias_off;          // slab index of the first bias sum (= count for the three-tap / per-tap slabs)
    int orient, Cb, Cs;     // orient 1: transposed image-end mapping
    int sub;                // 1 / 2: the slab holds the 16 cells of the sub-pixel form (MfmaWgradArgs::sub), folded into the 9 taps here
    long cc;                // Cin * Cout (sub only)
    int bias_parts;         // 2: two bias partials per slab, nbias floats apart (the nine-tap kernel's upsample form); 0 / 1: one
    int vec4;               // a thread sums FOUR consecutive filter elements with 16-byte loads (count, cc, stride multiples of 4, orient 0)
  } it[REDUCE_GROUP_MAX];
};
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__global__ void slab_reduce2_group_kernel(SlabReduceGroup g) {
  const SlabReduceGroup::Item& r = g.it[blockIdx.y];
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r.vec4) {
    // threads [0, count / 4): four elements each, the same per-element order of additions as the scalar form below (bit-identical sums);
    // threads behind them: the bias tail, one element each, through the scalar form
    const long nv = r.count >> 2;
    if (i < nv) {
      const long e = i << 2;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r.sub) {
        const int tap = (int)(e / r.cc), kh = tap / 3, kw = tap - kh * 3;
        const long rem = e - (long)tap * r.cc;
        long c4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int pa = q >> 1, pb = q & 1;
          const int sa = ((r.sub == 1) == (pa == 0)) ? (kh >= 1) : (kh >= 2);
          const int sb = ((r.sub == 1) == (pb == 0)) ? (kw >= 1) : (kw >= 2);
          c4[q] = (long)(q * 4 + sa * 2 + sb) * r.cc + rem;
        }
        int z = 0;
        for (; z + 2 <= r.nz; z += 2) {
          float4 v[2][4];
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[u][q] = *(const float4*)(r.slab + (long)(z + u) * r.stride + c4[q]);
#pragma unroll
          for (int u = 0; u < 2; ++u) s = f4add(s, f4add(f4add(v[u][0], v[u][1]), f4add(v[u][2], v[u][3])));
        }
        for (; z < r.nz; ++z) {
          const float* sl = r.slab + (long)z * r.stride;
          s = f4add(s, f4add(f4add(*(const float4*)(sl + c4[0]), *(const float4*)(sl + c4[1])), f4add(*(const float4*)(sl + c4[2]), *(const float4*)(sl + c4[3]))));
        }
        if (r.sub == 2) { s.x *= 0.25f; s.y *= 0.25f; s.z *= 0.25f; s.w *= 0.25f; }
      } else {
        int z = 0;
        for (; z + 8 <= r.nz; z += 8) {
          float4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(r.slab + (long)(z + u) * r.stride + e);
#pragma unroll
          for (int u = 0; u < 8; ++u) s = f4add(s, v[u]);
        }
#pragma unroll 4
        for (; z < r.nz; ++z) s = f4add(s, *(const float4*)(r.slab + (long)z * r.stride + e));
      }
      float4* o = (float4*)(r.out + e);
      if (r.accumulate) s = f4add(s, *o);
      *o = s;
      return;
    }
    i = r.count + (i - nv);       // a bias element (or past the end)
  }
  if (i >= r.count + r.nbias) return;
  const long si = i < r.count ? i : r.bias_off + (i - r.count);
  float s = 0.f;
  if (r.sub && i < r.count) {
    // dW[kh][kw] = scale * sum over the parities (pa, pb) of G[pa][pb][S(pa, kh)][S(pb, kw)]
    const int tap = (int)(i / r.cc), kh = tap / 3, kw = tap - kh * 3;
    const long rem = i - (long)tap * r.cc;
    long c4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pa = q >> 1, pb = q & 1;
      // S(sub 1): parity 0 -> 0,1,1; parity 1 -> 0,0,1.  S(sub 2): the other way round
      const int sa = ((r.sub == 1) == (pa == 0)) ? (kh >= 1) : (kh >= 2);
      const int sb = ((r.sub == 1) == (pb == 0)) ? (kw >= 1) : (kw >= 2);
      c4[q] = (long)(q * 4 + sa * 2 + sb) * r.cc + rem;
    }
    // (the loads of four slabs -- sixteen values -- are requested before the first of them is added: the loop is a chain of
    // memory round trips, not of additions)
    int z = 0;
    for (; z + 4 <= r.nz; z += 4) {
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[u][q] = r.slab[(long)(z + u) * r.stride + c4[q]];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    }
    for (; z < r.nz; ++z) {
      const float* sl = r.slab + (long)z * r.stride;
      s += (sl[c4[0]] + sl[c4[1]]) + (sl[c4[2]] + sl[c4[3]]);
    }
    if (r.sub == 2) s *= 0.25f;
  } else {
    int z = 0;
    for (; z + 16 <= r.nz; z += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = r.slab[(long)(z + u) * r.stride + si];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
#pragma unroll 4
    for (; z < r.nz; ++z) s += r.slab[(long)z * r.stride + si];
    if (i >= r.count && r.bias_parts == 2) {
#pragma unroll 4
      for (z = 0; z < r.nz; ++z) s += r.slab[(long)z * r.stride + si + r.nbias];
    }
  }
  float* o;
  if (i >= r.count) o = r.bias_out + (i - r.count);
  else if (r.orient == 0) o = r.out + i;
  else {
    const int k = (int)(i / r.Cb), n = (int)(i - (long)k * r.Cb);
    const int t = k / r.Cs, c = k - t * r.Cs;
    o = r.out + ((long)t * r.Cb + n) * r.Cs + c;
  }
  if (r.accumulate) s += *o;
  *o = s;
}

// Dynamic loss scaling (fp16 activations).  ls: DEVICE float[4] = {scale, good steps since the last change, non-finite flag, skipped steps}.
// With ls the optimiser kernels (a) return without touching anything when the flag is set -- the step is skipped, (b) divide the
// gradient by the CURRENT scale on top of grad_scale, (c) take t = *tdev + 1 (tdev counts the APPLIED updates: a skipped step does
// not advance the bias correction).
struct AdamDyn { const float* ls; const float* tdev; };

__global__ void adam_tf_kernel(size_t count, float* w, const float* g, float* m, float* v, const float* hyper, float lr_v, float t_v,
                               float beta1, float beta2, float eps, float clip, float grad_scale, AdamDyn dyn) {
  if (dyn.ls) {
    if (dyn.ls[2] != 0.f) return;
    grad_scale /= dyn.ls[0];
    t_v = dyn.tdev[0] + 1.f;
  }
  // the arithmetic form of TF's ApplyAdam kernel: alpha = lr*sqrt(1-b2^t)/(1-b1^t);
  // m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); w -= (m*alpha)/(sqrt(v)+eps), all in fp32
  // {lr, t}: device memory (a captured launch replayed with new values) or kernel arguments (hyper == nullptr)
  const float lr = hyper ? hyper[0] : lr_v, t = hyper ? hyper[1] : t_v;
  const float alpha = lr * sqrtf(1.f - powf(beta2, t)) / (1.f - powf(beta1, t));
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * grad_scale;
    float mi = m[i] + (gi - m[i]) * omb1;
    float vi = v[i] + (gi * gi - v[i]) * omb2;
    float wi = w[i] - (mi * alpha) / (sqrtf(vi) + eps);
    if (clip > 0.f) wi = fminf(fmaxf(wi, -clip), clip);
    m[i] = mi; v[i] = vi; w[i] = wi;
  }
}

// the same update on four consecutive parameters per thread (16-byte accesses, one memory round trip per thread for slabs of up to
// 8M parameters): ranges whose start and length are multiples of four floats
__global__ void adam_tf_vec4_kernel(size_t count4, float4* w, const float4* g, float4* m, float4* v, const float* hyper, float lr_v, float t_v,
                                    float beta1, float beta2, float eps, float clip, float grad_scale, AdamDyn dyn) {
  if (dyn.ls) {
    if (dyn.ls[2] != 0.f) return;
    grad_scale /= dyn.ls[0];
    t_v = dyn.tdev[0] + 1.f;
  }
  const float lr = hyper ? hyper[0] : lr_v, t = hyper ? hyper[1] : t_v;
  const float alpha = lr * sqrtf(1.f - powf(beta2, t)) / (1.f - powf(beta1, t));
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 g4 = g[i], m4 = m[i], v4 = v[i], w4 = w[i];
    const float gs[4] = {g4.x, g4.y, g4.z, g4.w}, ms[4] = {m4.x, m4.y, m4.z, m4.w}, vs[4] = {v4.x, v4.y, v4.z, v4.w}, ws[4] = {w4.x, w4.y, w4.z, w4.w};
    float mo[4], vo[4], wo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {      // (the operation sequence of adam_tf_kernel, element by element)
      const float gi = gs[q] * grad_scale;
      mo[q] = ms[q] + (gi - ms[q]) * omb1;
      vo[q] = vs[q] + (gi * gi - vs[q]) * omb2;
      wo[q] = ws[q] - (mo[q] * alpha) / (sqrtf(vo[q]) + eps);
      if (clip > 0.f) wo[q] = fminf(fmaxf(wo[q], -clip), clip);
    }
    m[i] = make_float4(mo[0], mo[1], mo[2], mo[3]);
    v[i] = make_float4(vo[0], vo[1], vo[2], vo[3]);
    w[i] = make_float4(wo[0], wo[1], wo[2], wo[3]);
  }
}

__global__ void set2_kernel(float* p, float a, float b) { p[0] = a; p[1] = b; }

// any non-finite value in g[0, count) raises the flag ls[2] (a benign race: every writer stores the same 1.0)
__global__ void grad_finite_check_kernel(size_t count, const float* g, float* ls) {
  bool bad = false;
  const size_t n4 = count / 4, stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = tid; i < n4; i += stride) {
    const float4 q = ((const float4*)g)[i];
    // x - x is 0 for finite x, NaN for +-inf and NaN
    const float z = (q.x - q.x) + (q.y - q.y) + (q.z - q.z) + (q.w - q.w);
    bad |= !(z == 0.f);
  }
  for (size_t i = n4 * 4 + tid; i < count; i += stride) bad |= !((g[i] - g[i]) == 0.f);
  if (bad) ls[2] = 1.f;
}

// one thread: skipped step -> halve the scale; applied step -> advance the groups' update counters, double the scale after
// `interval` applied steps in a row
__global__ void loss_scale_update_kernel(float* ls, float* t0, float* t1, float interval, float min_scale, float max_scale) {
  if (ls[2] != 0.f) {
    ls[0] = fmaxf(ls[0] * 0.5f, min_scale);
    ls[1] = 0.f; ls[2] = 0.f; ls[3] += 1.f;
  } else {
    if (t0) t0[0] += 1.f;
    if (t1) t1[0] += 1.f;
    ls[1] += 1.f;
    if (ls[1] >= interval) { if (ls[0] < max_scale) ls[0] = fminf(ls[0] * 2.f, max_scale); ls[1] = 0.f; }
  }
}

static void adam_launch(rcgan_ctx* ctx, size_t count, float* w, const float* g, float* m, float* v, const float* hyper, float lr, float t,
                        float beta1, float beta2, float eps, float clip, float grad_scale, AdamDyn dyn = {nullptr, nullptr}) {
  const bool vec = count % 4 == 0 && (((size_t)w | (size_t)g | (size_t)m | (size_t)v) & 15) == 0;
  if (vec) {
    size_t blocks = (count / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(adam_tf_vec4_kernel, dim3((int)blocks), dim3(256), 0, ctx->stream, count / 4, (float4*)w, (const float4*)g, (float4*)m,
                       (float4*)v, hyper, lr, t, beta1, beta2, eps, clip, grad_scale, dyn);
  } else {
    size_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_tf_kernel, dim3((int)blocks), dim3(256), 0, ctx->stream, count, w, g, m, v, hyper, lr, t, beta1, beta2, eps, clip,
                       grad_scale, dyn);
  }
}

static int g_use_tr = -1;   // -1 unknown, 0 no, 1 yes (decided by the self test)

static int ensure_selftest(rcgan_ctx* ctx) {
  if (g_use_tr >= 0) return RCGAN_OK;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(ctx->stream, &st);
  if (st != hipStreamCaptureStatusNone) RC_FAIL(ctx, RCGAN_EINVALID_ARG, "first MFMA wgrad call must happen outside graph capture");
  int res[2] = {0, 0};
  int rc = mfma_selftest(ctx, res);
  if (rc) return rc;
  if (res[0] != 0) RC_FAIL(ctx, RCGAN_EHIP, "MFMA 16x16x32 bf16 fragment layout probe failed (%d mismatches)", res[0]);
  g_use_tr = (res[1] == 0) ? 1 : 0;
  const char* e = getenv("RCGAN_NO_TR");
  if (e && e[0] == '1') g_use_tr = 0;
  return RCGAN_OK;
}

extern "C" {

const char* rcgan_version(void) { return RCGAN_HALF_FP16 ? "rcgan_hip 0.1 (gfx950, fp16 build)" : "rcgan_hip 0.1 (gfx950, bf16 build)"; }

int rcgan_create(rcgan_ctx** out, int device, void* stream) {
  if (!out) return RCGAN_EINVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return RCGAN_EHIP;
  if (hipSetDevice(device) != hipSuccess) return RCGAN_EHIP;
  rcgan_ctx* c = new rcgan_ctx();
  c->device = device;
  c->stream = (hipStream_t)stream;
  c->capturing = false;
  c->devtmp = nullptr;
  c->devtmp_bytes = 0;
  c->prof_which = 0;
  c->prof_flops = 0.0; c->prof_flops_exec = 0.0;
  c->main_stream = c->stream; c->side_stream = nullptr; c->fork_ev = nullptr; c->join_ev = nullptr; c->on_side = false;
  if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return RCGAN_EHIP;
  }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    c->num_cus = cus;
  }
  c->dbg_stamps = nullptr;
  c->head_stage = 0;
  c->comm = nullptr; c->comm_world = 1; c->comm_rank = 0; c->comm_stub = false; c->comm_stream = nullptr;
  c->comm_fork = nullptr; c->comm_join = nullptr; c->comm_pending = false;
  c->stub_bus_gbps = 0.0; c->stub_latency_us = 0.0; c->wall_clock_khz = 0;
  c->narrow_ws = nullptr; c->narrow_ws_bytes = 0; c->splitr_ws = nullptr; c->splitr_ws_bytes = 0;
  c->gscale_host = 1.f;
  c->gscale_dev = nullptr;
  c->zero_page = nullptr;
  if (hipMalloc(&c->zero_page, RC_ZERO_PAGE_BYTES) != hipSuccess || hipMemset(c->zero_page, 0, RC_ZERO_PAGE_BYTES) != hipSuccess) {
    delete c;
    return RCGAN_EHIP;
  }
  for (int i = 0; i < 64; ++i) c->event_made[i] = false;
  *out = c;
  return RCGAN_OK;
}

int rcgan_destroy(rcgan_ctx* ctx) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  (void)hipStreamSynchronize(ctx->stream);
  // graphs first: captured ncclAllReduce nodes hold references to the communicator, which RCCL wants released before ncclCommDestroy
  for (auto& g : ctx->graphs)
    if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
  (void)rcgan_comm_destroy(ctx);
  for (int i = 0; i < 64; ++i)
    if (ctx->event_made[i]) (void)hipEventDestroy(ctx->events[i]);
  if (ctx->zero_page) (void)hipFree(ctx->zero_page);
  if (ctx->narrow_ws) (void)hipFree(ctx->narrow_ws);
  if (ctx->splitr_ws) (void)hipFree(ctx->splitr_ws);
  for (void* p : ctx->retired_ws) (void)hipFree(p);
  if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
  if (ctx->join_ev) (void)hipEventDestroy(ctx->join_ev);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  delete ctx;
  return RCGAN_OK;
}

const char* rcgan_last_error(rcgan_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
int rcgan_half_dtype(void) { return RCGAN_H16; }

// CRC-32C, slice-by-8 tables built on first use (host only)
unsigned rcgan_crc32c(unsigned crc, const void* data, size_t n) {
  static uint32_t tab[8][256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) tab[t][i] = (tab[t - 1][i] >> 8) ^ tab[0][tab[t - 1][i] & 0xff];
    ready = true;
  }
  const unsigned char* p = (const unsigned char*)data;
  uint32_t c = ~crc;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = tab[7][lo & 0xff] ^ tab[6][(lo >> 8) & 0xff] ^ tab[5][(lo >> 16) & 0xff] ^ tab[4][lo >> 24] ^
        tab[3][hi & 0xff] ^ tab[2][(hi >> 8) & 0xff] ^ tab[1][(hi >> 16) & 0xff] ^ tab[0][hi >> 24];
    p += 8; n -= 8;
  }
  while (n--) c = (c >> 8) ^ tab[0][(c ^ *p++) & 0xff];
  return ~c;
}

int rcgan_set_stream(rcgan_ctx* ctx, void* stream) {
  RC_REQUIRE(ctx, !ctx->on_side, "set_stream inside a side section");
  ctx->stream = ctx->main_stream = (hipStream_t)stream;
  return RCGAN_OK;
}

// Fork: launches issued until rcgan_side_end go to the side stream, ordered after everything issued so far on the main
// stream.  Join: the main stream waits for the side section.  Between end and join the two streams run concurrently.
int rcgan_side_begin(rcgan_ctx* ctx) {
  RC_REQUIRE(ctx, !ctx->on_side, "side sections do not nest");
  RC_HIP(ctx, hipEventRecord(ctx->fork_ev, ctx->main_stream));
  RC_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->fork_ev, 0));
  ctx->stream = ctx->side_stream;
  ctx->on_side = true;
  return RCGAN_OK;
}

int rcgan_side_end(rcgan_ctx* ctx) {
  RC_REQUIRE(ctx, ctx->on_side, "no side section open");
  RC_HIP(ctx, hipEventRecord(ctx->join_ev, ctx->side_stream));
  ctx->stream = ctx->main_stream;
  ctx->on_side = false;
  return RCGAN_OK;
}

int rcgan_side_join(rcgan_ctx* ctx) {
  RC_REQUIRE(ctx, !ctx->on_side, "close the side section first");
  RC_HIP(ctx, hipStreamWaitEvent(ctx->main_stream, ctx->join_ev, 0));
  return RCGAN_OK;
}

int rcgan_stream_sync(rcgan_ctx* ctx) {
  RC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RCGAN_OK;
}

int rcgan_event_record(rcgan_ctx* ctx, int slot) {
  RC_REQUIRE(ctx, slot >= 0 && slot < 64, "slot %d", slot);
  if (!ctx->event_made[slot]) {
    RC_HIP(ctx, hipEventCreate(&ctx->events[slot]));
    ctx->event_made[slot] = true;
  }
  RC_HIP(ctx, hipEventRecord(ctx->events[slot], ctx->stream));
  return RCGAN_OK;
}

int rcgan_event_elapsed_ms(rcgan_ctx* ctx, int s0, int s1, float* ms) {
  RC_REQUIRE(ctx, s0 >= 0 && s0 < 64 && s1 >= 0 && s1 < 64 && ctx->event_made[s0] && ctx->event_made[s1], "bad slots");
  RC_HIP(ctx, hipEventSynchronize(ctx->events[s1]));
  RC_HIP(ctx, hipEventElapsedTime(ms, ctx->events[s0], ctx->events[s1]));
  return RCGAN_OK;
}

int rcgan_graph_begin(rcgan_ctx* ctx) {
  RC_REQUIRE(ctx, !ctx->capturing, "already capturing");
  RC_REQUIRE(ctx, ctx->stream != nullptr, "graph capture needs a non-null stream");
  RC_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  ctx->capturing = true;
  return RCGAN_OK;
}

int rcgan_reserve_scratch(rcgan_ctx* ctx, size_t bytes) {
  RC_REQUIRE(ctx, !ctx->capturing, "scratch cannot grow inside a capture");
  RC_HIP(ctx, ctx_grow_scratch(ctx, &ctx->splitr_ws, &ctx->splitr_ws_bytes, bytes));
  RC_HIP(ctx, ctx_grow_scratch(ctx, &ctx->narrow_ws, &ctx->narrow_ws_bytes, bytes));
  return RCGAN_OK;
}

int rcgan_scratch_bytes(rcgan_ctx* ctx, size_t* split_reduction_bytes, size_t* narrow_bytes) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  if (split_reduction_bytes) *split_reduction_bytes = ctx->splitr_ws_bytes;
  if (narrow_bytes) *narrow_bytes = ctx->narrow_ws_bytes;
  return RCGAN_OK;
}

int rcgan_graph_end(rcgan_ctx* ctx, int* graph_id) {
  RC_REQUIRE(ctx, ctx->capturing, "not capturing");
  hipGraph_t g = nullptr;
  ctx->capturing = false;
  RC_HIP(ctx, hipStreamEndCapture(ctx->stream, &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) RC_FAIL(ctx, RCGAN_EHIP, "hipGraphInstantiate -> %s", hipGetErrorString(e));
  ctx->graphs.push_back(ge);
  *graph_id = (int)ctx->graphs.size() - 1;
  return RCGAN_OK;
}

int rcgan_graph_abort(rcgan_ctx* ctx) {
  // Leave a capture that cannot be completed (a launch inside it failed): what was recorded is dropped, nothing was executed.
  if (!ctx) return RCGAN_EINVALID_ARG;
  if (!ctx->capturing) return RCGAN_OK;
  ctx->capturing = false;
  ctx->head_stage = 0;           // (deferred launches recorded into the dropped capture never ran; their buffers belong to the aborted step)
  ctx->comm_pending = false;     // (an asynchronous bucket's join event was recorded inside the dropped capture)
  hipGraph_t g = nullptr;
  (void)hipStreamEndCapture(ctx->stream, &g);
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  return RCGAN_OK;
}

int rcgan_graph_launch(rcgan_ctx* ctx, int id) {
  RC_REQUIRE(ctx, id >= 0 && id < (int)ctx->graphs.size() && ctx->graphs[id], "graph id %d", id);
  RC_HIP(ctx, hipGraphLaunch(ctx->graphs[id], ctx->stream));
  return RCGAN_OK;
}

int rcgan_graph_destroy(rcgan_ctx* ctx, int id) {
  RC_REQUIRE(ctx, id >= 0 && id < (int)ctx->graphs.size() && ctx->graphs[id], "graph id %d", id);
  (void)hipGraphExecDestroy(ctx->graphs[id]);
  ctx->graphs[id] = nullptr;
  return RCGAN_OK;
}

int rcgan_debug_stamps(rcgan_ctx* ctx, void* stamps) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  ctx->dbg_stamps = stamps;
  return RCGAN_OK;
}

int rcgan_prof_begin(rcgan_ctx* ctx, int which) {
  for (auto e : ctx->prof_ev) (void)hipEventDestroy(e);
  ctx->prof_ev.clear();
  ctx->prof_flops = 0.0;
  ctx->prof_flops_exec = 0.0;
  ctx->prof_bn_in = 0;
  ctx->prof_which = which;
  return RCGAN_OK;
}

int rcgan_prof_end(rcgan_ctx* ctx, int* launches, double* total_ms, double* total_flops) {
  ctx->prof_which = 0;
  RC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double ms = 0.0;
  size_t n = ctx->prof_ev.size() / 2;
  for (size_t i = 0; i < n; ++i) {
    float t = 0.f;
    RC_HIP(ctx, hipEventElapsedTime(&t, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
    ms += t;
  }
  for (auto e : ctx->prof_ev) (void)hipEventDestroy(e);
  ctx->prof_ev.clear();
  if (launches) *launches = (int)n;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = ctx->prof_flops;
  return RCGAN_OK;
}

int rcgan_prof_executed_flops(rcgan_ctx* ctx, double* executed_flops) {
  RC_REQUIRE(ctx, executed_flops != nullptr, "null argument");
  *executed_flops = ctx->prof_flops_exec;
  return RCGAN_OK;
}

int rcgan_prof_bn_in_launches(rcgan_ctx* ctx, int* launches) {
  if (!ctx || !launches) return RCGAN_EINVALID_ARG;
  *launches = ctx->prof_bn_in;
  return RCGAN_OK;
}

int rcgan_selftest(rcgan_ctx* ctx) {
  g_use_tr = -1;
  return ensure_selftest(ctx);
}

int rcgan_query(rcgan_ctx* ctx, int what) {
  (void)ctx;
  if (what == RCGAN_QUERY_TR_READ) return g_use_tr;
  return RCGAN_EINVALID_ARG;
}

// ------------------------------------------------------------------------------------------------
// convolution
// ------------------------------------------------------------------------------------------------
static int check_desc(rcgan_ctx* ctx, const rcgan_conv_desc* d) {
  RC_REQUIRE(ctx, d != nullptr, "null desc");
  RC_REQUIRE(ctx, d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, "bad dims");
  RC_REQUIRE(ctx, d->kh > 0 && d->kw > 0 && d->stride > 0, "bad kernel/stride");
  RC_REQUIRE(ctx, d->dtype == RCGAN_F32 || d->dtype == RCGAN_H16, "bad dtype");
  if (d->flags & RCGAN_CONV_IN_UPSAMPLE2X) RC_REQUIRE(ctx, d->h % 2 == 0 && d->w % 2 == 0, "upsampled size must be even");
  if (d->flags & RCGAN_CONV_OUT_MEANPOOL2)
    RC_REQUIRE(ctx, !(d->flags & RCGAN_CONV_IN_UPSAMPLE2X) && d->h % 2 == 0 && d->w % 2 == 0, "fused mean pool: even size, no upsampled input");
  return RCGAN_OK;
}

int rcgan_conv_fused_pool_ok(const rcgan_conv_desc* d) { return d && mfma_pool_ok(d) ? 1 : 0; }

size_t rcgan_conv_prepared_bytes(const rcgan_conv_desc* d) {
  size_t elems = (size_t)d->kh * d->kw * d->cin * d->cout;
  if (mfma_eligible(d)) return 2 * elems * sizeof(bf16_t) + (mfma_phase_filters(d) ? 32 * (size_t)d->cin * d->cout * sizeof(bf16_t) : 0) + 256;
  if (img_side(d)) return img_extra_offset(d) + img_extra_bytes(d);
  return elems * sizeof(float) + 256;
}

int rcgan_conv_prepare(rcgan_ctx* ctx, const rcgan_conv_desc* d, const float* w, const float* sigma, void* prepared) {
  int rc = check_desc(ctx, d);
  if (rc) return rc;
  const int T = d->kh * d->kw;
  size_t elems = (size_t)T * d->cin * d->cout;
  if (mfma_eligible(d)) {
    bf16_t* wt = (bf16_t*)prepared;
    rc = mfma_prepare_launch(ctx, w, sigma, wt, wt + elems, T, d->cin, d->cout);
    if (rc || !mfma_phase_filters(d)) return rc;
    bf16_t* wph = wt + 2 * elems;
    const int kind = (d->flags & RCGAN_CONV_OUT_MEANPOOL2) ? 1 : 0;
    return conv_prepare_phase_launch(ctx, 1, &w, &sigma, &wph, &d->cin, &d->cout, &kind);
  }
  rc = direct_prepare_launch(ctx, w, sigma, (float*)prepared, (long)elems);
  if (rc) return rc;
  return img_prepare_launch(ctx, d, w, sigma, prepared);
}

int rcgan_conv_prepare_batch(rcgan_ctx* ctx, const rcgan_prepare_item* items, int n_items) {
  RC_REQUIRE(ctx, items != nullptr && n_items >= 0, "bad items");
  for (int i = 0; i < n_items; ++i) {
    int rc = check_desc(ctx, &items[i].desc);
    if (rc) return rc;
  }
  return conv_prepare_batch_launch(ctx, items, n_items);
}

int rcgan_conv_prepare_batch_embed(rcgan_ctx* ctx, const rcgan_prepare_item* items, int n_items, const rcgan_embed_desc* e) {
  return rcgan_conv_prepare_batch_riders(ctx, items, n_items, e, nullptr);
}

int rcgan_conv_prepare_batch_riders(rcgan_ctx* ctx, const rcgan_prepare_item* items, int n_items, const rcgan_embed_desc* e,
                                    const rcgan_step_inputs_desc* si) {
  return rcgan_conv_prepare_batch_frags(ctx, items, n_items, e, si, nullptr, 0);
}

int rcgan_conv_prepare_batch_frags(rcgan_ctx* ctx, const rcgan_prepare_item* items, int n_items, const rcgan_embed_desc* e,
                                   const rcgan_step_inputs_desc* si, const rcgan_frag_item* frags, int n_frags) {
  if (e == nullptr && si == nullptr && n_frags == 0) return rcgan_conv_prepare_batch(ctx, items, n_items);
  RC_REQUIRE(ctx, items != nullptr && n_items >= 1, "the riders need at least one filter to ride with");
  RC_REQUIRE(ctx, n_frags >= 0 && n_frags <= 12 && (n_frags == 0 || frags != nullptr), "%d fragment copies (at most 12)", n_frags);
  for (int f = 0; f < n_frags; ++f) {
    RC_REQUIRE(ctx, frags[f].item >= 0 && frags[f].item < n_items && frags[f].fwd && frags[f].bwd, "fragment copy %d: bad item / null destination", f);
    const rcgan_conv_desc& d = items[frags[f].item].desc;
    RC_REQUIRE(ctx, mfma_eligible(&d) && d.kh == 3 && d.kw == 3 && frags[f].ctn >= 1 && frags[f].ss >= 1 && d.cout % (16 * frags[f].ctn) == 0 &&
                        d.cin % (16 * frags[f].ctn) == 0 && (9 * d.cin / 32) % frags[f].ss == 0 && (9 * d.cout / 32) % frags[f].ss == 0,
               "fragment copy %d: not a 16-bit 3x3 filter this layout divides (%d -> %d, %d tiles, %d steps)", f, d.cin, d.cout, frags[f].ctn, frags[f].ss);
  }
  for (int i = 0; i < n_items; ++i) {
    int rc = check_desc(ctx, &items[i].desc);
    if (rc) return rc;
  }
  SmallGemmArgs ge;
  if (e) {
    RC_REQUIRE(ctx, e->v >= 1 && e->v <= HEAD_MAX_V && e->e_dim >= 1 && e->d >= 1 && e->table && e->w_e && e->E, "bad embed desc");
    // E[l][j] = (sum_k table[l][k] W_e[k][j]) / sigma_e + b_e[j]   (the first small-left GEMM of rcgan_proj_head_fwd_bwd)
    ge = {e->v, e->e_dim, e->d, e->table, e->e_dim, 1, e->w_e, e->sigma_e, e->b_e, e->E, nullptr, 0};
  }
  StepInputsArgs sa;
  if (si) {
    RC_REQUIRE(ctx, si->n >= 1 && si->images && si->x && si->rng_state, "bad step-inputs desc");
    RC_REQUIRE(ctx, si->dtype == RCGAN_F32 || si->dtype == RCGAN_H16, "bad dtype");
    RC_REQUIRE(ctx, si->fill == nullptr || (si->fill_count % 4 == 0 && ((size_t)si->fill & 15) == 0), "fill range must be whole float4");
    sa.n = si->n; sa.is16 = si->dtype == RCGAN_H16 ? 1 : 0; sa.img = si->images; sa.x = si->x; sa.pooled = si->pooled;
    sa.lo = si->noise_lo; sa.hi = si->noise_hi; sa.seed = si->seed; sa.state = (uint64_t*)si->rng_state;
    sa.counter = ctx->counters() + RC_COUNTER_INPUTS;
    sa.fill = si->fill; sa.fill4 = si->fill ? si->fill_count / 4 : 0;
    sa.fakes = si->fakes; sa.slice = (unsigned*)si->fake_slice; sa.nslices = si->n_slices;
    RC_REQUIRE(ctx, si->fakes == nullptr || (si->fake_slice != nullptr && si->n_slices >= 1 && ((size_t)si->fakes & 15) == 0 && ((size_t)si->x & 15) == 0),
               "fake slices need a device counter and 16-byte aligned tensors");
  }
  return conv_prepare_batch_launch(ctx, items, n_items, e ? &ge : nullptr, si ? &sa : nullptr, frags, n_frags);
}

size_t rcgan_conv_workspace_bytes(const rcgan_conv_desc* d) {
  int oh, ow, p;
  same_pad(d->h, d->kh, d->stride, &oh, &p);
  same_pad(d->w, d->kw, d->stride, &ow, &p);
  long M = (long)d->n * oh * ow;
  size_t ws = direct_wgrad_ws_bytes(d);
  if (small_wgrad_kind(d)) {
    size_t s = small_wgrad_ws_bytes(d);
    if (s > ws) ws = s;
  }
  if (img_side(d)) {
    size_t s = img_wgrad_ws_bytes(d);
    if (s > ws) ws = s;
  }
  if (mfma_wgrad_eligible(d)) {
    size_t s = (size_t)mfma_wgrad_splits(d, M) * ((size_t)d->kh * d->kw * d->cin + 1) * d->cout * sizeof(float) + (size_t)(cdiv(M, 2048) + 1024) * d->cout * sizeof(float) + 256;
    if (s > ws) ws = s;
  }
  // upsample-folded data gradient: full-resolution dx scratch
  if (d->flags & RCGAN_CONV_IN_UPSAMPLE2X) {
    size_t s = (size_t)d->n * d->h * d->w * d->cin * dtype_size(d->dtype) + 256;
    if (s > ws) ws = s;
  }
  return ws;
}

static void fill_mfma_args(const rcgan_conv_desc* d, MfmaConvArgs& a) {
  int oh, ow, pt, pl;
  same_pad(d->h, d->kh, 1, &oh, &pt);
  same_pad(d->w, d->kw, 1, &ow, &pl);
  a.N = d->n; a.H = d->h; a.W = d->w; a.KH = d->kh; a.KW = d->kw; a.PT = pt; a.PL = pl;
  a.M = (long)d->n * d->h * d->w;
  a.lw = ilog2_exact(d->w); a.lh = ilog2_exact(d->h);
  if (a.lw < 0 || a.lh < 0) { a.lw = -1; a.lh = -1; }
  a.stamps = nullptr;
  a.wph = nullptr; a.phase = 0;
  a.resid_up = 0;
  a.stats = nullptr;
}

// RCGAN_CONV_RESID_UPSAMPLE2X: the matrix-core epilogue reads the half-resolution residual in place (power-of-two output grid)
int rcgan_conv_resid_up_ok(const rcgan_conv_desc* d) {
  return d && mfma_eligible(d) && !(d->flags & RCGAN_CONV_OUT_MEANPOOL2) && d->stride == 1 && ilog2_exact(d->w) >= 1 && ilog2_exact(d->h) >= 1 ? 1 : 0;
}

int rcgan_conv2d_fwd(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias, void* y) {
  return rcgan_conv2d_fwd_residual(ctx, d, x, prepared, bias, nullptr, y);
}

// ---- batch-norm statistics out of the producing convolution's epilogue (conv_mfma8.hip) -------------------------------------
// the forward launch rcgan_conv2d_fwd_residual would issue for this descriptor on the matrix-core path
static void stats_conv_args(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias,
                            const void* residual, void* y, MfmaConvArgs& a) {
  fill_mfma_args(d, a);
  a.in = (const bf16_t*)x; a.wt = (const bf16_t*)prepared; a.bias = bias; a.mask = nullptr; a.out = (bf16_t*)y;
  if (mfma_phase_filters(d) && (d->flags & RCGAN_CONV_IN_UPSAMPLE2X)) a.wph = (const bf16_t*)prepared + 2 * (size_t)d->kh * d->kw * d->cin * d->cout;
  a.resid = (const bf16_t*)residual;
  a.resid_up = (residual != nullptr && (d->flags & RCGAN_CONV_RESID_UPSAMPLE2X)) ? 1 : 0;
  a.zero = (const bf16_t*)(ctx ? ctx->zero_page : (void*)16);
  a.Cin = d->cin; a.Cout = d->cout;
  a.up = (d->flags & RCGAN_CONV_IN_UPSAMPLE2X) ? 1 : 0;
  a.relu_in = 0; a.accumulate = 0;
}

int rcgan_conv_stats_ok(const rcgan_conv_desc* d) {
  if (!d || !mfma_eligible(d) || d->stride != 1 || d->cout != 256) return 0;
  if (d->flags & (RCGAN_CONV_OUT_MEANPOOL2 | RCGAN_CONV_ACCUMULATE | RCGAN_CONV_IN_RELU | RCGAN_CONV_FORCE_DIRECT)) return 0;
  if (((long)d->n * d->h * d->w) % 256 != 0 || (long)d->n * d->h * d->w * (d->cin > d->cout ? d->cin : d->cout) >= (1L << 31)) return 0;
  if ((d->flags & RCGAN_CONV_RESID_UPSAMPLE2X) && !rcgan_conv_resid_up_ok(d)) return 0;
  MfmaConvArgs a;
  int dummy = 0;
  stats_conv_args(nullptr, d, nullptr, &dummy, nullptr, nullptr, nullptr, a);
  // (an upsample-3x3 layer must take its sub-pixel form or not be upsampled at all: the finisher knows these two tile orders)
  if (a.up && !(mfma_phase_filters(d) && mfma_conv8_phase_form(a))) return 0;
  return mfma_conv_is_p8(a) ? 1 : 0;
}

size_t rcgan_conv_stats_bytes(const rcgan_conv_desc* d) {
  return d ? (size_t)((long)d->n * d->h * d->w / 256) * d->cout * 2 * sizeof(float) : 0;
}

int rcgan_conv2d_fwd_stats(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias,
                           const void* residual, void* y, float* tile_sums) {
  int rc = check_desc(ctx, d);
  if (rc) return rc;
  RC_REQUIRE(ctx, tile_sums != nullptr, "null tile_sums");
  RC_REQUIRE(ctx, rcgan_conv_stats_ok(d), "this convolution does not produce tile statistics (rcgan_conv_stats_ok)");
  MfmaConvArgs a;
  stats_conv_args(ctx, d, x, prepared, bias, residual, y, a);
  a.stats = tile_sums;
  return mfma_conv_launch(ctx, a);
}

int rcgan_bn_stats_from_tiles(rcgan_ctx* ctx, const rcgan_conv_desc* d, int nseg, float eps, const float* tile_sums, float* mean, float* rstd) {
  int rc = check_desc(ctx, d);
  if (rc) return rc;
  RC_REQUIRE(ctx, rcgan_conv_stats_ok(d) && tile_sums && mean && rstd, "bad arguments");
  RC_REQUIRE(ctx, nseg >= 1 && d->n % nseg == 0, "%d samples in %d segments", d->n, nseg);
  const long px = (long)d->h * d->w;                  // output pixels per sample
  const double count = (double)(d->n / nseg) * (double)px;
  if (d->flags & RCGAN_CONV_IN_UPSAMPLE2X) {
    // sub-pixel form: tile index = (phase, sample, low-resolution pixels / 256)
    const long tps = px / 4 / 256;                    // tiles per sample and phase
    RC_REQUIRE(ctx, tps >= 1 && (px / 4) % 256 == 0, "sub-pixel tiles must hold whole 256-pixel runs of one sample");
    return bn_tile_stats_finish_launch(ctx, tile_sums, d->cout, nseg, (int)((d->n / nseg) * tps), 4, (long)d->n * tps, count, eps, mean, rstd);
  }
  RC_REQUIRE(ctx, px % 256 == 0 || 256 % px == 0, "tiles must not straddle segments");
  RC_REQUIRE(ctx, ((long)(d->n / nseg) * px) % 256 == 0, "a segment must be whole tiles");
  return bn_tile_stats_finish_launch(ctx, tile_sums, d->cout, nseg, (int)((long)(d->n / nseg) * px / 256), 1, 0, count, eps, mean, rstd);
}

// the MfmaConvArgs of an ordinary forward launch (no pool fold), as rcgan_conv2d_fwd_residual builds them
static void mfma_fwd_args(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias, const void* residual,
                          void* y, MfmaConvArgs& a) {
  fill_mfma_args(d, a);
  a.in = (const bf16_t*)x; a.wt = (const bf16_t*)prepared; a.bias = bias; a.mask = nullptr; a.out = (bf16_t*)y;
  if (mfma_phase_filters(d) && (d->flags & RCGAN_CONV_IN_UPSAMPLE2X)) a.wph = (const bf16_t*)prepared + 2 * (size_t)d->kh * d->kw * d->cin * d->cout;
  a.resid = (const bf16_t*)residual;
  a.resid_up = (residual != nullptr && (d->flags & RCGAN_CONV_RESID_UPSAMPLE2X)) ? 1 : 0;
  a.zero = ctx ? (const bf16_t*)ctx->zero_page : (const bf16_t*)prepared;      // (routing queries without a context: any non-null value)
  a.Cin = d->cin; a.Cout = d->cout;
  a.up = (d->flags & RCGAN_CONV_IN_UPSAMPLE2X) ? 1 : 0;
  a.relu_in = (d->flags & RCGAN_CONV_IN_RELU) ? 1 : 0;
  a.accumulate = (d->flags & RCGAN_CONV_ACCUMULATE) ? 1 : 0;
}

// the convolutions that can apply the batch norm in front of them to their staged input: the small-output image-end layers (G.Output)
// and (round 5) whatever the routing hands to a halo-patch kernel -- plain 3x3 / upsample-3x3 layers on 16- / 32-wide (low-resolution)
// images with enough tiles to fill the chip (G.Block.2.Conv2, G.Block.3.Conv1 / Conv2 at the bench batches)
static int mfma_bn_in_route(const rcgan_conv_desc* d) {
  if (!d || d->dtype != RCGAN_H16 || !mfma_eligible(d) || (d->flags & (RCGAN_CONV_OUT_MEANPOOL2 | RCGAN_CONV_IN_RELU | RCGAN_CONV_ACCUMULATE | RCGAN_CONV_FORCE_DIRECT)))
    return 0;
  if ((long)d->n * d->h * d->w * (d->cin > d->cout ? d->cin : d->cout) >= (1L << 31)) return 0;
  static const int on = [] { const char* e = getenv("RCGAN_BN_INTO_PATCH"); return e ? atoi(e) : 1; }();
  if (!on) return 0;
  MfmaConvArgs a;
  // (the summed sub-pixel filters exist whenever mfma_phase_filters(d): a non-null stand-in is enough for the routing question -- nothing
  // is read through it)
  static const bf16_t stand_in[8] = {0};
  mfma_fwd_args(nullptr, d, nullptr, stand_in, nullptr, nullptr, nullptr, a);
  return mfma_conv_bn_route(a);
}

int rcgan_conv_bn_in_ok(const rcgan_conv_desc* d) {
  return (d && d->dtype == RCGAN_H16 && (img_fwd_bn_ok(d) || mfma_bn_in_route(d))) ? 1 : 0;
}

int rcgan_conv2d_fwd_bn(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias, void* y,
                        int segments, const int32_t* labels, const float* gamma, const float* beta, const float* mean, const float* rstd,
                        int act) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  int rc = check_desc(ctx, d);
  if (rc) return rc;
  RC_REQUIRE(ctx, x && prepared && y, "null argument");
  return rcgan_conv2d_fwd_bn_residual(ctx, d, x, prepared, bias, nullptr, y, segments, labels, gamma, beta, mean, rstd, act);
}

int rcgan_conv2d_fwd_bn_residual(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias, const void* residual,
                                 void* y, int segments, const int32_t* labels, const float* gamma, const float* beta, const float* mean,
                                 const float* rstd, int act) {
  if (!ctx) return RCGAN_EINVALID_ARG;
  int rc = check_desc(ctx, d);
  if (rc) return rc;
  RC_REQUIRE(ctx, x && prepared && y && mean && rstd && gamma && beta, "null argument");
  RC_REQUIRE(ctx, segments >= 1 && d->n % segments == 0, "segments must divide the batch");
  if (!rcgan_conv_bn_in_ok(d)) RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "no kernel applies a batch norm to this convolution's staged input (rcgan_conv_bn_in_ok)");
  if (residual == nullptr && img_fwd_bn_ok(d)) return img_fwd_bn(ctx, d, x, prepared, bias, y, mean, rstd, gamma, beta, labels, segments, act);
  RC_REQUIRE(ctx, mfma_bn_in_route(d), "residual form needs a halo-patch kernel");
  if (residual != nullptr && (d->flags & RCGAN_CONV_RESID_UPSAMPLE2X))
    RC_REQUIRE(ctx, rcgan_conv_resid_up_ok(d), "half-resolution residual not available for this convolution (rcgan_conv_resid_up_ok)");
  rc = ensure_selftest(ctx);
  if (rc) return rc;
  MfmaConvArgs a;
  mfma_fwd_args(ctx, d, x, prepared, bias, residual, y, a);
  a.bn_mean = mean; a.bn_rstd = rstd; a.bn_gamma = gamma; a.bn_beta = beta; a.bn_labels = labels;
  a.bn_seg_samples = d->n / segments; a.bn_act = act;
  return mfma_conv_launch(ctx, a);
}

int rcgan_conv2d_fwd_residual(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias,
                              const void* residual, void* y) {
  int rc = check_desc(ctx, d);
  if (rc) return rc;
  if (residual != nullptr && (d->flags & RCGAN_CONV_RESID_UPSAMPLE2X))
    RC_REQUIRE(ctx, rcgan_conv_resid_up_ok(d), "half-resolution residual not available for this convolution (rcgan_conv_resid_up_ok)");
  if (residual != nullptr && !mfma_eligible(d)) {      // other kernels: plain forward, then y += residual
    RC_REQUIRE(ctx, residual != y, "residual must not alias the output");
    rc = rcgan_conv2d_fwd_residual(ctx, d, x, prepared, bias, nullptr, y);
    if (rc) return rc;
    int oh, ow, p;
    same_pad(d->h, d->kh, d->stride, &oh, &p);
    same_pad(d->w, d->kw, d->stride, &ow, &p);
    return rcgan_axpby(ctx, (size_t)d->n * oh * ow * d->cout, d->dtype, 1.f, residual, 1.f, y);
  }
  if (d->flags & RCGAN_CONV_OUT_MEANPOOL2) {
    // ConvMeanPool as one 4x4 stride-2 convolution over x (16 taps, summed filters x 1/4): y over the pooled grid
    RC_REQUIRE(ctx, mfma_pool_ok(d) && residual == nullptr, "fused mean pool not available for this convolution (rcgan_conv_fused_pool_ok)");
    if ((long)d->n * d->h * d->w * (d->cin > d->cout ? d->cin : d->cout) >= (1L << 31))
      RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "tensor exceeds the 32-bit element offsets of the MFMA kernels");
    MfmaConvArgs a;
    fill_mfma_args(d, a);
    const size_t elems = (size_t)d->kh * d->kw * d->cin * d->cout;
    a.in = (const bf16_t*)x; a.wt = (const bf16_t*)prepared; a.bias = bias; a.mask = nullptr; a.out = (bf16_t*)y; a.resid = nullptr;
    a.zero = (const bf16_t*)ctx->zero_page;
    a.Cin = d->cin; a.Cout = d->cout;
    a.up = 0;
    a.relu_in = (d->flags & RCGAN_CONV_IN_RELU) ? 1 : 0;
    a.accumulate = (d->flags & RCGAN_CONV_ACCUMULATE) ? 1 : 0;
    a.M = (long)d->n * (d->h / 2) * (d->w / 2);
    a.wph = (const bf16_t*)prepared + 2 * elems;        // gather layout [Cout][16 * Cin]
    a.phase = 2;
    return mfma_conv_launch(ctx, a);
  }
  if (mfma_eligible(d)) {
    if ((long)d->n * d->h * d->w * (d->cin > d->cout ? d->cin : d->cout) >= (1L << 31))
      RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "tensor exceeds the 32-bit element offsets of the MFMA kernels");
    MfmaConvArgs a;
    fill_mfma_args(d, a);
    a.in = (const bf16_t*)x; a.wt = (const bf16_t*)prepared; a.bias = bias; a.mask = nullptr; a.out = (bf16_t*)y;
    if (mfma_phase_filters(d) && (d->flags & RCGAN_CONV_IN_UPSAMPLE2X)) a.wph = (const bf16_t*)prepared + 2 * (size_t)d->kh * d->kw * d->cin * d->cout;
    a.resid = (const bf16_t*)residual;
    a.resid_up = (residual != nullptr && (d->flags & RCGAN_CONV_RESID_UPSAMPLE2X)) ? 1 : 0;
    a.zero = (const bf16_t*)ctx->zero_page;
    a.Cin = d->cin; a.Cout = d->cout;
    a.up = (d->flags & RCGAN_CONV_IN_UPSAMPLE2X) ? 1 : 0;
    a.relu_in = (d->flags & RCGAN_CONV_IN_RELU) ? 1 : 0;
    a.accumulate = (d->flags & RCGAN_CONV_ACCUMULATE) ? 1 : 0;
    return mfma_conv_launch(ctx, a);
  }
  if (img_side(d)) return img_fwd(ctx, d, x, prepared, bias, y);
  if (int kind = small_fwd_kind(d)) {
    RC_DISPATCH_DTYPE(ctx, d->dtype, return small_fwd<T>(ctx, d, kind, (const T*)x, (const float*)prepared, bias, (T*)y));
  }
  RC_DISPATCH_DTYPE(ctx, d->dtype, return direct_fwd<T>(ctx, d, (const T*)x, (const float*)prepared, nullptr, bias, (T*)y));
  return RCGAN_OK;
}

int rcgan_conv2d_bwd_data(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* dy, const void* prepared, const void* x, void* dx,
                          void* ws, size_t ws_bytes) {
  return rcgan_conv2d_bwd_data_residual(ctx, d, dy, prepared, x, nullptr, dx, ws, ws_bytes);
}

int rcgan_conv2d_bwd_data_residual(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* dy, const void* prepared, const void* x,
                                   const void* residual, void* dx, void* ws, size_t ws_bytes) {
  int rc = check_desc(ctx, d);
  if (rc) return rc;
  if (residual != nullptr) {
    RC_REQUIRE(ctx, residual != dx && !(d->flags & RCGAN_CONV_ACCUMULATE), "residual form is out of place and not accumulating");
    if (!mfma_eligible(d) || (d->flags & RCGAN_CONV_IN_UPSAMPLE2X)) {      // other kernels: plain data gradient, then dx += residual
      rc = rcgan_conv2d_bwd_data_residual(ctx, d, dy, prepared, x, nullptr, dx, ws, ws_bytes);
      if (rc) return rc;
      return rcgan_axpby(ctx, (size_t)d->n * d->h * d->w * d->cin, d->dtype, 1.f, residual, 1.f, dx);
    }
  }
  const bool up = d->flags & RCGAN_CONV_IN_UPSAMPLE2X;
  const bool relu = d->flags & RCGAN_CONV_IN_RELU;
  const int acc = (d->flags & RCGAN_CONV_ACCUMULATE) ? 1 : 0;
  RC_REQUIRE(ctx, !relu || x != nullptr, "IN_RELU needs x for the mask");
  if (d->flags & RCGAN_CONV_OUT_MEANPOOL2) {
    // dy lives on the pooled grid: dx (full resolution) in the sub-pixel form -- pixel (2i + ph, 2j + pw) gathers the 2x2 pooled
    // pixels around it with the transposed summed filters of its phase
    RC_REQUIRE(ctx, mfma_pool_ok(d) && residual == nullptr, "fused mean pool not available for this convolution (rcgan_conv_fused_pool_ok)");
    MfmaConvArgs a;
    fill_mfma_args(d, a);
    const size_t elems = (size_t)d->kh * d->kw * d->cin * d->cout;
    a.in = (const bf16_t*)dy; a.wt = (const bf16_t*)prepared; a.bias = nullptr; a.mask = relu ? (const bf16_t*)x : nullptr;
    a.resid = nullptr; a.out = (bf16_t*)dx;
    a.zero = (const bf16_t*)ctx->zero_page;
    a.Cin = d->cout; a.Cout = d->cin;
    a.up = 1;                                             // source grid = the pooled one (h/2 x w/2)
    a.relu_in = 0; a.accumulate = acc;
    a.wph = (const bf16_t*)prepared + 2 * elems + 16 * (size_t)d->cin * d->cout;        // phase layout [4][Cin][4 * Cout]
    a.phase = 1;
    return mfma_conv_launch(ctx, a);
  }
  void* target = dx;
  if (up) {
    size_t need = (size_t)d->n * d->h * d->w * d->cin * dtype_size(d->dtype);
    if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
    target = ws;
  }
  if (up && mfma_phase_dgrad_ok(d)) {
    // sub-pixel form: dx over the low-resolution grid straight from dy (16 taps at stride 2, transposed summed filters), the
    // ReLU mask and the accumulation in the epilogue -- no full-resolution scratch, no 2x2 sum pass
    MfmaConvArgs a;
    fill_mfma_args(d, a);
    const size_t elems = (size_t)d->kh * d->kw * d->cin * d->cout;
    a.in = (const bf16_t*)dy; a.wt = (const bf16_t*)prepared + elems; a.bias = nullptr; a.mask = relu ? (const bf16_t*)x : nullptr;
    a.resid = nullptr; a.out = (bf16_t*)dx;
    a.zero = (const bf16_t*)ctx->zero_page;
    a.Cin = d->cout; a.Cout = d->cin;
    a.up = 0; a.relu_in = 0; a.accumulate = acc;
    a.M = (long)d->n * (d->h / 2) * (d->w / 2);
    a.wph = (const bf16_t*)prepared + 2 * elems + 16 * (size_t)d->cin * d->cout;
    a.phase = 2;
    return mfma_conv_launch(ctx, a);
  }
  const void* mask = (relu && !up) ? x : nullptr;
  const int acc_now = up ? 0 : acc;
  if (mfma_eligible(d)) {
    MfmaConvArgs a;
    fill_mfma_args(d, a);
    size_t elems = (size_t)d->kh * d->kw * d->cin * d->cout;
    a.in = (const bf16_t*)dy; a.wt = (const bf16_t*)prepared + elems; a.bias = nullptr; a.mask = (const bf16_t*)mask;
    a.resid = (const bf16_t*)residual;
    a.out = (bf16_t*)target;
    a.zero = (const bf16_t*)ctx->zero_page;
    a.Cin = d->cout; a.Cout = d->cin;          // reduction over cout, output channels = cin
    a.PT = d->kh - 1 - a.PT; a.PL = d->kw - 1 - a.PL;
    a.up = 0; a.relu_in = 0; a.accumulate = acc_now;
    rc = mfma_conv_launch(ctx, a);
    if (rc) return rc;
  } else if (mask == nullptr && !up && img_side(d)) {
    rc = img_dgrad(ctx, d, dy, prepared, target, acc_now);
    if (rc) return rc;
  } else if (int kind = (mask == nullptr && !up) ? small_dgrad_kind(d) : 0) {
    RC_DISPATCH_DTYPE(ctx, d->dtype, rc = small_dgrad<T>(ctx, d, kind, (const T*)dy, (const float*)prepared, (T*)target, acc_now));
    if (rc) return rc;
  } else {
    RC_DISPATCH_DTYPE(ctx, d->dtype, rc = direct_dgrad<T>(ctx, d, (const T*)dy, (const float*)prepared, nullptr, nullptr, (const T*)mask, (T*)target, acc_now));
    if (rc) return rc;
  }
  if (up) {
    RC_DISPATCH_DTYPE(ctx, d->dtype, rc = sumpool2_masked_launch<T>(ctx, d->n, d->h, d->w, d->cin, (const T*)ws, relu ? (const T*)x : (const T*)nullptr, (T*)dx, acc));
    if (rc) return rc;
  }
  return RCGAN_OK;
}

// The matrix-core filter-gradient problem of a layer (everything but the slab).  Upsample-3x3 and ConvMeanPool layers the
// three-tap kernel takes are posed in their sub-pixel form (MfmaWgradArgs::sub): reduction over the low-resolution grid.
static void wgrad_args_from_desc(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* dy, bool want_bias, MfmaWgradArgs& a) {
  int oh, ow, pt, pl;
  same_pad(d->h, d->kh, 1, &oh, &pt);
  same_pad(d->w, d->kw, 1, &ow, &pl);
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.slab = nullptr;
  a.zero = (const bf16_t*)ctx->zero_page;
  a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.KH = d->kh; a.KW = d->kw; a.PT = pt; a.PL = pl;
  a.up = (d->flags & RCGAN_CONV_IN_UPSAMPLE2X) ? 1 : 0;
  a.relu_in = (d->flags & RCGAN_CONV_IN_RELU) ? 1 : 0;
  a.use_tr = g_use_tr;
  a.sub = mfma_wgrad_sub_kind(d, g_use_tr);
  if (a.sub == 1 || a.sub == 2) { a.H >>= 1; a.W >>= 1; a.up = 0; }
  a.cells = (a.sub == 1 || a.sub == 2) ? 16 : d->kh * d->kw;
  a.M = (long)d->n * a.H * a.W;
  a.lw = ilog2_exact(a.W); a.lh = ilog2_exact(a.H);
  if (a.lw < 0 || a.lh < 0) { a.lw = -1; a.lh = -1; }
  a.slab_stride = (long)a.cells * d->cin * d->cout + d->cout;
  a.want_bias = want_bias ? 1 : 0;
  a.m_chunk = 0;
}

static SlabReduceGroup::Item wgrad_reduce_item(const rcgan_conv_desc* d, const MfmaWgradArgs& a, float* dw, float* dbias, int nbias, int nz, int accumulate) {
  SlabReduceGroup::Item it = {};
  it.slab = a.slab; it.stride = a.slab_stride; it.out = dw; it.count = (long)d->kh * d->kw * d->cin * d->cout;
  it.bias_out = dbias; it.nbias = nbias; it.nz = nz; it.accumulate = accumulate;
  it.bias_off = (long)a.cells * d->cin * d->cout;
  it.sub = a.sub == 3 ? 0 : a.sub; it.cc = (long)d->cin * d->cout;
  it.bias_parts = a.slab_stride == it.bias_off + 2L * d->cout ? 2 : 1;       // (mfma_wgrad9_plan's upsample form)
  static const int vec4_on = [] { const char* e = getenv("RCGAN_SLAB_REDUCE_VEC4"); return e ? atoi(e) : 1; }();
  it.vec4 = vec4_on && it.count % 4 == 0 && it.cc % 4 == 0 && a.slab_stride % 4 == 0 && ((size_t)a.slab & 15) == 0 && ((size_t)dw & 15) == 0;
  return it;
}

int rcgan_conv_wgrad_pool_ok(const rcgan_conv_desc* d) {
  // (before the self test has run the transposing LDS read counts as available; rcgan_conv2d_bwd_weight re-checks)
  return d && (d->flags & RCGAN_CONV_OUT_MEANPOOL2) && mfma_wgrad_sub_kind(d, g_use_tr < 0 ? 1 : g_use_tr) == 2 ? 1 : 0;
}

int rcgan_conv2d_bwd_weight(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                            int accumulate, void* ws, size_t ws_bytes) {
  int rc = check_desc(ctx, d);
  if (rc) return rc;
  if (mfma_wgrad_eligible(d)) {
    rc = ensure_selftest(ctx);
    if (rc) return rc;
    MfmaWgradArgs a;
    wgrad_args_from_desc(ctx, d, x, dy, dbias != nullptr, a);
    if ((d->flags & RCGAN_CONV_OUT_MEANPOOL2) && a.sub != 2)
      RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "filter gradient from the pooled dy needs the sub-pixel three-tap kernel (rcgan_conv_wgrad_pool_ok)");
    a.slab = (float*)ws;
    int nz = a.sub ? mfma_wgrad_sub_splits(d, a.M) : mfma_wgrad_splits(d, a.M);
    long cnt = (long)d->kh * d->kw * d->cin * d->cout;
    size_t need = (size_t)nz * a.slab_stride * sizeof(float) + (size_t)(cdiv(a.M, 2048) + 1024) * d->cout * sizeof(float);
    if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
    bool bias_done = false;
    int nzz = mfma_wgrad_launch(ctx, a, nz, &bias_done, ws_bytes);
    if (nzz < 0) return nzz;
    const int nb = bias_done ? d->cout : 0;
    if (a.sub == 1 || a.sub == 2) {
      SlabReduceGroup g;
      for (int q = 0; q < REDUCE_GROUP_MAX; ++q) g.it[q] = wgrad_reduce_item(d, a, dw, dbias, nb, nzz, accumulate);
      hipLaunchKernelGGL(slab_reduce2_group_kernel, dim3(cdiv(cnt + nb, 256), 1), dim3(256), 0, ctx->stream, g);
    } else {
      hipLaunchKernelGGL(slab_reduce2_kernel, dim3(cdiv(cnt + nb, 256)), dim3(256), 0, ctx->stream, (const float*)a.slab, a.slab_stride, dw, cnt,
                         dbias, nb, nzz, accumulate);
    }
    RC_LAUNCH_CHECK(ctx);
    if (dbias && !bias_done) {
      float* part = (float*)((char*)ws + (size_t)nz * a.slab_stride * sizeof(float));
      rc = colsum_launch<bf16_t>(ctx, (const bf16_t*)dy, a.M, d->cout, dbias, accumulate, part);
      if (rc) return rc;
    }
    return RCGAN_OK;
  }
  if (img_side(d)) return img_wgrad(ctx, d, x, dy, dw, dbias, accumulate, ws, ws_bytes);
  if (int kind = small_wgrad_kind(d)) {
    RC_DISPATCH_DTYPE(ctx, d->dtype, return small_wgrad<T>(ctx, d, kind, (const T*)x, (const T*)dy, dw, dbias, accumulate, ws, ws_bytes));
  }
  RC_DISPATCH_DTYPE(ctx, d->dtype, return direct_wgrad<T>(ctx, d, (const T*)x, (const T*)dy, dw, dbias, accumulate, ws, ws_bytes));
  return RCGAN_OK;
}

// Filter gradients of several layers whose x / dy are all available (the end of a backward pass): the layers the
// three-tap matrix-core kernel takes run as ONE grouped launch per input-ReLU flavour + ONE grouped slab reduction; every
// other layer goes through rcgan_conv2d_bwd_weight as usual.
int rcgan_conv2d_bwd_weight_group(rcgan_ctx* ctx, int n, const rcgan_conv_desc* descs, const void* const* xs, const void* const* dys,
                                  float* const* dws, float* const* dbiases, int accumulate, void* ws, size_t ws_bytes) {
  RC_REQUIRE(ctx, n >= 0 && (n == 0 || (descs && xs && dys && dws && dbiases)), "null argument");
  static const int target_blocks = [] { const char* e = getenv("RCGAN_WGRAD_GROUP_BLOCKS"); return e ? atoi(e) : 448; }();      // (512 before the sub-pixel forms: 6.67 -> 6.64 ms with 384; round 4, the critic step's launch with its chunks' workgroups on one XCD and the generator step's big layers gone to the nine-tap kernel, same box: 384 5.65 ms, 448 5.59, 512 5.60, 576 5.75)
  std::vector<MfmaWgradArgs> cand(n);
  std::vector<char> takes(n, 0);
  // The nine-tap kernel (conv_wgrad9.hip) takes the group's 3x3 layers when they are big: its one workgroup per CU writes a slab of ALL nine
  // (sixteen) cells, and the riders of the three-tap launch (image-end layers, 1x1 shortcuts, the head) lose the workgroups they hid under.
  // Measured on the bench iteration: the critic step's layers (n = 128, 128 channels: 1152 pixels per workgroup) 120 -> 172 us with it, the
  // generator step's (256 channels: 12032 / 2560 pixels per workgroup) 478 -> 425 us.
  const long w9_minwork = [] { const char* e = getenv("RCGAN_WGRAD9_GROUP_MINWORK"); return e ? atol(e) : 1000000L; }();      // (per call: the tests force it; the critic step's group is 0.52 M, the generator step's 2.75 M, one 256-channel 32x32 layer 1.05 M)
  bool wgrad9_group_on = false;
  {
    double w9 = 0;
    for (int i = 0; i < n; ++i) {
      const rcgan_conv_desc* d = descs + i;
      if (check_desc(ctx, d) || !mfma_wgrad_eligible(d) || d->kh != 3) continue;
      w9 += (double)d->n * d->h * d->w * (d->cin / 64) * (d->cout / 128);       // (full-resolution pixels x channel tiles)
    }
    wgrad9_group_on = w9 >= (double)w9_minwork;
  }
  // pass 1: which layers the grouped kernel takes, and the pixels per workgroup that gives ~target_blocks workgroups in all
  double work = 0, work9[2] = {0, 0};        // (work9: the nine-tap kernel's plain / sub-pixel layers, one workgroup per CU each)
  for (int i = 0; i < n; ++i) {
    const rcgan_conv_desc* d = descs + i;
    int rc = check_desc(ctx, d);
    if (rc) return rc;
    if (!mfma_wgrad_eligible(d)) continue;
    rc = ensure_selftest(ctx);
    if (rc) return rc;
    MfmaWgradArgs& a = cand[i];
    wgrad_args_from_desc(ctx, d, xs[i], dys[i], dbiases[i] != nullptr, a);
    if ((d->flags & RCGAN_CONV_OUT_MEANPOOL2) && a.sub != 2)
      RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "filter gradient from the pooled dy needs the sub-pixel three-tap kernel (rcgan_conv_wgrad_pool_ok)");
    if (mfma_wgrad9_takes(a) && wgrad9_group_on) {
      takes[i] = a.sub ? 3 : 2;
      work9[a.sub ? 1 : 0] += (double)a.M * (a.Cin / 64) * (a.Cout / 128) * (a.sub ? 2 : 1);
    } else if (mfma_wgrad3_takes(a)) {
      takes[i] = 1;
      // workgroup-passes over a pixel: 3 filter rows of three taps, or 8 (parity, row shift) tiles of two taps
      work += (double)a.M * (a.sub == 3 ? 1.0 / 3.0 : (a.sub ? 8 * 2.0 / 3.0 : a.KH)) * (a.Cin / 64) * (a.Cout / 128);
    }
  }
  static const int px_max = [] { const char* e = getenv("RCGAN_WGRAD_GROUP_PXMAX"); return e ? atoi(e) : 6144; }();
  long px = (long)(work / target_blocks);
  px = (px + 63) / 64 * 64;
  if (px < 256) px = 256;
  if (px > px_max) px = px_max;      // big groups (the generator step): more workgroups rather than ever longer ones
  static const int target9 = [] { const char* e = getenv("RCGAN_WGRAD9_BLOCKS"); return e ? atoi(e) : 256; }();
  // one workgroup per CU and equal pixels per workgroup: the smallest chunk with which the layers' tiles x chunks fit ONE round of target9
  // workgroups (a 257th workgroup would double the launch's time); the plain and the sub-pixel layers are a launch each
  long px9[2] = {0, 0};
  for (int f = 0; f < 2; ++f) {
    px9[f] = ((long)(work9[f] / target9) + 127) / 128 * 128;
    if (px9[f] < 512) px9[f] = 512;
    for (int it = 0; it < 4096 && work9[f] > 0; ++it, px9[f] += 128) {
      long tot = 0;
      for (int i = 0; i < n; ++i)
        if (takes[i] == 2 + f) {
          MfmaWgradArgs b = cand[i];
          unsigned bgx = 0, bgy = 0;
          if (mfma_wgrad9_plan(b, 1 << 20, &bgx, &bgy, px9[f])) tot += (long)bgx * bgy;
        }
      if (tot <= target9) break;
    }
  }
  std::vector<MfmaWgradArgs> args9[4];           // nine-tap kernel: plain without / with input ReLU, sub-pixel forms without / with
  std::vector<unsigned> gxs9[4], gys9[4];
  // pass 2: slabs, grouped launches per input-ReLU flavour, everything else on its own
  std::vector<MfmaWgradArgs> args[3];            // three-tap kernel without / with input ReLU, per-tap kernel
  std::vector<MfmaWgradArgs> late;               // small 1x1 layers on the two-tap body: join one of the first two
  std::vector<unsigned> late_gx, late_gy;
  std::vector<unsigned> gxs[3], gys[3];
  std::vector<SlabReduceGroup::Item> red;
  size_t used = 0;
  // image-end layers ride in the first grouped launch (if there is one) and in the grouped reduction
  ImgWGroup img;
  img.n = 0;
  for (int q = 0; q <= IMG_GROUP_MAX; ++q) img.first[q] = 0;
  static const int img_group = [] { const char* e = getenv("RCGAN_WGRAD_GROUP_IMG"); return e ? atoi(e) : 1; }();
  bool any_group = false;
  for (int i = 0; i < n && img_group; ++i) any_group = any_group || takes[i] == 1;
  for (int i = 0; i < n; ++i) {
    const rcgan_conv_desc* d = descs + i;
    bool grouped = false;
    if (any_group && img.n < IMG_GROUP_MAX && img_side(d) && (d->cin <= 3 ? d->cout : d->cin) == 128) {
      ImgWArgs ia;
      int cb = 0, nwg = 0;
      static const int img_wgs = [] { const char* e = getenv("RCGAN_WGRAD_IMG_WGS"); return e ? atoi(e) : 128; }();       // (with 384 three-tap workgroups: 6.64 -> 6.62 ms with 96; round 5, the three-tap workgroups a quarter faster: 96 5.36 ms, 128 5.34, 160 5.40, 256 5.36)
      int rc = img_wgrad_plan(ctx, d, xs[i], dys[i], img_wgs, &ia, &cb, &nwg);
      if (rc) return rc;
      const long per = 32L * cb + 32;
      const size_t need = ((size_t)nwg * per * sizeof(float) + 255) / 256 * 256;
      if (used + need <= ws_bytes / 2) {
        ia.slab = (float*)((char*)ws + used);
        used += need;
        const int q = img.n++;
        img.a[q] = ia;
        for (int r = q + 1; r <= IMG_GROUP_MAX; ++r) img.first[r] = img.first[q] + (unsigned)nwg;
        const int side = img_side(d), cs = side == 1 ? d->cin : d->cout, T = d->kh * d->kw;
        SlabReduceGroup::Item it = {};
        it.slab = ia.slab; it.stride = per; it.out = dws[i]; it.count = (long)T * cs * cb; it.nz = nwg; it.accumulate = accumulate;
        it.Cb = cb; it.Cs = cs;
        if (side == 1) {           // small = conv input: slab order is dW order; bias gradient = the ones row (31)
          it.orient = 0; it.bias_out = dbiases[i]; it.nbias = dbiases[i] ? cb : 0; it.bias_off = 31L * cb;
        } else {                   // small = dy: transposed; bias gradient = column sums of the centre tap
          it.orient = 1; it.bias_out = dbiases[i]; it.nbias = dbiases[i] ? cs : 0; it.bias_off = 32L * cb + (T == 9 ? 4 : 0) * cs;
        }
        red.push_back(it);
        grouped = true;
      }
    }
    if (!grouped && mfma_wgrad_eligible(d)) {
      MfmaWgradArgs a = cand[i];
      const int nz = a.sub ? mfma_wgrad_sub_splits(d, a.M) : mfma_wgrad_splits(d, a.M);
      unsigned gx = 0, gy = 0;
      // (the grouped slabs are fitted into the workspace below: no a-priori bound on the nine-tap kernel's pixel chunks)
      const bool nine = takes[i] >= 2 && mfma_wgrad9_plan(a, 1 << 20, &gx, &gy, px9[takes[i] - 2]);
      const bool three = !nine && takes[i] && mfma_wgrad3_plan(a, nz, &gx, &gy, px);
      if (a.sub && !three && !nine) RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "sub-pixel filter gradient needs the three-tap kernel");
      if (nine || three || mfma_wgrad_tap_plan(a, nz, &gx, &gy)) {
        const size_t need = ((size_t)gy * a.slab_stride * sizeof(float) + 255) / 256 * 256;
        if (used + need <= ws_bytes / 2) {
          a.slab = (float*)((char*)ws + used);
          used += need;
          const int f = three ? (a.relu_in ? 1 : 0) : 2;
          if (nine) { const int f9 = (a.sub ? 2 : 0) + (a.relu_in ? 1 : 0); args9[f9].push_back(a); gxs9[f9].push_back(gx); gys9[f9].push_back(gy); }
          else if (three && a.sub == 3 && !a.relu_in) { late.push_back(a); late_gx.push_back(gx); late_gy.push_back(gy); }   // placed below
          else { args[f].push_back(a); gxs[f].push_back(gx); gys[f].push_back(gy); }
          red.push_back(wgrad_reduce_item(d, a, dws[i], dbiases[i], dbiases[i] ? d->cout : 0, (int)gy, accumulate));
          grouped = true;
        }
      }
    }
    if (!grouped) {          // its own launches, with the workspace half the grouped slabs do not use
      int rc = rcgan_conv2d_bwd_weight(ctx, d, xs[i], dys[i], dws[i], dbiases[i], accumulate, (char*)ws + ws_bytes / 2, ws_bytes - ws_bytes / 2);
      if (rc) return rc;
    }
  }
  // a riding 1x1 without input ReLU runs on either flavour of the three-tap kernel (the ReLU is a per-problem switch in the two-tap
  // body): it joins the launch with the most workgroups
  if (!late.empty()) {
    unsigned tot[2] = {0, 0};
    for (int f = 0; f < 2; ++f)
      for (size_t q = 0; q < args[f].size(); ++q) tot[f] += gxs[f][q] * gys[f][q];
    const int f = tot[1] > tot[0] ? 1 : 0;
    for (size_t q = 0; q < late.size(); ++q) { args[f].push_back(late[q]); gxs[f].push_back(late_gx[q]); gys[f].push_back(late_gy[q]); }
  }
  // the image-end workgroups go with the launch that has the most workgroups to hide under
  int img_f = -1;
  if (img.n) {
    unsigned best = 0;
    for (int f = 0; f < 3; ++f) {
      unsigned tot = 0;
      for (size_t q = 0; q < args[f].size() && q < WGRAD_GROUP_MAX_HOST; ++q) tot += gxs[f][q] * gys[f][q];
      if (!args[f].empty() && tot >= best) { best = tot; img_f = f; }
    }
    RC_REQUIRE(ctx, img_f >= 0, "image-end filter gradients planned without a grouped launch");
  }
  // the projection head's deferred parameter sums (head_rider.h) ride in the three-tap launch with the most workgroups
  int head_f = -1;
  if (ctx->head_stage == 2) {
    if (img_f == 0 || img_f == 1) head_f = img_f;
    else {
      unsigned best = 0;
      for (int f = 0; f < 2; ++f) {
        unsigned tot = 0;
        for (size_t q = 0; q < args[f].size() && q < WGRAD_GROUP_MAX_HOST; ++q) tot += gxs[f][q] * gys[f][q];
        if (!args[f].empty() && tot >= best) { best = tot; head_f = f; }
      }
    }
  }
  for (int f = 0; f < 4; ++f)
    if (!args9[f].empty()) {
      int rc = mfma_wgrad9_group_launch(ctx, (int)args9[f].size(), args9[f].data(), gxs9[f].data(), gys9[f].data());
      if (rc) return rc;
    }
  for (int f = 0; f < 3; ++f)
    if (!args[f].empty()) {
      int rc = mfma_wgrad3_group_launch(ctx, (int)args[f].size(), args[f].data(), gxs[f].data(), gys[f].data(), f == 2 ? 1 : 0,
                                        f == img_f ? &img : nullptr, f == head_f);
      if (rc) return rc;
    }
  for (size_t i0 = 0; i0 < red.size(); i0 += REDUCE_GROUP_MAX) {
    SlabReduceGroup g;
    const int m = (int)((red.size() - i0 < REDUCE_GROUP_MAX) ? red.size() - i0 : REDUCE_GROUP_MAX);
    long maxc = 0;
    for (int q = 0; q < m; ++q) {
      g.it[q] = red[i0 + q];
      const long threads = (g.it[q].vec4 ? g.it[q].count / 4 : g.it[q].count) + g.it[q].nbias;
      if (threads > maxc) maxc = threads;
    }
    for (int q = m; q < REDUCE_GROUP_MAX; ++q) g.it[q] = red[i0];
    hipLaunchKernelGGL(slab_reduce2_group_kernel, dim3(cdiv(maxc, 256), m), dim3(256), 0, ctx->stream, g);
    RC_LAUNCH_CHECK(ctx);
  }
  return RCGAN_OK;
}

// transposed conv: d describes the forward conv whose data-gradient it is (see header)
int rcgan_deconv2d_fwd(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const float* w, const float* bias, void* y) {
  int rc = check_desc(ctx, d);
  if (rc) return rc;
  RC_DISPATCH_DTYPE(ctx, d->dtype, return direct_dgrad<T>(ctx, d, (const T*)x, w, nullptr, bias, (const T*)nullptr, (T*)y, 0));
  return RCGAN_OK;
}

int rcgan_deconv2d_bwd_data(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* dy, const float* w, void* dx) {
  return rcgan_deconv2d_bwd_data_cols(ctx, d, dy, w, dx, 0);
}

int rcgan_deconv2d_bwd_data_cols(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* dy, const float* w, void* dx, int n_cols) {
  int rc = check_desc(ctx, d);
  if (rc) return rc;
  RC_REQUIRE(ctx, n_cols >= 0 && n_cols <= d->cout, "n_cols %d of %d input channels", n_cols, d->cout);
  RC_DISPATCH_DTYPE(ctx, d->dtype, return direct_fwd<T>(ctx, d, (const T*)dy, w, nullptr, nullptr, (T*)dx, n_cols));
  return RCGAN_OK;
}

int rcgan_deconv2d_bwd_weight(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                              int accumulate, void* ws, size_t ws_bytes) {
  int rc = check_desc(ctx, d);
  if (rc) return rc;
  // forward-conv input = deconv output gradient dy; forward-conv output gradient = deconv input x
  RC_DISPATCH_DTYPE(ctx, d->dtype, rc = direct_wgrad<T>(ctx, d, (const T*)dy, (const T*)x, dw, nullptr, accumulate, ws, ws_bytes));
  if (rc) return rc;
  if (dbias) {
    int oh, ow, p;
    same_pad(d->h, d->kh, d->stride, &oh, &p);
    same_pad(d->w, d->kw, d->stride, &ow, &p);
    (void)oh; (void)ow;
    long rows = (long)d->n * d->h * d->w;
    // partial column sums go to the tail of the workspace (the filter-gradient slabs sit at its head and are consumed)
    size_t need = ((size_t)(cdiv(rows, 2048) + 1024) * d->cin * sizeof(float) + 255) / 256 * 256;
    float* part = (ws != nullptr && ws_bytes >= direct_wgrad_ws_bytes(d) + need) ? (float*)((char*)ws + (ws_bytes - need) / 256 * 256) : nullptr;
    RC_DISPATCH_DTYPE(ctx, d->dtype, rc = colsum_launch<T>(ctx, (const T*)dy, rows, d->cin, dbias, accumulate, part));
    if (rc) return rc;
  }
  return RCGAN_OK;
}

size_t rcgan_deconv2d_bwd_weight_concat_bytes(const rcgan_conv_desc* d, int n_cols) {
  if (!d || n_cols <= 0 || n_cols >= d->cout) return 0;
  const long rows = (long)d->n * d->h * d->w;
  return direct_wgrad_cols_ws_bytes(d, n_cols) + ((size_t)(cdiv(rows, 2048) + 1024) * d->cin * sizeof(float) + 255) / 256 * 256 + 256;
}

int rcgan_deconv2d_bwd_weight_concat(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* dy, int n_cols, const float* yb,
                                     float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes) {
  int rc = check_desc(ctx, d);
  if (rc) return rc;
  RC_REQUIRE(ctx, n_cols > 0 && n_cols < d->cout && yb != nullptr, "%d real columns of %d, yb %p", n_cols, d->cout, (const void*)yb);
  const size_t need = rcgan_deconv2d_bwd_weight_concat_bytes(d, n_cols);
  if (ws == nullptr || ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
  // forward-conv input = deconv output gradient dy; forward-conv output gradient = deconv input x = [t ; yb]
  RC_DISPATCH_DTYPE(ctx, d->dtype, rc = direct_wgrad_cols<T>(ctx, d, (const T*)dy, (const T*)x, n_cols, yb, dw, accumulate, ws, ws_bytes));
  if (rc) return rc;
  if (dbias) {
    const long rows = (long)d->n * d->h * d->w;
    const size_t tail = ((size_t)(cdiv(rows, 2048) + 1024) * d->cin * sizeof(float) + 255) / 256 * 256;
    float* part = (float*)((char*)ws + (ws_bytes - tail) / 256 * 256);
    RC_DISPATCH_DTYPE(ctx, d->dtype, rc = colsum_launch<T>(ctx, (const T*)dy, rows, d->cin, dbias, accumulate, part));
    if (rc) return rc;
  }
  return RCGAN_OK;
}

// ------------------------------------------------------------------------------------------------
// dense layers: plain row-major GEMMs on the shared fp32 GEMM core (conv_direct.hip)
// ------------------------------------------------------------------------------------------------
static int check_lin(rcgan_ctx* ctx, int m, int k, int n, int dtype) {
  RC_REQUIRE(ctx, m > 0 && k > 0 && n > 0, "bad linear shape [%d,%d]x[%d,%d]", m, k, k, n);
  RC_REQUIRE(ctx, dtype == RCGAN_F32 || dtype == RCGAN_H16, "bad dtype %d", dtype);
  return RCGAN_OK;
}

size_t rcgan_linear_workspace_bytes(int m, int k, int n) { return linear_wgrad_ws_bytes(m, k, n); }

int rcgan_linear_fwd(rcgan_ctx* ctx, int m, int k, int n, int dtype, const void* x, const float* w, const float* sigma,
                     const float* bias, void* y) {
  int rc = check_lin(ctx, m, k, n, dtype);
  if (rc) return rc;
  RC_DISPATCH_DTYPE(ctx, dtype, return linear_fwd<T>(ctx, m, k, n, (const T*)x, w, sigma, bias, (T*)y));
  return RCGAN_OK;
}

int rcgan_linear_bwd_data(rcgan_ctx* ctx, int m, int k, int n, int dtype, const void* dy, const float* w, const float* sigma,
                          void* dx, int accumulate) {
  int rc = check_lin(ctx, m, k, n, dtype);
  if (rc) return rc;
  RC_DISPATCH_DTYPE(ctx, dtype, return linear_dgrad<T>(ctx, m, k, n, (const T*)dy, w, sigma, (T*)dx, accumulate));
  return RCGAN_OK;
}

int rcgan_linear_bwd_weight(rcgan_ctx* ctx, int m, int k, int n, int dtype, const void* x, const void* dy, float* dw, float* dbias,
                            int accumulate, void* ws, size_t ws_bytes) {
  int rc = check_lin(ctx, m, k, n, dtype);
  if (rc) return rc;
  RC_DISPATCH_DTYPE(ctx, dtype, return linear_wgrad<T>(ctx, m, k, n, (const T*)x, (const T*)dy, dw, dbias, accumulate, ws, ws_bytes));
  return RCGAN_OK;
}

// ------------------------------------------------------------------------------------------------
// optimiser
// ------------------------------------------------------------------------------------------------
int rcgan_adam_tf(rcgan_ctx* ctx, size_t count, float* w, const float* g, float* m, float* v, const float* hyper, float beta1,
                  float beta2, float eps, float clip, float grad_scale) {
  if (count == 0) return RCGAN_OK;
  RC_REQUIRE(ctx, hyper != nullptr, "null hyper (rcgan_adam_tf_host takes {lr, t} by value)");
  adam_launch(ctx, count, w, g, m, v, hyper, 0.f, 0.f, beta1, beta2, eps, clip, grad_scale);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_adam_tf_host(rcgan_ctx* ctx, size_t count, float* w, const float* g, float* m, float* v, float lr, float t, float beta1,
                       float beta2, float eps, float clip, float grad_scale) {
  if (count == 0) return RCGAN_OK;
  adam_launch(ctx, count, w, g, m, v, nullptr, lr, t, beta1, beta2, eps, clip, grad_scale);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_set2_f32(rcgan_ctx* ctx, float* p, float a, float b) {
  RC_REQUIRE(ctx, p != nullptr, "null pointer");
  hipLaunchKernelGGL(set2_kernel, dim3(1), dim3(1), 0, ctx->stream, p, a, b);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_set_grad_scale(rcgan_ctx* ctx, float host_scale, const float* dev_scale) {
  RC_REQUIRE(ctx, host_scale > 0.f, "grad scale %g", (double)host_scale);
  ctx->gscale_host = host_scale;
  ctx->gscale_dev = dev_scale;
  return RCGAN_OK;
}

int rcgan_grad_finite_check(rcgan_ctx* ctx, size_t count, const float* g, float* ls_state) {
  RC_REQUIRE(ctx, g != nullptr && ls_state != nullptr && ((size_t)g & 15) == 0, "null / misaligned argument");
  if (count == 0) return RCGAN_OK;
  size_t blocks = (count / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(grad_finite_check_kernel, dim3((int)blocks), dim3(256), 0, ctx->stream, count, g, ls_state);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_adam_tf_dyn(rcgan_ctx* ctx, size_t count, float* w, const float* g, float* m, float* v, float lr, const float* t_dev, float beta1,
                      float beta2, float eps, float clip, float grad_scale, const float* ls_state) {
  RC_REQUIRE(ctx, t_dev != nullptr && ls_state != nullptr, "null t_dev / ls_state");
  if (count == 0) return RCGAN_OK;
  adam_launch(ctx, count, w, g, m, v, nullptr, lr, 0.f, beta1, beta2, eps, clip, grad_scale, AdamDyn{ls_state, t_dev});
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_loss_scale_update(rcgan_ctx* ctx, float* ls_state, float* t_dev0, float* t_dev1, float growth_interval, float min_scale,
                            float max_scale) {
  RC_REQUIRE(ctx, ls_state != nullptr && growth_interval >= 1.f && min_scale > 0.f && max_scale >= min_scale, "bad loss-scale arguments");
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(1), 0, ctx->stream, ls_state, t_dev0, t_dev1, growth_interval, min_scale, max_scale);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

}  // extern "C"
