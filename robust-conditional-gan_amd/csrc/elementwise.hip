// HBM-bound elementwise / resampling kernels (grid-stride, 4 elements per thread).
#include "common.h"
#include "rng.h"

#define EW_BLOCK 256
static inline int ew_grid(size_t count, int per_thread = 1) {
  size_t b = (count + (size_t)EW_BLOCK * per_thread - 1) / ((size_t)EW_BLOCK * per_thread);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

template <typename T>
__global__ void act_fwd_kernel(size_t count, int act, const T* x, T* y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    Elem<T>::st(y + i, act_apply(act, Elem<T>::ld(x + i)));
}

template <typename T>
__global__ void act_bwd_kernel(size_t count, int act, const T* s, const T* dy, T* dx, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    float v = Elem<T>::ld(dy + i) * act_grad(act, Elem<T>::ld(s + i));
    if (accumulate) v += Elem<T>::ld(dx + i);
    Elem<T>::st(dx + i, v);
  }
}

template <typename T>
__global__ void axpby_kernel(size_t count, float alpha, const T* a, float beta, const T* b, T* y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    float v = alpha * Elem<T>::ld(a + i);
    if (beta != 0.f) v += beta * Elem<T>::ld(b + i);
    Elem<T>::st(y + i, v);
  }
}

template <typename S, typename D>
__global__ void cast_kernel(size_t count, const S* s, D* d) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    Elem<D>::st(d + i, Elem<S>::ld(s + i));
}

__global__ void fill_kernel(size_t count, float* p, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}


// ---- 8-channel vectorised resampling kernels (c % 8 == 0, < 2^32 chunks): one 16-B (bf16) access per window pixel,
//      32-bit index arithmetic ----------------------------------------------------------------------------------
__device__ __forceinline__ void rs_ld8(const float* p, float* v) {
  float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void rs_ld8(const bf16_t* p, float* v) {
  uint4 a = *(const uint4*)p;
  uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[2 * j] = bf16_to_f32((bf16_t)(w[j] & 0xffff)); v[2 * j + 1] = bf16_to_f32((bf16_t)(w[j] >> 16)); }
}
__device__ __forceinline__ void rs_st8(float* p, const float* v) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
  *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void rs_st8(bf16_t* p, const float* v) {
  uint4 pk;
  pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  pk.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
  pk.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
  *(uint4*)p = pk;
}

// thread = 8 channels of one LOW-resolution pixel.  MODE 0: low = scale * sum(2x2 window of hi) [, masked by xmask > 0]
//                                                  MODE 1: the 2x2 window of hi (=|+=) scale * low
template <typename T, int MODE>
__global__ void resample2_vec_kernel(unsigned nchunks, int oh, int ow, int c, float scale, const T* src, const T* xmask, T* dst,
                                     int accumulate) {
  const unsigned cpr = (unsigned)c >> 3;
  const size_t rowpitch = (size_t)2 * ow * c;          // one hi-res image row
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += gridDim.x * blockDim.x) {
    const unsigned ch = (i % cpr) << 3;
    unsigned p = i / cpr;
    const unsigned x2 = p % (unsigned)ow;
    p /= (unsigned)ow;
    const unsigned y2 = p % (unsigned)oh, b = p / (unsigned)oh;
    const size_t lo = (size_t)i << 3;
    const size_t hi = (((size_t)b * 2 * oh + 2 * y2) * 2 * ow + 2 * x2) * c + ch;
    if (MODE == 0) {
      float a0[8], a1[8], a2[8], a3[8], o[8];
      rs_ld8(src + hi, a0); rs_ld8(src + hi + rowpitch, a1); rs_ld8(src + hi + c, a2); rs_ld8(src + hi + rowpitch + c, a3);
      // add_n order of gan_resnet.py:239-240: [::2,::2] + [1::2,::2] + [::2,1::2] + [1::2,1::2]
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (((a0[j] + a1[j]) + a2[j]) + a3[j]) * scale;
      if (xmask) {
        float m[8];
        rs_ld8(xmask + lo, m);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (!(m[j] > 0.f)) o[j] = 0.f;
      }
      if (accumulate) {
        float d[8];
        rs_ld8(dst + lo, d);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += d[j];
      }
      rs_st8(dst + lo, o);
    } else {
      float v[8];
      rs_ld8(src + lo, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= scale;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t off = hi + (q & 1) * (size_t)c + (q >> 1) * rowpitch;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = v[j];
        if (accumulate) {
          float d[8];
          rs_ld8(dst + off, d);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += d[j];
        }
        rs_st8(dst + off, o);
      }
    }
  }
}

static inline bool resample_vec_ok(int n, int oh, int ow, int c) {
  return c % 8 == 0 && (size_t)n * oh * ow * (c / 8) < ((size_t)1 << 31);
}
static inline int resample_grid(size_t nchunks) {
  size_t b = (nchunks + 255) / 256;
  if (b > 16384) b = 16384;
  return b < 1 ? 1 : (int)b;
}

// y[n][h/2][w/2][c] = mean of the 2x2 window
template <typename T>
__global__ void meanpool2_fwd_kernel(int n, int h, int w, int c, const T* x, T* y) {
  const int oh = h >> 1, ow = w >> 1;
  size_t total = (size_t)n * oh * ow * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int ch = (int)(i % c);
    size_t p = i / c;
    int x2 = (int)(p % ow);
    size_t q = p / ow;
    int y2 = (int)(q % oh);
    int b = (int)(q / oh);
    const T* s = x + (((size_t)b * h + 2 * y2) * w + 2 * x2) * c + ch;
    // add_n order of gan_resnet.py:239-240: [::2,::2] + [1::2,::2] + [::2,1::2] + [1::2,1::2]
    float v = Elem<T>::ld(s) + Elem<T>::ld(s + (size_t)w * c) + Elem<T>::ld(s + c) + Elem<T>::ld(s + (size_t)w * c + c);
    Elem<T>::st(y + i, v * 0.25f);
  }
}

// dx[n][h][w][c] (=|+=) scale * dy[n][h/2][w/2][c]    (adjoint of meanpool with scale .25; nearest upsample with scale 1)
template <typename T>
__global__ void expand2_kernel(int n, int h, int w, int c, float scale, const T* dy, T* dx, int accumulate) {
  size_t total = (size_t)n * h * w * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int ch = (int)(i % c);
    size_t p = i / c;
    int xx = (int)(p % w);
    size_t q = p / w;
    int yy = (int)(q % h);
    int b = (int)(q / h);
    float v = scale * Elem<T>::ld(dy + (((size_t)b * (h >> 1) + (yy >> 1)) * (w >> 1) + (xx >> 1)) * c + ch);
    if (accumulate) v += Elem<T>::ld(dx + i);
    Elem<T>::st(dx + i, v);
  }
}

// dx[n][h/2][w/2][c] (=|+=) sum of the 2x2 window of dy[n][h][w][c], optionally masked by xmask>0 (fused ReLU backward)
template <typename T>
__global__ void sumpool2_kernel(int n, int h, int w, int c, const T* dy, const T* xmask, T* dx, int accumulate) {
  const int oh = h >> 1, ow = w >> 1;
  size_t total = (size_t)n * oh * ow * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int ch = (int)(i % c);
    size_t p = i / c;
    int x2 = (int)(p % ow);
    size_t q = p / ow;
    int y2 = (int)(q % oh);
    int b = (int)(q / oh);
    const T* s = dy + (((size_t)b * h + 2 * y2) * w + 2 * x2) * c + ch;
    float v = Elem<T>::ld(s) + Elem<T>::ld(s + (size_t)w * c) + Elem<T>::ld(s + c) + Elem<T>::ld(s + (size_t)w * c + c);
    if (xmask && !(Elem<T>::ld(xmask + i) > 0.f)) v = 0.f;
    if (accumulate) v += Elem<T>::ld(dx + i);
    Elem<T>::st(dx + i, v);
  }
}

template <typename T>
__global__ void concat_channels_fwd_kernel(int n, int hw, int c1, int c2, const T* x, const float* yb, T* y) {
  const int c = c1 + c2;
  size_t total = (size_t)n * hw * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int ch = (int)(i % c);
    size_t p = i / c;
    int b = (int)(p / hw);
    float v = ch < c1 ? Elem<T>::ld(x + p * c1 + ch) : yb[(size_t)b * c2 + (ch - c1)];
    Elem<T>::st(y + i, v);
  }
}

template <typename T>
__global__ void concat_channels_bwd_kernel(int n, int hw, int c1, int c2, const T* dy, T* dx) {
  const int c = c1 + c2;
  size_t total = (size_t)n * hw * c1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int ch = (int)(i % c1);
    size_t p = i / c1;
    Elem<T>::st(dx + i, Elem<T>::ld(dy + p * c + ch));
  }
}

// The batch repeated `reps` times back to back, y[r][i] = x[i] -- one discriminator pass over every label instead of the reference's
// ten discriminator() calls on the same images (mnist/model.py:152-163,187-197) -- and its adjoint dx[i] (+)= sum_r dy[r][i]
// (fp32 sum, rounded once).
template <typename T>
__global__ void tile_rows_fwd_kernel(size_t count, int reps, const T* x, T* y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const T v = x[i];
    for (int r = 0; r < reps; ++r) y[(size_t)r * count + i] = v;
  }
}

template <typename T>
__global__ void tile_rows_bwd_kernel(size_t count, int reps, const T* dy, T* dx, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    float s = accumulate ? Elem<T>::ld(dx + i) : 0.f;
    for (int r = 0; r < reps; ++r) s += Elem<T>::ld(dy + (size_t)r * count + i);
    Elem<T>::st(dx + i, s);
  }
}

// y[c][r] (+)= x[r][c], fp32: the [labels][samples] logits of that pass as the [samples][labels] matrix the weighted loss terms take
__global__ void transpose_f32_kernel(int rows, int cols, const float* x, float* y, int accumulate) {
  const size_t total = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    const size_t o = (size_t)c * rows + r;
    y[o] = accumulate ? y[o] + x[i] : x[i];
  }
}

// dst[i] = src[i], 4-byte words (16-byte pieces where both sides are 16-byte aligned): a step's packed input batch into the step's
// static input slab (the feed_dict of one session.run, gan_resnet.py:931,938) as an ordinary kernel on the step's stream -- the
// runtime's own device-to-device copy sits between two graph launches as a different kind of packet and costs a longer turnaround
__global__ void copy_words_kernel(size_t count, const uint32_t* src, uint32_t* dst) {
  const size_t n4 = ((((size_t)src | (size_t)dst) & 15) == 0) ? count / 4 : 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    ((uint4*)dst)[i] = ((const uint4*)src)[i];
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// gan_resnet.py:548-551
template <typename T>
__global__ void preprocess_cifar_kernel(int n, const int32_t* img, const float* noise, T* y) {
  size_t total = (size_t)n * 3072;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int ch = (int)(i % 3);
    size_t p = i / 3;
    int px = (int)(p % 1024);
    int b = (int)(p / 1024);
    size_t src = (size_t)b * 3072 + (size_t)ch * 1024 + px;
    float v = 2.f * ((float)img[src] / 256.f - .5f);
    v += noise[src];
    Elem<T>::st(y + i, v);
  }
}

// ---- counter-based RNG (Philox4x32-10): stands in for tf.random_normal (gan_resnet.py:359) and
// tf.random_uniform (gan_resnet.py:549).  state[0..1] = 64-bit stream offset kept on the device so a
// captured graph draws fresh numbers on every replay; rng_advance_kernel bumps it after each use.
// kind 0: uniform [lo, hi)   kind 1: normal(mean=lo, std=hi)
template <typename T>
__global__ void rng_fill_kernel(size_t count, int kind, float lo, float hi, uint64_t seed, const uint64_t* state, T* y) {
  const uint64_t base = state ? state[0] : 0;
  const size_t nquad = (count + 3) / 4;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (size_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4(base + q, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    float v[4];
    if (kind == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = philox_uniform(r[i], lo, hi);
    } else {
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        float u1 = ((float)(r[i] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        float u2 = ((float)(r[i + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        float rad = sqrtf(-2.f * logf(u1));
        v[i] = lo + hi * rad * cosf(6.28318530718f * u2);
        v[i + 1] = lo + hi * rad * sinf(6.28318530718f * u2);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (q * 4 + i < count) Elem<T>::st(y + q * 4 + i, v[i]);
  }
}

__global__ void rng_advance_kernel(uint64_t* state, uint64_t n) { state[0] += n; }

// y[row][before + c] = x[row][c], zero in the `before` leading and `after` trailing channels (tf.pad on the channel axis:
// the option-A shortcut of the label-classifier ResNet, resnet-110/graph_optimized.pb nodes conv{2,3}_0/Pad)
template <typename T>
__global__ void pad_channels_kernel(size_t rows, int c, int before, int after, const T* x, T* y) {
  const int co = before + c + after;
  const size_t total = rows * (size_t)co;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % co) - before;
    const size_t r = i / co;
    Elem<T>::st(y + i, (ch >= 0 && ch < c) ? Elem<T>::ld(x + r * c + ch) : 0.f);
  }
}

extern "C" {

int rcgan_rng_fill(rcgan_ctx* ctx, size_t count, int dtype, int kind, float lo, float hi, uint64_t seed, void* state, void* y) {
  RC_REQUIRE(ctx, kind == 0 || kind == 1, "kind %d", kind);
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(rng_fill_kernel<T>, dim3(ew_grid((count + 3) / 4)), dim3(EW_BLOCK), 0, ctx->stream, count, kind, lo, hi, seed, (const uint64_t*)state, (T*)y));
  RC_LAUNCH_CHECK(ctx);
  if (state) {
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, ctx->stream, (uint64_t*)state, (uint64_t)((count + 3) / 4));
    RC_LAUNCH_CHECK(ctx);
  }
  return RCGAN_OK;
}

int rcgan_act_fwd(rcgan_ctx* ctx, size_t count, int dtype, int act, const void* x, void* y) {
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(act_fwd_kernel<T>, dim3(ew_grid(count)), dim3(EW_BLOCK), 0, ctx->stream, count, act, (const T*)x, (T*)y));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_act_bwd(rcgan_ctx* ctx, size_t count, int dtype, int act, const void* s, const void* dy, void* dx, int accumulate) {
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(ew_grid(count)), dim3(EW_BLOCK), 0, ctx->stream, count, act, (const T*)s, (const T*)dy, (T*)dx, accumulate));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_add(rcgan_ctx* ctx, size_t count, int dtype, const void* a, const void* b, void* y) {
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(axpby_kernel<T>, dim3(ew_grid(count)), dim3(EW_BLOCK), 0, ctx->stream, count, 1.f, (const T*)a, 1.f, (const T*)b, (T*)y));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// y = alpha*a + beta*y
int rcgan_axpby(rcgan_ctx* ctx, size_t count, int dtype, float alpha, const void* a, float beta, void* y) {
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(axpby_kernel<T>, dim3(ew_grid(count)), dim3(EW_BLOCK), 0, ctx->stream, count, alpha, (const T*)a, beta, (const T*)y, (T*)y));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_cast(rcgan_ctx* ctx, size_t count, int sd, const void* s, int dd, void* d) {
  dim3 g(ew_grid(count)), b(EW_BLOCK);
  if (sd == RCGAN_F32 && dd == RCGAN_H16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, b, 0, ctx->stream, count, (const float*)s, (bf16_t*)d);
  else if (sd == RCGAN_H16 && dd == RCGAN_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, b, 0, ctx->stream, count, (const bf16_t*)s, (float*)d);
  else if (sd == RCGAN_F32 && dd == RCGAN_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, ctx->stream, count, (const float*)s, (float*)d);
  else if (sd == RCGAN_H16 && dd == RCGAN_H16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, b, 0, ctx->stream, count, (const bf16_t*)s, (bf16_t*)d);
  else RC_FAIL(ctx, RCGAN_EINVALID_ARG, "bad dtypes %d %d", sd, dd);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_copy_words(rcgan_ctx* ctx, size_t count, const void* src, void* dst) {
  RC_REQUIRE(ctx, src && dst, "null pointer");
  if (count == 0) return RCGAN_OK;
  hipLaunchKernelGGL(copy_words_kernel, dim3(ew_grid((count + 3) / 4)), dim3(EW_BLOCK), 0, ctx->stream, count, (const uint32_t*)src, (uint32_t*)dst);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_fill_f32(rcgan_ctx* ctx, size_t count, float* p, float v) {
  hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(count)), dim3(EW_BLOCK), 0, ctx->stream, count, p, v);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_meanpool2_fwd(rcgan_ctx* ctx, int n, int h, int w, int c, int dtype, const void* x, void* y) {
  RC_REQUIRE(ctx, (h % 2 == 0) && (w % 2 == 0), "odd spatial size %dx%d", h, w);
  size_t cnt = (size_t)n * (h / 2) * (w / 2) * c;
  if (resample_vec_ok(n, h / 2, w / 2, c)) {
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL((resample2_vec_kernel<T, 0>), dim3(resample_grid(cnt / 8)), dim3(256), 0, ctx->stream,
                                                     (unsigned)(cnt / 8), h / 2, w / 2, c, 0.25f, (const T*)x, (const T*)nullptr, (T*)y, 0));
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(meanpool2_fwd_kernel<T>, dim3(ew_grid(cnt)), dim3(EW_BLOCK), 0, ctx->stream, n, h, w, c, (const T*)x, (T*)y));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_meanpool2_bwd(rcgan_ctx* ctx, int n, int h, int w, int c, int dtype, const void* dy, void* dx, int accumulate) {
  size_t cnt = (size_t)n * h * w * c;
  if (h % 2 == 0 && w % 2 == 0 && resample_vec_ok(n, h / 2, w / 2, c)) {
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL((resample2_vec_kernel<T, 1>), dim3(resample_grid(cnt / 32)), dim3(256), 0, ctx->stream,
                                                     (unsigned)(cnt / 32), h / 2, w / 2, c, 0.25f, (const T*)dy, (const T*)nullptr, (T*)dx, accumulate));
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(expand2_kernel<T>, dim3(ew_grid(cnt)), dim3(EW_BLOCK), 0, ctx->stream, n, h, w, c, 0.25f, (const T*)dy, (T*)dx, accumulate));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// h, w: OUTPUT (upsampled) size
int rcgan_upsample2_fwd(rcgan_ctx* ctx, int n, int h, int w, int c, int dtype, const void* x, void* y) {
  size_t cnt = (size_t)n * h * w * c;
  if (h % 2 == 0 && w % 2 == 0 && resample_vec_ok(n, h / 2, w / 2, c)) {
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL((resample2_vec_kernel<T, 1>), dim3(resample_grid(cnt / 32)), dim3(256), 0, ctx->stream,
                                                     (unsigned)(cnt / 32), h / 2, w / 2, c, 1.f, (const T*)x, (const T*)nullptr, (T*)y, 0));
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(expand2_kernel<T>, dim3(ew_grid(cnt)), dim3(EW_BLOCK), 0, ctx->stream, n, h, w, c, 1.f, (const T*)x, (T*)y, 0));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_upsample2_bwd(rcgan_ctx* ctx, int n, int h, int w, int c, int dtype, const void* dy, void* dx, int accumulate) {
  size_t cnt = (size_t)n * (h / 2) * (w / 2) * c;
  if (resample_vec_ok(n, h / 2, w / 2, c)) {
    RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL((resample2_vec_kernel<T, 0>), dim3(resample_grid(cnt / 8)), dim3(256), 0, ctx->stream,
                                                     (unsigned)(cnt / 8), h / 2, w / 2, c, 1.f, (const T*)dy, (const T*)nullptr, (T*)dx, accumulate));
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(sumpool2_kernel<T>, dim3(ew_grid(cnt)), dim3(EW_BLOCK), 0, ctx->stream, n, h, w, c, (const T*)dy, (const T*)nullptr, (T*)dx, accumulate));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_pad_channels(rcgan_ctx* ctx, size_t rows, int c, int before, int after, int dtype, const void* x, void* y) {
  RC_REQUIRE(ctx, c > 0 && before >= 0 && after >= 0, "bad channel padding %d + %d + %d", before, c, after);
  const size_t cnt = rows * (size_t)(before + c + after);
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(pad_channels_kernel<T>, dim3(ew_grid(cnt)), dim3(EW_BLOCK), 0, ctx->stream, rows, c, before, after,
                                                   (const T*)x, (T*)y));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_concat_channels_fwd(rcgan_ctx* ctx, int n, int hw, int c1, int c2, int dtype, const void* x, const float* yb, void* y) {
  size_t cnt = (size_t)n * hw * (c1 + c2);
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(concat_channels_fwd_kernel<T>, dim3(ew_grid(cnt)), dim3(EW_BLOCK), 0, ctx->stream, n, hw, c1, c2, (const T*)x, yb, (T*)y));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_concat_channels_bwd(rcgan_ctx* ctx, int n, int hw, int c1, int c2, int dtype, const void* dy, void* dx) {
  size_t cnt = (size_t)n * hw * c1;
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(concat_channels_bwd_kernel<T>, dim3(ew_grid(cnt)), dim3(EW_BLOCK), 0, ctx->stream, n, hw, c1, c2, (const T*)dy, (T*)dx));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_tile_rows_fwd(rcgan_ctx* ctx, size_t count, int reps, int dtype, const void* x, void* y) {
  RC_REQUIRE(ctx, reps >= 1 && x && y, "bad arguments");
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(tile_rows_fwd_kernel<T>, dim3(ew_grid(count)), dim3(EW_BLOCK), 0, ctx->stream, count, reps, (const T*)x, (T*)y));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_tile_rows_bwd(rcgan_ctx* ctx, size_t count, int reps, int dtype, const void* dy, void* dx, int accumulate) {
  RC_REQUIRE(ctx, reps >= 1 && dy && dx, "bad arguments");
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(tile_rows_bwd_kernel<T>, dim3(ew_grid(count)), dim3(EW_BLOCK), 0, ctx->stream, count, reps, (const T*)dy, (T*)dx, accumulate));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_transpose_f32(rcgan_ctx* ctx, int rows, int cols, const float* x, float* y, int accumulate) {
  RC_REQUIRE(ctx, rows >= 1 && cols >= 1 && x && y, "bad arguments");
  hipLaunchKernelGGL(transpose_f32_kernel, dim3(ew_grid((size_t)rows * cols)), dim3(EW_BLOCK), 0, ctx->stream, rows, cols, x, y, accumulate);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int rcgan_preprocess_cifar(rcgan_ctx* ctx, int n, const int32_t* img, const float* noise, int dtype, void* y) {
  size_t cnt = (size_t)n * 3072;
  RC_DISPATCH_DTYPE(ctx, dtype, hipLaunchKernelGGL(preprocess_cifar_kernel<T>, dim3(ew_grid(cnt)), dim3(EW_BLOCK), 0, ctx->stream, n, img, noise, (T*)y));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

}  // extern "C"

// used by the conv API for the upsample-folded data gradient
template <typename T>
int sumpool2_masked_launch(rcgan_ctx* ctx, int n, int h, int w, int c, const T* dy, const T* xmask, T* dx, int accumulate) {
  size_t cnt = (size_t)n * (h / 2) * (w / 2) * c;
  if (resample_vec_ok(n, h / 2, w / 2, c)) {
    hipLaunchKernelGGL((resample2_vec_kernel<T, 0>), dim3(resample_grid(cnt / 8)), dim3(256), 0, ctx->stream, (unsigned)(cnt / 8), h / 2, w / 2, c,
                       1.f, dy, xmask, dx, accumulate);
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  hipLaunchKernelGGL(sumpool2_kernel<T>, dim3(ew_grid(cnt)), dim3(EW_BLOCK), 0, ctx->stream, n, h, w, c, dy, xmask, dx, accumulate);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}
template int sumpool2_masked_launch<float>(rcgan_ctx*, int, int, int, int, const float*, const float*, float*, int);
template int sumpool2_masked_launch<bf16_t>(rcgan_ctx*, int, int, int, int, const bf16_t*, const bf16_t*, bf16_t*, int);
