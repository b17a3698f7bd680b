// Convolutions with a 3-channel (<= 4) side: the image ends of both CIFAR nets (D.Block.1.Conv1 /
// Shortcut: 3 -> 128; G.Output: 256 -> 3) and their gradients.  None of them is GEMM-shaped enough for
// MFMA (K = 27 or N = 3): all are HBM-bound on the big-channel tensor, so each kernel streams that tensor
// exactly once with coalesced accesses and keeps the small side in registers / LDS.
//   family K  "small reduction":  out[m][n]   = sum_{t, c<Cs} in[pix(m,t)][c] * w(t,c,n)      (N = 128/256)
//   family S  "small output":     out[m][n<4] = sum_{t, c<C}  in[pix(m,t)][c] * w(t,c,n)      (C = 128/256)
//   family W  "small-side wgrad": dW[t][cs][n] = sum_m S[pix(m,+-t)][cs] * Bg[m][n]
// Reference call sites: cifar10/gan_resnet.py:337-352 (D.Block.1), :368 (G.Output) via conv2d.py:181-187.
#include "common.h"

// packed 8-wide store / CPL-wide load helpers (bf16: one 16-B / 8-B / 4-B access instead of scalar shorts)
__device__ __forceinline__ void store8(float* o, const float* v, int accumulate) {
  float4 a = make_float4(v[0], v[1], v[2], v[3]), b = make_float4(v[4], v[5], v[6], v[7]);
  if (accumulate) {
    float4 x = *(const float4*)o, y = *(const float4*)(o + 4);
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
  }
  *(float4*)o = a;
  *(float4*)(o + 4) = b;
}
__device__ __forceinline__ void store8(bf16_t* o, const float* v, int accumulate) {
  float t[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = v[j];
  if (accumulate) {
    uint4 x = *(const uint4*)o;
    uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[2 * j] += bf16_to_f32((bf16_t)(w[j] & 0xffff)); t[2 * j + 1] += bf16_to_f32((bf16_t)(w[j] >> 16)); }
  }
  uint4 pk;
  pk.x = (uint32_t)f32_to_bf16(t[0]) | ((uint32_t)f32_to_bf16(t[1]) << 16);
  pk.y = (uint32_t)f32_to_bf16(t[2]) | ((uint32_t)f32_to_bf16(t[3]) << 16);
  pk.z = (uint32_t)f32_to_bf16(t[4]) | ((uint32_t)f32_to_bf16(t[5]) << 16);
  pk.w = (uint32_t)f32_to_bf16(t[6]) | ((uint32_t)f32_to_bf16(t[7]) << 16);
  *(uint4*)o = pk;
}
template <int CPL> __device__ __forceinline__ void loadv(const float* p, float* a) {
  if (CPL == 4) { float4 v = *(const float4*)p; a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w; }
  else { float2 v = *(const float2*)p; a[0] = v.x; a[1] = v.y; }
}
template <int CPL> __device__ __forceinline__ void loadv(const bf16_t* p, float* a) {
  if (CPL == 4) {
    uint2 v = *(const uint2*)p;
    a[0] = bf16_to_f32((bf16_t)(v.x & 0xffff)); a[1] = bf16_to_f32((bf16_t)(v.x >> 16));
    a[2] = bf16_to_f32((bf16_t)(v.y & 0xffff)); a[3] = bf16_to_f32((bf16_t)(v.y >> 16));
  } else {
    uint32_t v = *(const uint32_t*)p;
    a[0] = bf16_to_f32((bf16_t)(v & 0xffff)); a[1] = bf16_to_f32((bf16_t)(v >> 16));
  }
}

struct SmallGeom {
  int N, H, W;          // batch, spatial size of the OUTPUT pixel grid (stride-s conv: output grid)
  int IH, IW;           // spatial size of the gathered tensor
  int KH, KW, S, PT, PL;
  int Cs, Cb;           // small / big channel counts
  long M;               // N*H*W
};

// ------------------------------------------------------------------------------------------------
// family K: each thread produces 8 consecutive big channels of one pixel
// WMODE 0: w(t,c,n) = W[(t*Cs+c)*Cb + n]            (forward, HWIO with Cin = Cs, Cout = Cb)
// WMODE 1: w(t,c,n) = W[((T-1-t)*Cb + n)*Cs + c]    (data gradient of a conv with Cin = Cb, Cout = Cs)
// ------------------------------------------------------------------------------------------------
// The <= 4-channel operand of a 64-pixel chunk (all taps, zero-filled halo) is gathered once into LDS
// (Ss[p][tap*4+c]); the main loop then only issues LDS broadcast reads, FMAs and one 16/32-byte store.
#define SM_PIX 64
template <typename T, int TT>
__device__ __forceinline__ void stage_small(const SmallGeom& g, const T* S, long m0, long me, int sign, float* Ss) {
  for (int idx = threadIdx.x; idx < SM_PIX * TT; idx += 256) {
    const int p = idx / TT, t = idx - p * TT;
    const long m = m0 + p;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < me) {
      int ow = (int)(m % g.W);
      long q = m / g.W;
      int oh = (int)(q % g.H);
      int b = (int)(q / g.H);
      const int kh = TT == 9 ? t / 3 : 0, kw = TT == 9 ? t % 3 : 0;
      int ih = sign > 0 ? oh * g.S + kh - g.PT : oh - kh + g.PT;
      int iw = sign > 0 ? ow * g.S + kw - g.PL : ow - kw + g.PL;
      if (ih >= 0 && ih < g.IH && iw >= 0 && iw < g.IW) {
        const T* ptr = S + (((long)b * g.IH + ih) * g.IW + iw) * g.Cs;
        for (int c = 0; c < g.Cs; ++c) v[c] = Elem<T>::ld(ptr + c);
      }
    }
    *(float4*)(Ss + p * (TT * 4) + t * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <typename T, int WMODE, int TT>
__global__ __launch_bounds__(256) void conv_smallk_kernel(SmallGeom g, const T* in, const float* w, const float* bias,
                                                           T* out, int accumulate, long pix_per_block) {
  extern __shared__ __attribute__((aligned(16))) float smemf[];
  float* Ws = smemf;                        // [TT*4][Cb]  (rows of unused channels are zero)
  float* Ss = smemf + TT * 4 * g.Cb;        // [SM_PIX][TT*4]
  for (int i = threadIdx.x; i < TT * 4 * g.Cb; i += 256) {
    int n = i % g.Cb, k4 = i / g.Cb;
    int c = k4 & 3, t = k4 >> 2;
    float v = 0.f;
    if (c < g.Cs) v = WMODE == 0 ? w[((long)t * g.Cs + c) * g.Cb + n] : w[((long)(TT - 1 - t) * g.Cb + n) * g.Cs + c];
    Ws[i] = v;
  }
  const int tpp = g.Cb / 8;                // threads per pixel
  const int ppp = 256 / tpp;               // pixels per pass
  const int n0 = (threadIdx.x % tpp) * 8;
  const int pl = threadIdx.x / tpp;
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bv[j] = bias ? bias[n0 + j] : 0.f;
  const long mb = (long)blockIdx.x * pix_per_block;
  long me = mb + pix_per_block;
  if (me > g.M) me = g.M;
  for (long m0 = mb; m0 < me; m0 += SM_PIX) {
    __syncthreads();
    stage_small<T, TT>(g, in, m0, me, 1, Ss);
    __syncthreads();
    for (int p = pl; p < SM_PIX; p += ppp) {
      const long m = m0 + p;
      if (m >= me) break;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = bv[j];
      const float* sp = Ss + p * (TT * 4);
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const float4 sv = *(const float4*)(sp + t * 4);
        const float a4[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c >= 3 && g.Cs <= 3) break;
          const float* wr = Ws + (t * 4 + c) * g.Cb + n0;
          const float4 w0 = *(const float4*)wr, w1 = *(const float4*)(wr + 4);
          acc[0] = fmaf(a4[c], w0.x, acc[0]); acc[1] = fmaf(a4[c], w0.y, acc[1]);
          acc[2] = fmaf(a4[c], w0.z, acc[2]); acc[3] = fmaf(a4[c], w0.w, acc[3]);
          acc[4] = fmaf(a4[c], w1.x, acc[4]); acc[5] = fmaf(a4[c], w1.y, acc[5]);
          acc[6] = fmaf(a4[c], w1.z, acc[6]); acc[7] = fmaf(a4[c], w1.w, acc[7]);
        }
      }
      store8(out + m * g.Cb + n0, acc, accumulate);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// family S: one wavefront per pixel, lanes over the big channels, <= 4 outputs reduced across the wave
// WMODE 0: w(t,c,n) = W[(t*Cb+c)*Cs + n]            (forward, Cin = Cb, Cout = Cs)
// WMODE 1: w(t,c,n) = W[((T-1-t)*Cs + n)*Cb + c]    (data gradient of a conv with Cin = Cs, Cout = Cb)
// CPL = big channels per lane (Cb = 64*CPL)
// ------------------------------------------------------------------------------------------------
template <typename T, int WMODE, int CPL, int TT>
__global__ __launch_bounds__(256) void conv_smalln_kernel(SmallGeom g, const T* in, const float* w, const float* bias,
                                                           T* out, int accumulate, int relu_in) {
  constexpr int Tt = TT;
  const int lane = threadIdx.x & 63;
  const int c0 = lane * CPL;
  // this lane's weights: [t][j<CPL][n<4] kept in registers for 3x3 / 1x1 filters (Tt <= 9)
  float wr[TT][CPL][4];
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int j = 0; j < CPL; ++j)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float v = 0.f;
        if (n < g.Cs)
          v = WMODE == 0 ? w[((long)t * g.Cb + c0 + j) * g.Cs + n] : w[((long)(Tt - 1 - t) * g.Cs + n) * g.Cb + c0 + j];
        wr[t][j][n] = v;
      }
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwave = (long)gridDim.x * 4;
  for (long m = wave; m < g.M; m += nwave) {
    int ow = (int)(m % g.W);
    long q = m / g.W;
    int oh = (int)(q % g.H);
    int b = (int)(q / g.H);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const int kh = TT == 9 ? t / 3 : 0, kw = TT == 9 ? t % 3 : 0;
      int ih = oh + kh - g.PT, iw = ow + kw - g.PL;
      const bool ok = ih >= 0 && ih < g.IH && iw >= 0 && iw < g.IW;
      if (!ok) { ih = 0; iw = 0; }
      const T* p = in + (((long)b * g.IH + ih) * g.IW + iw) * g.Cb + c0;
      float av[CPL];
      loadv<CPL>(p, av);
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        float a = ok ? av[j] : 0.f;
        if (relu_in) a = a > 0.f ? a : 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = fmaf(a, wr[t][j][n], acc[n]);
      }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = wave_sum(acc[n]);
    if (lane < g.Cs) {
      float v = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
      if (bias) v += bias[lane];
      T* o = out + m * g.Cs + lane;
      if (accumulate) v += Elem<T>::ld(o);
      Elem<T>::st(o, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// family W: partial[blk][k<K][n<Cb] = sum over the block's pixels of S[pix(m,sign*t)][cs] * Bg[m][n]
// SIGN +1: S gathered at (oh + kh - PT)   (filter gradient with the small tensor as conv INPUT)
// SIGN -1: S gathered at (oh - kh + PT)   (filter gradient with the small tensor as conv OUTPUT gradient)
// also partial[blk][K][n] = sum Bg[m][n] and partial[blk][K+1][c<Cs] = sum S[m][c]  (bias gradients)
// ------------------------------------------------------------------------------------------------
template <typename T, int SIGN, int TT>
__global__ __launch_bounds__(256) void conv_smallw_kernel(SmallGeom g, const T* S, const T* Bg, float* partial,
                                                           long pix_per_block, int relu_big) {
  __shared__ __attribute__((aligned(16))) float Ss[SM_PIX * TT * 4];
  __shared__ float red[256];
  constexpr int K4 = TT * 4;
  constexpr int CTR = TT == 9 ? 4 : 0;                  // the un-shifted tap
  const int K = TT * g.Cs;
  const int npl = 256 / g.Cb > 0 ? 256 / g.Cb : 1;      // pixel lanes (Cb = 128 -> 2, 256 -> 1)
  const int n = threadIdx.x % g.Cb;
  const int pl = threadIdx.x / g.Cb;
  const bool active = pl < npl;
  float acc[K4];
#pragma unroll
  for (int k = 0; k < K4; ++k) acc[k] = 0.f;
  float bsum = 0.f, ssum = 0.f;
  const long mb = (long)blockIdx.x * pix_per_block;
  long me = mb + pix_per_block;
  if (me > g.M) me = g.M;
  for (long m0 = mb; m0 < me; m0 += SM_PIX) {
    __syncthreads();
    stage_small<T, TT>(g, S, m0, me, SIGN, Ss);
    __syncthreads();
    if (active) {
      for (int p = pl; p < SM_PIX; p += npl) {
        const long m = m0 + p;
        if (m >= me) break;
        float bg = Elem<T>::ld(Bg + m * g.Cb + n);
        if (relu_big) bg = bg > 0.f ? bg : 0.f;
        bsum += bg;
        const float* sp = Ss + p * K4;
        if (n < 4) ssum += sp[CTR * 4 + n];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const float4 sv = *(const float4*)(sp + t * 4);
          acc[t * 4 + 0] = fmaf(sv.x, bg, acc[t * 4 + 0]);
          acc[t * 4 + 1] = fmaf(sv.y, bg, acc[t * 4 + 1]);
          acc[t * 4 + 2] = fmaf(sv.z, bg, acc[t * 4 + 2]);
          acc[t * 4 + 3] = fmaf(sv.w, bg, acc[t * 4 + 3]);
        }
      }
    }
  }
  // reduce the pixel lanes through LDS and write the block's partial slab
  float* slab = partial + (long)blockIdx.x * (K + 2) * g.Cb;
#pragma unroll
  for (int k4 = 0; k4 < K4 + 2; ++k4) {
    const int t = k4 / 4, c = k4 % 4;
    if (k4 < K4 && c >= g.Cs) continue;              // uniform across the block
    float v = k4 < K4 ? acc[k4 < K4 ? k4 : 0] : (k4 == K4 ? bsum : ssum);
    const int k = k4 < K4 ? t * g.Cs + c : (k4 == K4 ? K : K + 1);
    red[threadIdx.x] = active ? v : 0.f;
    __syncthreads();
    if (pl == 0) {
      float tot = 0.f;
      for (int q = 0; q < npl; ++q) tot += red[q * g.Cb + n];
      slab[(long)k * g.Cb + n] = tot;
    }
    __syncthreads();
  }
}

// out (= or +=) sum_blk partial[blk][k][n], written as dW in HWIO order.
// ORIENT 0: dW[(t*Cs+c)*Cb + n], dbias[n<Cb] from slot K   ORIENT 1: dW[(t*Cb+n)*Cs + c], dbias[c<Cs] from slot K+1.
__global__ __launch_bounds__(256) void conv_smallw_reduce_kernel(const float* partial, int nblk, int K, int Cs, int Cb, int orient,
                                                                 float* dw, float* dbias, int accumulate) {
  __shared__ float red[4][64];
  const int total = (K + 2) * Cb;
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int bl = threadIdx.x >> 6;
  float s = 0.f;
  if (i < total)
    for (int b = bl; b < nblk; b += 4) s += partial[(long)b * total + i];
  red[bl][threadIdx.x & 63] = s;
  __syncthreads();
  if (bl != 0 || i >= total) return;
  s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  const int k = i / Cb, n = i % Cb;
  if (k >= K) {
    const bool mine = (orient == 0 && k == K) || (orient == 1 && k == K + 1 && n < Cs);
    if (!mine || !dbias) return;
    if (accumulate) s += dbias[n];
    dbias[n] = s;
    return;
  }
  const int t = k / Cs, c = k % Cs;
  const long o = orient == 0 ? (long)k * Cb + n : ((long)t * Cb + n) * Cs + c;
  if (accumulate) s += dw[o];
  dw[o] = s;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static SmallGeom small_geom(const rcgan_conv_desc* d) {
  SmallGeom g;
  int oh, ow, pt, pl;
  same_pad(d->h, d->kh, d->stride, &oh, &pt);
  same_pad(d->w, d->kw, d->stride, &ow, &pl);
  g.N = d->n; g.H = oh; g.W = ow; g.IH = d->h; g.IW = d->w;
  g.KH = d->kh; g.KW = d->kw; g.S = d->stride; g.PT = pt; g.PL = pl;
  g.M = (long)d->n * oh * ow;
  g.Cs = 0; g.Cb = 0;
  return g;
}

static bool plain(const rcgan_conv_desc* d) {
  return !(d->flags & (RCGAN_CONV_IN_UPSAMPLE2X | RCGAN_CONV_FORCE_DIRECT)) && d->kh == d->kw && (d->kh == 3 || d->kh == 1);
}

// forward eligibility
int small_fwd_kind(const rcgan_conv_desc* d) {
  if (!plain(d)) return 0;
  if (d->cin <= 4 && (d->cout == 128 || d->cout == 256) && !(d->flags & RCGAN_CONV_IN_RELU)) return 1;               // family K
  if (d->cout <= 4 && (d->cin == 128 || d->cin == 256) && d->stride == 1) return 2;                                  // family S
  return 0;
}
int small_dgrad_kind(const rcgan_conv_desc* d) {
  if (!plain(d) || d->stride != 1 || (d->flags & RCGAN_CONV_IN_RELU)) return 0;
  if (d->cout <= 4 && (d->cin == 128 || d->cin == 256)) return 1;    // dX[m][big] from dY[m][small]: family K, WMODE 1
  if (d->cin <= 4 && (d->cout == 128 || d->cout == 256)) return 2;   // dX[m][small] from dY[m][big]: family S, WMODE 1
  return 0;
}
int small_wgrad_kind(const rcgan_conv_desc* d) {
  if (!plain(d) || d->stride != 1) return 0;
  if (d->cin <= 4 && (d->cout == 128 || d->cout == 256) && !(d->flags & RCGAN_CONV_IN_RELU)) return 1;   // S = x, Bg = dy
  if (d->cout <= 4 && (d->cin == 128 || d->cin == 256)) return 2;                                         // S = dy, Bg = x
  return 0;
}

size_t small_wgrad_ws_bytes(const rcgan_conv_desc* d) {
  long M = (long)d->n * d->h * d->w;
  int cs = d->cin <= 4 ? d->cin : d->cout, cb = d->cin <= 4 ? d->cout : d->cin;
  long nblk = (M + 255) / 256;
  if (nblk > 512) nblk = 512;
  return (size_t)nblk * (d->kh * d->kw * cs + 2) * cb * sizeof(float) + 256;
}

template <typename T>
int small_fwd(rcgan_ctx* ctx, const rcgan_conv_desc* d, int kind, const T* x, const float* w, const float* bias, T* y) {
  SmallGeom g = small_geom(d);
  const int acc = (d->flags & RCGAN_CONV_ACCUMULATE) ? 1 : 0;
  if (kind == 1) {
    g.Cs = d->cin; g.Cb = d->cout;
    const int TT = d->kh * d->kw;
    size_t lds = ((size_t)TT * 4 * g.Cb + (size_t)SM_PIX * TT * 4) * sizeof(float);
    long ppb = 256;
    while ((g.M + ppb - 1) / ppb > 2048) ppb *= 2;
    int blocks = cdiv(g.M, ppb);
    if (TT == 9) hipLaunchKernelGGL((conv_smallk_kernel<T, 0, 9>), dim3(blocks), dim3(256), lds, ctx->stream, g, x, w, bias, y, acc, ppb);
    else hipLaunchKernelGGL((conv_smallk_kernel<T, 0, 1>), dim3(blocks), dim3(256), lds, ctx->stream, g, x, w, bias, y, acc, ppb);
  } else {
    g.Cs = d->cout; g.Cb = d->cin;
    long blocks = (g.M + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    const int relu = (d->flags & RCGAN_CONV_IN_RELU) ? 1 : 0;
    const bool k3 = d->kh == 3;
    if (g.Cb == 256 && k3) hipLaunchKernelGGL((conv_smalln_kernel<T, 0, 4, 9>), dim3((int)blocks), dim3(256), 0, ctx->stream, g, x, w, bias, y, acc, relu);
    else if (g.Cb == 256) hipLaunchKernelGGL((conv_smalln_kernel<T, 0, 4, 1>), dim3((int)blocks), dim3(256), 0, ctx->stream, g, x, w, bias, y, acc, relu);
    else if (k3) hipLaunchKernelGGL((conv_smalln_kernel<T, 0, 2, 9>), dim3((int)blocks), dim3(256), 0, ctx->stream, g, x, w, bias, y, acc, relu);
    else hipLaunchKernelGGL((conv_smalln_kernel<T, 0, 2, 1>), dim3((int)blocks), dim3(256), 0, ctx->stream, g, x, w, bias, y, acc, relu);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}
template int small_fwd<float>(rcgan_ctx*, const rcgan_conv_desc*, int, const float*, const float*, const float*, float*);
template int small_fwd<bf16_t>(rcgan_ctx*, const rcgan_conv_desc*, int, const bf16_t*, const float*, const float*, bf16_t*);

template <typename T>
int small_dgrad(rcgan_ctx* ctx, const rcgan_conv_desc* d, int kind, const T* dy, const float* w, T* dx, int accumulate) {
  SmallGeom g = small_geom(d);     // stride 1: output grid == input grid
  g.PT = d->kh - 1 - g.PT; g.PL = d->kw - 1 - g.PL;     // gather offsets of the flipped filter
  if (kind == 1) {
    g.Cs = d->cout; g.Cb = d->cin;
    const int TT = d->kh * d->kw;
    size_t lds = ((size_t)TT * 4 * g.Cb + (size_t)SM_PIX * TT * 4) * sizeof(float);
    long ppb = 256;
    while ((g.M + ppb - 1) / ppb > 2048) ppb *= 2;
    int blocks = cdiv(g.M, ppb);
    const float* nb0 = nullptr;
    if (TT == 9) hipLaunchKernelGGL((conv_smallk_kernel<T, 1, 9>), dim3(blocks), dim3(256), lds, ctx->stream, g, dy, w, nb0, dx, accumulate, ppb);
    else hipLaunchKernelGGL((conv_smallk_kernel<T, 1, 1>), dim3(blocks), dim3(256), lds, ctx->stream, g, dy, w, nb0, dx, accumulate, ppb);
  } else {
    g.Cs = d->cin; g.Cb = d->cout;
    long blocks = (g.M + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    const bool k3 = d->kh == 3;
    const float* nb = nullptr;
    if (g.Cb == 256 && k3) hipLaunchKernelGGL((conv_smalln_kernel<T, 1, 4, 9>), dim3((int)blocks), dim3(256), 0, ctx->stream, g, dy, w, nb, dx, accumulate, 0);
    else if (g.Cb == 256) hipLaunchKernelGGL((conv_smalln_kernel<T, 1, 4, 1>), dim3((int)blocks), dim3(256), 0, ctx->stream, g, dy, w, nb, dx, accumulate, 0);
    else if (k3) hipLaunchKernelGGL((conv_smalln_kernel<T, 1, 2, 9>), dim3((int)blocks), dim3(256), 0, ctx->stream, g, dy, w, nb, dx, accumulate, 0);
    else hipLaunchKernelGGL((conv_smalln_kernel<T, 1, 2, 1>), dim3((int)blocks), dim3(256), 0, ctx->stream, g, dy, w, nb, dx, accumulate, 0);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}
template int small_dgrad<float>(rcgan_ctx*, const rcgan_conv_desc*, int, const float*, const float*, float*, int);
template int small_dgrad<bf16_t>(rcgan_ctx*, const rcgan_conv_desc*, int, const bf16_t*, const float*, bf16_t*, int);

template <typename T>
int small_wgrad(rcgan_ctx* ctx, const rcgan_conv_desc* d, int kind, const T* x, const T* dy, float* dw, float* dbias,
                int accumulate, void* ws, size_t ws_bytes) {
  SmallGeom g = small_geom(d);
  size_t need = small_wgrad_ws_bytes(d);
  if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
  long nblk = (g.M + 255) / 256;
  if (nblk > 512) nblk = 512;
  long ppb = (g.M + nblk - 1) / nblk;
  nblk = (g.M + ppb - 1) / ppb;
  float* partial = (float*)ws;
  if (kind == 1) {
    g.Cs = d->cin; g.Cb = d->cout;
    if (d->kh == 3) hipLaunchKernelGGL((conv_smallw_kernel<T, 1, 9>), dim3((int)nblk), dim3(256), 0, ctx->stream, g, x, dy, partial, ppb, 0);
    else hipLaunchKernelGGL((conv_smallw_kernel<T, 1, 1>), dim3((int)nblk), dim3(256), 0, ctx->stream, g, x, dy, partial, ppb, 0);
    RC_LAUNCH_CHECK(ctx);
    int K = d->kh * d->kw * g.Cs;
    hipLaunchKernelGGL(conv_smallw_reduce_kernel, dim3(cdiv((K + 2) * g.Cb, 64)), dim3(256), 0, ctx->stream, (const float*)partial,
                       (int)nblk, K, g.Cs, g.Cb, 0, dw, dbias, accumulate);
    RC_LAUNCH_CHECK(ctx);
  } else {
    g.Cs = d->cout; g.Cb = d->cin;
    const int relu = (d->flags & RCGAN_CONV_IN_RELU) ? 1 : 0;
    if (d->kh == 3) hipLaunchKernelGGL((conv_smallw_kernel<T, -1, 9>), dim3((int)nblk), dim3(256), 0, ctx->stream, g, dy, x, partial, ppb, relu);
    else hipLaunchKernelGGL((conv_smallw_kernel<T, -1, 1>), dim3((int)nblk), dim3(256), 0, ctx->stream, g, dy, x, partial, ppb, relu);
    RC_LAUNCH_CHECK(ctx);
    int K = d->kh * d->kw * g.Cs;
    hipLaunchKernelGGL(conv_smallw_reduce_kernel, dim3(cdiv((K + 2) * g.Cb, 64)), dim3(256), 0, ctx->stream, (const float*)partial,
                       (int)nblk, K, g.Cs, g.Cb, 1, dw, dbias, accumulate);
    RC_LAUNCH_CHECK(ctx);
  }
  return RCGAN_OK;
}
template int small_wgrad<float>(rcgan_ctx*, const rcgan_conv_desc*, int, const float*, const float*, float*, float*, int, void*, size_t);
template int small_wgrad<bf16_t>(rcgan_ctx*, const rcgan_conv_desc*, int, const bf16_t*, const bf16_t*, float*, float*, int, void*, size_t);
