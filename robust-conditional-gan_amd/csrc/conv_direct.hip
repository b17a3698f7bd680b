// Generic fp32-accumulate implicit-GEMM convolution on the vector ALUs ("direct" path).
//
// One LDS-tiled 64x64x16 GEMM skeleton, C[i][j] = sum_r A(i,r) * B(r,j), with the operand fetch
// expressed as gather functors.  It serves every shape the MFMA path does not take: fp32 parity
// mode, the 3-channel ends of the CIFAR nets (Cin=3 / Cout=3), the MNIST 5x5 stride-2 convs and
// transposed convs, and all dense layers.  Same math as tf.nn.conv2d / conv2d_backprop_input /
// conv2d_backprop_filter with SAME padding (reference call sites: mnist/ops.py:62,78;
// cifar10/common/ops/conv2d.py:181-187).
#include "common.h"
#include <type_traits>

struct ConvGeom {
  int N, H, W, Cin;      // logical conv input (post-upsample)
  int OH, OW, Cout;
  int KH, KW, S, PT, PL;
  int up;                // input tensor is stored at (H/2, W/2) and read through nearest upsample
  int relu_in;
};

static ConvGeom make_geom(const rcgan_conv_desc* d) {
  ConvGeom g;
  g.N = d->n; g.H = d->h; g.W = d->w; g.Cin = d->cin; g.Cout = d->cout;
  g.KH = d->kh; g.KW = d->kw; g.S = d->stride;
  same_pad(d->h, d->kh, d->stride, &g.OH, &g.PT);
  same_pad(d->w, d->kw, d->stride, &g.OW, &g.PL);
  g.up = (d->flags & RCGAN_CONV_IN_UPSAMPLE2X) ? 1 : 0;
  g.relu_in = (d->flags & RCGAN_CONV_IN_RELU) ? 1 : 0;
  return g;
}

// Operand functors of the gather GEMM  C[i][j] = sum_r A(i,r) * B(r,j).  Each provides
//   Row row(i)                          the row decoded once per thread (a thread fetches one row of A for the whole launch)
//   a8(row, r, r_end, raw[8], mask)     8 consecutive reduction indices of that row: raw element loads from addresses
//                                       that are always inside the tensor + a validity bit mask (zero-fill, ReLU and
//                                       conversion happen at the LDS write, after the MFMAs of the previous step)
//   bbase(r, stride)                    B(r, j) = bbase[j * stride]
//   store(i, j, v, z)
// Pixel indices are 32-bit (a tensor has < 2^32 pixels); the element offset is one 64-bit multiply-add.  When the run
// dimension (channels) is >= 8 a run of 8 touches at most two filter taps, so the tap decode, the bounds test and the
// pixel address are computed twice per run instead of eight times.
__device__ __forceinline__ unsigned run_mask(bool okA, bool okB, int nA, long r, long r_end) {
  const unsigned mA = nA >= 8 ? 0xffu : ((1u << nA) - 1u);
  unsigned m = (okA ? mA : 0u) | (okB ? (0xffu & ~mA) : 0u);
  const long left = r_end - r;
  m &= left >= 8 ? 0xffu : (left > 0 ? ((1u << (int)left) - 1u) : 0u);
  return m;
}

// ---- forward: i = output pixel, j = cout, r = (kh,kw,ci) ---------------------------------------
template <typename T> struct FwdOp {
  typedef T AT; typedef float BT;
  ConvGeom g; const T* x; const float* w; const float* bias; T* y; int accumulate;
  const float* wscale;   // optional device scalar: filter is divided by it (spectral norm sigma)
  long M, N, R, r_chunk;
  struct Row { int n, ih0, iw0, ok; };
  __device__ __forceinline__ Row row(long i) const {
    Row rw;
    rw.ok = i < M;
    const unsigned ii = rw.ok ? (unsigned)i : 0u;
    const unsigned ow = ii % (unsigned)g.OW, t = ii / (unsigned)g.OW;
    rw.n = (int)(t / (unsigned)g.OH);
    rw.ih0 = (int)(t % (unsigned)g.OH) * g.S - g.PT;
    rw.iw0 = (int)ow * g.S - g.PL;
    return rw;
  }
  __device__ __forceinline__ bool a_relu() const { return g.relu_in; }
  // element pointer of tap (kh,kw), channel offset c0 (may be negative: the second tap of a run is addressed at q >= nA)
  __device__ __forceinline__ const T* tap(const Row& rw, int kh, int kw, int c0, bool& ok) const {
    const int Hs = g.up ? (g.H >> 1) : g.H, Ws = g.up ? (g.W >> 1) : g.W;
    const int ih = rw.ih0 + kh, iw = rw.iw0 + kw;
    ok = rw.ok && kh < g.KH && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
    const int sh = g.up ? (ih >> 1) : ih, sw = g.up ? (iw >> 1) : iw;
    const unsigned pix = ((unsigned)rw.n * (unsigned)Hs + (unsigned)sh) * (unsigned)Ws + (unsigned)sw;
    return ok ? x + ((long)pix * g.Cin + c0) : x;
  }
  __device__ __forceinline__ void a8(const Row& rw, long r, long r_end, T* raw, unsigned& mask) const {
    const unsigned kk = (unsigned)r / (unsigned)g.Cin;
    int ci = (int)((unsigned)r - kk * (unsigned)g.Cin);
    int kh = (int)(kk / (unsigned)g.KW), kw = (int)(kk - (unsigned)kh * (unsigned)g.KW);
    if (g.Cin >= 8) {
      const int nA = g.Cin - ci;
      bool okA, okB;
      const T* pA = tap(rw, kh, kw, ci, okA);
      int kw2 = kw + 1, kh2 = kh;
      if (kw2 == g.KW) { kw2 = 0; ++kh2; }
      const T* pB = tap(rw, kh2, kw2, -nA, okB);
#pragma unroll
      for (int q = 0; q < 8; ++q) raw[q] = (q < nA ? pA : pB)[q];
      mask = run_mask(okA, okB, nA, r, r_end);
    } else {
      mask = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        bool ok;
        const T* p = tap(rw, kh, kw, ci, ok);
        ok = ok && r + q < r_end;
        raw[q] = *(ok ? p : x);
        mask |= (unsigned)ok << q;
        if (++ci == g.Cin) { ci = 0; if (++kw == g.KW) { kw = 0; ++kh; } }
      }
    }
  }
  __device__ __forceinline__ const float* bbase(long r, long& stride) const { stride = 1; return w + r * g.Cout; }
  __device__ __forceinline__ void store(long i, long j, float v, int) const {
    if (bias) v += bias[j];
    T* p = y + i * g.Cout + j;
    if (accumulate) v += Elem<T>::ld(p);
    Elem<T>::st(p, v);
  }
};

// ---- data gradient: i = input pixel (n,ih,iw) at the logical resolution, j = ci, r = (kh,kw,co) ---
template <typename T> struct DgradOp {
  typedef T AT; typedef float BT;
  ConvGeom g; const T* dy; const float* w; const float* bias; T* dx; const T* xmask; int accumulate;
  const float* wscale;
  long M, N, R, r_chunk;
  struct Row { int n, ih, iw, ok; };
  __device__ __forceinline__ Row row(long i) const {
    Row rw;
    rw.ok = i < M;
    const unsigned ii = rw.ok ? (unsigned)i : 0u;
    rw.iw = (int)(ii % (unsigned)g.W);
    const unsigned t = ii / (unsigned)g.W;
    rw.ih = (int)(t % (unsigned)g.H);
    rw.n = (int)(t / (unsigned)g.H);
    return rw;
  }
  __device__ __forceinline__ bool a_relu() const { return false; }
  __device__ __forceinline__ const T* tap(const Row& rw, int kh, int kw, int c0, bool& ok) const {
    const int th = rw.ih + g.PT - kh, tw = rw.iw + g.PL - kw;
    ok = rw.ok && kh < g.KH && th >= 0 && tw >= 0;
    int oh = th, ow = tw;
    if (g.S == 2) { ok = ok && !((th | tw) & 1); oh = th >> 1; ow = tw >> 1; }
    else if (g.S > 2) { ok = ok && (th % g.S == 0) && (tw % g.S == 0); oh = th / g.S; ow = tw / g.S; }
    ok = ok && oh < g.OH && ow < g.OW;
    const unsigned pix = ((unsigned)rw.n * (unsigned)g.OH + (unsigned)oh) * (unsigned)g.OW + (unsigned)ow;
    return ok ? dy + ((long)pix * g.Cout + c0) : dy;
  }
  __device__ __forceinline__ void a8(const Row& rw, long r, long r_end, T* raw, unsigned& mask) const {
    const unsigned kk = (unsigned)r / (unsigned)g.Cout;
    int co = (int)((unsigned)r - kk * (unsigned)g.Cout);
    int kh = (int)(kk / (unsigned)g.KW), kw = (int)(kk - (unsigned)kh * (unsigned)g.KW);
    if (g.Cout >= 8) {
      const int nA = g.Cout - co;
      bool okA, okB;
      const T* pA = tap(rw, kh, kw, co, okA);
      int kw2 = kw + 1, kh2 = kh;
      if (kw2 == g.KW) { kw2 = 0; ++kh2; }
      const T* pB = tap(rw, kh2, kw2, -nA, okB);
#pragma unroll
      for (int q = 0; q < 8; ++q) raw[q] = (q < nA ? pA : pB)[q];
      mask = run_mask(okA, okB, nA, r, r_end);
    } else {
      mask = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        bool ok;
        const T* p = tap(rw, kh, kw, co, ok);
        ok = ok && r + q < r_end;
        raw[q] = *(ok ? p : dy);
        mask |= (unsigned)ok << q;
        if (++co == g.Cout) { co = 0; if (++kw == g.KW) { kw = 0; ++kh; } }
      }
    }
  }
  __device__ __forceinline__ const float* bbase(long r, long& stride) const {
    const unsigned kk = (unsigned)r / (unsigned)g.Cout, co = (unsigned)r - kk * (unsigned)g.Cout;
    stride = g.Cout;
    return w + ((long)kk * g.Cin * g.Cout + co);
  }
  __device__ __forceinline__ void store(long i, long j, float v, int) const {
    if (bias) v += bias[j];          // used by the transposed-conv forward
    long off = i * g.Cin + j;
    if (xmask) { float xv = Elem<T>::ld(xmask + off); if (!(xv > 0.f)) v = 0.f; }
    T* p = dx + off;
    if (accumulate) v += Elem<T>::ld(p);
    Elem<T>::st(p, v);
  }
};

// ---- data gradient of a stride-2 conv (= the MNIST transposed convs and the backward of its 5x5 stride-2 convs),
//      one input-pixel PARITY CLASS (ih % 2, iw % 2) per launch.  For a pixel of class (ph, pw) only the taps with
//      kh = kh0 + 2*jh, kw = kw0 + 2*jw (kh0 = (ph + PT) % 2) reach an output pixel, so the reduction runs over
//      ceil(KH/2) x ceil(KW/2) x Cout instead of KH x KW x Cout with three quarters of the gathers returning zero.
//      i = (n, ih/2, iw/2) inside the class, j = ci, r = (jh, jw, co).
template <typename T> struct DgradS2Op {
  typedef T AT; typedef float BT;
  ConvGeom g; const T* dy; const float* w; const float* bias; T* dx; const T* xmask; int accumulate;
  const float* wscale;
  long M, N, R, r_chunk;
  int ph, pw, Hp, Wp, kh0, kw0, nkh, nkw, dh, dwc;
  struct Row { int n, ih2, iw2, ok; };
  __device__ __forceinline__ Row row(long i) const {
    Row rw;
    rw.ok = i < M;
    const unsigned ii = rw.ok ? (unsigned)i : 0u;
    rw.iw2 = (int)(ii % (unsigned)Wp);
    const unsigned t = ii / (unsigned)Wp;
    rw.ih2 = (int)(t % (unsigned)Hp);
    rw.n = (int)(t / (unsigned)Hp);
    return rw;
  }
  __device__ __forceinline__ bool a_relu() const { return false; }
  __device__ __forceinline__ const T* tap(const Row& rw, int jh, int jw, int c0, bool& ok) const {
    const int oh = rw.ih2 + dh - jh, ow = rw.iw2 + dwc - jw;
    ok = rw.ok && jh < nkh && oh >= 0 && oh < g.OH && ow >= 0 && ow < g.OW;
    const unsigned pix = ((unsigned)rw.n * (unsigned)g.OH + (unsigned)oh) * (unsigned)g.OW + (unsigned)ow;
    return ok ? dy + ((long)pix * g.Cout + c0) : dy;
  }
  __device__ __forceinline__ void a8(const Row& rw, long r, long r_end, T* raw, unsigned& mask) const {
    const unsigned jj = (unsigned)r / (unsigned)g.Cout;
    int co = (int)((unsigned)r - jj * (unsigned)g.Cout);
    int jh = (int)(jj / (unsigned)nkw), jw = (int)(jj - (unsigned)jh * (unsigned)nkw);
    if (g.Cout >= 8) {
      const int nA = g.Cout - co;
      bool okA, okB;
      const T* pA = tap(rw, jh, jw, co, okA);
      int jw2 = jw + 1, jh2 = jh;
      if (jw2 == nkw) { jw2 = 0; ++jh2; }
      const T* pB = tap(rw, jh2, jw2, -nA, okB);
#pragma unroll
      for (int q = 0; q < 8; ++q) raw[q] = (q < nA ? pA : pB)[q];
      mask = run_mask(okA, okB, nA, r, r_end);
    } else {
      mask = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        bool ok;
        const T* p = tap(rw, jh, jw, co, ok);
        ok = ok && r + q < r_end;
        raw[q] = *(ok ? p : dy);
        mask |= (unsigned)ok << q;
        if (++co == g.Cout) { co = 0; if (++jw == nkw) { jw = 0; ++jh; } }
      }
    }
  }
  __device__ __forceinline__ const float* bbase(long r, long& stride) const {
    const unsigned jj = (unsigned)r / (unsigned)g.Cout, co = (unsigned)r - jj * (unsigned)g.Cout;
    const int jh = (int)(jj / (unsigned)nkw), jw = (int)(jj - (unsigned)jh * (unsigned)nkw);
    const int kh = kh0 + 2 * jh, kw = kw0 + 2 * jw;
    stride = g.Cout;
    return w + ((long)(kh * g.KW + kw) * g.Cin * g.Cout + co);
  }
  __device__ __forceinline__ void store(long i, long j, float v, int) const {
    const unsigned ii = (unsigned)i;
    const int iw2 = (int)(ii % (unsigned)Wp);
    const unsigned t = ii / (unsigned)Wp;
    const int ih2 = (int)(t % (unsigned)Hp), n = (int)(t / (unsigned)Hp);
    if (bias) v += bias[j];
    const long off = (((long)n * g.H + 2 * ih2 + ph) * g.W + 2 * iw2 + pw) * g.Cin + j;
    if (xmask) { float xv = Elem<T>::ld(xmask + off); if (!(xv > 0.f)) v = 0.f; }
    T* p = dx + off;
    if (accumulate) v += Elem<T>::ld(p);
    Elem<T>::st(p, v);
  }
};

// ---- filter gradient: i = (kh,kw,ci), j = cout, r = output pixel; split over r into fp32 slabs ----
template <typename T> struct WgradOp {
  typedef T AT; typedef T BT;
  ConvGeom g; const T* x; const T* dy; float* slab;
  const float* wscale;   // always null (the filter gradient has no filter operand)
  long M, N, R, r_chunk;
  // the row is a (kh,kw,ci) filter position, the reduction walks 8 consecutive output pixels
  struct Row { int kh, kw, ci, ok; };
  __device__ __forceinline__ Row row(long i) const {
    Row rw;
    rw.ok = i < M;
    const unsigned ii = rw.ok ? (unsigned)i : 0u;
    const unsigned kk = ii / (unsigned)g.Cin;
    rw.ci = (int)(ii - kk * (unsigned)g.Cin);
    rw.kh = (int)(kk / (unsigned)g.KW);
    rw.kw = (int)(kk - (unsigned)rw.kh * (unsigned)g.KW);
    return rw;
  }
  __device__ __forceinline__ bool a_relu() const { return g.relu_in; }
  __device__ __forceinline__ void a8(const Row& rw, long r, long r_end, T* raw, unsigned& mask) const {
    const unsigned t = (unsigned)r / (unsigned)g.OW;
    int ow = (int)((unsigned)r - t * (unsigned)g.OW);
    int n = (int)(t / (unsigned)g.OH), oh = (int)(t - (unsigned)n * (unsigned)g.OH);
    const int Hs = g.up ? (g.H >> 1) : g.H, Ws = g.up ? (g.W >> 1) : g.W;
    mask = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int ih = oh * g.S + rw.kh - g.PT, iw = ow * g.S + rw.kw - g.PL;
      const bool ok = rw.ok && r + q < r_end && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
      const int sh = g.up ? (ih >> 1) : ih, sw = g.up ? (iw >> 1) : iw;
      const unsigned pix = ((unsigned)n * (unsigned)Hs + (unsigned)sh) * (unsigned)Ws + (unsigned)sw;
      raw[q] = x[ok ? (long)pix * g.Cin + rw.ci : 0];
      mask |= (unsigned)ok << q;
      if (++ow == g.OW) { ow = 0; if (++oh == g.OH) { oh = 0; ++n; } }
    }
  }
  __device__ __forceinline__ const T* bbase(long r, long& stride) const { stride = 1; return dy + r * g.Cout; }
  __device__ __forceinline__ void store(long i, long j, float v, int z) const {
    slab[(long)z * M * N + i * N + j] = v;
  }
};

// ---- fully connected layers: the same GEMM core with plain row-major operands:
//      y[m][n] = x[m][:] . w[:][n];  dx[m][k] = dy[m][:] . w[k][:];  dw[k][n] = x[:][k] . dy[:][n]
__device__ __forceinline__ unsigned tail_mask(bool ok, long r, long r_end) {
  const long left = r_end - r;
  return ok ? (left >= 8 ? 0xffu : (left > 0 ? ((1u << (int)left) - 1u) : 0u)) : 0u;
}
template <typename T> struct LinFwdOp {
  typedef T AT; typedef float BT;
  const T* x; const float* w; const float* bias; T* y; const float* wscale;
  long M, N, R, r_chunk;
  struct Row { const T* p; int ok; };
  __device__ __forceinline__ Row row(long i) const { Row rw; rw.ok = i < M; rw.p = x + (rw.ok ? i : 0) * R; return rw; }
  __device__ __forceinline__ bool a_relu() const { return false; }
  __device__ __forceinline__ void a8(const Row& rw, long r, long r_end, T* raw, unsigned& mask) const {
#pragma unroll
    for (int q = 0; q < 8; ++q) raw[q] = rw.p[r + q < r_end ? r + q : r_end - 1];
    mask = tail_mask(rw.ok, r, r_end);
  }
  __device__ __forceinline__ const float* bbase(long r, long& stride) const { stride = 1; return w + r * N; }
  __device__ __forceinline__ void store(long i, long j, float v, int) const {
    if (bias) v += bias[j];
    Elem<T>::st(y + i * N + j, v);
  }
};
template <typename T> struct LinDgradOp {
  typedef T AT; typedef float BT;
  const T* dy; const float* w; T* dx; int accumulate; const float* wscale;
  long M, N, R, r_chunk;          // N = in features, R = out features
  struct Row { const T* p; int ok; };
  __device__ __forceinline__ Row row(long i) const { Row rw; rw.ok = i < M; rw.p = dy + (rw.ok ? i : 0) * R; return rw; }
  __device__ __forceinline__ bool a_relu() const { return false; }
  __device__ __forceinline__ void a8(const Row& rw, long r, long r_end, T* raw, unsigned& mask) const {
#pragma unroll
    for (int q = 0; q < 8; ++q) raw[q] = rw.p[r + q < r_end ? r + q : r_end - 1];
    mask = tail_mask(rw.ok, r, r_end);
  }
  __device__ __forceinline__ const float* bbase(long r, long& stride) const { stride = R; return w + r; }
  __device__ __forceinline__ void store(long i, long j, float v, int) const {
    T* p = dx + i * N + j;
    if (accumulate) v += Elem<T>::ld(p);
    Elem<T>::st(p, v);
  }
};
template <typename T> struct LinWgradOp {
  typedef T AT; typedef T BT;
  const T* x; const T* dy; float* out; int accumulate; int direct; const float* wscale;
  long M, N, R, r_chunk;          // M = in features, N = out features, R = batch rows
  struct Row { const T* p; int ok; };
  __device__ __forceinline__ Row row(long i) const { Row rw; rw.ok = i < M; rw.p = x + (rw.ok ? i : 0); return rw; }
  __device__ __forceinline__ bool a_relu() const { return false; }
  __device__ __forceinline__ void a8(const Row& rw, long r, long r_end, T* raw, unsigned& mask) const {
#pragma unroll
    for (int q = 0; q < 8; ++q) raw[q] = rw.p[(r + q < r_end ? r + q : r_end - 1) * M];
    mask = tail_mask(rw.ok, r, r_end);
  }
  __device__ __forceinline__ const T* bbase(long r, long& stride) const { stride = 1; return dy + r * N; }
  __device__ __forceinline__ void store(long i, long j, float v, int z) const {
    if (direct) {                  // single r-chunk: straight into the gradient
      float* p = out + i * N + j;
      *p = accumulate ? *p + v : v;
    } else {
      out[(long)z * M * N + i * N + j] = v;
    }
  }
};

// 64 x 64 output tile, K-step 32, fp32 matrix cores (v_mfma_f32_32x32x2_f32: IEEE fp32 products and sums, only the
// summation order differs from a scalar loop).  Four waves, one 32x32 quadrant each: per K-step a wave reads its
// operands with 4 ds_read_b128 + 16 ds_read_b32 and issues 16 MFMAs -- 8 KB of LDS traffic per wave-step where the
// FMA formulation moved 64 KB and stalled on every LDS->FMA dependency at the one-wave-per-SIMD occupancy of the
// few-workgroup dense layers.  The K-slot a lane feeds to MFMA s is 16*(lane>>5)+s, so the A operand of all 16 steps is
// one contiguous 64-byte run per lane.  The operand elements of step s+1 are fetched into registers before the MFMAs of
// step s and written to the other LDS buffer after them: one barrier per step.
// KS groups of four waves walk interleaved K-steps (group g takes steps g, g+KS, ...) on private LDS buffers and their
// accumulators are summed through LDS at the end: the dense layers of the MNIST nets launch 16..400 workgroups with
// 50..200 sequential K-steps each, and a step is latency (address decode -> loads -> LDS -> MFMA chain), not throughput.
template <class Op, int KS>
__global__ __launch_bounds__(256 * KS) void gemm_gather_kernel(Op op) {
  extern __shared__ __attribute__((aligned(16))) float gg_smem[];
  const int kg = threadIdx.x >> 8;
  typedef float AsT[2][64][36];   // [buf][row][k]
  typedef float BsT[2][32][68];   // [buf][k][col]
  AsT& As = *(AsT*)(gg_smem + (size_t)kg * (2 * 64 * 36 + 2 * 32 * 68));
  BsT& Bs = *(BsT*)(gg_smem + (size_t)kg * (2 * 64 * 36 + 2 * 32 * 68) + 2 * 64 * 36);
  const int tid = threadIdx.x & 255;
  const int lane = tid & 63, wv = tid >> 6, wr = wv >> 1, wc = wv & 1, l31 = lane & 31, hh = lane >> 5;
  const long i0 = (long)blockIdx.y * 64, j0 = (long)blockIdx.x * 64;
  const long r_begin = (long)blockIdx.z * op.r_chunk;
  long r_end = r_begin + op.r_chunk;
  if (r_end > op.R) r_end = op.R;
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 acc;
#pragma unroll
  for (int p = 0; p < 16; ++p) acc[p] = 0.f;

  // spectral-norm division W / sigma: sigma is loaded once per thread, not once per operand element
  const float bscale = op.wscale ? 1.f / *op.wscale : 1.f;
  const long ai = i0 + (tid >> 2);
  const int ar = (tid & 3) * 8;
  const int br = tid >> 4;
  const long bj = j0 + (tid & 15) * 4;
  // Operand fetch is branch-free (addresses always inside the tensors, validity kept in bit masks) and the zero-fill /
  // ReLU / sigma scale are applied at the LDS write: every load of a step is in flight together and nothing waits on
  // them until after the MFMAs (a select or max right behind each guarded load serialised the load latencies).
  typename Op::AT ra[8];
  typename Op::BT rb[8];
  unsigned amask = 0, bmask = 0;
  const auto arow = op.row(ai);
  auto fetch = [&](long r0) {
    op.a8(arow, r0 + ar, r_end, ra, amask);
    bmask = 0;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const long r = r0 + br + 16 * h2;
      long st;
      const typename Op::BT* pb = op.bbase(r < r_end ? r : r_end - 1, st);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const long j = bj + c;
        rb[h2 * 4 + c] = pb[(j < op.N ? j : op.N - 1) * st];
        bmask |= (unsigned)(r < r_end && j < op.N) << (h2 * 4 + c);
      }
    }
  };
  auto stash = [&](int buf) {
    float fa[8], fb[8];
    const bool relu = op.a_relu();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = Elem<typename Op::AT>::ld(&ra[q]);
      if (relu) v = v > 0.f ? v : 0.f;
      fa[q] = ((amask >> q) & 1u) ? v : 0.f;
      fb[q] = ((bmask >> q) & 1u) ? Elem<typename Op::BT>::ld(&rb[q]) * bscale : 0.f;
    }
    *(float4*)&As[buf][tid >> 2][ar] = make_float4(fa[0], fa[1], fa[2], fa[3]);
    *(float4*)&As[buf][tid >> 2][ar + 4] = make_float4(fa[4], fa[5], fa[6], fa[7]);
    *(float4*)&Bs[buf][br][(tid & 15) * 4] = make_float4(fb[0], fb[1], fb[2], fb[3]);
    *(float4*)&Bs[buf][br + 16][(tid & 15) * 4] = make_float4(fb[4], fb[5], fb[6], fb[7]);
  };
  const long nsteps = (r_end - r_begin + 31) / 32;
  const long T = (nsteps + KS - 1) / KS;          // per-group steps (same for every group: the barriers are block-wide)
  if (T > 0) { fetch(r_begin + 32 * kg); stash(0); }
  __syncthreads();
  int buf = 0;
  for (long t = 0; t < T; ++t) {
    const bool more = t + 1 < T;
    if (more) fetch(r_begin + 32 * (kg + KS * (t + 1)));
    float av[16], bv[16];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float4 t4 = *(const float4*)&As[buf][wr * 32 + l31][16 * hh + 4 * v];
      av[4 * v] = t4.x; av[4 * v + 1] = t4.y; av[4 * v + 2] = t4.z; av[4 * v + 3] = t4.w;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) bv[q] = Bs[buf][16 * hh + q][wc * 32 + l31];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
    if (more) stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if constexpr (KS > 1) {                          // sum the groups' accumulators: [group-1][wave][reg][lane]
    float* red = gg_smem;
    if (kg > 0) {
#pragma unroll
      for (int p = 0; p < 16; ++p) red[(((kg - 1) * 4 + wv) * 16 + p) * 64 + lane] = acc[p];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int g2 = 0; g2 < KS - 1; ++g2)
#pragma unroll
      for (int p = 0; p < 16; ++p) acc[p] += red[((g2 * 4 + wv) * 16 + p) * 64 + lane];
  }
  const long j = j0 + wc * 32 + l31;
  if (j < op.N) {
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const long i = i0 + wr * 32 + 8 * (p >> 2) + 4 * hh + (p & 3);
      if (i < op.M) op.store(i, j, acc[p], blockIdx.z);
    }
  }
}

// out[i] (= or +=) sum_z slab[z][i]
__global__ void slab_reduce_kernel(const float* slab, float* out, long count, int nz, int accumulate) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int z = 0; z < nz; ++z) s += slab[(long)z * count + i];
  if (accumulate) s += out[i];
  out[i] = s;
}

// column sums of a [rows][c] matrix, one block per 64 columns; deterministic.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, long rows, int c, float* out, int accumulate) {
  __shared__ float red[4][64];
  int col = blockIdx.x * 64 + (threadIdx.x & 63);
  int lane_r = threadIdx.x >> 6;
  float s = 0.f;
  if (col < c)
    for (long r = lane_r; r < rows; r += 4) s += Elem<T>::ld(x + r * c + col);
  red[lane_r][threadIdx.x & 63] = s;
  __syncthreads();
  if (threadIdx.x < 64 && col < c) {
    float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (accumulate) t += out[col];
    out[col] = t;
  }
}

// two-level column sum for tall matrices: partial[blk][c] then reduce
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* x, long rows, int c, long rows_per_blk, float* partial) {
  __shared__ float red[4][64];
  int col = blockIdx.x * 64 + (threadIdx.x & 63);
  int lane_r = threadIdx.x >> 6;
  long rb = (long)blockIdx.y * rows_per_blk;
  long re = rb + rows_per_blk;
  if (re > rows) re = rows;
  float s = 0.f;
  if (col < c)
    for (long r = rb + lane_r; r < re; r += 4) s += Elem<T>::ld(x + r * c + col);
  red[lane_r][threadIdx.x & 63] = s;
  __syncthreads();
  if (threadIdx.x < 64 && col < c)
    partial[(long)blockIdx.y * c + col] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// narrow matrices (c <= 8, e.g. the single-channel image bias of the MNIST generator): every thread walks whole rows,
// the 256 row lanes are reduced through LDS
template <typename T>
__global__ __launch_bounds__(256) void colsum_narrow_partial_kernel(const T* x, long rows, int c, long rows_per_blk, float* partial) {
  __shared__ float red[4];
  const long rb = (long)blockIdx.x * rows_per_blk;
  long re = rb + rows_per_blk;
  if (re > rows) re = rows;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long r = rb + threadIdx.x; r < re; r += 256)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < c) s[j] += Elem<T>::ld(x + r * c + j);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j >= c) break;
    const float v = block_sum256(s[j], red);
    if (threadIdx.x == 0) partial[(long)blockIdx.x * c + j] = v;
    __syncthreads();
  }
}

// vectorised partial column sums: each thread owns 8 consecutive columns (one 16-B bf16 / two 16-B fp32
// loads per row), row lanes reduced through LDS.  Requires c % 8 == 0 and (c/8) | 256.
__device__ __forceinline__ void load8(const float* p, float* v) {
  float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float* v) {
  uint4 a = *(const uint4*)p;
  uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[2 * j] = bf16_to_f32((bf16_t)(w[j] & 0xffff)); v[2 * j + 1] = bf16_to_f32((bf16_t)(w[j] >> 16)); }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_partial_kernel(const T* x, long rows, int c, long rows_per_blk, float* partial) {
  __shared__ float red[2048];
  const int chunks = c / 8, nrl = 256 / chunks;
  const int ch = threadIdx.x % chunks, rl = threadIdx.x / chunks;
  const long rb = (long)blockIdx.x * rows_per_blk;
  long re = rb + rows_per_blk;
  if (re > rows) re = rows;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (long r = rb + rl; r < re; r += nrl) {
    float v[8];
    load8(x + r * c + ch * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl * c + ch * 8 + j] = acc[j];
  __syncthreads();
  for (int col = threadIdx.x; col < c; col += 256) {
    float s = 0.f;
    for (int q = 0; q < nrl; ++q) s += red[q * c + col];
    partial[(long)blockIdx.x * c + col] = s;
  }
}

static int gg_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

template <class Op, int KS>
static int launch_gemm_ks(rcgan_ctx* ctx, Op& op, dim3 grid) {
  static bool attr_set = false;
  const size_t lds = (size_t)KS * (2 * 64 * 36 + 2 * 32 * 68) * sizeof(float);
  if (!attr_set) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)gemm_gather_kernel<Op, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_gather_kernel<Op, KS>), grid, dim3(256 * KS), lds, ctx->stream, op);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

template <class Op>
static int launch_gemm(rcgan_ctx* ctx, Op& op, int nz) {
  dim3 grid(cdiv(op.N, 64), cdiv(op.M, 64), nz);
  if (grid.y > 65535u || grid.z > 65535u) RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "M too large");
  static const int ks_min_steps = gg_env_int("RCGAN_GG_KS_MINSTEPS", 8);
  const long steps = (op.r_chunk + 31) / 32;
  if (steps >= ks_min_steps) return launch_gemm_ks<Op, 4>(ctx, op, grid);
  return launch_gemm_ks<Op, 1>(ctx, op, grid);
}

// pick the number of r-splits for a filter-gradient GEMM so the grid fills the chip
static int wgrad_splits(long K, long Cout, long M) {
  long tiles = (long)cdiv(K, 64) * cdiv(Cout, 64);
  long want = (1024 + tiles - 1) / tiles;
  long maxs = (M + 255) / 256;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  return (int)want;
}

size_t direct_wgrad_ws_bytes(const rcgan_conv_desc* d) {
  ConvGeom g = make_geom(d);
  long K = (long)g.KH * g.KW * g.Cin, M = (long)g.N * g.OH * g.OW;
  int nz = wgrad_splits(K, g.Cout, M);
  size_t bias_part = (size_t)(cdiv(M, 2048) + 1024) * g.Cout * sizeof(float);
  return (size_t)nz * K * g.Cout * sizeof(float) + bias_part + 256;
}

template <typename T>
int colsum_launch(rcgan_ctx* ctx, const T* x, long rows, int c, float* out, int accumulate, float* partial_ws) {
  if (partial_ws != nullptr && rows >= 2048 && c % 8 == 0 && c <= 256 && 256 % (c / 8) == 0) {
    long rpb = 256;
    while ((rows + rpb - 1) / rpb > 1024) rpb *= 2;
    int nb = cdiv(rows, rpb);          // <= 1024 <= the cdiv(rows, 2048)*... slots? see colsum_ws_rows()
    hipLaunchKernelGGL(colsum_vec_partial_kernel<T>, dim3(nb), dim3(256), 0, ctx->stream, x, rows, c, rpb, partial_ws);
    RC_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(colsum_kernel<float>, dim3(cdiv(c, 64)), dim3(256), 0, ctx->stream, (const float*)partial_ws, (long)nb, c, out, accumulate);
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  if (partial_ws != nullptr && rows >= 4096 && c <= 8) {
    long rpb = 1024;
    while ((rows + rpb - 1) / rpb > 1024) rpb *= 2;
    int nb = cdiv(rows, rpb);
    hipLaunchKernelGGL(colsum_narrow_partial_kernel<T>, dim3(nb), dim3(256), 0, ctx->stream, x, rows, c, rpb, partial_ws);
    RC_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(colsum_kernel<float>, dim3(cdiv(c, 64)), dim3(256), 0, ctx->stream, (const float*)partial_ws, (long)nb, c, out, accumulate);
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  if (rows <= 4096 || partial_ws == nullptr) {
    hipLaunchKernelGGL(colsum_kernel<T>, dim3(cdiv(c, 64)), dim3(256), 0, ctx->stream, x, rows, c, out, accumulate);
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  int nb = cdiv(rows, 2048);
  hipLaunchKernelGGL(colsum_partial_kernel<T>, dim3(cdiv(c, 64), nb), dim3(256), 0, ctx->stream, x, rows, c, (long)2048, partial_ws);
  RC_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(colsum_kernel<float>, dim3(cdiv(c, 64)), dim3(256), 0, ctx->stream, (const float*)partial_ws, (long)nb, c, out, accumulate);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}
template int colsum_launch<float>(rcgan_ctx*, const float*, long, int, float*, int, float*);
template int colsum_launch<bf16_t>(rcgan_ctx*, const bf16_t*, long, int, float*, int, float*);

// ---- tiny dense layers (a few thousand outputs): one thread per output element.  The 64x64-tile GEMM would run
//      them in 1-4 workgroups, i.e. on 1-4 of the 256 CUs, for the whole reduction; here the outputs spread over
//      the chip and each thread walks the reduction with independent, cache-resident loads.
//      MODE 0: y[m][n] = x[m][:] . w[:][n] (/sigma) + b[n]      MODE 1: dx[m][k] (+)= dy[m][:] . w[k][:] (/sigma)
//      MODE 2: dw[k][n] (+)= x[:][k] . dy[:][n]
template <typename T, int MODE>
__global__ __launch_bounds__(256) void linear_tiny_kernel(int m, int k, int n, const T* a, const void* bptr, const float* wscale,
                                                          const float* bias, void* out, int accumulate) {
  // workgroup = 64 outputs x 4 slices of the reduction (summed through LDS in a fixed order): four times shorter
  // dependent-load chains than one thread per output
  __shared__ float part[4][64];
  const int rows = MODE == 2 ? k : m, cols = MODE == 1 ? k : n;
  const int red = MODE == 0 ? k : (MODE == 1 ? n : m);
  const int idx = blockIdx.x * 64 + (threadIdx.x & 63), kp = threadIdx.x >> 6;
  const bool live = idx < rows * cols;
  const int r = live ? idx / cols : 0, c = live ? idx - r * cols : 0;
  const int chunk = (red + 3) / 4;
  const int q0 = kp * chunk, q1 = min(red, q0 + chunk);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (live) {
    int q = q0;
    if (MODE == 0) {
      const T* x = a + (long)r * k;
      const float* w = (const float*)bptr + c;
      for (; q + 4 <= q1; q += 4) {
        s0 = fmaf(Elem<T>::ld(x + q), w[(long)q * n], s0);
        s1 = fmaf(Elem<T>::ld(x + q + 1), w[(long)(q + 1) * n], s1);
        s2 = fmaf(Elem<T>::ld(x + q + 2), w[(long)(q + 2) * n], s2);
        s3 = fmaf(Elem<T>::ld(x + q + 3), w[(long)(q + 3) * n], s3);
      }
      for (; q < q1; ++q) s0 = fmaf(Elem<T>::ld(x + q), w[(long)q * n], s0);
    } else if (MODE == 1) {
      const T* dy = a + (long)r * n;
      const float* w = (const float*)bptr + (long)c * n;
      for (; q + 4 <= q1; q += 4) {
        s0 = fmaf(Elem<T>::ld(dy + q), w[q], s0);
        s1 = fmaf(Elem<T>::ld(dy + q + 1), w[q + 1], s1);
        s2 = fmaf(Elem<T>::ld(dy + q + 2), w[q + 2], s2);
        s3 = fmaf(Elem<T>::ld(dy + q + 3), w[q + 3], s3);
      }
      for (; q < q1; ++q) s0 = fmaf(Elem<T>::ld(dy + q), w[q], s0);
    } else {
      const T* x = a + r;
      const T* dy = (const T*)bptr + c;
      for (; q + 4 <= q1; q += 4) {
        s0 = fmaf(Elem<T>::ld(x + (long)q * k), Elem<T>::ld(dy + (long)q * n), s0);
        s1 = fmaf(Elem<T>::ld(x + (long)(q + 1) * k), Elem<T>::ld(dy + (long)(q + 1) * n), s1);
        s2 = fmaf(Elem<T>::ld(x + (long)(q + 2) * k), Elem<T>::ld(dy + (long)(q + 2) * n), s2);
        s3 = fmaf(Elem<T>::ld(x + (long)(q + 3) * k), Elem<T>::ld(dy + (long)(q + 3) * n), s3);
      }
      for (; q < q1; ++q) s0 = fmaf(Elem<T>::ld(x + (long)q * k), Elem<T>::ld(dy + (long)q * n), s0);
    }
  }
  part[kp][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kp != 0 || !live) return;
  const int t = threadIdx.x;
  float v = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
  if (MODE != 2 && wscale) v /= *wscale;
  if (MODE == 0) {
    if (bias) v += bias[c];
    Elem<T>::st((T*)out + idx, v);
  } else if (MODE == 1) {
    T* p = (T*)out + idx;
    if (accumulate) v += Elem<T>::ld(p);
    Elem<T>::st(p, v);
  } else {
    float* p = (float*)out + idx;
    *p = accumulate ? *p + v : v;
  }
}

static inline bool linear_tiny(long outputs, long red) { return outputs <= 65536 && red <= 4096; }

// y[row][n <= 16]: the GEMM tiling would put the whole reduction in one workgroup (one 64-column tile); here one
// workgroup per row splits K over its 256 threads (the RCGAN permutation classifier: [B,3072] x [3072,10]).
template <typename T>
__global__ __launch_bounds__(256) void linear_skinny_fwd_kernel(int k, int n, const T* x, const float* w, const float* wscale,
                                                                const float* bias, T* y) {
  __shared__ float red[4][16];
  const int row = blockIdx.x, t = threadIdx.x;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  const T* xr = x + (long)row * k;
  for (int kk = t; kk < k; kk += 256) {
    const float xv = Elem<T>::ld(xr + kk);
    const float* wr = w + (long)kk * n;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < n) acc[j] = fmaf(xv, wr[j], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float v = wave_sum(acc[j]);
    if ((t & 63) == 0) red[t >> 6][j] = v;
  }
  __syncthreads();
  if (t < n) {
    float v = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    if (wscale) v /= *wscale;
    if (bias) v += bias[t];
    Elem<T>::st(y + (long)row * n + t, v);
  }
}

template <typename T>
int linear_fwd(rcgan_ctx* ctx, long m, long k, long n, const T* x, const float* w, const float* wscale, const float* bias, T* y) {
  if (n <= 16 && k >= 512 && m <= 65535) {
    hipLaunchKernelGGL(linear_skinny_fwd_kernel<T>, dim3((int)m), dim3(256), 0, ctx->stream, (int)k, (int)n, x, w, wscale, bias, y);
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  if (linear_tiny(m * n, k)) {
    hipLaunchKernelGGL((linear_tiny_kernel<T, 0>), dim3(cdiv(m * n, 64)), dim3(256), 0, ctx->stream, (int)m, (int)k, (int)n, x, (const void*)w,
                       wscale, bias, (void*)y, 0);
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  LinFwdOp<T> op;
  op.x = x; op.w = w; op.bias = bias; op.y = y; op.wscale = wscale;
  op.M = m; op.N = n; op.R = k; op.r_chunk = k;
  return launch_gemm(ctx, op, 1);
}
template int linear_fwd<float>(rcgan_ctx*, long, long, long, const float*, const float*, const float*, const float*, float*);
template int linear_fwd<bf16_t>(rcgan_ctx*, long, long, long, const bf16_t*, const float*, const float*, const float*, bf16_t*);

template <typename T>
int linear_dgrad(rcgan_ctx* ctx, long m, long k, long n, const T* dy, const float* w, const float* wscale, T* dx, int accumulate) {
  if (linear_tiny(m * k, n)) {
    hipLaunchKernelGGL((linear_tiny_kernel<T, 1>), dim3(cdiv(m * k, 64)), dim3(256), 0, ctx->stream, (int)m, (int)k, (int)n, dy, (const void*)w,
                       wscale, (const float*)nullptr, (void*)dx, accumulate);
    RC_LAUNCH_CHECK(ctx);
    return RCGAN_OK;
  }
  LinDgradOp<T> op;
  op.dy = dy; op.w = w; op.dx = dx; op.accumulate = accumulate; op.wscale = wscale;
  op.M = m; op.N = k; op.R = n; op.r_chunk = n;
  return launch_gemm(ctx, op, 1);
}
template int linear_dgrad<float>(rcgan_ctx*, long, long, long, const float*, const float*, const float*, float*, int);
template int linear_dgrad<bf16_t>(rcgan_ctx*, long, long, long, const bf16_t*, const float*, const float*, bf16_t*, int);

size_t linear_wgrad_ws_bytes(long m, long k, long n) {
  int nz = m <= 1024 ? 1 : wgrad_splits(k, n, m);
  return (size_t)(nz > 1 ? nz : 0) * k * n * sizeof(float) + (size_t)(cdiv(m, 2048) + 1024) * n * sizeof(float) + 256;
}

template <typename T>
int linear_wgrad(rcgan_ctx* ctx, long m, long k, long n, const T* x, const T* dy, float* dw, float* dbias, int accumulate,
                 void* ws, size_t ws_bytes) {
  if (linear_tiny(k * n, m)) {
    if (ws_bytes < linear_wgrad_ws_bytes(m, k, n)) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", linear_wgrad_ws_bytes(m, k, n), ws_bytes);
    hipLaunchKernelGGL((linear_tiny_kernel<T, 2>), dim3(cdiv(k * n, 64)), dim3(256), 0, ctx->stream, (int)m, (int)k, (int)n, x, (const void*)dy,
                       (const float*)nullptr, (const float*)nullptr, (void*)dw, accumulate);
    RC_LAUNCH_CHECK(ctx);
    if (dbias) return colsum_launch<T>(ctx, dy, m, (int)n, dbias, accumulate, (float*)ws);
    return RCGAN_OK;
  }
  LinWgradOp<T> op;
  op.x = x; op.dy = dy; op.wscale = nullptr; op.accumulate = accumulate;
  op.M = k; op.N = n; op.R = m;
  int nz = m <= 1024 ? 1 : wgrad_splits(k, n, m);
  size_t slab_bytes = (size_t)(nz > 1 ? nz : 0) * k * n * sizeof(float);
  if (ws_bytes < linear_wgrad_ws_bytes(m, k, n)) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", linear_wgrad_ws_bytes(m, k, n), ws_bytes);
  if (nz == 1) {
    op.direct = 1; op.out = dw; op.r_chunk = m;
    int rc = launch_gemm(ctx, op, 1);
    if (rc) return rc;
  } else {
    op.direct = 0; op.out = (float*)ws;
    op.r_chunk = ((m + nz - 1) / nz + 15) / 16 * 16;
    nz = cdiv(m, op.r_chunk);
    int rc = launch_gemm(ctx, op, nz);
    if (rc) return rc;
    long cnt = k * n;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(cnt, 256)), dim3(256), 0, ctx->stream, (const float*)op.out, dw, cnt, nz, accumulate);
    RC_LAUNCH_CHECK(ctx);
  }
  if (dbias) {
    float* part = (float*)((char*)ws + slab_bytes);
    return colsum_launch<T>(ctx, dy, m, (int)n, dbias, accumulate, part);
  }
  return RCGAN_OK;
}
template int linear_wgrad<float>(rcgan_ctx*, long, long, long, const float*, const float*, float*, float*, int, void*, size_t);
template int linear_wgrad<bf16_t>(rcgan_ctx*, long, long, long, const bf16_t*, const bf16_t*, float*, float*, int, void*, size_t);

template <typename T>
int direct_fwd(rcgan_ctx* ctx, const rcgan_conv_desc* d, const T* x, const float* w, const float* wscale, const float* bias, T* y) {
  FwdOp<T> op;
  op.g = make_geom(d); op.x = x; op.w = w; op.wscale = wscale; op.bias = bias; op.y = y;
  op.accumulate = (d->flags & RCGAN_CONV_ACCUMULATE) ? 1 : 0;
  op.M = (long)op.g.N * op.g.OH * op.g.OW; op.N = op.g.Cout; op.R = (long)op.g.KH * op.g.KW * op.g.Cin; op.r_chunk = op.R;
  return launch_gemm(ctx, op, 1);
}
template int direct_fwd<float>(rcgan_ctx*, const rcgan_conv_desc*, const float*, const float*, const float*, const float*, float*);
template int direct_fwd<bf16_t>(rcgan_ctx*, const rcgan_conv_desc*, const bf16_t*, const float*, const float*, const float*, bf16_t*);

// dgrad at the LOGICAL input resolution (no upsample folding here; the caller pools afterwards).
template <typename T>
int direct_dgrad(rcgan_ctx* ctx, const rcgan_conv_desc* d, const T* dy, const float* w, const float* wscale, const float* bias,
                 const T* xmask, T* dx, int accumulate) {
  {
    ConvGeom g = make_geom(d);
    if (g.S == 2) {                      // one launch per input-pixel parity class
      for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
          DgradS2Op<T> op;
          op.g = g; op.g.up = 0; op.dy = dy; op.w = w; op.wscale = wscale; op.bias = bias; op.dx = dx; op.xmask = xmask; op.accumulate = accumulate;
          op.ph = ph; op.pw = pw; op.Hp = (g.H - ph + 1) / 2; op.Wp = (g.W - pw + 1) / 2;
          if (op.Hp <= 0 || op.Wp <= 0) continue;
          op.kh0 = (ph + g.PT) % 2; op.kw0 = (pw + g.PL) % 2;
          op.nkh = g.KH > op.kh0 ? (g.KH - op.kh0 + 1) / 2 : 0;
          op.nkw = g.KW > op.kw0 ? (g.KW - op.kw0 + 1) / 2 : 0;
          op.dh = (ph + g.PT - op.kh0) / 2; op.dwc = (pw + g.PL - op.kw0) / 2;
          if (op.nkw == 0) { op.nkw = 1; op.nkh = 0; }          // keeps the divisions defined; R = 0: outputs are bias / 0
          op.M = (long)g.N * op.Hp * op.Wp; op.N = g.Cin; op.R = (long)op.nkh * op.nkw * g.Cout; op.r_chunk = op.R;
          int rc = launch_gemm(ctx, op, 1);
          if (rc) return rc;
        }
      return RCGAN_OK;
    }
  }
  DgradOp<T> op;
  op.g = make_geom(d); op.g.up = 0; op.dy = dy; op.w = w; op.wscale = wscale; op.bias = bias; op.dx = dx; op.xmask = xmask; op.accumulate = accumulate;
  op.M = (long)op.g.N * op.g.H * op.g.W; op.N = op.g.Cin; op.R = (long)op.g.KH * op.g.KW * op.g.Cout; op.r_chunk = op.R;
  return launch_gemm(ctx, op, 1);
}
template int direct_dgrad<float>(rcgan_ctx*, const rcgan_conv_desc*, const float*, const float*, const float*, const float*, const float*, float*, int);
template int direct_dgrad<bf16_t>(rcgan_ctx*, const rcgan_conv_desc*, const bf16_t*, const float*, const float*, const float*, const bf16_t*, bf16_t*, int);

template <typename T>
int direct_wgrad(rcgan_ctx* ctx, const rcgan_conv_desc* d, const T* x, const T* dy, float* dw, float* dbias,
                 int accumulate, void* ws, size_t ws_bytes) {
  WgradOp<T> op;
  op.g = make_geom(d); op.x = x; op.dy = dy; op.wscale = nullptr;
  long K = (long)op.g.KH * op.g.KW * op.g.Cin, M = (long)op.g.N * op.g.OH * op.g.OW;
  int nz = wgrad_splits(K, op.g.Cout, M);
  size_t need = direct_wgrad_ws_bytes(d);
  if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
  op.slab = (float*)ws;
  op.M = K; op.N = op.g.Cout; op.R = M;
  op.r_chunk = ((M + nz - 1) / nz + 15) / 16 * 16;
  nz = cdiv(M, op.r_chunk);
  int rc = launch_gemm(ctx, op, nz);
  if (rc) return rc;
  long cnt = K * op.g.Cout;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(cnt, 256)), dim3(256), 0, ctx->stream, (const float*)op.slab, dw, cnt, nz, accumulate);
  RC_LAUNCH_CHECK(ctx);
  if (dbias) {
    float* part = (float*)((char*)ws + (size_t)wgrad_splits(K, op.g.Cout, M) * K * op.g.Cout * sizeof(float));
    rc = colsum_launch<T>(ctx, dy, M, op.g.Cout, dbias, accumulate, part);
    if (rc) return rc;
  }
  return RCGAN_OK;
}
template int direct_wgrad<float>(rcgan_ctx*, const rcgan_conv_desc*, const float*, const float*, float*, float*, int, void*, size_t);
template int direct_wgrad<bf16_t>(rcgan_ctx*, const rcgan_conv_desc*, const bf16_t*, const bf16_t*, float*, float*, int, void*, size_t);
