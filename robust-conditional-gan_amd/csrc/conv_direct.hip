// Generic fp32 implicit-GEMM convolution ("direct" path): a gather GEMM on the fp32 matrix cores.
//
// One LDS-tiled 64x64x32 GEMM skeleton, C[i][j] = sum_r A(i,r) * B(r,j), on v_mfma_f32_32x32x2_f32, with the operand
// fetch expressed as functors.  It serves every shape the 16-bit MFMA path does not take: fp32 mode (MNIST config 2 and
// the fp32 parity runs), the MNIST 5x5 stride-2 convs and transposed convs, and all dense layers.  Same math as
// tf.nn.conv2d / conv2d_backprop_input / conv2d_backprop_filter with SAME padding (reference call sites:
// mnist/ops.py:62,78; cifar10/common/ops/conv2d.py:181-187).
#include <algorithm>

#include "common.h"
#include <type_traits>

// Division by a launch-invariant 32-bit divisor as multiply-high + shifts (Granlund-Montgomery, branch-free form):
// the operand decode of every K-step divides by the channel count and the filter width, ~30 VALU ops each when the
// compiler has to expand a general division.
struct FastDiv {
  unsigned d, m, s;
  __host__ __device__ __forceinline__ unsigned div(unsigned n) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned t = __umulhi(n, m);
#else
    const unsigned t = (unsigned)(((unsigned long long)n * m) >> 32);
#endif
    const unsigned q = (t + ((n - t) >> 1)) >> s;
    return d == 1 ? n : q;
  }
};
static FastDiv make_fastdiv(unsigned d) {
  FastDiv f; f.d = d; f.m = 0; f.s = 0;
  if (d <= 1) return f;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;                       // ceil(log2 d)
  f.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s = l - 1;
  return f;
}

struct ConvGeom {
  int N, H, W, Cin;      // logical conv input (post-upsample)
  int OH, OW, Cout;
  int KH, KW, S, PT, PL;
  int up;                // input tensor is stored at (H/2, W/2) and read through nearest upsample
  int relu_in;
  FastDiv dCin, dCout, dKW, dOW, dOH;
};

static ConvGeom make_geom(const rcgan_conv_desc* d) {
  ConvGeom g;
  g.N = d->n; g.H = d->h; g.W = d->w; g.Cin = d->cin; g.Cout = d->cout;
  g.KH = d->kh; g.KW = d->kw; g.S = d->stride;
  same_pad(d->h, d->kh, d->stride, &g.OH, &g.PT);
  same_pad(d->w, d->kw, d->stride, &g.OW, &g.PL);
  g.up = (d->flags & RCGAN_CONV_IN_UPSAMPLE2X) ? 1 : 0;
  g.relu_in = (d->flags & RCGAN_CONV_IN_RELU) ? 1 : 0;
  g.dCin = make_fastdiv(g.Cin); g.dCout = make_fastdiv(g.Cout); g.dKW = make_fastdiv(g.KW);
  g.dOW = make_fastdiv(g.OW); g.dOH = make_fastdiv(g.OH);
  return g;
}

// Operand functors of the gather GEMM  C[i][j] = sum_r A(i,r) * B(r,j).  An operand is fetched along the dimension
// it is contiguous in, in 8- or 16-byte vectors where the channel count and the base pointer allow:
//   K-major (X_KMAJOR = true):  a thread owns one row (col) of the tile and fetches runs of 8 consecutive reduction
//       indices.  row(i) / brow(j) decode the row once per launch; a8 / b8 produce the raw elements.
//       LDS layout [row][k], the MFMA operand is read with ds_read_b128.
//   N-major (X_KMAJOR = false): a thread fetches 4 consecutive rows (cols) at one reduction index (two per K-step).
//       arun(i) decodes the 4-row group once; a4 / b4 produce the raw elements.  LDS layout [k][row], operand read
//       with ds_read_b32.
// Loads are branch-free and mask-free: an invalid element (padding tap, row / column / reduction index past the end)
// is read from a page of zeros, so the LDS write after the MFMAs of the previous step is a plain fp32 conversion (+ReLU).
// Pixel indices are 32-bit (a tensor has < 2^32 pixels); the element offset is one 64-bit multiply-add.  When the run
// dimension (channels) is >= 8 a run of 8 touches at most two filter taps, so the tap decode, the bounds test and the
// pixel address are computed twice per run instead of eight times.
struct NoAux {};
// real vector types: a struct of four scalars is split into four scalar loads by SROA and never re-vectorised
template <typename T> struct VecOf { typedef T v4 __attribute__((ext_vector_type(4))); typedef T v2 __attribute__((ext_vector_type(2))); };
template <typename T> struct Vec4 {
  typename VecOf<T>::v4 v;
  __device__ __forceinline__ static Vec4 load(const T* p) { Vec4 r; r.v = *(const typename VecOf<T>::v4*)p; return r; }
};
template <typename T> struct Vec2 {
  typename VecOf<T>::v2 v;
  __device__ __forceinline__ static Vec2 load(const T* p) { Vec2 r; r.v = *(const typename VecOf<T>::v2*)p; return r; }
};

template <typename T> static int vec_of(const T* p, long c) {
  const uintptr_t a = (uintptr_t)p;
  if (c % 4 == 0 && a % (4 * sizeof(T)) == 0) return 4;
  if (c % 2 == 0 && a % (2 * sizeof(T)) == 0) return 2;
  return 1;
}

// 4 contiguous elements
template <typename T> __device__ __forceinline__ void ld4(const T* p, int vec, T* out) {
  if (vec == 4) {
    const Vec4<T> t = Vec4<T>::load(p);
    out[0] = t.v[0]; out[1] = t.v[1]; out[2] = t.v[2]; out[3] = t.v[3];
  } else if (vec == 2) {
    const Vec2<T> a = Vec2<T>::load(p), b = Vec2<T>::load(p + 2);
    out[0] = a.v[0]; out[1] = a.v[1]; out[2] = b.v[0]; out[3] = b.v[1];
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = p[q];
  }
}
// 8 elements: element q comes from pA[q] while q < nA, from pB[q] after (nA is a multiple of vec)
template <typename T> __device__ __forceinline__ void ld_run8(const T* pA, const T* pB, int nA, int vec, T* raw) {
  if (vec == 4) {
    const Vec4<T> a = Vec4<T>::load(pA), b = Vec4<T>::load((nA >= 8 ? pA : pB) + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) { raw[q] = a.v[q]; raw[4 + q] = b.v[q]; }
  } else if (vec == 2) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Vec2<T> t = Vec2<T>::load((2 * k < nA ? pA : pB) + 2 * k);
      raw[2 * k] = t.v[0]; raw[2 * k + 1] = t.v[1];
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) raw[q] = (q < nA ? pA : pB)[q];
  }
}
// plain row-major rows (dense layers): 8 consecutive elements of row p.  The tail of the reduction is clamped (A operand:
// the matching B elements are zero) or read from the zero page (B operand)
template <typename T> __device__ __forceinline__ void lin_run8(const T* p, long r, long r_end, int vec, const T* zp, bool zero_tail, T* raw) {
  const long left = r_end - r;
  if (left >= 8 && vec > 1) {
    ld_run8<T>(p + r, p + r, 8, vec, raw);
  } else if (zero_tail) {
#pragma unroll
    for (int q = 0; q < 8; ++q) raw[q] = *(r + q < r_end ? p + r + q : zp);
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) raw[q] = p[r + q < r_end ? r + q : r_end - 1];
  }
}
// 4 consecutive columns j..j+3 of a row-major matrix row p (n columns); rows past the reduction end and columns past n
// read the zero page
template <typename T> __device__ __forceinline__ void lin_run4(const T* p, bool rok, long j, long n, int vec, const T* zp, T* raw) {
  if ((n & 3) == 0) {
    ld4<T>(rok && j < n ? p + j : zp, vec, raw);
  } else if ((n & 1) == 0 && vec >= 2) {
    const Vec2<T> a = Vec2<T>::load(rok && j < n ? p + j : zp), b = Vec2<T>::load(rok && j + 2 < n ? p + j + 2 : zp);
    raw[0] = a.v[0]; raw[1] = a.v[1]; raw[2] = b.v[0]; raw[3] = b.v[1];
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) raw[c] = *(rok && j + c < n ? p + j + c : zp);
  }
}

// ---- forward: i = output pixel, j = cout, r = (kh,kw,ci) ---------------------------------------
template <typename T> struct FwdOp {
  typedef T AT; typedef float BT;
  typedef NoAux Aux;
  static constexpr bool A_KMAJOR = true, B_KMAJOR = false;
  ConvGeom g; const T* x; const float* w; const float* bias; T* y; int accumulate;
  const float* wscale;   // optional device scalar: filter is divided by it (spectral norm sigma)
  long M, N, R, r_chunk;
  int avec, bvec;
  int ldy;               // elements between two output pixels: g.Cout, or N when only the first N output channels are produced
  float* slab;           // split reduction (launch_gemm_split_r): chunk z writes its partial sums to slab[z][M][ldy]; bias / accumulate later
  const void* zp;        // >= 32 bytes of device zeros: where invalid operand elements are read from
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
    return ok ? x + ((long)pix * g.Cin + c0) : (const T*)zp;
  }
  __device__ __forceinline__ void a8(const Row& rw, long r, long r_end, T* raw) const {
    const unsigned kk = g.dCin.div((unsigned)r);
    int ci = (int)((unsigned)r - kk * (unsigned)g.Cin);
    int kh = (int)g.dKW.div(kk), kw = (int)(kk - (unsigned)kh * (unsigned)g.KW);
    if (g.Cin >= 8) {
      const int nA = g.Cin - ci;
      bool okA, okB;
      const T* pA = tap(rw, kh, kw, ci, okA);
      const T* pB = pA;
      okB = false;
      if (nA < 8) {                        // second tap only for the runs that cross into it
        int kw2 = kw + 1, kh2 = kh;
        if (kw2 == g.KW) { kw2 = 0; ++kh2; }
        pB = tap(rw, kh2, kw2, -nA, okB);
      }
      ld_run8<T>(pA, pB, nA, avec, raw);
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        bool ok;
        raw[q] = *tap(rw, kh, kw, ci, ok);
        if (++ci == g.Cin) { ci = 0; if (++kw == g.KW) { kw = 0; ++kh; } }
      }
    }
  }
  __device__ __forceinline__ void b4(long r, long r_end, long j, float* raw) const {
    const bool rok = r < r_end;
    lin_run4<float>(w + (rok ? r : 0) * g.Cout, rok, j, N, bvec, (const float*)zp, raw);
  }
  __device__ __forceinline__ void store(long i, long j, float v, int z) const {
    if (slab) { slab[((long)z * M + i) * ldy + j] = v; return; }
    if (bias) v += bias[j];
    T* p = y + i * ldy + j;
    if (accumulate) v += Elem<T>::ld(p);
    Elem<T>::st(p, v);
  }
};

// filter element run for the data gradients: B(r, j) = w[tap(r)][j][co(r)], contiguous along co for a fixed ci = j
struct WRow { long j; int ok; };

// ---- data gradient: i = input pixel (n,ih,iw) at the logical resolution, j = ci, r = (kh,kw,co) ---
template <typename T> struct DgradOp {
  typedef T AT; typedef float BT;
  typedef NoAux Aux;
  static constexpr bool A_KMAJOR = true, B_KMAJOR = true;
  ConvGeom g; const T* dy; const float* w; const float* bias; T* dx; const T* xmask; int accumulate;
  const float* wscale;
  long M, N, R, r_chunk;
  int avec, bvec;
  const void* zp;        // >= 32 bytes of device zeros: where invalid operand elements are read from
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
    return ok ? dy + ((long)pix * g.Cout + c0) : (const T*)zp;
  }
  __device__ __forceinline__ void a8(const Row& rw, long r, long r_end, T* raw) const {
    const unsigned kk = g.dCout.div((unsigned)r);
    int co = (int)((unsigned)r - kk * (unsigned)g.Cout);
    int kh = (int)g.dKW.div(kk), kw = (int)(kk - (unsigned)kh * (unsigned)g.KW);
    if (g.Cout >= 8) {
      const int nA = g.Cout - co;
      bool okA, okB;
      const T* pA = tap(rw, kh, kw, co, okA);
      const T* pB = pA;
      okB = false;
      if (nA < 8) {
        int kw2 = kw + 1, kh2 = kh;
        if (kw2 == g.KW) { kw2 = 0; ++kh2; }
        pB = tap(rw, kh2, kw2, -nA, okB);
      }
      ld_run8<T>(pA, pB, nA, avec, raw);
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        bool ok;
        raw[q] = *tap(rw, kh, kw, co, ok);
        if (++co == g.Cout) { co = 0; if (++kw == g.KW) { kw = 0; ++kh; } }
      }
    }
  }
  __device__ __forceinline__ WRow brow(long j) const { WRow b; b.ok = j < N; b.j = b.ok ? j : 0; return b; }
  __device__ __forceinline__ const float* wtap(const WRow& b, unsigned kk, int c0, bool& ok) const {
    ok = b.ok && kk < (unsigned)(g.KH * g.KW);
    return ok ? w + (((long)kk * g.Cin + b.j) * g.Cout + c0) : (const float*)zp;
  }
  __device__ __forceinline__ void b8(const WRow& b, long r, long r_end, float* raw) const {
    const unsigned kk = g.dCout.div((unsigned)r);
    int co = (int)((unsigned)r - kk * (unsigned)g.Cout);
    if (g.Cout >= 8) {
      const int nA = g.Cout - co;
      bool okA, okB;
      const float* pA = wtap(b, kk, co, okA);
      const float* pB = pA;
      okB = false;
      if (nA < 8) pB = wtap(b, kk + 1, -nA, okB);
      ld_run8<float>(pA, pB, nA, bvec, raw);
    } else {
      unsigned k2 = kk;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        bool ok;
        raw[q] = *wtap(b, k2, co, ok);
        if (++co == g.Cout) { co = 0; ++k2; }
      }
    }
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
struct S2Cls { int ph, pw, Hp, Wp, kh0, kw0, nkh, nkw, dh, dwc; long M, R; FastDiv dnkw; };
struct S2Table { S2Cls cls[4]; };
template <typename T> struct DgradS2Op {
  typedef T AT; typedef float BT;
  static constexpr bool A_KMAJOR = true, B_KMAJOR = true;
  ConvGeom g; const T* dy; const float* w; const float* bias; T* dx; const T* xmask; int accumulate;
  const float* wscale;
  long M, N, R, r_chunk;
  int avec, bvec;
  const void* zp;        // >= 32 bytes of device zeros: where invalid operand elements are read from
  int ph, pw, Hp, Wp, kh0, kw0, nkh, nkw, dh, dwc;
  FastDiv dnkw;
  // the (up to) four parity classes of one layer run in ONE launch, class = blockIdx.z.  The class table is its own
  // kernel argument: a dynamically indexed array inside the functor would push the whole by-value copy into scratch.
  typedef S2Table Aux;
  __device__ __forceinline__ void select(const S2Table& tab, int z) {
    const S2Cls& c = tab.cls[z];
    ph = c.ph; pw = c.pw; Hp = c.Hp; Wp = c.Wp; kh0 = c.kh0; kw0 = c.kw0; nkh = c.nkh; nkw = c.nkw; dh = c.dh; dwc = c.dwc;
    M = c.M; R = c.R; r_chunk = c.R; dnkw = c.dnkw;
  }
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
    return ok ? dy + ((long)pix * g.Cout + c0) : (const T*)zp;
  }
  __device__ __forceinline__ void a8(const Row& rw, long r, long r_end, T* raw) const {
    const unsigned jj = g.dCout.div((unsigned)r);
    int co = (int)((unsigned)r - jj * (unsigned)g.Cout);
    int jh = (int)dnkw.div(jj), jw = (int)(jj - (unsigned)jh * (unsigned)nkw);
    if (g.Cout >= 8) {
      const int nA = g.Cout - co;
      bool okA, okB;
      const T* pA = tap(rw, jh, jw, co, okA);
      const T* pB = pA;
      okB = false;
      if (nA < 8) {
        int jw2 = jw + 1, jh2 = jh;
        if (jw2 == nkw) { jw2 = 0; ++jh2; }
        pB = tap(rw, jh2, jw2, -nA, okB);
      }
      ld_run8<T>(pA, pB, nA, avec, raw);
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        bool ok;
        raw[q] = *tap(rw, jh, jw, co, ok);
        if (++co == g.Cout) { co = 0; if (++jw == nkw) { jw = 0; ++jh; } }
      }
    }
  }
  __device__ __forceinline__ WRow brow(long j) const { WRow b; b.ok = j < N; b.j = b.ok ? j : 0; return b; }
  __device__ __forceinline__ const float* wtap(const WRow& b, int jh, int jw, int c0, bool& ok) const {
    ok = b.ok && jh < nkh;
    const int kh = kh0 + 2 * jh, kw = kw0 + 2 * jw;
    return ok ? w + (((long)(kh * g.KW + kw) * g.Cin + b.j) * g.Cout + c0) : (const float*)zp;
  }
  __device__ __forceinline__ void b8(const WRow& b, long r, long r_end, float* raw) const {
    const unsigned jj = g.dCout.div((unsigned)r);
    int co = (int)((unsigned)r - jj * (unsigned)g.Cout);
    int jh = (int)dnkw.div(jj), jw = (int)(jj - (unsigned)jh * (unsigned)nkw);
    if (g.Cout >= 8) {
      const int nA = g.Cout - co;
      bool okA, okB;
      const float* pA = wtap(b, jh, jw, co, okA);
      const float* pB = pA;
      okB = false;
      if (nA < 8) {
        int jw2 = jw + 1, jh2 = jh;
        if (jw2 == nkw) { jw2 = 0; ++jh2; }
        pB = wtap(b, jh2, jw2, -nA, okB);
      }
      ld_run8<float>(pA, pB, nA, bvec, raw);
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        bool ok;
        raw[q] = *wtap(b, jh, jw, co, ok);
        if (++co == g.Cout) { co = 0; if (++jw == nkw) { jw = 0; ++jh; } }
      }
    }
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

// ---- filter gradient: i = (kh,kw,ci), j = cout, r = output pixel; split over r into fp32 slabs.
//      Both operands are contiguous along the tile dimensions (ci / cout), strided along r: N-major fetch.
template <typename T> struct WgradOp {
  typedef T AT; typedef T BT;
  typedef NoAux Aux;
  static constexpr bool A_KMAJOR = false, B_KMAJOR = false;
  ConvGeom g; const T* x; const T* dy; float* slab;
  const float* wscale;   // always null (the filter gradient has no filter operand)
  long M, N, R, r_chunk;
  int avec, bvec;
  const void* zp;        // >= 32 bytes of device zeros: where invalid operand elements are read from
  // four consecutive filter positions (kh,kw,ci); with Cin % 4 == 0 they share the tap and only e[0] is used
  struct ARun { unsigned e[4]; };     // kh | kw << 8 | ci << 16 | valid << 31
  __device__ __forceinline__ ARun arun(long i) const {
    ARun a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = i + q < M;
      const unsigned ii = ok ? (unsigned)(i + q) : 0u;
      const unsigned kk = ii / (unsigned)g.Cin, ci = ii - kk * (unsigned)g.Cin;
      const unsigned kh = kk / (unsigned)g.KW, kw = kk - kh * (unsigned)g.KW;
      a.e[q] = kh | (kw << 8) | (ci << 16) | ((unsigned)ok << 31);
    }
    return a;
  }
  __device__ __forceinline__ bool a_relu() const { return g.relu_in; }
  __device__ __forceinline__ void a4(const ARun& a, long r, long r_end, T* raw) const {
    const bool rok = r < r_end;
    const unsigned rr = rok ? (unsigned)r : 0u;
    const unsigned t = g.dOW.div(rr), ow = rr - t * (unsigned)g.OW;
    const unsigned n = g.dOH.div(t), oh = t - n * (unsigned)g.OH;
    const int Hs = g.up ? (g.H >> 1) : g.H, Ws = g.up ? (g.W >> 1) : g.W;
    const int ihb = (int)oh * g.S - g.PT, iwb = (int)ow * g.S - g.PL;
    auto elem = [&](unsigned e, bool& ok) -> const T* {
      const int ih = ihb + (int)(e & 0xff), iw = iwb + (int)((e >> 8) & 0xff);
      ok = rok && (e >> 31) && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
      const int sh = g.up ? (ih >> 1) : ih, sw = g.up ? (iw >> 1) : iw;
      const unsigned pix = (n * (unsigned)Hs + (unsigned)sh) * (unsigned)Ws + (unsigned)sw;
      return ok ? x + ((long)pix * g.Cin + ((e >> 16) & 0x7fff)) : (const T*)zp;
    };
    if ((g.Cin & 3) == 0) {
      bool ok;
      ld4<T>(elem(a.e[0], ok), avec, raw);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bool ok;
        raw[q] = *elem(a.e[q], ok);
      }
    }
  }
  __device__ __forceinline__ void b4(long r, long r_end, long j, T* raw) const {
    const bool rok = r < r_end;
    lin_run4<T>(dy + (rok ? r : 0) * g.Cout, rok, j, N, bvec, (const T*)zp, raw);
  }
  __device__ __forceinline__ void store(long i, long j, float v, int z) const {
    slab[(long)z * M * N + i * N + j] = v;
  }
};

// ---- fully connected layers: the same GEMM core with plain row-major operands:
//      y[m][n] = x[m][:] . w[:][n];  dx[m][k] = dy[m][:] . w[k][:];  dw[k][n] = x[:][k] . dy[:][n]
template <typename T> struct LinRow { const T* p; int ok; };
template <typename T> struct LinFwdOp {
  typedef T AT; typedef float BT;
  typedef NoAux Aux;
  static constexpr bool A_KMAJOR = true, B_KMAJOR = false;
  const T* x; const float* w; const float* bias; T* y; const float* wscale;
  long M, N, R, r_chunk;
  int avec, bvec;
  const void* zp;        // >= 32 bytes of device zeros: where invalid operand elements are read from
  typedef LinRow<T> Row;
  __device__ __forceinline__ Row row(long i) const { Row rw; rw.ok = i < M; rw.p = x + (rw.ok ? i : 0) * R; return rw; }
  __device__ __forceinline__ bool a_relu() const { return false; }
  __device__ __forceinline__ void a8(const Row& rw, long r, long r_end, T* raw) const {
    lin_run8<T>(rw.p, r, r_end, avec, (const T*)zp, false, raw);
  }
  __device__ __forceinline__ void b4(long r, long r_end, long j, float* raw) const {
    const bool rok = r < r_end;
    lin_run4<float>(w + (rok ? r : 0) * N, rok, j, N, bvec, (const float*)zp, raw);
  }
  __device__ __forceinline__ void store(long i, long j, float v, int) const {
    if (bias) v += bias[j];
    Elem<T>::st(y + i * N + j, v);
  }
};
template <typename T> struct LinDgradOp {
  typedef T AT; typedef float BT;
  typedef NoAux Aux;
  static constexpr bool A_KMAJOR = true, B_KMAJOR = true;
  const T* dy; const float* w; T* dx; int accumulate; const float* wscale;
  float* slab;           // split reduction: chunk z writes slab[z][M][N]
  long M, N, R, r_chunk;          // N = in features, R = out features
  int avec, bvec;
  const void* zp;        // >= 32 bytes of device zeros: where invalid operand elements are read from
  typedef LinRow<T> Row;
  __device__ __forceinline__ Row row(long i) const { Row rw; rw.ok = i < M; rw.p = dy + (rw.ok ? i : 0) * R; return rw; }
  __device__ __forceinline__ bool a_relu() const { return false; }
  __device__ __forceinline__ void a8(const Row& rw, long r, long r_end, T* raw) const {
    lin_run8<T>(rw.p, r, r_end, avec, (const T*)zp, false, raw);
  }
  __device__ __forceinline__ LinRow<float> brow(long j) const { LinRow<float> b; b.ok = j < N; b.p = w + (b.ok ? j : 0) * R; return b; }
  __device__ __forceinline__ void b8(const LinRow<float>& b, long r, long r_end, float* raw) const {
    lin_run8<float>(b.p, r, r_end, bvec, (const float*)zp, true, raw);
  }
  __device__ __forceinline__ void store(long i, long j, float v, int z) const {
    if (slab) { slab[((long)z * M + i) * N + j] = v; return; }
    T* p = dx + i * N + j;
    if (accumulate) v += Elem<T>::ld(p);
    Elem<T>::st(p, v);
  }
};
template <typename T> struct LinWgradOp {
  typedef T AT; typedef T BT;
  typedef NoAux Aux;
  static constexpr bool A_KMAJOR = false, B_KMAJOR = false;
  const T* x; const T* dy; float* out; int accumulate; int direct; const float* wscale;
  long M, N, R, r_chunk;          // M = in features, N = out features, R = batch rows
  int avec, bvec;
  const void* zp;        // >= 32 bytes of device zeros: where invalid operand elements are read from
  struct ARun { long i; };
  __device__ __forceinline__ ARun arun(long i) const { ARun a; a.i = i; return a; }
  __device__ __forceinline__ bool a_relu() const { return false; }
  __device__ __forceinline__ void a4(const ARun& a, long r, long r_end, T* raw) const {
    const bool rok = r < r_end;
    lin_run4<T>(x + (rok ? r : 0) * M, rok, a.i, M, avec, (const T*)zp, raw);
  }
  __device__ __forceinline__ void b4(long r, long r_end, long j, T* raw) const {
    const bool rok = r < r_end;
    lin_run4<T>(dy + (rok ? r : 0) * N, rok, j, N, bvec, (const T*)zp, raw);
  }
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
// operands from LDS and issues 16 MFMAs.  The K-slot a lane feeds to MFMA s is 16*(lane>>5)+s, so a K-major operand
// of all 16 steps is one contiguous 64-byte run per lane.  The operand elements of step s+1 are fetched into registers
// before the MFMAs of step s and written to the other LDS buffer after them: one barrier per step.
// KS groups of four waves walk interleaved K-steps (group g takes steps g, g+KS, ...) on private LDS buffers and their
// accumulators are summed through LDS at the end: the dense layers of the MNIST nets launch 16..400 workgroups with
// 50..200 sequential K-steps each.
template <class Op, class = void> struct has_select : std::false_type {};
template <class Op> struct has_select<Op, std::void_t<decltype(&Op::select)>> : std::true_type {};
template <class Op> __device__ __forceinline__ int gg_select(Op& op, const typename Op::Aux& aux, int z) {
  if constexpr (has_select<Op>::value) { op.select(aux, z); return 0; }
  else return z;
}
template <class Op> __device__ __forceinline__ auto a_init(const Op& op, long i0, int tid) {
  if constexpr (Op::A_KMAJOR) return op.row(i0 + (tid >> 2));
  else return op.arun(i0 + (tid & 15) * 4);
}
template <class Op> __device__ __forceinline__ auto b_init(const Op& op, long j0, int tid) {
  if constexpr (Op::B_KMAJOR) return op.brow(j0 + (tid >> 2));
  else return j0 + (tid & 15) * 4;
}

#define GG_XBUF 2304   /* floats per operand buffer: max(64 x 36 K-major, 32 x 68 N-major) */

template <class Op, int KS>
__global__ __launch_bounds__(256 * KS) void gemm_gather_kernel(Op op_in, typename Op::Aux aux) {
  extern __shared__ __attribute__((aligned(16))) float gg_smem[];
  Op op = op_in;
  const int zc = gg_select(op, aux, (int)blockIdx.z);      // r-chunk index (ops that use grid.z for something else return 0)
  if ((long)blockIdx.y * 64 >= op.M) return;
  const int kg = threadIdx.x >> 8;
  float* const lds = gg_smem + (size_t)kg * 4 * GG_XBUF;      // A buf 0/1, B buf 0/1
  const int tid = threadIdx.x & 255;
  const int lane = tid & 63, wv = tid >> 6, wr = wv >> 1, wc = wv & 1, l31 = lane & 31, hh = lane >> 5;
  const long i0 = (long)blockIdx.y * 64, j0 = (long)blockIdx.x * 64;
  const long r_begin = (long)zc * op.r_chunk;
  long r_end = r_begin + op.r_chunk;
  if (r_end > op.R) r_end = op.R;
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 acc;
#pragma unroll
  for (int p = 0; p < 16; ++p) acc[p] = 0.f;

  // spectral-norm division W / sigma: sigma is loaded once per thread, not once per operand element
  const float bscale = op.wscale ? 1.f / *op.wscale : 1.f;
  typename Op::AT ra[8];
  typename Op::BT rb[8];
  const auto ast = a_init(op, i0, tid);
  const auto bst = b_init(op, j0, tid);
  auto fetch = [&](long r0) {
    if constexpr (Op::A_KMAJOR) {
      op.a8(ast, r0 + (tid & 3) * 8, r_end, ra);
    } else {
      op.a4(ast, r0 + (tid >> 4), r_end, ra);
      op.a4(ast, r0 + (tid >> 4) + 16, r_end, ra + 4);
    }
    if constexpr (Op::B_KMAJOR) {
      op.b8(bst, r0 + (tid & 3) * 8, r_end, rb);
    } else {
      op.b4(r0 + (tid >> 4), r_end, bst, rb);
      op.b4(r0 + (tid >> 4) + 16, r_end, bst, rb + 4);
    }
  };
  auto put = [&](float* xb, bool kmajor, const float* f) {
    if (kmajor) {
      float* p = xb + (tid >> 2) * 36 + (tid & 3) * 8;
      *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
      *(float4*)(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
    } else {
      float* p = xb + (tid >> 4) * 68 + (tid & 15) * 4;
      *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
      *(float4*)(p + 16 * 68) = make_float4(f[4], f[5], f[6], f[7]);
    }
  };
  // Invalid operand elements (padding taps, rows / columns / reduction indices past the end) were READ as zeros from the
  // zero page, so the LDS write is a plain conversion: no masks.  B is exactly zero past the end of the reduction, which
  // covers the (finite) A elements there; rows and columns past M / N are never stored.
  const bool relu = op.a_relu();
  auto stash = [&](int buf) {
    float fa[8], fb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      fa[q] = Elem<typename Op::AT>::ld(&ra[q]);
      fb[q] = Elem<typename Op::BT>::ld(&rb[q]);
    }
    if (relu) {
#pragma unroll
      for (int q = 0; q < 8; ++q) fa[q] = fmaxf(fa[q], 0.f);
    }
    put(lds + buf * GG_XBUF, Op::A_KMAJOR, fa);
    put(lds + (2 + buf) * GG_XBUF, Op::B_KMAJOR, fb);
  };
  auto get = [&](const float* xb, bool kmajor, int w32, float* v) {
    if (kmajor) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 t4 = *(const float4*)(xb + (w32 + l31) * 36 + 16 * hh + 4 * c);
        v[4 * c] = t4.x; v[4 * c + 1] = t4.y; v[4 * c + 2] = t4.z; v[4 * c + 3] = t4.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = xb[(16 * hh + q) * 68 + w32 + l31];
    }
  };
  const long nsteps = (r_end - r_begin + 31) / 32;
  const long T = (nsteps + KS - 1) / KS;          // per-group steps (same for every group: the barriers are block-wide)
  auto step_r = [&](long t) { return r_begin + 32 * (kg + KS * t); };
  // a wavefront whose 32 x 32 block lies wholly past M or N (the 10 label columns of a 138- / 74- / 1034-wide operand fill a sixth of their
  // tile) stages operands and keeps the barriers but leaves the matrix pipe and the LDS reads to the others
  const bool block_live = i0 + wr * 32 < op.M && j0 + wc * 32 < op.N;
  auto mfma_step = [&](int buf) {
    if (!block_live) return;
    float av[16], bv[16];
    get(lds + buf * GG_XBUF, Op::A_KMAJOR, wr * 32, av);
    get(lds + (2 + buf) * GG_XBUF, Op::B_KMAJOR, wc * 32, bv);
    // all operand reads land before the MFMA chain starts: left alone the scheduler feeds every second MFMA from an
    // LDS read issued right before it and the chain pays the LDS latency sixteen times
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // The waves that share a SIMD belong to different K-groups and run the same fetch -> MFMA -> LDS-write cycle; left
  // in phase they all decode addresses together (matrix core idle) and then queue on the matrix core together (vector
  // ALU idle).  Odd groups run the cycle rotated by one stage (MFMA first, then write + fetch two steps ahead), so one
  // half of a SIMD's waves is in its MFMA chain while the other half does the vector work.
  const bool rotated = KS > 1 && (kg & 1);
  if (T > 0) { fetch(step_r(0)); stash(0); }
  if (rotated && T > 1) fetch(step_r(1));
  __syncthreads();
  int buf = 0;
  if (!rotated) {
    for (long t = 0; t < T; ++t) {
      const bool more = t + 1 < T;
      if (more) fetch(step_r(t + 1));
      mfma_step(buf);
      if (more) stash(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  } else {
    for (long t = 0; t < T; ++t) {
      mfma_step(buf);
      if (t + 1 < T) stash(buf ^ 1);
      if (t + 2 < T) fetch(step_r(t + 2));
      __syncthreads();
      buf ^= 1;
    }
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
      if (i < op.M) op.store(i, j, acc[p] * bscale, zc);      // the filter's 1/sigma, applied once per output
    }
  }
}

// Stride-2 data gradient / transposed conv with 1..4 output channels (the MNIST generator's image layer, Cout = 1):
// a 64-column GEMM tile would be 1/64 occupied.  Sixteen lanes share an output pixel and split the channel runs of its
// reachable taps (coalesced, VEC elements per lane), the filter slice lives in LDS, the lane partials meet in a
// 4-step butterfly.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void dgrad_s2_narrow_kernel(DgradS2Op<T> op_in, S2Table tab) {
  extern __shared__ __attribute__((aligned(16))) float nw_smem[];     // [tap][j][Cout]
  DgradS2Op<T> op = op_in;
  op.select(tab, (int)blockIdx.y);
  if ((long)blockIdx.x * 16 >= op.M) return;
  const int C = op.g.Cout, NJ = (int)op.N, ntaps = op.nkh * op.nkw;
  const float bscale = op.wscale ? 1.f / *op.wscale : 1.f;
  for (int e = threadIdx.x; e < ntaps * NJ * C; e += 256) {
    const int co = e % C, t = e / C, j = t % NJ, tp = t / NJ;
    const int kh = op.kh0 + 2 * (tp / op.nkw), kw = op.kw0 + 2 * (tp % op.nkw);
    nw_smem[e] = op.w[((long)(kh * op.g.KW + kw) * op.g.Cin + j) * C + co] * bscale;
  }
  __syncthreads();
  const int sub = threadIdx.x & 15;
  const long i = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const auto rw = op.row(i);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int jh = 0; jh < op.nkh; ++jh)
    for (int jw = 0; jw < op.nkw; ++jw) {
      bool ok;
      const T* p = op.tap(rw, jh, jw, 0, ok);
      if (!ok) continue;
      const float* wl = nw_smem + (size_t)(jh * op.nkw + jw) * NJ * C;
      for (int c0 = sub * VEC; c0 < C; c0 += 16 * VEC) {
        T raw[VEC];
        if constexpr (VEC == 4) { const Vec4<T> t4 = Vec4<T>::load(p + c0); raw[0] = t4.v[0]; raw[1] = t4.v[1]; raw[2] = t4.v[2]; raw[3] = t4.v[3]; }
        else if constexpr (VEC == 2) { const Vec2<T> t2 = Vec2<T>::load(p + c0); raw[0] = t2.v[0]; raw[1] = t2.v[1]; }
        else raw[0] = p[c0];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          const float a = Elem<T>::ld(&raw[q]);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < NJ) acc[j] = fmaf(a, wl[j * C + c0 + q], acc[j]);
        }
      }
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = acc[j];
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    if (sub == 0 && j < NJ && i < op.M) op.store(i, j, v, 0);
  }
}

template <typename T>
static int launch_dgrad_s2_narrow(rcgan_ctx* ctx, DgradS2Op<T>& op, const S2Table& tab, int ncls, long maxM, long maxR) {
  const size_t lds = (size_t)maxR * op.N * sizeof(float);
  dim3 grid(cdiv(maxM, 16), ncls);
  op.zp = ctx->zero_page;
  if (op.avec == 4) hipLaunchKernelGGL((dgrad_s2_narrow_kernel<T, 4>), grid, dim3(256), lds, ctx->stream, op, tab);
  else if (op.avec == 2) hipLaunchKernelGGL((dgrad_s2_narrow_kernel<T, 2>), grid, dim3(256), lds, ctx->stream, op, tab);
  else hipLaunchKernelGGL((dgrad_s2_narrow_kernel<T, 1>), grid, dim3(256), lds, ctx->stream, op, tab);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// out[i] (= or +=) sum_z slab[z][i].  A block owns 64 outputs; its four wavefronts take the slabs z = w, w + 4, ... (four
// independent loads in flight each) and meet in LDS in a fixed order: the narrow layers' filter gradients come as up to 256 slabs of a
// few thousand outputs (g_h3: 3450 outputs x 256 slabs), where one thread per output was a chain of 256 dependent loads (48 us).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* slab, float* out, long count, int nz, int accumulate) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < count) {
    int z = w;
    for (; z + 12 < nz; z += 16) {
      s0 += slab[(long)z * count + i];
      s1 += slab[(long)(z + 4) * count + i];
      s2 += slab[(long)(z + 8) * count + i];
      s3 += slab[(long)(z + 12) * count + i];
    }
    for (; z < nz; z += 4) s0 += slab[(long)z * count + i];
  }
  part[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && i < count) {
    float s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (accumulate) s += out[i];
    out[i] = s;
  }
}

// out[i] (= or +=) bias[i % n] + sum_z slab[z][i]  (the epilogue of a split-reduction forward / data-gradient GEMM)
__global__ void slab_reduce_bias_kernel(const float* slab, float* out, long count, int nz, int accumulate, const float* bias, int n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s = bias ? bias[i % n] : 0.f;
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
static int launch_gemm_ks(rcgan_ctx* ctx, Op& op, dim3 grid, const typename Op::Aux& aux) {
  static bool attr_set = false;
  const size_t lds = (size_t)KS * 4 * GG_XBUF * sizeof(float);
  if (!attr_set) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)gemm_gather_kernel<Op, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  {
    // multiply-adds the launch performs: M x N outputs over the whole reduction (grid.z splits R), or -- ops that use grid.z for the
    // parity classes of a stride-2 data gradient -- over every class's share
    const double fl = 2.0 * (double)op.M * (double)op.N * (double)op.R * (has_select<Op>::value ? (double)grid.z : 1.0);
    ProfScope ps(ctx, RCGAN_PROF_GATHER_F32, fl);
    hipLaunchKernelGGL((gemm_gather_kernel<Op, KS>), grid, dim3(256 * KS), lds, ctx->stream, op, aux);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

template <class Op>
static int launch_gemm(rcgan_ctx* ctx, Op& op, int nz, const typename Op::Aux& aux = typename Op::Aux()) {
  op.zp = ctx->zero_page;
  dim3 grid(cdiv(op.N, 64), cdiv(op.M, 64), nz);
  if (grid.y > 65535u || grid.z > 65535u) RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "M too large");
  static const int ks_min_steps = gg_env_int("RCGAN_GG_KS_MINSTEPS", 8);
  const long steps = (op.r_chunk + 31) / 32;
  static const int ks_force = gg_env_int("RCGAN_GG_KS", 0);
  if (ks_force == 2) return launch_gemm_ks<Op, 2>(ctx, op, grid, aux);
  if (ks_force == 1) return launch_gemm_ks<Op, 1>(ctx, op, grid, aux);
  if (steps >= ks_min_steps) {
    // four K-slices per workgroup = 1024 threads = ONE workgroup per CU: a grid of more workgroups than CUs runs in whole rounds (392
    // workgroups: two, the second half empty).  Two slices = two workgroups per CU, the same K-steps per wavefront pair -- measured on the
    // MNIST iteration (B = 256), per launch: 1568 workgroups 214 -> 177 us, 1666 111 -> 84, 392 170 -> 161; but 196 workgroups 46 -> 53,
    // 128 44 -> 51 (fewer workgroups than CUs: the four slices are the parallelism).
    static const long ks2_min_wgs = gg_env_int("RCGAN_GG_KS2_MINWGS", 256);
    if ((long)grid.x * grid.y * grid.z >= ks2_min_wgs) return launch_gemm_ks<Op, 2>(ctx, op, grid, aux);
    return launch_gemm_ks<Op, 4>(ctx, op, grid, aux);
  }
  return launch_gemm_ks<Op, 1>(ctx, op, grid, aux);
}

// Forward / data-gradient GEMMs with few output tiles and a long reduction (the MNIST critic's 4x4 and 2x2 layers: 64 / 16 tiles over
// K = 1600; the generator's 6272 -> 1024 dense data gradient: 64 tiles over K = 6272) leave most of the chip idle behind a serial chain
// of K-steps.  Here the reduction is cut into nz chunks (grid.z) that write fp32 partial tiles to a scratch slab; a second launch adds
// them up in a fixed order (+ bias, + the accumulate target).  fp32 outputs only; the scratch is the context's grow-only buffer.
static int split_r_enabled() {
  static const int v = gg_env_int("RCGAN_GG_SPLIT_R", 1);
  return v;
}
static int split_r_chunks(long M, long N, long R, int kind = 1) {      // kind 1: convolution forward, 2: dense data gradient
  const long tiles = (long)cdiv(M, 64) * cdiv(N, 64), steps = (R + 31) / 32;
  const int en = split_r_enabled();        // (debug: 2 = convolutions only, 3 = dense layers only)
  if (!en || (en == 2 && kind != 1) || (en == 3 && kind != 2) || tiles >= 128 || steps < 32) return 1;
  long nz = 256 / tiles;
  if (nz > steps / 8) nz = steps / 8;       // >= 8 K-steps per chunk
  if (nz > 8) nz = 8;
  return nz < 2 ? 1 : (int)nz;
}
template <class Op>
static int launch_gemm_split_r(rcgan_ctx* ctx, Op& op, int nz, float* out, long ld_count, const float* bias, int ncols, int accumulate) {
  const size_t need = (size_t)nz * ld_count * sizeof(float);
  RC_HIP(ctx, ctx_grow_scratch(ctx, &ctx->splitr_ws, &ctx->splitr_ws_bytes, need));
  op.slab = (float*)ctx->splitr_ws;
  op.r_chunk = (((op.R + nz - 1) / nz) + 31) / 32 * 32;
  nz = (int)cdiv(op.R, op.r_chunk);
  int rc = launch_gemm(ctx, op, nz);
  if (rc != RCGAN_OK) return rc;
  hipLaunchKernelGGL(slab_reduce_bias_kernel, dim3((unsigned)cdiv(ld_count, 256)), dim3(256), 0, ctx->stream, (const float*)op.slab, out, ld_count, nz,
                     accumulate, bias, ncols);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// pick the number of r-splits for a filter-gradient GEMM so the grid fills the chip
static int wgrad_splits(long K, long Cout, long M) {
  long tiles = (long)cdiv(K, 64) * cdiv(Cout, 64);
  long want = (1024 + tiles - 1) / tiles;
  long maxs = (M + 255) / 256;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  // (round 6) whole rounds: such a grid runs two 512-thread workgroups per CU (launch_gemm: two K-slices, 74 KB of LDS each), 512 at
  // a time on 256 CUs.  tiles x splits just above a multiple of 512 -- the MNIST generator's 5x5 transposed convolution: 150 tiles
  // x 7 splits = 1050 -- pays a whole extra round for a few workgroups; one split less (900: two rounds of 17 % longer workgroups)
  // is the shorter launch.  RCGAN_GG_WGRAD_FIT=0: the plain ceil(1024 / tiles).
  static const int fit = gg_env_int("RCGAN_GG_WGRAD_FIT", 1);
  if (fit && want > 1 && tiles * want >= 512) {
    const long over = (tiles * want) % 512;
    if (over != 0 && over * 4 < 512) {               // the last round less than a quarter full
      long w2 = want;
      while (w2 > 1 && (tiles * w2) % 512 != 0 && ((tiles * w2) % 512) * 4 < 512 && tiles * w2 > 512) --w2;
      // accept when the rounds saved outweigh the longer workgroups: rounds(w2) / w2 < rounds(want) / want
      const long r1 = (tiles * want + 511) / 512, r2 = (tiles * w2 + 511) / 512;
      if (r2 * want < r1 * w2) want = w2;
    }
  }
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
  op.avec = vec_of(x, k); op.bvec = vec_of(w, n);
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
  op.dy = dy; op.w = w; op.dx = dx; op.accumulate = accumulate; op.wscale = wscale; op.slab = nullptr;
  op.M = m; op.N = k; op.R = n; op.r_chunk = n;
  op.avec = vec_of(dy, n); op.bvec = vec_of(w, n);
  if constexpr (std::is_same<T, float>::value) {
    const int nz = split_r_chunks(m, k, n, 2);
    if (nz > 1) return launch_gemm_split_r(ctx, op, nz, (float*)dx, m * k, nullptr, (int)k, accumulate);
  }
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
  op.avec = vec_of(x, k); op.bvec = vec_of(dy, n);
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
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(cnt, 64)), dim3(256), 0, ctx->stream, (const float*)op.out, dw, cnt, nz, accumulate);
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

// n_cols > 0: only the first n_cols output channels are produced, densely ([pixels][n_cols]) -- the data gradient of a transposed
// convolution whose input was a channel concatenation x || labels needs the x part only (mnist/ops.py:46-51: the labels are data)
template <typename T>
int direct_fwd(rcgan_ctx* ctx, const rcgan_conv_desc* d, const T* x, const float* w, const float* wscale, const float* bias, T* y, int n_cols) {
  FwdOp<T> op;
  op.g = make_geom(d); op.x = x; op.w = w; op.wscale = wscale; op.bias = bias; op.y = y;
  op.accumulate = (d->flags & RCGAN_CONV_ACCUMULATE) ? 1 : 0;
  op.M = (long)op.g.N * op.g.OH * op.g.OW; op.N = n_cols > 0 ? n_cols : op.g.Cout; op.R = (long)op.g.KH * op.g.KW * op.g.Cin; op.r_chunk = op.R;
  op.ldy = (int)op.N; op.slab = nullptr;
  op.avec = vec_of(x, op.g.Cin); op.bvec = vec_of(w, op.g.Cout);
  if constexpr (std::is_same<T, float>::value) {
    const int nz = split_r_chunks(op.M, op.N, op.R);
    if (nz > 1) return launch_gemm_split_r(ctx, op, nz, (float*)y, op.M * op.ldy, bias, op.ldy, op.accumulate);
  }
  return launch_gemm(ctx, op, 1);
}
template int direct_fwd<float>(rcgan_ctx*, const rcgan_conv_desc*, const float*, const float*, const float*, const float*, float*, int);
template int direct_fwd<bf16_t>(rcgan_ctx*, const rcgan_conv_desc*, const bf16_t*, const float*, const float*, const float*, bf16_t*, int);

// ---- narrow stride-2 data gradient / transposed convolution in two steps (fp32) ------------------------------------------------
// The MNIST generator's image layer (g_h3: 138 -> 1 channel, 14x14 -> 28x28, model.py:726) and the data gradient of the critic's first
// convolution (1 <- 64 channels) produce ONE channel per pixel: as a gather GEMM a 64-column tile is 1/64 occupied, and the
// 16-lanes-per-pixel kernel above spends its time on tap arithmetic (118 us for 0.35 GFLOP at B = 256).  Because the output is narrow
// the products can be formed per SOURCE pixel instead: P[m][(kh, kw, ci)] = sum_co dy[m][co] * w[kh][kw][ci][co] is a dense
// [N*OH*OW, Cout] x [Cout, KH*KW*Cin] GEMM on the matrix cores (linear_dgrad: 25 columns for one output channel), and the transposed
// convolution is then a col2im: every output pixel adds the <= ceil(K/2)^2 products that land on it (+ bias).
__global__ __launch_bounds__(256) void col2im_s2_kernel(long total, ConvGeom g, const float* P, const float* bias, float* dx, int accumulate) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int ci = (int)(e % g.Cin);
  long t = e / g.Cin;
  const int iw = (int)(t % g.W); t /= g.W;
  const int ih = (int)(t % g.H);
  const int n = (int)(t / g.H);
  const int ntap = g.KH * g.KW * g.Cin;
  float v = bias ? bias[ci] : 0.f;
  for (int kh = (ih + g.PT) % g.S; kh < g.KH; kh += g.S) {
    const int oh = (ih + g.PT - kh) / g.S;
    if (ih + g.PT - kh < 0 || oh >= g.OH) continue;
    for (int kw = (iw + g.PL) % g.S; kw < g.KW; kw += g.S) {
      const int ow = (iw + g.PL - kw) / g.S;
      if (iw + g.PL - kw < 0 || ow >= g.OW) continue;
      v += P[(((long)n * g.OH + oh) * g.OW + ow) * ntap + (kh * g.KW + kw) * g.Cin + ci];
    }
  }
  dx[e] = accumulate ? dx[e] + v : v;
}

static int narrow_two_step_enabled() {
  static const int v = gg_env_int("RCGAN_NARROW_TWO_STEP", 1);
  return v;
}

// dgrad at the LOGICAL input resolution (no upsample folding here; the caller pools afterwards).
template <typename T>
int direct_dgrad(rcgan_ctx* ctx, const rcgan_conv_desc* d, const T* dy, const float* w, const float* wscale, const float* bias,
                 const T* xmask, T* dx, int accumulate) {
  {
    ConvGeom g = make_geom(d);
    if (g.S == 2) {                      // the four input-pixel parity classes, one launch (class = grid.z)
      DgradS2Op<T> op;
      op.g = g; op.g.up = 0; op.dy = dy; op.w = w; op.wscale = wscale; op.bias = bias; op.dx = dx; op.xmask = xmask; op.accumulate = accumulate;
      op.N = g.Cin;
      op.avec = vec_of(dy, g.Cout); op.bvec = vec_of(w, g.Cout);
      S2Table tab;
      int ncls = 0;
      long maxM = 0, maxR = 0;
      for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
          S2Cls c;
          c.ph = ph; c.pw = pw; c.Hp = (g.H - ph + 1) / 2; c.Wp = (g.W - pw + 1) / 2;
          if (c.Hp <= 0 || c.Wp <= 0) continue;
          c.kh0 = (ph + g.PT) % 2; c.kw0 = (pw + g.PL) % 2;
          c.nkh = g.KH > c.kh0 ? (g.KH - c.kh0 + 1) / 2 : 0;
          c.nkw = g.KW > c.kw0 ? (g.KW - c.kw0 + 1) / 2 : 0;
          c.dh = (ph + g.PT - c.kh0) / 2; c.dwc = (pw + g.PL - c.kw0) / 2;
          if (c.nkw == 0) { c.nkw = 1; c.nkh = 0; }          // keeps the divisions defined; R = 0: outputs are bias / 0
          c.dnkw = make_fastdiv(c.nkw);
          c.M = (long)g.N * c.Hp * c.Wp; c.R = (long)c.nkh * c.nkw * g.Cout;
          if (c.M > maxM) maxM = c.M;
          if (c.R > maxR) maxR = c.R;
          tab.cls[ncls++] = c;
        }
      if (ncls == 0) return RCGAN_OK;
      // (round 6) longest class first: grid.z is the slowest dispatch index, and with a 5x5 filter the classes reduce over 4, 6, 6 and 9
      // taps -- in (ph, pw) order the 9-tap workgroups started last and the launch ended on them alone.  RCGAN_S2_LPT=0: (ph, pw) order.
      static const int lpt = gg_env_int("RCGAN_S2_LPT", 1);
      if (lpt) std::stable_sort(tab.cls, tab.cls + ncls, [](const S2Cls& a, const S2Cls& b) { return a.R * a.M > b.R * b.M; });
      for (int q = ncls; q < 4; ++q) tab.cls[q] = tab.cls[0];
      // launch-shape fields (the kernel re-selects per class)
      op.M = maxM; op.R = maxR; op.r_chunk = maxR;
      { const auto& c = tab.cls[0]; op.ph = c.ph; op.pw = c.pw; op.Hp = c.Hp; op.Wp = c.Wp; op.kh0 = c.kh0; op.kw0 = c.kw0;
        op.nkh = c.nkh; op.nkw = c.nkw; op.dh = c.dh; op.dwc = c.dwc; op.dnkw = c.dnkw; }
      if constexpr (std::is_same<T, float>::value) {
        // one or two output channels, fp32, no ReLU mask: products per source pixel on the matrix cores + col2im (above)
        const long m = (long)g.N * g.OH * g.OW, kcols = (long)g.KH * g.KW * g.Cin;
        const size_t need = (size_t)m * kcols * sizeof(float);
        if (narrow_two_step_enabled() && op.N <= 2 && xmask == nullptr && g.Cout >= 32 && need <= ((size_t)1 << 30)) {
          RC_HIP(ctx, ctx_grow_scratch(ctx, &ctx->narrow_ws, &ctx->narrow_ws_bytes, need));
          int rc = linear_dgrad<float>(ctx, m, kcols, g.Cout, (const float*)dy, w, wscale, (float*)ctx->narrow_ws, 0);
          if (rc != RCGAN_OK) return rc;
          const long total = (long)g.N * g.H * g.W * g.Cin;
          hipLaunchKernelGGL(col2im_s2_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, ctx->stream, total, g, (const float*)ctx->narrow_ws, bias,
                             (float*)dx, accumulate);
          RC_LAUNCH_CHECK(ctx);
          return RCGAN_OK;
        }
      }
      if (op.N <= 4 && maxR > 0 && (size_t)maxR * op.N * sizeof(float) <= 48 * 1024) return launch_dgrad_s2_narrow(ctx, op, tab, ncls, maxM, maxR);
      return launch_gemm(ctx, op, ncls, tab);
    }
  }
  DgradOp<T> op;
  op.g = make_geom(d); op.g.up = 0; op.dy = dy; op.w = w; op.wscale = wscale; op.bias = bias; op.dx = dx; op.xmask = xmask; op.accumulate = accumulate;
  op.M = (long)op.g.N * op.g.H * op.g.W; op.N = op.g.Cin; op.R = (long)op.g.KH * op.g.KW * op.g.Cout; op.r_chunk = op.R;
  op.avec = vec_of(dy, op.g.Cout); op.bvec = vec_of(w, op.g.Cout);
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
  op.avec = vec_of(x, op.g.Cin); op.bvec = vec_of(dy, op.g.Cout);
  op.r_chunk = ((M + nz - 1) / nz + 15) / 16 * 16;
  nz = cdiv(M, op.r_chunk);
  int rc = launch_gemm(ctx, op, nz);
  if (rc) return rc;
  long cnt = K * op.g.Cout;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(cnt, 64)), dim3(256), 0, ctx->stream, (const float*)op.slab, dw, cnt, nz, accumulate);
  RC_LAUNCH_CHECK(ctx);
  if (dbias) {
    float* part = (float*)((char*)ws + (size_t)wgrad_splits(K, op.g.Cout, M) * K * op.g.Cout * sizeof(float));
    rc = colsum_launch<T>(ctx, dy, M, op.g.Cout, dbias, accumulate, part);
    if (rc) return rc;
  }
  return RCGAN_OK;
}
// ---- filter gradient of a convolution whose OUTPUT-gradient rows end in label columns (round 6) -------------------------------------
// dy = conv_cond_concat(t, yb) (mnist/ops.py:46-51) seen from the transposed convolution that consumes it (the MNIST generator's g_h2 / g_h3,
// model.py:722-731: the layer's input is [t ; yb broadcast over the pixels]): columns j >= c1 of an output-gradient row are yb[n][j - c1],
// the same for every pixel of sample n.  The gather GEMM over all Cout = c1 + c2 columns pays a whole 64-wide column tile for the c2 = 10
// label columns (150 output tiles instead of 100 for the 5x5x128x138 filter).  Here the GEMM runs over the c1 real columns only and the
// label columns come from what they are:
//     dW[kh][kw][c][c1 + l] = sum_n yb[n][l] * D[n][kh][kw][c],     D[n][kh][kw][c] = sum over the output pixels (oh, ow) whose tap (kh, kw)
//                                                                    lands inside the image of x[n][oh*S - PT + kh][ow*S - PL + kw][c]
// -- D is a sum over a sub-grid of the image, separable in rows and columns: ONE pass over x (wgrad_label_sums_kernel), then a
// [c2 x N] x [N x K] product in sample chunks (wgrad_label_cols_kernel), and one reduction that writes the c1 columns from the GEMM's
// slabs and the c2 columns from the chunk partials with the filter's own row stride.
#define WGRAD_LABEL_RG 4        /* row groups: a sample's image rows ih = rg, rg + 4, ... go to workgroup (n, rg) */
// which image columns (rows) a tap column kw (row kh) reaches: bit iw of col[kw] is set iff (iw + PL - kw) is a multiple of S inside the
// output grid -- computed on the host, so the kernel's inner loop is a mask test instead of two integer divisions per (pixel, tap)
struct LabelMasks { unsigned col[5], row[5]; };
template <typename T, int WMAX>
__global__ __launch_bounds__(128) void wgrad_label_sums_kernel(ConvGeom g, LabelMasks mk, const T* x, float* D) {      // D[rg][n][KH*KW][Cin]; H <= 32, W <= WMAX
  const int n = blockIdx.x, rg = blockIdx.y;
  for (int c = threadIdx.x; c < g.Cin; c += blockDim.x) {
    float d[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) d[t] = 0.f;
    // the workgroup's rows ih = rg, rg + 4, ... in batches of four: every load of a batch (4 x W) is in flight before the first sum
    for (int ih0 = rg; ih0 < g.H; ih0 += 4 * WGRAD_LABEL_RG) {
      float v[4][WMAX];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ih = ih0 + b * WGRAD_LABEL_RG;
        const T* row = x + ((long)(n * g.H + (ih < g.H ? ih : 0)) * g.W) * g.Cin + c;
#pragma unroll
        for (int iw = 0; iw < WMAX; ++iw) v[b][iw] = (ih < g.H && iw < g.W) ? Elem<T>::ld(row + (long)iw * g.Cin) : 0.f;
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ih = ih0 + b * WGRAD_LABEL_RG;
        if (ih >= g.H) break;
        float cs[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) {
          const unsigned m = mk.col[kw];
#pragma unroll
          for (int iw = 0; iw < WMAX; ++iw) cs[kw] += ((m >> iw) & 1u) ? v[b][iw] : 0.f;
        }
#pragma unroll
        for (int kh = 0; kh < 5; ++kh) {
          if ((mk.row[kh] >> ih) & 1u) {
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) d[kh * 5 + kw] += cs[kw];
          }
        }
      }
    }
    const long K = (long)g.KH * g.KW * g.Cin;
    for (int kh = 0; kh < g.KH; ++kh)
      for (int kw = 0; kw < g.KW; ++kw) D[((long)rg * g.N + n) * K + (long)(kh * g.KW + kw) * g.Cin + c] = d[kh * 5 + kw];
  }
}

// partial[chunk][i][l] = sum over the chunk's samples (at most 8) of yb[n][l] * D[n][i]   (i = (kh, kw, c), K of them; c2 <= 16)
// (D arrives as WGRAD_LABEL_RG row-group partials [rg][n][K], summed here in a fixed order; every load of the chunk in flight at once)
#define WGRAD_LABEL_PER_CHUNK 8
__global__ __launch_bounds__(128) void wgrad_label_cols_kernel(const float* D, const float* yb, int N, long K, int c2, float* partial) {
  __shared__ float yb_s[WGRAD_LABEL_PER_CHUNK * 16];
  const int n0 = blockIdx.y * WGRAD_LABEL_PER_CHUNK;
  for (int e = threadIdx.x; e < WGRAD_LABEL_PER_CHUNK * c2; e += blockDim.x) {
    const int q = e / c2, l = e - q * c2;
    yb_s[q * 16 + l] = n0 + q < N ? yb[(long)(n0 + q) * c2 + l] : 0.f;
  }
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K) return;
  float dv[WGRAD_LABEL_PER_CHUNK][WGRAD_LABEL_RG];
#pragma unroll
  for (int q = 0; q < WGRAD_LABEL_PER_CHUNK; ++q)
#pragma unroll
    for (int rg = 0; rg < WGRAD_LABEL_RG; ++rg) dv[q][rg] = n0 + q < N ? D[((long)rg * N + n0 + q) * K + i] : 0.f;
  float acc[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) acc[l] = 0.f;
#pragma unroll
  for (int q = 0; q < WGRAD_LABEL_PER_CHUNK; ++q) {
    float d = 0.f;
#pragma unroll
    for (int rg = 0; rg < WGRAD_LABEL_RG; ++rg) d += dv[q][rg];
#pragma unroll
    for (int l = 0; l < 16; ++l)
      if (l < c2) acc[l] += yb_s[q * 16 + l] * d;
  }
#pragma unroll
  for (int l = 0; l < 16; ++l)
    if (l < c2) partial[((long)blockIdx.y * K + i) * c2 + l] = acc[l];
}

// out[i][j] (= or +=)  j < c1: sum_z slab[z][i][j] (slabs [K][c1])   |   j >= c1: sum_q partial[q][i][j - c1]      (out rows of c1 + c2 floats)
// one thread per four real columns (16-byte slab loads; c1 % 4 == 0) or per label column
__global__ void slab_reduce_cols_kernel(const float* slab, int nz, const float* partial, int nq, float* out, long K, int c1, int c2, int accumulate) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = (c1 >> 2) + c2, ld = c1 + c2;
  if (e >= K * per_row) return;
  const long i = e / per_row;
  const int q = (int)(e - i * per_row);
  if (q < (c1 >> 2)) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < nz; ++z) {
      const float4 t = *(const float4*)(slab + ((long)z * K + i) * c1 + 4 * q);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    float* o = out + i * ld + 4 * q;         // (rows of c1 + c2 floats: 8-byte aligned at best)
    if (accumulate) { s.x += o[0]; s.y += o[1]; s.z += o[2]; s.w += o[3]; }
    o[0] = s.x; o[1] = s.y; o[2] = s.z; o[3] = s.w;
  } else {
    const int l = q - (c1 >> 2);
    float s = 0.f;
    for (int z = 0; z < nq; ++z) s += partial[((long)z * K + i) * c2 + l];
    float* o = out + i * ld + c1 + l;
    *o = accumulate ? *o + s : s;
  }
}

size_t direct_wgrad_cols_ws_bytes(const rcgan_conv_desc* d, int c1) {
  ConvGeom g = make_geom(d);
  const long K = (long)g.KH * g.KW * g.Cin, M = (long)g.N * g.OH * g.OW;
  const int c2 = g.Cout - c1;
  const int nz = wgrad_splits(K, c1, M);
  return ((size_t)nz * K * c1 + (size_t)WGRAD_LABEL_RG * g.N * K + (size_t)cdiv(g.N, WGRAD_LABEL_PER_CHUNK) * K * c2) * sizeof(float) + 1024;
}

// dy rows: [c1 real columns ; c2 = Cout - c1 label columns = yb[n][:]]  (yb: fp32 [N][c2], c2 <= 16; KH, KW <= 5)
template <typename T>
int direct_wgrad_cols(rcgan_ctx* ctx, const rcgan_conv_desc* d, const T* x, const T* dy, int c1, const float* yb, float* dw, int accumulate,
                      void* ws, size_t ws_bytes) {
  WgradOp<T> op;
  op.g = make_geom(d); op.x = x; op.dy = dy; op.wscale = nullptr;
  const int c2 = op.g.Cout - c1;
  if (c1 <= 0 || (c1 & 3) || c2 <= 0 || c2 > 16 || op.g.KH > 5 || op.g.KW > 5 || op.g.W > 32 || op.g.H > 32 || op.g.up || yb == nullptr)
    RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "label-column filter gradient: %d + %d columns, %d x %d filter", c1, c2, op.g.KH, op.g.KW);
  const long K = (long)op.g.KH * op.g.KW * op.g.Cin, M = (long)op.g.N * op.g.OH * op.g.OW;
  int nz = wgrad_splits(K, c1, M);
  const size_t need = direct_wgrad_cols_ws_bytes(d, c1);
  if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
  float* const slab = (float*)ws;
  float* const Dn = slab + (size_t)nz * K * c1;
  float* const partial = Dn + (size_t)WGRAD_LABEL_RG * op.g.N * K;
  // the label columns' two small launches first (they only read x and yb), then the GEMM over the real columns
  LabelMasks mk;
  for (int t = 0; t < 5; ++t) {
    mk.col[t] = 0; mk.row[t] = 0;
    for (int p = 0; p < 32; ++p) {
      const int qc = p + op.g.PL - t, qr = p + op.g.PT - t;
      if (t < op.g.KW && p < op.g.W && qc >= 0 && qc % op.g.S == 0 && qc / op.g.S < op.g.OW) mk.col[t] |= 1u << p;
      if (t < op.g.KH && p < op.g.H && qr >= 0 && qr % op.g.S == 0 && qr / op.g.S < op.g.OH) mk.row[t] |= 1u << p;
    }
  }
  if (op.g.W <= 16) hipLaunchKernelGGL((wgrad_label_sums_kernel<T, 16>), dim3(op.g.N, WGRAD_LABEL_RG), dim3(128), 0, ctx->stream, op.g, mk, x, Dn);
  else hipLaunchKernelGGL((wgrad_label_sums_kernel<T, 32>), dim3(op.g.N, WGRAD_LABEL_RG), dim3(128), 0, ctx->stream, op.g, mk, x, Dn);
  RC_LAUNCH_CHECK(ctx);
  const int nq = cdiv(op.g.N, WGRAD_LABEL_PER_CHUNK);
  hipLaunchKernelGGL(wgrad_label_cols_kernel, dim3(cdiv(K, 128), nq), dim3(128), 0, ctx->stream, (const float*)Dn, yb, op.g.N, K, c2, partial);
  RC_LAUNCH_CHECK(ctx);
  op.slab = slab;
  op.M = K; op.N = c1; op.R = M;               // (the row stride of dy stays g.Cout: b4 reads columns j < N of a Cout-wide row)
  op.avec = vec_of(x, op.g.Cin); op.bvec = vec_of(dy, op.g.Cout);
  op.r_chunk = ((M + nz - 1) / nz + 15) / 16 * 16;
  nz = cdiv(M, op.r_chunk);
  int rc = launch_gemm(ctx, op, nz);
  if (rc) return rc;
  const long cnt = K * ((c1 >> 2) + c2);
  hipLaunchKernelGGL(slab_reduce_cols_kernel, dim3(cdiv(cnt, 256)), dim3(256), 0, ctx->stream, (const float*)slab, nz, (const float*)partial, nq, dw, K, c1, c2,
                     accumulate);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}
template int direct_wgrad_cols<float>(rcgan_ctx*, const rcgan_conv_desc*, const float*, const float*, int, const float*, float*, int, void*, size_t);
template int direct_wgrad_cols<bf16_t>(rcgan_ctx*, const rcgan_conv_desc*, const bf16_t*, const bf16_t*, int, const float*, float*, int, void*, size_t);

template int direct_wgrad<float>(rcgan_ctx*, const rcgan_conv_desc*, const float*, const float*, float*, float*, int, void*, size_t);
template int direct_wgrad<bf16_t>(rcgan_ctx*, const rcgan_conv_desc*, const bf16_t*, const bf16_t*, float*, float*, int, void*, size_t);
