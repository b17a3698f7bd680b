// Image-end convolutions on the matrix cores (bf16): the layers with a 3-channel side -- D.Block.1.Conv1 /
// Shortcut (3 -> 128) and G.Output (256 -> 3) of the CIFAR nets (cifar10/gan_resnet.py:337-352, :368) -- and
// their gradients.  None of them has enough arithmetic to matter (K = 27 or N = 3); all are bound by streaming
// the big-channel tensor once, so every kernel here reads/writes that tensor exactly once, coalesced, and pads
// the small side to one MFMA tile (K = 32 or N = 16):
//   small-reduction  out[m][n<Cb]  = sum_{k<32} col[m][k] * wK[n][k]          (forward of 3->Cb, dX of Cb->3)
//   small-output     out[m][n<Cs]  = sum_{t,c<Cb} in[pix(m,t)][c] * wS[t][n][c] (forward of Cb->3, dX of 3->Cb)
//   small-side wgrad dW[k<32][n<Cb] = sum_m col[m][k] * big[m][n]             (+ bias gradients)
// col[m][k = t*Cs + c] is the im2col row of the <=3-channel tensor (zero outside the image), gathered with
// scalar loads -- that tensor is ~1 MB and lives in L2.
// The fp32 path and the shapes these kernels do not take (non power-of-two images, Cs = 4) stay on conv_small.hip.
#include "conv_mfma.h"
#include "mfma_util.h"
#include "conv_image.h"

// ---------------------------------------------------------------------------------------------------------
// eligibility + prepared-filter layouts
// ---------------------------------------------------------------------------------------------------------
int img_side(const rcgan_conv_desc* d) {
  if (d->dtype != RCGAN_H16 || d->stride != 1 || d->kh != d->kw || (d->kh != 1 && d->kh != 3)) return 0;
  if (d->flags & (RCGAN_CONV_IN_UPSAMPLE2X | RCGAN_CONV_FORCE_DIRECT)) return 0;
  const int lw = ilog2_exact(d->w), lh = ilog2_exact(d->h);
  if (lw < 3 || lw > 5 || lh < 0 || d->h * d->w < 256) return 0;      // W in {8,16,32}; whole 256-pixel bands
  const int T = d->kh * d->kw;
  if (d->cin <= 3 && T * d->cin <= 31 && (d->cout == 128 || d->cout == 256) && !(d->flags & RCGAN_CONV_IN_RELU)) return 1;
  if (d->cout <= 3 && T * d->cout <= 31 && (d->cin == 128 || d->cin == 256)) return 2;
  return 0;
}

size_t img_extra_offset(const rcgan_conv_desc* d) {
  size_t b = (size_t)d->kh * d->kw * d->cin * d->cout * sizeof(float);
  return (b + 255) / 256 * 256;
}

size_t img_extra_bytes(const rcgan_conv_desc* d) {
  const int cb = d->cin <= 3 ? d->cout : d->cin;
  return ((size_t)cb * 32 + (size_t)d->kh * d->kw * 16 * cb) * sizeof(bf16_t) + 256;
}

__global__ void img_prepare_kernel(int side, int T, int Cin, int Cout, const float* w, const float* sigma, bf16_t* extra) {
  const int Cb = side == 1 ? Cout : Cin;
  const long total = (long)Cb * 32 + (long)T * 16 * Cb;
  const float inv = sigma ? 1.f / *sigma : 1.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    extra[e] = img_prepare_elem(e, side, T, Cin, Cout, w, inv);
}

int img_prepare_launch(rcgan_ctx* ctx, const rcgan_conv_desc* d, const float* w, const float* sigma, void* prepared) {
  const int side = img_side(d);
  if (!side) return RCGAN_OK;
  const int T = d->kh * d->kw, cb = side == 1 ? d->cout : d->cin;
  const long total = (long)cb * 32 + (long)T * 16 * cb;
  hipLaunchKernelGGL(img_prepare_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, side, T, d->cin, d->cout, w, sigma,
                     (bf16_t*)((char*)prepared + img_extra_offset(d)));
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// small reduction: out[m][n] = sum_k col[m][k] * wK[n][k].  One wavefront per 16-pixel tile; the filter
// fragments (Cb/16 x 4 registers) stay in registers.  Filter rows are permuted so that a lane ends up with
// 8 consecutive channels of its pixel in two accumulators: one 16-byte store.
// ---------------------------------------------------------------------------------------------------------
struct ImgKArgs {
  ColGeom g;
  const bf16_t* wK;      // [Cb][32]
  const float* bias;     // [Cb] or null
  bf16_t* out;           // [M][Cb]
  int Cb, accumulate;
};

template <int NF>      // Cb / 16
__global__ __launch_bounds__(256) void conv_img_small_red_kernel(ImgKArgs a) {
  const int lane = threadIdx.x & 63, g4 = lane >> 4, px = lane & 15;
  // A operand (filter): fragment i, MFMA row j = lane&15, k = 8*g4..+7.  Row j of fragment pair (2q, 2q+1) maps to
  // channel q*32 + (j>>2)*8 + (i&1)*4 + (j&3): D rows 4*g4..+3 of the pair are channels q*32 + g4*8 + 0..7.
  bf16x8_t wf[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int ch = (i >> 1) * 32 + (px >> 2) * 8 + (i & 1) * 4 + (px & 3);
    wf[i] = *(const bf16x8_t*)(a.wK + ch * 32 + g4 * 8);
  }
  int dh[8], dw[8], cc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) col_tap(a.g, g4 * 8 + e, dh[e], dw[e], cc[e]);
  float bv[NF / 2][8];
#pragma unroll
  for (int q = 0; q < NF / 2; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[q][j] = a.bias ? a.bias[q * 32 + g4 * 8 + j] : 0.f;

  const long ntile = (a.g.M + 15) >> 4;
  const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (long)gridDim.x * 4;
  uint32_t vn[8];
  if (wave0 < ntile) {
#pragma unroll
    for (int e = 0; e < 8; ++e) vn[e] = col_load(a.g, wave0 * 16 + px, dh[e], dw[e], cc[e]);
  }
  for (long tile = wave0; tile < ntile; tile += nwave) {
    const long m = tile * 16 + px;
    uint32_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = vn[e];
    if (tile + nwave < ntile) {          // the next tile's columns are requested before this tile's stores
#pragma unroll
      for (int e = 0; e < 8; ++e) vn[e] = col_load(a.g, (tile + nwave) * 16 + px, dh[e], dw[e], cc[e]);
    }
    const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)));
    {   // M is a multiple of 256 on this path (whole 16-pixel tiles): the MFMAs always run with every lane active
      bf16_t* orow = a.out + m * a.Cb + g4 * 8;
#pragma unroll
      for (int q = 0; q < NF / 2; ++q) {
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        const f32x4_t lo = mfma16(wf[2 * q], xf, z);
        const f32x4_t hi = mfma16(wf[2 * q + 1], xf, z);
        float o[8] = {lo[0] + bv[q][0], lo[1] + bv[q][1], lo[2] + bv[q][2], lo[3] + bv[q][3],
                      hi[0] + bv[q][4], hi[1] + bv[q][5], hi[2] + bv[q][6], hi[3] + bv[q][7]};
        uint4* dst = (uint4*)(orow + q * 32);
        if (a.accumulate) {
          const uint4 p = *dst;
          const uint32_t w[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) { o[2 * j] += bf16_to_f32((bf16_t)(w[j] & 0xffff)); o[2 * j + 1] += bf16_to_f32((bf16_t)(w[j] >> 16)); }
        }
        uint4 pk;
        pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
        pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
        pk.z = (uint32_t)f32_to_bf16(o[4]) | ((uint32_t)f32_to_bf16(o[5]) << 16);
        pk.w = (uint32_t)f32_to_bf16(o[6]) | ((uint32_t)f32_to_bf16(o[7]) << 16);
        *dst = pk;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// small output: out[m][n<Cs] = sum_{t, c<Cb} in[pix(m,t)][c] * wS[t][n][c].
// One workgroup = a band of 256 pixels (256/W image rows) of one image.  Per 64-channel chunk the band WITH its
// halo (zero page outside the image) and the chunk's filter rows are deposited in LDS by LDS-DMA (double
// buffered); the 9 taps are row offsets of the fragment reads, so the big tensor is fetched once instead of once
// per tap.  16-B slot s of LDS row r holds channel chunk s ^ (r & 7) (source-side swizzle).
// ---------------------------------------------------------------------------------------------------------
struct ImgSArgs {
  const bf16_t* in;      // [N][H][W][Cb]
  const bf16_t* wS;      // [TT][16][Cb]
  const float* bias;     // [Cs] or null
  bf16_t* out;           // [N][H][W][Cs]
  const bf16_t* zero;
  int N, H, W, lw, lh, Cb, Cs, relu_in, accumulate;
  // optional: the (conditional) batch norm + activation in FRONT of the convolution applied to the staged pixels in LDS -- the
  // normalised tensor is never written (rcgan_conv2d_fwd_bn).  mean / rstd: [segments][Cb]; gamma / beta: [n_labels][Cb]
  const float *bn_mean, *bn_rstd, *bn_gamma, *bn_beta;
  const int32_t* bn_labels;         // [N] or null (label 0)
  int bn_seg_samples, bn_act;
};

// the batch-norm affine exactly as bn.hip applies it (bn_pre / bn_c0 there): inv = rstd*gamma, c0 = fma(-mean, inv, beta), pre = fma(x, inv, c0)
__device__ __forceinline__ float img_bn_c0(float mean, float inv, float beta) { return __fmaf_rn(-mean, inv, beta); }
__device__ __forceinline__ float img_bn_pre(float x, float inv, float c0) { return __fmaf_rn(x, inv, c0); }

// NST: stages per workgroup.  2 = the next chunk's DMA runs under this chunk's MFMAs, one workgroup per CU (128 KB of LDS);
// 1 = one stage, TWO workgroups per CU that cover each other's DMA round trips and in-LDS transforms.
template <int TT, int NST = 2>
__global__ __launch_bounds__(256) void conv_img_small_out_kernel(ImgSArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HALO = TT == 9 ? 1 : 0;
  constexpr int XROWS = TT == 9 ? 352 : 256;      // LDS pixel rows per stage (>= (R+2)*(W+2)), 128 B each
  constexpr int NXI = XROWS / 32;                 // pixel deposits per wavefront (8 rows each)
  constexpr int NWI = TT == 9 ? 5 : 1;            // filter deposits per wavefront
  constexpr int STAGE = (XROWS + NWI * 32) * 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R = 256 >> a.lw, bands = a.H / R;
  const int b = blockIdx.x / bands, r0 = (blockIdx.x - b * bands) * R;
  const int cols = a.W + 2 * HALO, npt = (R + 2 * HALO) * cols;

  int xoff[NXI], woff[NWI];
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    const int tr = (i * 4 + wave) * 8 + (lane >> 3);
    xoff[i] = -1;
    if (tr < npt) {
      const int ty = tr / cols, tx = tr - ty * cols;
      const int ih = r0 + ty - HALO, iw = tx - HALO;
      if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
        xoff[i] = ((b * a.H + ih) * a.W + iw) * a.Cb + (((lane & 7) ^ (tr & 7)) << 3);
    }
  }
#pragma unroll
  for (int i = 0; i < NWI; ++i) {
    const int wr = (i * 4 + wave) * 8 + (lane >> 3);
    woff[i] = wr < TT * 16 ? wr * a.Cb + (((lane & 7) ^ (wr & 7)) << 3) : -1;
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // staged-pixel transform (below): 0 = none, 1 = input ReLU, 2 = batch norm + activation.  Tables inv[Cb], c0[Cb] behind the stages.
  const int xmode = a.bn_mean ? 2 : (a.relu_in ? 1 : 0);
  float* const bn_s = (float*)(smem + NST * STAGE);
  if (xmode == 2) {
    const int lab = a.bn_labels ? a.bn_labels[b] : 0, seg = b / a.bn_seg_samples;
    for (int c = tid; c < a.Cb; c += 256) {
      const float inv = a.bn_rstd[seg * a.Cb + c] * a.bn_gamma[lab * a.Cb + c];
      bn_s[c] = inv;
      bn_s[a.Cb + c] = img_bn_c0(a.bn_mean[seg * a.Cb + c], inv, a.bn_beta[lab * a.Cb + c]);
    }
    __syncthreads();
  }
  auto issue = [&](int c0, int buf) {
    const unsigned stage = lds0 + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NXI; ++i) glds16_asm(xoff[i] >= 0 ? a.in + xoff[i] + c0 : a.zero, stage + (i * 4 + wave) * 1024);
#pragma unroll
    for (int i = 0; i < NWI; ++i) glds16_asm(woff[i] >= 0 ? a.wS + woff[i] + c0 : a.zero, stage + XROWS * 128 + (i * 4 + wave) * 1024);
  };

  // fragment geometry: pixel fragment f of this wavefront = band pixels wave*64 + f*16 + (lane&15)
  const int px = lane & 15, g4 = lane >> 4;
  int tr0[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int p = wave * 64 + f * 16 + px;
    tr0[f] = ((p >> a.lw) + HALO) * cols + (p & (a.W - 1)) + HALO;
  }
  f32x4_t acc[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) acc[f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nch = a.Cb / 64;
  issue(0, 0);
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = NST == 2 ? (ch & 1) : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (xmode) {
      // The input ReLU / batch norm ONCE per staged element instead of once per tap on the fragments (nine taps read every pixel:
      // 288 packed-max instructions per chunk and wavefront sat in the MFMA loop).  Every thread transforms exactly the slots its own
      // DMA lanes deposited (landed: vmcnt(0) above): slot (lane & 7) of row tr holds source chunk (lane & 7) ^ (tr & 7), and
      // tr & 7 = lane >> 3 for every deposit of this thread -- the same 8 channels each time.  Halo slots (zero page) stay zero.
      unsigned char* const st = smem + buf * STAGE;
      const int cch = ch * 64 + (((lane & 7) ^ (lane >> 3)) << 3);
      float inv8[8], c08[8];
      const float relu_floor = a.bn_act == RCGAN_ACT_RELU ? 0.f : -INFINITY;
      if (xmode == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { inv8[j] = bn_s[cch + j]; c08[j] = bn_s[a.Cb + cch + j]; }
      }
#pragma unroll
      for (int i = 0; i < NXI; ++i) {
        if (xoff[i] < 0) continue;
        uint4* const slot = (uint4*)(st + (i * 4 + wave) * 1024 + lane * 16);
        uint4 v = *slot;
        if (xmode == 1) {
          v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w);
        } else {
          uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // (ReLU or none -- img_fwd_bn refuses the rest: act_apply's run-time switch in here cost more than the pass it replaces)
            float lo = img_bn_pre(h16_lo(w4[q]), inv8[2 * q], c08[2 * q]);
            float hi = img_bn_pre(h16_hi(w4[q]), inv8[2 * q + 1], c08[2 * q + 1]);
            lo = fmaxf(lo, relu_floor); hi = fmaxf(hi, relu_floor);
            w4[q] = pack_h16x2(lo, hi);
          }
          v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
        *slot = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (NST == 2 && ch + 1 < nch) issue((ch + 1) * 64, buf ^ 1);
    const unsigned char* xs = smem + buf * STAGE;
    const unsigned char* ws = xs + XROWS * 128;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const int dtr = TT == 9 ? (t / 3 - 1) * cols + (t % 3 - 1) : 0;
      const int wr = t * 16 + px;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int kc = ks * 4 + g4;
        const bf16x8_t wf = *(const bf16x8_t*)(ws + wr * 128 + ((kc ^ (wr & 7)) << 4));
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int tr = tr0[f] + dtr;
          acc[f] = mfma16(wf, *(const bf16x8_t*)(xs + tr * 128 + ((kc ^ (tr & 7)) << 4)), acc[f]);
        }
      }
    }
    if (NST == 1 && ch + 1 < nch) {        // one stage: everybody has read it before the next chunk lands in it
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      issue((ch + 1) * 64, 0);
    }
  }
  // D[row = n][col = pixel]: lanes 0..15 hold n = 0..3 of their pixel
  if (g4 == 0) {
    const long mbase = ((long)b * a.H + r0) * a.W + wave * 64;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      bf16_t* o = a.out + (mbase + f * 16 + px) * a.Cs;
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        if (n < a.Cs) {
          float v = acc[f][n] + (a.bias ? a.bias[n] : 0.f);
          if (a.accumulate) v += bf16_to_f32(o[n]);
          o[n] = f32_to_bf16(v);
        }
      }
    }
  }
}

// dW (HWIO) and the bias gradient from the workgroup slabs.  thread block = 16 outputs x 16 slab lanes.
//   orient 0 (small = conv input):   dW[k*Cb + n],  k = t*Cs + c;  dbias[n<Cb] = row 31 (ones column)
//   orient 1 (small = dy):           dW[(t*Cb + n)*Cs + c];        dbias[c<Cs] = column sums of the centre tap
__global__ __launch_bounds__(256) void conv_img_wgrad_reduce_kernel(const float* slab, int nblk, int Cb, int Cs, int TT, int orient,
                                                                    float* dw, float* dbias, int accumulate) {
  __shared__ float red[16][17];
  const int per = 32 * Cb + 32;
  const int i = blockIdx.x * 16 + (threadIdx.x & 15), bl = threadIdx.x >> 4;
  float s = 0.f;
  if (i < per)
    for (int b = bl; b < nblk; b += 16) s += slab[(long)b * per + i];
  red[bl][threadIdx.x & 15] = s;
  __syncthreads();
  if (bl != 0 || i >= per) return;
  s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += red[q][threadIdx.x];
  const int K = TT * Cs;
  float* o = nullptr;
  if (i < 32 * Cb) {
    const int k = i / Cb, n = i - k * Cb;
    if (k < K) {
      const int t = k / Cs, c = k - t * Cs;
      o = orient == 0 ? dw + (long)k * Cb + n : dw + ((long)t * Cb + n) * Cs + c;
    } else if (orient == 0 && k == 31 && dbias) {
      o = dbias + n;
    }
  } else if (orient == 1 && dbias) {
    const int k = i - 32 * Cb, ctr = TT == 9 ? 4 : 0;
    if (k >= ctr * Cs && k < ctr * Cs + Cs) o = dbias + (k - ctr * Cs);
  }
  if (!o) return;
  if (accumulate) s += *o;
  *o = s;
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static ColGeom col_geom(const rcgan_conv_desc* d, const bf16_t* s, int Cs, int sign, int pt, int pl) {
  ColGeom g;
  g.s = s; g.H = d->h; g.W = d->w; g.lw = ilog2_exact(d->w); g.lh = ilog2_exact(d->h); g.Cs = Cs; g.TT = d->kh * d->kw;
  g.PT = pt; g.PL = pl; g.sign = sign; g.M = (long)d->n * d->h * d->w;
  return g;
}

static const bf16_t* img_wk(const rcgan_conv_desc* d, const void* prepared) { return (const bf16_t*)((const char*)prepared + img_extra_offset(d)); }
static const bf16_t* img_ws(const rcgan_conv_desc* d, const void* prepared) {
  return img_wk(d, prepared) + (size_t)(d->cin <= 3 ? d->cout : d->cin) * 32;
}

static int launch_small_red(rcgan_ctx* ctx, ImgKArgs& a) {
  long tiles = (a.g.M + 15) / 16;
  long blocks = (tiles + 3) / 4;
  static const int maxb = [] { const char* e = getenv("RCGAN_IMG_RED_MAXBLK"); return e ? atoi(e) : 512; }();
  // (512 workgroups whose wavefronts walk four tiles each with the next tile's columns in flight, instead of one short-lived
  // wavefront per tile: 6.62 -> 6.585 ms per iteration)
  if (blocks > maxb) blocks = maxb;
  if (a.Cb == 128) hipLaunchKernelGGL(conv_img_small_red_kernel<8>, dim3((int)blocks), dim3(256), 0, ctx->stream, a);
  else hipLaunchKernelGGL(conv_img_small_red_kernel<16>, dim3((int)blocks), dim3(256), 0, ctx->stream, a);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

static int launch_small_out(rcgan_ctx* ctx, const rcgan_conv_desc* d, ImgSArgs& a) {
  a.N = d->n; a.H = d->h; a.W = d->w; a.lw = ilog2_exact(d->w); a.lh = ilog2_exact(d->h);
  a.zero = (const bf16_t*)ctx->zero_page;
  const int R = 256 / d->w;
  dim3 grid(d->n * (d->h / R));
  // one stage, two workgroups per CU (default): G.Output forward 30.7 -> 22.0 us, D.Block.1.Conv1's data gradient 17.8 -> 11.9 us at n = 128,
  // the critic steps' generator forwards 1.301 -> 1.280 ms (profiles/r03_microbench.txt); RCGAN_IMG_OUT_STAGES=2: double-buffered, one per CU
  static const int nst = [] { const char* e = getenv("RCGAN_IMG_OUT_STAGES"); return e ? atoi(e) : 1; }();
  if (d->kh == 3 && nst == 1) {
    static bool attr = false;
    const size_t lds = (size_t)(352 + 160) * 128 + 2 * 256 * sizeof(float);
    if (!attr) { RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_img_small_out_kernel<9, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
    hipLaunchKernelGGL((conv_img_small_out_kernel<9, 1>), grid, dim3(256), lds, ctx->stream, a);
  } else if (d->kh == 3) {
    static bool attr = false;
    const size_t lds = (size_t)2 * (352 + 160) * 128 + 2 * 256 * sizeof(float);
    if (!attr) { RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_img_small_out_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
    hipLaunchKernelGGL(conv_img_small_out_kernel<9>, grid, dim3(256), lds, ctx->stream, a);
  } else {
    static bool attr = false;
    const size_t lds = (size_t)2 * (256 + 32) * 128 + 2 * 256 * sizeof(float);
    if (!attr) { RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_img_small_out_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
    hipLaunchKernelGGL(conv_img_small_out_kernel<1>, grid, dim3(256), lds, ctx->stream, a);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int img_fwd(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias, void* y) {
  int pt, pl, oh, ow;
  same_pad(d->h, d->kh, 1, &oh, &pt);
  same_pad(d->w, d->kw, 1, &ow, &pl);
  const int acc = (d->flags & RCGAN_CONV_ACCUMULATE) ? 1 : 0;
  if (img_side(d) == 1) {
    ImgKArgs a;
    a.g = col_geom(d, (const bf16_t*)x, d->cin, +1, pt, pl);
    a.wK = img_wk(d, prepared); a.bias = bias; a.out = (bf16_t*)y; a.Cb = d->cout; a.accumulate = acc;
    return launch_small_red(ctx, a);
  }
  ImgSArgs a = {};
  a.in = (const bf16_t*)x; a.wS = img_ws(d, prepared); a.bias = bias; a.out = (bf16_t*)y;
  a.Cb = d->cin; a.Cs = d->cout; a.relu_in = (d->flags & RCGAN_CONV_IN_RELU) ? 1 : 0; a.accumulate = acc;
  return launch_small_out(ctx, d, a);
}

// y = conv(act(batch_norm(x))) with the normalisation applied to the staged pixels (ImgSArgs::bn_*): the small-output side only
bool img_fwd_bn_ok(const rcgan_conv_desc* d) {
  return img_side(d) == 2 && d->cin % 64 == 0 && d->cin <= 256 && (d->flags & ~RCGAN_CONV_ACCUMULATE) == 0;
}
int img_fwd_bn(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* prepared, const float* bias, void* y,
               const float* mean, const float* rstd, const float* gamma, const float* beta, const int32_t* labels, int segments, int act) {
  RC_REQUIRE(ctx, img_fwd_bn_ok(d), "not a small-output image-end shape");
  RC_REQUIRE(ctx, mean && rstd && gamma && beta && segments >= 1 && d->n % segments == 0, "bad batch-norm arguments");
  RC_REQUIRE(ctx, act == RCGAN_ACT_NONE || act == RCGAN_ACT_RELU, "activation %d: ReLU or none", act);
  ImgSArgs a = {};
  a.in = (const bf16_t*)x; a.wS = img_ws(d, prepared); a.bias = bias; a.out = (bf16_t*)y;
  a.Cb = d->cin; a.Cs = d->cout; a.relu_in = 0; a.accumulate = (d->flags & RCGAN_CONV_ACCUMULATE) ? 1 : 0;
  a.bn_mean = mean; a.bn_rstd = rstd; a.bn_gamma = gamma; a.bn_beta = beta; a.bn_labels = labels;
  a.bn_seg_samples = d->n / segments; a.bn_act = act;
  return launch_small_out(ctx, d, a);
}

// dX = conv of dY with the rotated filter (gather offsets PT' = k-1-PT); no input-ReLU mask here (the caller keeps
// masked data gradients on the generic path)
int img_dgrad(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* dy, const void* prepared, void* dx, int accumulate) {
  int pt, pl, oh, ow;
  same_pad(d->h, d->kh, 1, &oh, &pt);
  same_pad(d->w, d->kw, 1, &ow, &pl);
  pt = d->kh - 1 - pt; pl = d->kw - 1 - pl;
  if (img_side(d) == 2) {            // dX[m][Cin big] from dY[m][Cout small]
    ImgKArgs a;
    a.g = col_geom(d, (const bf16_t*)dy, d->cout, +1, pt, pl);
    a.wK = img_wk(d, prepared); a.bias = nullptr; a.out = (bf16_t*)dx; a.Cb = d->cin; a.accumulate = accumulate;
    return launch_small_red(ctx, a);
  }
  if (pt != (d->kh == 3 ? 1 : 0) || pl != pt) RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "asymmetric padding");
  ImgSArgs a = {};                   // dX[m][Cin small] from dY[m][Cout big]
  a.in = (const bf16_t*)dy; a.wS = img_ws(d, prepared); a.bias = nullptr; a.out = (bf16_t*)dx;
  a.Cb = d->cout; a.Cs = d->cin; a.relu_in = 0; a.accumulate = accumulate;
  return launch_small_out(ctx, d, a);
}

// standalone launch of the filter-gradient body (conv_image.h); in a backward pass the same body rides in the grouped launches
template <int CB>
__global__ __launch_bounds__(256) void conv_img_wgrad_kernel(ImgWArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  img_wgrad_body<CB>(a, blockIdx.x, smem);
}

// pixel blocks per workgroup for about `target` workgroups; workgroups = slabs
static void img_wgrad_split(const rcgan_conv_desc* d, int target, int* nsub, int* nwg) {
  const int cb = d->cin <= 3 ? d->cout : d->cin;
  const long M = (long)d->n * d->h * d->w, pb = 16384 / cb, blocks = (M + pb - 1) / pb;
  long ns = (blocks + target - 1) / target;
  if (ns < 1) ns = 1;
  *nsub = (int)ns;
  *nwg = (int)((blocks + ns - 1) / ns);
}

#define IMG_WGRAD_ALONE_WGS 512
size_t img_wgrad_ws_bytes(const rcgan_conv_desc* d) {
  const int cb = d->cin <= 3 ? d->cout : d->cin;
  int nsub, nwg;
  img_wgrad_split(d, IMG_WGRAD_ALONE_WGS, &nsub, &nwg);
  return (size_t)nwg * (32 * (size_t)cb + 32) * sizeof(float) + 256;
}

// Kernel arguments of one image-end filter gradient cut into about target_wgs workgroups (a->slab is left to the caller:
// *nwg slabs of 32*cb + 32 floats)
int img_wgrad_plan(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* dy, int target_wgs, ImgWArgs* a, int* cb_out, int* nwg) {
  int pt, pl, oh, ow;
  same_pad(d->h, d->kh, 1, &oh, &pt);
  same_pad(d->w, d->kw, 1, &ow, &pl);
  const int side = img_side(d);
  RC_REQUIRE(ctx, side != 0, "not an image-end convolution");
  const int cb = side == 1 ? d->cout : d->cin, cs = side == 1 ? d->cin : d->cout;
  // side 1: col = im2col(x) (+), big = dy;  side 2: col = dy gathered at (oh - kh + PT), big = x
  a->g = side == 1 ? col_geom(d, (const bf16_t*)x, cs, +1, pt, pl) : col_geom(d, (const bf16_t*)dy, cs, -1, pt, pl);
  a->big = side == 1 ? (const bf16_t*)dy : (const bf16_t*)x;
  a->zero = (const bf16_t*)ctx->zero_page;
  a->slab = nullptr;
  a->ones_col = side == 1 ? 1 : 0;
  a->relu_big = (side == 2 && (d->flags & RCGAN_CONV_IN_RELU)) ? 1 : 0;
  img_wgrad_split(d, target_wgs, &a->nsub, nwg);
  *cb_out = cb;
  return RCGAN_OK;
}

int img_wgrad(rcgan_ctx* ctx, const rcgan_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, int accumulate,
              void* ws, size_t ws_bytes) {
  const size_t need = img_wgrad_ws_bytes(d);
  if (ws_bytes < need) RC_FAIL(ctx, RCGAN_EWORKSPACE_TOO_SMALL, "need %zu have %zu", need, ws_bytes);
  ImgWArgs a;
  int cb, nwg;
  int rc = img_wgrad_plan(ctx, d, x, dy, IMG_WGRAD_ALONE_WGS, &a, &cb, &nwg);
  if (rc) return rc;
  a.slab = (float*)ws;
  const int side = img_side(d), cs = side == 1 ? d->cin : d->cout;
  if (cb == 128) {
    hipLaunchKernelGGL(conv_img_wgrad_kernel<128>, dim3(nwg), dim3(256), ImgWGeom<128>::LDS, ctx->stream, a);
  } else {
    hipLaunchKernelGGL(conv_img_wgrad_kernel<256>, dim3(nwg), dim3(256), ImgWGeom<256>::LDS, ctx->stream, a);
  }
  RC_LAUNCH_CHECK(ctx);
  const int per = 32 * cb + 32;
  hipLaunchKernelGGL(conv_img_wgrad_reduce_kernel, dim3(cdiv(per, 16)), dim3(256), 0, ctx->stream, (const float*)a.slab, nwg, cb, cs,
                     d->kh * d->kw, side == 1 ? 0 : 1, dw, dbias, accumulate);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}
