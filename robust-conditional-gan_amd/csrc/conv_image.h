// Device code of the image-end filter gradient shared by conv_image.hip (its own launch) and conv_mfma.hip (as extra
// workgroups of the grouped filter-gradient launches): im2col geometry of the <= 3-channel tensor and the workgroup body.
#pragma once
#include "mfma_util.h"

// ---------------------------------------------------------------------------------------------------------
// im2col geometry of the small tensor
// ---------------------------------------------------------------------------------------------------------
struct ColGeom {
  const bf16_t* s;       // [N][H][W][Cs]
  int H, W, lw, lh, Cs, TT, PT, PL, sign;   // sign +1: (oh + kh - PT, ow + kw - PL);  -1: (oh - kh + PT, ow - kw + PL)
  long M;
};

// (row offset, column offset, channel) of im2col column k; dh = 1 << 20 marks a padding column (k >= TT*Cs)
__device__ __forceinline__ void col_tap(const ColGeom& g, int k, int& dh, int& dw, int& c) {
  if (k >= g.TT * g.Cs) { dh = 1 << 20; dw = 0; c = 0; return; }
  const int t = k / g.Cs;
  c = k - t * g.Cs;
  const int kh = g.TT == 9 ? t / 3 : 0, kw = g.TT == 9 ? t - 3 * (t / 3) : 0;
  dh = g.sign > 0 ? kh - g.PT : g.PT - kh;
  dw = g.sign > 0 ? kw - g.PL : g.PL - kw;
}

__device__ __forceinline__ uint32_t col_load(const ColGeom& g, long m, int dh, int dw, int c) {
  if (m >= g.M) return 0u;
  const unsigned mm = (unsigned)m;
  const int ow = (int)(mm & (unsigned)(g.W - 1)) + dw;
  const int oh = (int)((mm >> g.lw) & (unsigned)(g.H - 1)) + dh;
  if (oh < 0 || oh >= g.H || ow < 0 || ow >= g.W) return 0u;
  const unsigned n = mm >> (g.lw + g.lh);
  return (uint32_t)g.s[(((n << g.lh) + oh) << g.lw | (unsigned)ow) * (unsigned)g.Cs + c];
}

// ---------------------------------------------------------------------------------------------------------
// small-side filter gradient: slab[wg][k<32][n<Cb] = sum over the workgroup's pixels of col[m][k] * big[m][n].
// A workgroup walks nsub blocks of PB = 16384/Cb pixels: the big rows arrive by LDS-DMA ([pixel][128 channels] per half,
// read back transposed), the im2col columns are gathered to LDS as [k][pixel]; the accumulators stay in registers across
// blocks.  Column 31 can be forced to ones: row 31 of the result is then the column sum of the big tensor (bias gradient
// when the big tensor is dy); an extra MFMA against an all-ones A operand gives the column sums of col (bias gradient
// when the small tensor is dy).  LDS: 32 KiB + 32 * (2 PB + 16) B <= 41.5 KiB -- below what the grouped three-tap and
// per-tap kernels allocate, so these workgroups ride in those launches without lowering their occupancy.
// ---------------------------------------------------------------------------------------------------------
struct ImgWArgs {
  ColGeom g;
  const bf16_t* big;     // [M][Cb]
  const bf16_t* zero;
  float* slab;           // [workgroups][32*Cb + 32]
  int ones_col, relu_big;
  int nsub;              // pixel blocks per workgroup
};

template <int CB> struct ImgWGeom {
  static constexpr int PB = 16384 / CB;                 // pixels per block: 128 (Cb = 128) or 64 (Cb = 256)
  static constexpr int NH = CB / 128;                   // 128-channel halves
  static constexpr int CPITCH = PB * 2 + 16;            // bytes per im2col row (+16: the 16 rows of a fragment read hit 16 bank groups)
  static constexpr int LDS = NH * PB * 256 + 32 * CPITCH;
};

template <int CB>
__device__ __forceinline__ void img_wgrad_body(const ImgWArgs& a, unsigned wg, unsigned char* smem) {
  constexpr int PB = ImgWGeom<CB>::PB, NH = ImgWGeom<CB>::NH, CPITCH = ImgWGeom<CB>::CPITCH;
  constexpr int NFW = CB / 64;                    // 16-channel fragments per wavefront
  unsigned char* bigs = smem;                     // [NH][PB][256 B]
  unsigned char* cols = smem + NH * PB * 256;     // [32][CPITCH]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // wavefront w: channels w*(CB/4) .. +CB/4 (inside one half); both 16-column groups of the 32 im2col columns
  const int g4 = lane >> 4, li = lane & 15;
  const int chw = wave * (CB / 4);
  const unsigned char* bh = bigs + (chw / 128) * PB * 256;
  const int slotw = (chw % 128) / 8;
  const uint32_t relu_lb = a.relu_big ? 0u : 0x80008000u;
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(H16_ONE_X2, H16_ONE_X2, H16_ONE_X2, H16_ONE_X2));
  f32x4_t acc[NFW][2], accs[2];
#pragma unroll
  for (int i = 0; i < NFW; ++i) { acc[i][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
  accs[0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; accs[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // im2col column of this thread: k = tid/8, pixels (tid%8)*(PB/8) .. +PB/8 of every block
  const int ck = tid >> 3, pg = tid & 7;
  int dh, dw, cc;
  col_tap(a.g, ck, dh, dw, cc);
  const bool ones_k = a.ones_col && ck == 31;

  for (int sub = 0; sub < a.nsub; ++sub) {
    const long m0 = ((long)wg * a.nsub + sub) * PB;
    if (m0 >= a.g.M) break;
    if (sub) __syncthreads();                     // the previous block's fragments have been read
    // big rows: deposits of 4 rows x 256 B; NH*PB/4 deposits, split over the 4 wavefronts
    constexpr int NDEP = NH * PB / 16;            // per wavefront
#pragma unroll 4
    for (int i = 0; i < NDEP; ++i) {
      const int dep = i * 4 + wave;               // 0 .. NH*PB/4
      const int half = dep / (PB / 4), row = (dep - half * (PB / 4)) * 4 + (lane >> 4);
      const long m = m0 + row;
      const int slot = (lane & 15) ^ ((row & 7) << 1);
      const bf16_t* p = m < a.g.M ? a.big + (unsigned)((unsigned)m * CB + half * 128 + slot * 8) : a.zero;
      glds16_asm(p, lds0 + dep * 1024);
    }
#pragma unroll
    for (int j = 0; j < PB / 64; ++j) {
      uint32_t v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const long m = m0 + pg * (PB / 8) + j * 8 + e;
        v[e] = ones_k ? (m < a.g.M ? H16_ONE : 0u) : col_load(a.g, m, dh, dw, cc);
      }
      *(uint4*)(cols + ck * CPITCH + (pg * (PB / 8) + j * 8) * 2) =
          make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 2
    for (int ks = 0; ks < PB / 32; ++ks) {
      bf16x8_t cf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // the transposing read hands lane group g4 the pixels {4*g4..+3} U {16+4*g4..+3} of the 32-pixel step: same order here
        const unsigned char* cp = cols + (j * 16 + li) * CPITCH + (ks * 32 + g4 * 4) * 2;
        const uint2 lo = *(const uint2*)cp, hi = *(const uint2*)(cp + 32);
        cf[j] = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
      }
      if (wave == 0) {
        accs[0] = mfma16(ones, cf[0], accs[0]);
        accs[1] = mfma16(ones, cf[1], accs[1]);
      }
#pragma unroll
      for (int i = 0; i < NFW; ++i) {
        const int row = ks * 32 + g4 * 4 + (li >> 2);
        const int slot0 = slotw + i * 2;
        const unsigned char* p = bh + row * 256 + (((slot0 + ((li & 3) >> 1)) ^ ((row & 7) << 1)) << 4) + (li & 1) * 8;
        uint4 v = __builtin_bit_cast(uint4, tr_pair(p, 16 * 256));
        v.x = pk_max_i16(v.x, relu_lb); v.y = pk_max_i16(v.y, relu_lb); v.z = pk_max_i16(v.z, relu_lb); v.w = pk_max_i16(v.w, relu_lb);
        const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, v);
        acc[i][0] = mfma16(bf, cf[0], acc[i][0]);
        acc[i][1] = mfma16(bf, cf[1], acc[i][1]);
      }
    }
  }
  // D[row = channel (4*g4 + r)][col = im2col column li (+16 j)]
  float* slab = a.slab + (long)wg * (32 * CB + 32);
#pragma unroll
  for (int i = 0; i < NFW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *(float4*)(slab + (j * 16 + li) * CB + chw + i * 16 + g4 * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  if (wave == 0 && g4 == 0) { slab[32 * CB + li] = accs[0][0]; slab[32 * CB + 16 + li] = accs[1][0]; }
}

// Image-end problems riding in a grouped filter-gradient launch: workgroups [0, first[n]) of the launch.  Cb = 128 only: that
// body needs 42 VGPRs + 24 AGPRs, inside the three-tap kernel's 64 + 96; the Cb = 256 body (74 + 48) would take the three-tap
// workgroups from three to two per SIMD.
#define IMG_GROUP_MAX 3
struct ImgWGroup {
  int n;
  unsigned first[IMG_GROUP_MAX + 1];
  ImgWArgs a[IMG_GROUP_MAX];
};

__device__ __forceinline__ void img_wgrad_group_body(const ImgWGroup& ig, unsigned b, unsigned char* smem) {
  int p = 0;
#pragma unroll
  for (int q = 1; q < IMG_GROUP_MAX; ++q)
    if (q < ig.n && b >= ig.first[q]) p = q;
  img_wgrad_body<128>(ig.a[p], b - ig.first[p], smem);
}
