// Filter gradient of a plain 3x3 stride-1 SAME convolution with ALL NINE taps in one workgroup (round 4).
//
// What round 4 measured on the three-tap kernel (conv_mfma.hip wgrad3_body; scripts/exp_wgrad_ablation.sh, profiles/r04_exp_wgrad_ablation.txt),
// 256-channel 32x32 layer at n = 128: 254 us as built, 192 us with the MFMAs removed, 193 us with MFMAs AND fragment reads removed, 97 us
// with the LDS-DMA stream removed -- the kernel is bound by the bytes its workgroups pull from L2 into LDS (13 KB per 32-pixel stage for
// 96 MFMAs, three workgroups per CU each streaming its own copy: 1.3 GB per launch), not by its matrix or LDS-read schedule.  The three
// workgroups that share a (channel tile, pixel chunk) and differ only in the filter row read the SAME dy pixels and the same x pixels one
// image row apart.  Here they are one workgroup: 12 wavefronts = 3 filter rows x (2 x 2) wavefront tiles of 64 output x 32 input channels
// x 3 column taps (the three-tap kernel's wavefront tile and accumulators: 96 registers), dy staged once, x staged once as a RING of image
// rows with one zero halo pixel left and right (the column taps are +-128-byte address shifts, no edge masks): 12 KB per stage for 288
// MFMAs, 3.25x fewer bytes per multiply-add.
//
//   workgroup      : 768 threads; tile = 128 output x 64 input channels x 9 taps over a chunk of whole 32-pixel K-steps
//   K-step         : 32 consecutive pixels of the (n, h, w) order = KR = 32 / W image rows, W = 8, 16 or 32
//   LDS            : NS stages of dy (32 pixels x 256 B, the three-tap kernel's swizzle) + a ring of NS * KR + 2 rows of x, row pitch PC =
//                    W + 8 pixels x 128 B (PC = 0 mod 8: the swizzle key of a pixel does not depend on its row); ring row r + 1 holds the
//                    r-th row of the NS stages in flight, rows 0 and NS * KR + 1 are copies of the last / first of them, so that the filter-row
//                    shift (kh - 1 rows) is an address offset without a wrap-around case: 102 .. 121 KB
//   LDS-DMA        : one global_load_lds_dwordx4 per wavefront and stage (wavefronts 0-7: dy, 8-11: x), issued NS - 2 stages ahead; a step
//                    needs the stage AFTER it as well (its first row is the kh = 2 source of the step's last row): vmcnt(NS - 4)
//   image borders  : the ring is the continuous pixel stream; at a step whose first (last) row is row 0 (H - 1) of an image the kh = 0
//                    (kh = 2) wavefronts clear that row's part of their dy fragments (wave-uniform branch)
//   bias gradient  : workgroups of input-channel tile 0, one extra MFMA against a ones fragment in 8 of the 12 wavefronts
// The slab layout is the three-tap kernel's (cell kh * 3 + kw, [ci][co]; bias tail), reduced by slab_reduce2_group_kernel.
#include "conv_mfma.h"
#include "mfma_util.h"

#ifndef WG9_ABLATE
#define WG9_ABLATE 0      /* timing only: 1 = no LDS-DMA after the prologue, 2 = no MFMAs */
#endif

namespace {

template <int N> __device__ __forceinline__ void w9_wait_vm() {
  if constexpr (N <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else static_assert(N <= 4, "unsupported vmcnt immediate");
}
// at most min(newer, MAXN) loads of this wavefront still in flight
template <int MAXN> __device__ __forceinline__ void w9_wait_newer(int newer) {
  if (newer >= MAXN) w9_wait_vm<MAXN>();
  else if (MAXN > 3 && newer == 3) w9_wait_vm<3>();
  else if (MAXN > 2 && newer == 2) w9_wait_vm<2>();
  else if (MAXN > 1 && newer == 1) w9_wait_vm<1>();
  else w9_wait_vm<0>();
}

__device__ __forceinline__ void w9_glds16(const void* sbase /* wave-uniform */, unsigned voff, unsigned lds_byte_addr /* wave-uniform */) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

// transposing 8-byte LDS read from an absolute byte address + a compile-time offset (the ds_read immediate)
__device__ __forceinline__ s16x4_t w9_tr8(int byte_addr, int imm) {
  typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
  __builtin_assume(byte_addr >= 0 && byte_addr < (1 << 18) && (byte_addr & 7) == 0);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)((lds_bytes)(size_t)(unsigned)byte_addr + imm));
}
__device__ __forceinline__ bf16x8_t w9_tr_pair(int byte_addr, int imm, int hi_off) {
  s16x4_t lo = w9_tr8(byte_addr, imm), hi = w9_tr8(byte_addr, imm + hi_off);
  s16x8_t r = (s16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

constexpr int w9_ns(int lw) { return lw == 3 ? 6 : 8; }
constexpr int w9_pc(int lw) { return (1 << lw) + 8; }
constexpr int w9_xring(int lw) { return (w9_ns(lw) * (32 >> lw) + 2) * w9_pc(lw) * 128; }
constexpr int W9_YT = 32 * 256;
constexpr int w9_lds_bytes(int lw) { return w9_xring(lw) + w9_ns(lw) * W9_YT; }
constexpr int W9_LDS_MAX = w9_lds_bytes(4) > w9_lds_bytes(5) ? (w9_lds_bytes(4) > w9_lds_bytes(3) ? w9_lds_bytes(4) : w9_lds_bytes(3))
                                                           : (w9_lds_bytes(5) > w9_lds_bytes(3) ? w9_lds_bytes(5) : w9_lds_bytes(3));

template <int LW, bool RELU>
__device__ __forceinline__ void wgrad9_body(const MfmaWgradArgs& a, const unsigned bx, const unsigned by, unsigned char* smem) {
  constexpr int W = 1 << LW, KR = 32 >> LW, NS = w9_ns(LW), PC = w9_pc(LW);
  constexpr int NRR = NS * KR + 2, XRING = NRR * PC * 128, YT = W9_YT, YOFF = XRING;
  constexpr int HI_X = LW == 5 ? 16 * 128 : (16 >> LW) * PC * 128;      // the second 16 pixels of a K-step inside the ring
  static_assert(XRING < 65536 && NS * YT <= 65536, "stage / tap offsets are ds_read immediates");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wave >> 2, wi = wave & 1, wo = (wave >> 1) & 1;
  const int nco = a.Cout / 128;
  const int cot = (int)bx % nco, cit = (int)bx / nco;
  const int ci0 = cit * 64, co0 = cot * 128;
  const long mb = (long)by * a.m_chunk;
  long me = mb + a.m_chunk;
  if (me > a.M) me = a.M;
  const int KT = (int)((me - mb) >> 5);
  const int g = lane >> 4, li = lane & 15;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  if (lds0 != 0) __builtin_trap();                  // (fragment addresses are absolute: the dynamic array is the kernel's only LDS)

  // ---- the ring starts as zeros (the halo pixels stay zero: no load ever writes them)
  for (int o = tid * 16; o < XRING; o += 768 * 16) *(uint4*)(smem + o) = make_uint4(0u, 0u, 0u, 0u);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- LDS-DMA roles.  dy (wavefronts 0-7): wavefront q deposits pixels 4 q + lane / 16, 16-byte slot lane % 16.
  //      x (wavefronts 8-11): wavefront 8 + q deposits the 8 pixels 8 q + lane / 8 of the stage, slot lane % 8
  const bool ydma = wave < 8;
  const int q = ydma ? wave : wave - 8;
  const int y_row = q * 4 + (lane >> 4);
  const int k8 = q * 8 + (lane >> 3);
  const int jq = (q * 8) >> LW, colq = (q * 8) & (W - 1);               // ring row inside the stage / first column of this wavefront's pixels
  // (PC = 0 mod 8: the swizzle key of ring pixel P = row * PC + 1 + column is ((1 + column) >> 1) & 3, whatever the row)
  const int xkey = ((1 + (k8 & (W - 1))) >> 1) & 3;
  const bf16_t* const sbase = ydma ? a.dy + co0 : a.x + ci0;
  const unsigned stride = 32u * 2u * (unsigned)(ydma ? a.Cout : a.Cin);      // bytes per stage
  unsigned voff = ydma ? 2u * (((unsigned)mb + (unsigned)y_row) * (unsigned)a.Cout + (unsigned)(((lane & 15) ^ ((y_row & 7) << 1)) * 8))
                       : 2u * (((unsigned)mb + (unsigned)k8) * (unsigned)a.Cin + (unsigned)(((lane & 7) ^ (xkey << 1)) * 8));
  const bool tail_ok = me < a.M;
  const unsigned xdst0 = lds0 + (unsigned)(((jq + 1) * PC + 1 + colq) * 128);                 // + slot * KR * PC * 128
  const unsigned xdst_lo = lds0 + (unsigned)((1 + colq) * 128);                                // ring row 0
  const unsigned xdst_hi = lds0 + (unsigned)(((NS * KR + 1) * PC + 1 + colq) * 128);           // ring row NS * KR + 1
  const unsigned ydst0 = lds0 + YOFF + q * 1024;                                               // + slot * YT
  auto issue = [&](int st, int slot) __attribute__((always_inline)) {      // stage st (0 .. KT) into slot = st % NS
    // (stage KT exists for its first x row only; past the end of the tensor it re-reads stage KT - 1: finite values nobody uses)
    const unsigned v = (st == KT && !tail_ok) ? voff - stride : voff;
    if (ydma) w9_glds16(sbase, v, ydst0 + slot * YT);
    else {
      w9_glds16(sbase, v, xdst0 + slot * KR * PC * 128);
      if (slot == NS - 1 && jq == KR - 1) w9_glds16(sbase, v, xdst_lo);
      if (slot == 0 && jq == 0) w9_glds16(sbase, v, xdst_hi);
    }
    voff += stride;
  };

  // ---- fragment addresses.  Transposing read: lane (g, li) reads pixel k = 4 g + li / 4 (and k + 16), 8 bytes = channels 4 (li % 4) .. + 3
  const int k = g * 4 + (li >> 2);
  int offy[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int slot0 = wo * 8 + i * 2;
    offy[i] = YOFF + k * 256 + (((slot0 + ((li & 3) >> 1)) ^ ((k & 7) << 1)) << 4) + (li & 1) * 8;
  }
  // x: ring pixel (stage row + kh) * PC + column + kw; the stage's first ring row is the immediate
  int AD[3][2];
  {
    const int rowl = k >> LW, col = k & (W - 1);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int P = (rowl + kh) * PC + col + t;
        const int slot0 = wi * 4 + j * 2;
        AD[t][j] = P * 128 + (((slot0 + ((li & 3) >> 1)) ^ ((((col + t) >> 1) & 3) << 1)) << 4) + (li & 1) * 8;
      }
  }
  // image-border rows inside a step: which halves / lanes of a dy fragment belong to the step's first and last image row
  const bool lane_first_lo = KR < 4 || g < 2, lane_last_hi = KR < 4 || g >= 2;

  f32x4_t acc[3][4][2];   // [kw][co subtile][ci subtile]
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  f32x4_t accb = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = a.want_bias && cit == 0 && kh < 2;
  const int ib = (kh * 2 + wi) & 3;
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(H16_ONE_X2, H16_ONE_X2, H16_ONE_X2, H16_ONE_X2));

  // ---- prologue: the image row in front of the chunk (ring row 0), stages 0 .. NS - 3
  if (!ydma && jq == KR - 1) {
    // pixels mb - W .. mb - 1; in front of the tensor (mb = 0: row 0 of an image, its kh = 0 products are cleared) the chunk's own first row
    const unsigned back = mb >= W ? 2u * (unsigned)W * (unsigned)a.Cin : 0u;
    const unsigned fwd = KR > 1 ? 2u * (unsigned)((KR - 1) * W) * (unsigned)a.Cin : 0u;      // this wavefront's pixels sit in the stage's LAST row
    w9_glds16(sbase, voff - fwd - back, xdst_lo);
  }
#pragma unroll
  for (int st = 0; st < NS - 2; ++st)
    if (st <= KT) issue(st, st);

  const int Hm = a.H - 1;
  const int h0 = (int)((mb >> LW) & (long)Hm);             // image row of the chunk's first row
  for (int s0 = 0; s0 < KT; s0 += NS) {
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      const int s = s0 + sidx;
      if (s < KT) {
        // stages <= s + 1 have landed once at most the NS - 4 stages behind them (if that many were issued) are in flight
        w9_wait_newer<NS - 4>(KT - s - 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s + NS - 2 <= KT && !(WG9_ABLATE & 1)) issue(s + NS - 2, (sidx + NS - 2) % NS);
        bf16x8_t yf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) yf[i] = w9_tr_pair(offy[i], sidx * YT, 16 * 256);
        if (do_bias) {        // (a uniform branch per fragment: a run-time index into the fragment array becomes a select chain)
          if (ib == 0) accb = mfma16(yf[0], ones, accb);
          else if (ib == 1) accb = mfma16(yf[1], ones, accb);
          else if (ib == 2) accb = mfma16(yf[2], ones, accb);
          else accb = mfma16(yf[3], ones, accb);
        }
        const int hf = (h0 + s * KR) & Hm;                 // image row of the step's first row
        if ((kh == 0 && hf == 0) || (kh == 2 && hf + KR - 1 == Hm)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            uint4 v = __builtin_bit_cast(uint4, yf[i]);
            if (kh == 0) {
              if (KR == 1) { v.x = 0u; v.y = 0u; v.z = 0u; v.w = 0u; }
              else { v.x = lane_first_lo ? 0u : v.x; v.y = lane_first_lo ? 0u : v.y; }
            } else {
              if (KR == 1) { v.x = 0u; v.y = 0u; v.z = 0u; v.w = 0u; }
              else { v.z = lane_last_hi ? 0u : v.z; v.w = lane_last_hi ? 0u : v.w; }
            }
            yf[i] = __builtin_bit_cast(bf16x8_t, v);
          }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          bf16x8_t xf[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            uint4 v = __builtin_bit_cast(uint4, w9_tr_pair(AD[t][j], sidx * KR * PC * 128, HI_X));
            if (RELU) { v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w); }
            xf[j] = __builtin_bit_cast(bf16x8_t, v);
          }
          if (WG9_ABLATE & 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(xf[j]));
            continue;
          }
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[t][i][j] = mfma16(yf[i], xf[j], acc[t][i][j]);
          __builtin_amdgcn_s_setprio(0);
        }
      }
    }
  }

  // D[row = co (4*(lane>>4)+r)][col = ci (lane&15)]
  float* slab = a.slab + (long)by * a.slab_stride;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int cell = kh * 3 + t;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int co = co0 + wo * 64 + i * 16 + (lane >> 4) * 4;
        const int ci = ci0 + wi * 32 + j * 16 + (lane & 15);
        *(float4*)(slab + ((long)cell * a.Cin + ci) * a.Cout + co) = make_float4(acc[t][i][j][0], acc[t][i][j][1], acc[t][i][j][2], acc[t][i][j][3]);
      }
  }
  if (do_bias && (lane & 15) == 0) {
    float* bs = slab + (long)a.cells * a.Cin * a.Cout;
    *(float4*)(bs + co0 + wo * 64 + ib * 16 + (lane >> 4) * 4) = make_float4(accb[0], accb[1], accb[2], accb[3]);
  }
}

}  // namespace

// several layers in one launch (the grouping of conv_mfma.hip's three-tap kernel): workgroup b -> problem p with first[p] <= b < first[p + 1],
// tile (b - first[p]) % gx[p], pixel chunk (b - first[p]) / gx[p]
struct Wgrad9Group {
  int n;
  unsigned first[WGRAD9_GROUP_MAX + 1];
  unsigned gx[WGRAD9_GROUP_MAX];
  MfmaWgradArgs a[WGRAD9_GROUP_MAX];
};
static_assert(sizeof(Wgrad9Group) <= 4096, "kernel argument block");

template <bool RELU>
__global__ __launch_bounds__(768) void conv_wgrad9_group_kernel(Wgrad9Group g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // workgroups that share a pixel chunk (consecutive b) on one XCD: they read the same dy / x pixels through one L2
  unsigned b = blockIdx.x;
  if ((gridDim.x & 7) == 0) b = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  int p = 0;
#pragma unroll
  for (int qq = 1; qq < WGRAD9_GROUP_MAX; ++qq)
    if (qq < g.n && b >= g.first[qq]) p = qq;
  const unsigned l = b - g.first[p];
  const unsigned gxp = g.gx[p];
  const MfmaWgradArgs& a = g.a[p];
  if (a.lw == 5) wgrad9_body<5, RELU>(a, l % gxp, l / gxp, smem);
  else if (a.lw == 4) wgrad9_body<4, RELU>(a, l % gxp, l / gxp, smem);
  else wgrad9_body<3, RELU>(a, l % gxp, l / gxp, smem);
}

static int wgrad9_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("RCGAN_WGRAD9"); v = e ? atoi(e) : 1; }
  return v;
}

// plain 3x3 stride-1 SAME filter gradient on an 8-, 16- or 32-pixel-wide power-of-two grid
bool mfma_wgrad9_takes(const MfmaWgradArgs& a) {
  if (!wgrad9_enabled() || !a.use_tr || a.zero == nullptr || a.sub || a.up) return false;
  if (a.KH != 3 || a.KW != 3 || a.PT != 1 || a.PL != 1 || a.lw < 3 || a.lw > 5 || a.lh < 0) return false;
  if (a.H < (32 >> a.lw) || a.M % 32 || a.Cin % 64 || a.Cout % 128) return false;
  return a.M * (long)a.Cin * 2 < (1L << 32) && a.M * (long)a.Cout * 2 < (1L << 32);       // 32-bit byte offsets
}

// grid of one problem: tiles x pixel chunks of ~px_per_block pixels (0: enough chunks for ~256 workgroups), at most nz chunks
bool mfma_wgrad9_plan(MfmaWgradArgs& a, int nz, unsigned* gx, unsigned* gy, long px_per_block) {
  if (!mfma_wgrad9_takes(a)) return false;
  const long tiles = (long)(a.Cin / 64) * (a.Cout / 128);
  long want = px_per_block > 0 ? cdiv(a.M, px_per_block) : (256 + tiles - 1) / tiles;
  const long maxs = a.M / 256 > 0 ? a.M / 256 : 1;
  if (want > maxs) want = maxs;
  if (want > nz) want = nz;
  if (want < 1) want = 1;
  a.m_chunk = ((a.M + want - 1) / want + 63) / 64 * 64;
  *gx = (unsigned)tiles;
  *gy = (unsigned)cdiv(a.M, a.m_chunk);
  return true;
}

template <bool RELU>
static int launch_wgrad9_group(rcgan_ctx* ctx, const Wgrad9Group& g) {
  static bool attr = false;
  if (!attr) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_wgrad9_group_kernel<RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, W9_LDS_MAX));
    attr = true;
  }
  double fl = 0;
  for (int p = 0; p < g.n; ++p) fl += 2.0 * (double)g.a[p].M * 9 * g.a[p].Cin * g.a[p].Cout;
  {
    ProfScope ps(ctx, RCGAN_PROF_WGRAD_MFMA, fl, fl);
    hipLaunchKernelGGL((conv_wgrad9_group_kernel<RELU>), dim3(g.first[g.n]), dim3(768), W9_LDS_MAX, ctx->stream, g);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// args[i] planned by mfma_wgrad9_plan, all with the same relu_in
int mfma_wgrad9_group_launch(rcgan_ctx* ctx, int n, const MfmaWgradArgs* args, const unsigned* gx, const unsigned* gy) {
  for (int i0 = 0; i0 < n; i0 += WGRAD9_GROUP_MAX) {
    Wgrad9Group g;
    g.n = (n - i0 < WGRAD9_GROUP_MAX) ? n - i0 : WGRAD9_GROUP_MAX;
    unsigned tot = 0;
    for (int p = 0; p < g.n; ++p) {
      g.a[p] = args[i0 + p];
      g.gx[p] = gx[i0 + p];
      g.first[p] = tot;
      tot += gx[i0 + p] * gy[i0 + p];
    }
    for (int p = g.n; p <= WGRAD9_GROUP_MAX; ++p) g.first[p] = tot;
    for (int p = g.n; p < WGRAD9_GROUP_MAX; ++p) { g.gx[p] = 1; g.a[p] = args[i0]; }
    int rc = args[i0].relu_in ? launch_wgrad9_group<true>(ctx, g) : launch_wgrad9_group<false>(ctx, g);
    if (rc) return rc;
  }
  return RCGAN_OK;
}
