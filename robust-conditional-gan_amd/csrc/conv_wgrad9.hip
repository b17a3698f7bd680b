// Filter gradient of a plain 3x3 stride-1 SAME convolution with ALL NINE taps in one workgroup (round 4).
//
// What round 4 measured on the three-tap kernel (conv_mfma.hip wgrad3_body; scripts/exp_wgrad_ablation.sh, profiles/r04_exp_wgrad_ablation.txt),
// 256-channel 32x32 layer at n = 128: 254 us as built, 192 us with the MFMAs removed, 193 us with MFMAs AND fragment reads removed, 97 us
// with the LDS-DMA stream removed -- the kernel is bound by the bytes its workgroups pull from L2 into LDS (13 KB per 32-pixel stage for
// 96 MFMAs, three workgroups per CU each streaming its own copy: 1.3 GB per launch), not by its matrix or LDS-read schedule.  The three
// workgroups that share a (channel tile, pixel chunk) and differ only in the filter row read the SAME dy pixels and the same x pixels one
// image row apart.  Here they are one workgroup: 12 wavefronts = 3 filter rows x (2 x 2) wavefront tiles of 64 output x 32 input channels
// x 3 column taps (the three-tap kernel's wavefront tile and accumulators: 96 registers), dy staged once, x staged once as image rows
// with one zero halo pixel left and right (the column taps are +-128-byte address shifts, no edge masks): 14 KB per K-step for 288 MFMAs,
// 2.8x fewer bytes per multiply-add.
//
//   workgroup      : 768 threads; tile = 128 output x 64 input channels x 9 taps over a chunk of whole 32-pixel K-steps
//   K-step         : 32 consecutive pixels of the (n, h, w) order = KR = 32 / W image rows, W = 8, 16 or 32
//   macro-step     : 4 K-steps = one barrier.  Its LDS buffer holds 4 K-steps of dy (32 pixels x 256 B each, the three-tap kernel's swizzle)
//                    and the 4 KR image rows of x it covers + the row in front + the row behind (self-contained: the filter-row shift kh - 1
//                    is an address offset), row pitch PC = W + 8 pixels x 128 B (PC = 0 mod 8: the swizzle key of a pixel does not depend on
//                    its row), one zero halo pixel left and right of a row.  Two buffers: 124 .. 136 KB
//   LDS-DMA        : the whole next macro-step (50 .. 56 global_load_lds_dwordx4, 4-5 per wavefront) is issued right after the barrier into the
//                    buffer everybody has just finished with and lands during the 96 MFMAs per wavefront of this one: vmcnt(0) at the barrier.
//                    Wave-uniform base + per-lane offset: no address arithmetic in the loop.
//                    (first version: a ring of rows and one barrier per K-step -- twelve wavefronts in lockstep, fragment reads and MFMAs of a
//                    SIMD's three wavefronts end to end: 1.4 us per K-step against 0.55 us of MFMA time)
//   image borders  : the rows are the continuous pixel stream; at a step whose first (last) row is row 0 (H - 1) of an image the kh = 0
//                    (kh = 2) wavefronts clear that row's part of their dy fragments (wave-uniform branch)
//   bias gradient  : workgroups of input-channel tile 0, one extra MFMA against a ones fragment in 8 of the 12 wavefronts
// The slab layout is the three-tap kernel's (cell kh * 3 + kw, [ci][co]; bias tail), reduced by slab_reduce2_group_kernel.
#include "conv_mfma.h"
#include "mfma_util.h"

#ifndef WG9_ABLATE
#define WG9_ABLATE 0      /* timing only: 1 = no LDS-DMA after the prologue, 2 = no MFMAs */
#endif

namespace {

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

constexpr int W9_YT = 32 * 256;                       // one K-step of dy (one column parity of it in the upsample form)
// MODE 0: plain 3x3.  MODE 1: the sub-pixel form of an upsample-3x3 layer (MfmaWgradArgs::sub = 1), MODE 2: of a ConvMeanPool layer (sub = 2) --
// conv_mfma.h.  The sub-pixel forms run one workgroup per ROW parity pa with 8 wavefronts = 2 row shifts s x (2 x 2) tiles of 64 output x 32
// input channels x the 4 cells (pb, d) (128 accumulator registers): the strided operand (dy of an upsample layer, x of a pooled one) is staged
// DE-INTERLEAVED, the even columns of a full-resolution row behind each other and then the odd ones, so that every cell reads 16 consecutive
// staged pixels like a column tap of the plain form.
constexpr int w9_ms(int mode) { return mode == 0 ? 4 : 2; }                       // K-steps (32 pixels each) per macro-step = per barrier
constexpr int w9_nw(int mode) { return mode == 0 ? 12 : 8; }                      // wavefronts that work
constexpr int w9_pc(int lw) { return (1 << lw) + 8; }
constexpr int w9_xrows(int lw, int mode) { return w9_ms(mode) * (32 >> lw) + (mode == 2 ? 1 : 2); }      // rows of the macro-step + the shifted ones
constexpr int w9_xrowb(int lw, int mode) { return (mode == 2 ? 2 : 1) * w9_pc(lw) * 128; }
constexpr int w9_xsz(int lw, int mode) { return w9_xrows(lw, mode) * w9_xrowb(lw, mode); }
constexpr int w9_yst(int mode) { return (mode == 1 ? 2 : 1) * W9_YT; }
constexpr int w9_buf(int lw, int mode) { return w9_xsz(lw, mode) + w9_ms(mode) * w9_yst(mode); }         // [x rows][dy K-steps]
constexpr int w9_lds_bytes(int lw, int mode) { return 2 * w9_buf(lw, mode); }
constexpr int w9_max(int a, int b) { return a > b ? a : b; }
constexpr int w9_lds_max_mode(int mode) { return w9_max(w9_lds_bytes(3, mode), w9_max(w9_lds_bytes(4, mode), w9_lds_bytes(5, mode))); }
constexpr int W9_LDS_MAX = w9_max(w9_lds_max_mode(0), w9_max(w9_lds_max_mode(1), w9_lds_max_mode(2)));
static_assert(W9_LDS_MAX <= 160 * 1024, "LDS");

template <int LW, bool RELU, int MODE>
__device__ __forceinline__ void wgrad9_body(const MfmaWgradArgs& a, unsigned bx, const unsigned by, unsigned char* smem) {
  constexpr int W = 1 << LW, KR = 32 >> LW, MS = w9_ms(MODE), NW = w9_nw(MODE), PC = w9_pc(LW);
  constexpr int XROWS = w9_xrows(LW, MODE), XROWB = w9_xrowb(LW, MODE), XSZ = w9_xsz(LW, MODE), YT = W9_YT, YST = w9_yst(MODE), BUF = w9_buf(LW, MODE);
  constexpr int HI_X = LW == 5 ? 16 * 128 : (16 >> LW) * XROWB;         // the second 16 pixels of a K-step inside the x rows
  constexpr int NT = MODE == 2 ? 4 : 3;                                 // distinct pixel fragments per K-step: column taps (d + pb), or cells (pb, d)
  constexpr int NC = MODE == 0 ? 3 : 4;                                 // accumulator cells per wavefront
  constexpr int NYP = MS * (MODE == 1 ? 16 : 8);                        // dy pieces (4 pixels x 256 B) per macro-step
  constexpr int XPR = (MODE == 2 ? 2 : 1) * (W / 8);                    // x pieces (8 pixels x 128 B) per row
  constexpr int NXP = XROWS * XPR;
  constexpr int NI = (NYP + NXP + NW - 1) / NW;                         // LDS-DMA instructions per wavefront and macro-step
  static_assert(XSZ < 65536 && MS * YST <= 65536, "K-step / tap offsets are ds_read immediates");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rs = wave >> 2, wi = wave & 1, wo = (wave >> 1) & 1;        // rs: filter row kh (plain), row shift s (sub-pixel forms)
  const int Cin = a.Cin, Cout = a.Cout;
  const int nco = Cout / 128, nci = Cin / 64;
  int pa = 0;
  if (MODE != 0) { pa = (int)bx / (nco * nci); bx = bx % (unsigned)(nco * nci); }
  const int cot = (int)bx % nco, cit = (int)bx / nco;
  const int ci0 = cit * 64, co0 = cot * 128;
  const long M = a.M;
  const long mb = (long)by * a.m_chunk;
  long me = mb + a.m_chunk;
  if (me > M) me = M;
  const int KT = (int)((me - mb) >> 5);
  const int NM = KT / MS;                           // (mfma_wgrad9_plan: chunks are whole macro-steps)
  const int g = lane >> 4, li = lane & 15;
  const int lh = a.lh, Hm = a.H - 1;
  const int NH = a.N << lh;                         // rows of the reduction grid
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  if (lds0 != 0) __builtin_trap();                  // (fragment addresses are absolute: the dynamic array is the kernel's only LDS)
  // row shift of this wavefront: plain kh - 1; upsample form s - 1 + pa; pooled form s - pa.  As a buffer-row offset (the buffer's first x
  // row is the one in front of the macro-step, in the pooled form row -pa): kh, s + pa, s
  const int dh = MODE == 0 ? rs - 1 : (MODE == 1 ? rs - 1 + pa : rs - pa);
  const int brow = MODE == 0 ? rs : (MODE == 1 ? rs + pa : rs);

  // ---- both buffers' x rows start as zeros (the halo pixels stay zero: no load ever writes them; a row outside the tensor is not loaded)
  for (int o = tid * 16; o < XSZ; o += NW * 64 * 16) {
    *(uint4*)(smem + o) = make_uint4(0u, 0u, 0u, 0u);
    *(uint4*)(smem + BUF + o) = make_uint4(0u, 0u, 0u, 0u);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- LDS-DMA: piece e = wave + NW j of a macro-step.  e < NYP: dy, 4 pixels x 256 B (lane / 16 = pixel, lane % 16 = 16-byte slot); then x,
  //      8 pixels x 128 B (lane / 8, lane % 8).  A piece lies in ONE row of its tensor: row address, first column and destination are scalar work,
  //      the per-lane part of the address (pixel inside the piece, swizzled slot) is the same for every piece of a kind -- two registers, no
  //      vector ALU work in the loop.  The strided operand's lanes step two pixels (its pieces are one column parity).
  //      (the dy swizzle key of K-step pixel k is k % 8 = 4 (piece % 2) + lane / 16, and piece % 2 = wave % 2: NW is even)
  //      The full-resolution row of reduction-grid row R at row parity pa is 2 R + pa (N H rows of the grid = 2 N H of the tensor): every
  //      piece's address is linear in the macro-step -- a wave-uniform base that advances + a per-lane offset per piece.
  const int ypx = lane >> 4, xpx = lane >> 3;
  unsigned voff[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int e = wave + NW * j;
    if (e < NYP) {
      const int i = MODE == 1 ? e >> 4 : e >> 3, pb = MODE == 1 ? (e >> 3) & 1 : 0, q = e & 7;
      const int rr = i * KR + ((4 * q) >> LW), col = ((4 * q) & (W - 1)) + ypx;
      const int px = MODE == 1 ? 2 * rr * 2 * W + 2 * col + pb : rr * W + col;
      voff[j] = 2u * ((unsigned)px * (unsigned)Cout + (unsigned)(((lane & 15) ^ (((4 * q + ypx) & 7) << 1)) * 8));
    } else {
      const int xe = e - NYP, rr = xe / XPR, rem = xe % XPR;
      const int pb = MODE == 2 ? rem / (W / 8) : 0, col = 8 * (MODE == 2 ? rem % (W / 8) : rem) + xpx;
      const int px = MODE == 2 ? 2 * rr * 2 * W + 2 * col + pb : rr * W + col;
      // (PC = 0 mod 8: the swizzle key of staged position 1 + column is ((1 + column) >> 1) & 3, whatever the row)
      voff[j] = 2u * ((unsigned)px * (unsigned)Cin + (unsigned)(((lane & 7) ^ ((((1 + col) >> 1) & 3) << 1)) * 8));
    }
  }
  const long R0 = mb >> LW;                                           // first row of the chunk
  // macro-step 0's first dy pixel / the first pixel of its first staged x row (row R0 - 1, or R0 - pa of the parity grid)
  const bf16_t* ybase = a.dy + co0 + (MODE == 1 ? ((2 * R0 + pa) << (LW + 1)) : (R0 << LW)) * Cout;
  const bf16_t* xbase = a.x + ci0 + (MODE == 2 ? ((2 * (R0 - pa) + pa) << (LW + 1)) : ((R0 - 1) << LW)) * Cin;
  const long ystep = (long)(MODE == 1 ? 4 : 1) * MS * 32 * Cout, xstep = (long)(MODE == 2 ? 4 : 1) * MS * 32 * Cin;
  auto issue = [&](int m, int b) __attribute__((always_inline)) {     // macro-step m (the next one in order) into buffer b
    const long Rm = R0 + (long)m * (MS * KR);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int e = wave + NW * j;                                    // (wave-uniform: destination and in-tensor test are scalar work)
      if (e < NYP) {
        const int i = MODE == 1 ? e >> 4 : e >> 3, pb = MODE == 1 ? (e >> 3) & 1 : 0, q = e & 7;
        if (Rm + i * KR + ((4 * q) >> LW) < NH)
          w9_glds16(ybase, voff[j], (unsigned)__builtin_amdgcn_readfirstlane(b * BUF + XSZ + i * YST + pb * YT + q * 1024));
      } else if (e < NYP + NXP) {
        const int xe = e - NYP, rr = xe / XPR, rem = xe % XPR;
        const int pb = MODE == 2 ? rem / (W / 8) : 0, c8 = MODE == 2 ? rem % (W / 8) : rem;
        const long R = Rm + rr - (MODE == 2 ? pa : 1);
        if (R >= 0 && R < NH)
          w9_glds16(xbase, voff[j], (unsigned)__builtin_amdgcn_readfirstlane(b * BUF + rr * XROWB + pb * PC * 128 + (1 + 8 * c8) * 128));
      }
    }
    ybase += ystep;
    xbase += xstep;
  };

  // ---- fragment addresses.  Transposing read: lane (g, li) reads pixel k = 4 g + li / 4 (and k + 16), 8 bytes = channels 4 (li % 4) .. + 3
  const int k = g * 4 + (li >> 2);
  int offy[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int slot0 = wo * 8 + i * 2;
    offy[i] = XSZ + k * 256 + (((slot0 + ((li & 3) >> 1)) ^ ((k & 7) << 1)) << 4) + (li & 1) * 8;
  }
  // x: buffer row (K-step row + brow), position 1 + column + column shift inside the row (or inside its parity half); the K-step's first row is
  // the immediate.  Plain / upsample form: fragment t = column tap, shift t - 1.  Pooled form: fragment t = 2 pb + d, shift d - pb in half pb
  int AD[NT][2];
  {
    const int rowl = k >> LW, col = k & (W - 1);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int pos = MODE == 2 ? 1 + col + (t & 1) - (t >> 1) : col + t;      // (PC = 0 mod 8: the swizzle key depends on pos % 8 only)
        const int slot0 = wi * 4 + j * 2;
        AD[t][j] = (rowl + brow) * XROWB + (MODE == 2 ? (t >> 1) * PC * 128 : 0) + pos * 128 +
                   (((slot0 + ((li & 3) >> 1)) ^ (((pos >> 1) & 3) << 1)) << 4) + (li & 1) * 8;
      }
  }
  // image-border rows inside a K-step: which lanes of a dy fragment half belong to the step's first and last image row
  const bool lane_first_lo = KR < 4 || g < 2, lane_last_hi = KR < 4 || g >= 2;

  f32x4_t acc[NC][4][2];   // [cell][co subtile][ci subtile]; cell = column tap kw (plain) or 2 pb + d
#pragma unroll
  for (int t = 0; t < NC; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  f32x4_t accb = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of dy: the workgroups of input-channel tile 0; one fragment per wavefront (plain: of the filter rows 0 and 1).
  // Pooled form: row parity 0 only (both parities read the same dy).  Upsample form: a workgroup sees the dy rows of ITS row parity, both column
  // parities -- two partial sums per chunk, side by side in a bias tail of 2 Cout floats (mfma_wgrad9_plan, SlabReduceGroup::Item::bias_parts)
  const bool do_bias = a.want_bias && cit == 0 && (MODE == 0 ? rs < 2 : (MODE == 2 ? pa == 0 : true));
  const int ib = (rs * 2 + wi) & 3;
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(H16_ONE_X2, H16_ONE_X2, H16_ONE_X2, H16_ONE_X2));

  issue(0, 0);
  const int h0 = (int)(R0 & (long)Hm);                     // image row of the chunk's first row
  for (int m = 0; m < NM; ++m) {
    // macro-step m was issued a whole macro-step ago; everybody has finished reading the other buffer
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (m + 1 < NM && !(WG9_ABLATE & 1)) issue(m + 1, (m + 1) & 1);
    // Software pipeline inside the macro-step (the chunk is whole macro-steps: no inactive K-step): the pixel fragments of the NEXT tap are
    // read before the MFMAs of this one (two fragment buffers), the dy fragments of the next K-step half by half as soon as the last
    // tap's MFMAs that use them have been issued.  (Without it a wavefront ran read -> wait -> 8 MFMAs three times per K-step with
    // the LDS latency exposed every time: 201 us for the 256-channel 32x32 layer, 105 us of it with the MFMAs removed.)
    constexpr int NYF = MODE == 1 ? 8 : 4;                // dy fragments per K-step: [column parity][co subtile]
    bf16x8_t yf[NYF], xq[2][2];
    auto load_y = [&](int i, int c) __attribute__((always_inline)) { yf[c] = w9_tr_pair(offy[c & 3], i * YST + (c >> 2) * YT, 16 * 256); };
    auto load_x = [&](int i, int t, int pbuf) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        uint4 v = __builtin_bit_cast(uint4, w9_tr_pair(AD[t][j], i * KR * XROWB, HI_X));
        if (RELU) { v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w); }
        xq[pbuf][j] = __builtin_bit_cast(bf16x8_t, v);
      }
    };
    // the MFMAs of pixel fragment t (in buffer pbuf) against dy subtile c: one cell (plain, pooled form) or the cells (pb, d = t - pb) of the
    // upsample form (t = 0: cell (0, 0); t = 1: (0, 1) and (1, 0); t = 2: (1, 1))
    auto mma = [&](int t, int pbuf, int c) __attribute__((always_inline)) {
      if (WG9_ABLATE & 2) { asm volatile("" :: "v"(xq[pbuf][0]), "v"(xq[pbuf][1]), "v"(yf[c])); return; }
      if (MODE != 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[t][c][j] = mfma16(yf[c], xq[pbuf][j], acc[t][c][j]);
      } else {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          const int d = t - pb;
          if (d < 0 || d > 1) continue;
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[2 * pb + d][c][j] = mfma16(yf[4 * pb + c], xq[pbuf][j], acc[2 * pb + d][c][j]);
        }
      }
    };
#pragma unroll
    for (int c = 0; c < NYF; ++c) load_y(0, c);
    load_x(0, 0, 0);
    int pcur = 0;                                           // (compile-time after unrolling) fragment buffer of the next tap
#pragma unroll
    for (int i = 0; i < MS; ++i) {
      const int s = m * MS + i;
      if (do_bias) {        // (a uniform branch per fragment: a run-time index into the fragment array becomes a select chain)
        if (ib == 0) accb = mfma16(yf[0], ones, accb);
        else if (ib == 1) accb = mfma16(yf[1], ones, accb);
        else if (ib == 2) accb = mfma16(yf[2], ones, accb);
        else accb = mfma16(yf[3], ones, accb);
        if (MODE == 1) {
          if (ib == 0) accb = mfma16(yf[4], ones, accb);
          else if (ib == 1) accb = mfma16(yf[5], ones, accb);
          else if (ib == 2) accb = mfma16(yf[6], ones, accb);
          else accb = mfma16(yf[7], ones, accb);
        }
      }
      const int hf = (h0 + s * KR) & Hm;                  // image row of the step's first row
      if ((dh < 0 && hf == 0) || (dh > 0 && hf + KR - 1 == Hm)) {
#pragma unroll
        for (int c = 0; c < NYF; ++c) {
          uint4 v = __builtin_bit_cast(uint4, yf[c]);
          if (dh < 0) {
            if (KR == 1) { v.x = 0u; v.y = 0u; v.z = 0u; v.w = 0u; }
            else { v.x = lane_first_lo ? 0u : v.x; v.y = lane_first_lo ? 0u : v.y; }
          } else {
            if (KR == 1) { v.x = 0u; v.y = 0u; v.z = 0u; v.w = 0u; }
            else { v.z = lane_last_hi ? 0u : v.z; v.w = lane_last_hi ? 0u : v.w; }
          }
          yf[c] = __builtin_bit_cast(bf16x8_t, v);
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bool last = t == NT - 1;
        if (!last) load_x(i, t + 1, pcur ^ 1);
        else if (i + 1 < MS) load_x(i + 1, 0, pcur ^ 1);
        if (!last || MODE == 1) {
#pragma unroll
          for (int c = 0; c < 4; ++c) mma(t, pcur, c);
          if (last && i + 1 < MS) {
#pragma unroll
            for (int c = 0; c < NYF; ++c) load_y(i + 1, c);
          }
        } else {
          mma(t, pcur, 0); mma(t, pcur, 1);
          if (i + 1 < MS) { load_y(i + 1, 0); load_y(i + 1, 1); }
          mma(t, pcur, 2); mma(t, pcur, 3);
          if (i + 1 < MS) { load_y(i + 1, 2); load_y(i + 1, 3); }
        }
        pcur ^= 1;
      }
    }
    // the other buffer (a select of two constants keeps the addresses provably aligned)
    const int delta = (m & 1) ? -BUF : BUF;
#pragma unroll
    for (int c = 0; c < 4; ++c) offy[c] += delta;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) AD[t][j] += delta;
  }

  // D[row = co (4*(lane>>4)+r)][col = ci (lane&15)]
  float* slab = a.slab + (long)by * a.slab_stride;
#pragma unroll
  for (int t = 0; t < NC; ++t) {
    // slab cell: tap (kh, kw = t), or [(pa*2 + pb)*4 + s*2 + d] with t = 2 pb + d
    const int cell = MODE == 0 ? rs * 3 + t : (pa * 2 + (t >> 1)) * 4 + rs * 2 + (t & 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int co = co0 + wo * 64 + i * 16 + (lane >> 4) * 4;
        const int ci = ci0 + wi * 32 + j * 16 + (lane & 15);
        *(float4*)(slab + ((long)cell * Cin + ci) * Cout + co) = make_float4(acc[t][i][j][0], acc[t][i][j][1], acc[t][i][j][2], acc[t][i][j][3]);
      }
  }
  if (do_bias && (lane & 15) == 0) {
    float* bs = slab + (long)a.cells * Cin * Cout + (MODE == 1 ? pa * Cout : 0);
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

// SUBK: the sub-pixel forms (8 wavefronts with 128 accumulator registers each: a kernel of its own, the plain form's 12 wavefronts have 170
// registers to live in)
template <bool RELU, bool SUBK>
__global__ __launch_bounds__(SUBK ? 512 : 768) void conv_wgrad9_group_kernel(Wgrad9Group g) {
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
  if (!SUBK) {
    if (a.lw == 5) wgrad9_body<5, RELU, 0>(a, l % gxp, l / gxp, smem);
    else if (a.lw == 4) wgrad9_body<4, RELU, 0>(a, l % gxp, l / gxp, smem);
    else wgrad9_body<3, RELU, 0>(a, l % gxp, l / gxp, smem);
  } else if (a.sub == 1) {
    if (a.lw == 5) wgrad9_body<5, RELU, 1>(a, l % gxp, l / gxp, smem);
    else if (a.lw == 4) wgrad9_body<4, RELU, 1>(a, l % gxp, l / gxp, smem);
    else wgrad9_body<3, RELU, 1>(a, l % gxp, l / gxp, smem);
  } else {
    if (a.lw == 5) wgrad9_body<5, RELU, 2>(a, l % gxp, l / gxp, smem);
    else if (a.lw == 4) wgrad9_body<4, RELU, 2>(a, l % gxp, l / gxp, smem);
    else wgrad9_body<3, RELU, 2>(a, l % gxp, l / gxp, smem);
  }
}

static int wgrad9_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("RCGAN_WGRAD9"); v = e ? atoi(e) : 3; }      // bit 0: the plain form, bit 1: the sub-pixel forms
  return v;
}

// plain 3x3 stride-1 SAME filter gradient on an 8-, 16- or 32-pixel-wide power-of-two grid, or the sub-pixel form (sub = 1, 2) of an
// upsample-3x3 / ConvMeanPool layer whose LOW-resolution grid is one
bool mfma_wgrad9_takes(const MfmaWgradArgs& a) {
  if (!a.use_tr || a.zero == nullptr || a.sub == 3 || a.up) return false;
  if (!(wgrad9_enabled() & (a.sub ? 2 : 1))) return false;
  if (a.KH != 3 || a.KW != 3 || a.lw < 3 || a.lw > 5 || a.lh < 0) return false;
  if (!a.sub && (a.PT != 1 || a.PL != 1)) return false;
  const int ms = a.sub ? 2 : 4;
  if (a.H < (32 >> a.lw) || a.M % (ms * 32) || a.Cin % 64 || a.Cout % 128) return false;      // a pixel chunk is whole macro-steps
  // 32-bit per-lane byte offsets only inside a piece; the row addresses are 64-bit scalar work
  return true;
}

// grid of one problem: tiles (x 2 row parities in the sub-pixel forms) x pixel chunks of ~px_per_block pixels (0: enough chunks for ~256
// workgroups), at most nz chunks
bool mfma_wgrad9_plan(MfmaWgradArgs& a, int nz, unsigned* gx, unsigned* gy, long px_per_block) {
  if (!mfma_wgrad9_takes(a)) return false;
  const long tiles = (long)(a.Cin / 64) * (a.Cout / 128) * (a.sub ? 2 : 1);
  const int ms32 = (a.sub ? 2 : 4) * 32;
  long want = px_per_block > 0 ? cdiv(a.M, px_per_block) : (256 + tiles - 1) / tiles;
  const long maxs = a.M / 256 > 0 ? a.M / 256 : 1;
  if (want > maxs) want = maxs;
  if (want > nz) want = nz;
  if (want < 1) want = 1;
  a.m_chunk = ((a.M + want - 1) / want + ms32 - 1) / ms32 * ms32;
  // (upsample form: two bias partials per chunk, one per row parity; the slab reduction adds both -- SlabReduceGroup::Item::bias_parts)
  a.slab_stride = (long)a.cells * a.Cin * a.Cout + (a.sub == 1 ? 2 : 1) * a.Cout;
  *gx = (unsigned)tiles;
  *gy = (unsigned)cdiv(a.M, a.m_chunk);
  return true;
}

template <bool RELU, bool SUBK>
static int launch_wgrad9_group(rcgan_ctx* ctx, const Wgrad9Group& g) {
  static bool attr = false;
  constexpr int lds = SUBK ? w9_max(w9_lds_max_mode(1), w9_lds_max_mode(2)) : w9_lds_max_mode(0);
  if (!attr) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_wgrad9_group_kernel<RELU, SUBK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr = true;
  }
  double fl = 0, fx = 0;
  for (int p = 0; p < g.n; ++p) {
    fl += 2.0 * (double)g.a[p].M * wgrad3_alg_taps(g.a[p]) * g.a[p].Cin * g.a[p].Cout;
    fx += 2.0 * (double)g.a[p].M * wgrad3_exec_taps(g.a[p]) * g.a[p].Cin * g.a[p].Cout;
  }
  {
    ProfScope ps(ctx, RCGAN_PROF_WGRAD_MFMA, fl, fx);
    hipLaunchKernelGGL((conv_wgrad9_group_kernel<RELU, SUBK>), dim3(g.first[g.n]), dim3(SUBK ? 512 : 768), lds, ctx->stream, g);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// args[i] planned by mfma_wgrad9_plan, all with the same relu_in and all plain or all sub-pixel forms
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
    const bool subk = args[i0].sub != 0, relu = args[i0].relu_in != 0;
    int rc = subk ? (relu ? launch_wgrad9_group<true, true>(ctx, g) : launch_wgrad9_group<false, true>(ctx, g))
                  : (relu ? launch_wgrad9_group<true, false>(ctx, g) : launch_wgrad9_group<false, false>(ctx, g));
    if (rc) return rc;
  }
  return RCGAN_OK;
}
