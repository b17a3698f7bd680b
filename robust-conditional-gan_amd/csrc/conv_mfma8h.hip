// 256 x 256-tile, 8-wavefront convolution with the PIXEL operand held as a zero-padded image patch in LDS (round 4).
//
// What round 4 measured on the tile-per-tap kernel (conv_mfma8.hip; scripts/build_p8_ablate.sh, profiles/r04_exp_p8_ablation.txt):
// per K-tile 1.52 us as built, 0.98 us with MFMAs and barriers only (the matrix pipe at the clock the chip sustains under this
// load), 1.02 us with the LDS-DMA stream and barriers only -- and 0.99 us when only the FILTER half of the stream is issued.  The
// kernel is bound by the bytes it moves from L2 into LDS (64 KB per K-tile and CU: at the ~0.6 us such a burst takes to land,
// the 2-3 half-tiles its LDS can keep in flight do not cover it), not by its matrix or LDS-read schedule.  Half of those bytes
// are the pixel tile, fetched again for every one of the nine taps although the taps read the same pixels shifted by a row or a
// column.  Here the pixels of a 64-channel chunk are fetched ONCE as a patch (the tile's TR image rows + one halo row above and
// below, one halo column left and right: (TR + 2) x (W + 2) pixels x 128 B = 43 KB for W = 32), the nine taps are row offsets into
// it, K runs chunk-major, and only the filter tiles stream per K-tile: 36.8 KB per K-tile instead of 64.  No tap-source table.
//
//   workgroup tile : 256 pixels (TR = 256 / W whole image rows, W = 16 or 32) x 256 output channels; K-tile = one tap x 64 channels
//   wavefronts     : 8 = 2 (pixels, 128 each) x 4 (channels, 64 each); 4 x 8 accumulator tiles, the epilogue of the other kernels
//   LDS            : 2 patches (chunk c, c + 1) + 2 filter K-tiles (t, t + 1 by parity) of 2 x 16 KiB halves (C0 / C1: the first /
//                    second 32 channels of every wavefront column) = 2 x 43.0 + 64 KiB = 150 KiB
//   LDS-DMA        : filters run 1.5 K-tiles ahead -- C0(t+2) is issued in phase 3 of tile t, C1(t+2) in phase 4 (their slots
//                    were last read in phases 1 / 2 of tile t); the next chunk's patch rides one piece per wavefront and K-tile
//                    behind C1 in phase 4 of the chunk's first taps.  Counted waits: vmcnt(4 + patch pieces issued since) in front of
//                    phase 1's first barrier (C1 of this tile has landed), vmcnt(6 + ...) in front of phase 4's (C0 of the next tile).
//                    Loads retire in order, so the patch of chunk c + 1 (issued by tap 5) has landed when C0 of its first K-tile
//                    (issued at tap 7) has; the counts are exact (a per-wavefront mask of the pieces that have lanes to fetch): a
//                    conservative count made every wait stall on a filter burst issued two phases earlier (0.15 us per K-tile).
//   schedule       : the ping-pong form of conv_mfma8.hip (PP): the wavefronts of a SIMD run one barrier apart, every phase has a
//                    barrier between its fragment reads and its MFMAs; a burst is read one barrier after the wait + barrier that
//                    retires it (the other group runs a barrier behind).
//   fragment reads : pixel fragment f of tap (kh, kw) = 16 consecutive patch pixels from (tr + kh) * PC + tc + kw; the 16-byte
//                    slots of a patch pixel P are XOR-swizzled with (P >> 1) & 7 (source side, as everywhere), so a window that
//                    starts at a multiple of four pixels reads conflict-free and the others 2-way on a quarter of their banks.
#include "conv_mfma.h"
#include "mfma_util.h"

// timing-only ablations (scripts/build_p8_ablate.sh h<k>; results are wrong by construction): 1 = every tap reads the patch at tap (0, 0)
// (aligned, conflict-free windows), 2 = no patch LDS-DMA after the prologue, 4 = no filter LDS-DMA after the prologue, 8 = no pixel
// fragment reads after the first K-tile, 16 = no MFMAs
#ifndef H8_ABLATE
#define H8_ABLATE 0
#endif

namespace {

template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else static_assert(N == 0, "unsupported vmcnt immediate");
}
// vmcnt(BASE + extra), extra in 0 .. MAXX wave-uniform: the patch pieces that were really issued behind the burst waited for
template <int BASE, int MAXX> __device__ __forceinline__ void wait_vm_plus(int extra) {
  if (extra == 0) wait_vm<BASE>();
  else if (MAXX < 2 || extra == 1) wait_vm<BASE + 1>();
  else if (MAXX < 3 || extra == 2) wait_vm<BASE + 2>();
  else if (MAXX < 4 || extra == 3) wait_vm<BASE + 3>();
  else if (MAXX < 5 || extra == 4) wait_vm<BASE + 4>();
  else if (MAXX < 6 || extra == 5) wait_vm<BASE + 5>();
  else wait_vm<BASE + 6>();
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

constexpr int H8_HALF = 128 * 128;                   // one filter half-tile: 128 rows x 128 B
#ifndef H8_PC_EXTRA
#define H8_PC_EXTRA 2      /* patch columns beyond the image width: 2 = the halo; 4 = two more (idle) columns, row pitch = 2 mod 4 pixels */
#endif
constexpr int h8_patch_rows(int lw) { return (((256 >> lw) + 2) * ((1 << lw) + H8_PC_EXTRA) + 7) / 8 * 8; }      // patch pixels, padded to whole DMA pieces
constexpr int h8_lds_bytes(int lw) { return 2 * h8_patch_rows(lw) * 128 + 4 * H8_HALF; }      // [filter buffers 0, 1][patch 0][patch 1]

}  // namespace

// (glds16_sbase -- LDS-DMA with a wave-uniform base in SGPRs and a 32-bit per-lane byte offset -- lives in mfma_util.h)

// 16 bytes from an absolute LDS byte address + a compile-time offset (the ds_read immediate): no `smem +` pointer arithmetic, which
// costs a vector add per read (the dynamic-LDS base is a relocation the compiler does not fold)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));     // (a native vector: HIP's uint4 struct is loaded as two 8-byte halves -> ds_read2_b64)
__device__ __forceinline__ u32x4_t lds_read16(int byte_addr, int imm) {
  typedef __attribute__((address_space(3))) const unsigned char* lds_bytes;
  // (LLVM folds a DS offset only onto a base it knows to be non-negative, and splits a 16-byte read it cannot prove aligned into ds_read2_b64)
  __builtin_assume(byte_addr >= 0 && byte_addr < (1 << 18) && (byte_addr & 15) == 0);
  return *(__attribute__((address_space(3))) const u32x4_t*)((lds_bytes)(size_t)(unsigned)byte_addr + imm);
}

// x ^ 64 where it is used: a loop-invariant the compiler would otherwise hoist into a register of its own (sixteen of them: spills)
__device__ __forceinline__ int xor64(int x) {
  int y;
  asm volatile("v_xor_b32 %0, 64, %1" : "=v"(y) : "v"(x));
  return y;
}

// ---- batch norm + activation applied to the staged patch (MfmaConvArgs::bn_*; the forward-only generator passes) -------------------
// Every thread transforms exactly the 16-byte slots its own LDS-DMA lanes deposited (once it knows they have landed): slot `pos` of
// patch pixel q holds source channels ((pos ^ ((q >> 1) & 7)) * 8 .. + 7 of the chunk, and (q >> 1) & 7 = (wave & 1) * 4 + (lrow >> 1) for
// every piece (wave + 8 j) of this thread -- the same eight channels each time.  inv = rstd * gamma[label], c0 = fma(-mean, inv, beta[label])
// per channel sit in an LDS table built once per workgroup (a tile lies inside ONE image); the arithmetic and the 16-bit rounding are
// bn.hip's apply pass (fma(x, inv, c0), ReLU, round to nearest even), so the convolution sees bit for bit what it would have read
// from the written-out tensor.  Halo slots (zero padding of the NORMALISED tensor) are never touched.
__device__ __forceinline__ void h8_bn_table(const MfmaConvArgs& a, float* tab /* [2][Cin] in LDS */, unsigned n_img, int tid) {
  const int lab = a.bn_labels ? a.bn_labels[n_img] : 0, seg = (int)n_img / a.bn_seg_samples;
  for (int c = tid; c < a.Cin; c += 512) {
    const float inv = a.bn_rstd[seg * a.Cin + c] * a.bn_gamma[lab * a.Cin + c];
    tab[c] = inv;
    tab[a.Cin + c] = __fmaf_rn(-a.bn_mean[seg * a.Cin + c], inv, a.bn_beta[lab * a.Cin + c]);
  }
}
// `keep`: per lane, false = a halo slot -- the zeros it holds are written back as they are (no branch: the straight-line form lets the
// compiler schedule this work between the MFMAs of the segment it sits in)
__device__ __forceinline__ void h8_bn_slot(unsigned char* slot, const float* tab, int cin, int ch0, float relu_floor, bool keep = true) {
  const float4 i0 = *(const float4*)(tab + ch0), i1 = *(const float4*)(tab + ch0 + 4);
  const float4 c0 = *(const float4*)(tab + cin + ch0), c1 = *(const float4*)(tab + cin + ch0 + 4);
  const float inv8[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w}, c08[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  uint4 v = *(uint4*)slot;
  uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float lo = __fmaf_rn(h16_lo(w4[q]), inv8[2 * q], c08[2 * q]);
    float hi = __fmaf_rn(h16_hi(w4[q]), inv8[2 * q + 1], c08[2 * q + 1]);
    lo = fmaxf(lo, relu_floor); hi = fmaxf(hi, relu_floor);
    w4[q] = keep ? pack_h16x2(lo, hi) : w4[q];
  }
  *(uint4*)slot = make_uint4(w4[0], w4[1], w4[2], w4[3]);
}

// The transform dealt between the four groups of four MFMAs of a segment (h8_bn_stage): word q of a slot (two channels' pair of
// 16-bit values) needs table entries 2q, 2q + 1.  The stages are pinned between the MFMA groups with sched_barrier(0): left to itself the
// compiler issues a piece's ~45 vector-ALU instructions as one block behind the 15th MFMA (a sched_group_barrier pipeline is not honoured
// across the LDS-read dependency), which made the segment ~200 cycles longer; pinned, a stage's ~10 instructions issue while the four
// MFMAs in front of it occupy the matrix pipe.
struct H8BnTab { float4 i0, i1, c0, c1; };
__device__ __forceinline__ H8BnTab h8_bn_tab_load(const float* tab, int cin, int ch0) {
  return H8BnTab{*(const float4*)(tab + ch0), *(const float4*)(tab + ch0 + 4), *(const float4*)(tab + cin + ch0), *(const float4*)(tab + cin + ch0 + 4)};
}
__device__ __forceinline__ uint32_t h8_bn_word(uint32_t w, float ia, float ib, float ca, float cb, float relu_floor, bool keep) {
  float lo = __fmaf_rn(h16_lo(w), ia, ca), hi = __fmaf_rn(h16_hi(w), ib, cb);
  lo = fmaxf(lo, relu_floor); hi = fmaxf(hi, relu_floor);
  return keep ? pack_h16x2(lo, hi) : w;
}
__device__ __forceinline__ void h8_bn_stage(int q, uint4& v, const H8BnTab& t, float relu_floor, bool keep) {
  if (q == 0) v.x = h8_bn_word(v.x, t.i0.x, t.i0.y, t.c0.x, t.c0.y, relu_floor, keep);
  if (q == 1) v.y = h8_bn_word(v.y, t.i0.z, t.i0.w, t.c0.z, t.c0.w, relu_floor, keep);
  if (q == 2) v.z = h8_bn_word(v.z, t.i1.x, t.i1.y, t.c1.x, t.c1.y, relu_floor, keep);
  if (q == 3) v.w = h8_bn_word(v.w, t.i1.z, t.i1.w, t.c1.z, t.c1.w, relu_floor, keep);
}

// PHM: the sub-pixel form of a 3x3 convolution behind the nearest 2x upsample (MfmaConvArgs::wph, conv_mfma8.hip): a tile is 256
// pixels of ONE phase (ph, pw) over the LOW-resolution grid (W = its width), the reduction runs over the 2 x 2 taps (a, b) that read
// low-resolution pixel (i + a - 1 + ph, j + b - 1 + pw) with the phase's summed filters -- patch rows / columns (a + ph, b + pw), so the
// one (TR + 2) x (W + 2) patch serves every phase; the epilogue scatters the rows to output pixel (2 i + ph, 2 j + pw).  Four K-tiles per
// chunk instead of nine: the patch pieces go out behind the filter bursts of the chunk's first two taps.
template <int LW, bool RELU, bool PHM, bool BNIN = false>
__global__ __launch_bounds__(512) void conv_mfma_h8_kernel(MfmaConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int W = 1 << LW, TR = 256 >> LW;        // (low-resolution) image width, image rows per tile
  constexpr int NT = PHM ? 4 : 9;                   // taps = K-tiles per 64-channel chunk
  // ... of which the first NTI carry patch pieces (landed two K-tiles later: NTI <= NT - 2).  BNIN's sub-pixel form sends them all behind
  // the first tap: the pieces are transformed under the MFMAs of the taps that follow their landing, and there are only two of those
  constexpr int NTI = PHM ? (BNIN ? 1 : 2) : 6;
  constexpr int PC = W + H8_PC_EXTRA, PR = TR + 2;  // patch columns / rows
  constexpr int NPX = PR * PC, NROWS = h8_patch_rows(LW), NP = NROWS / 8;      // patch pixels, padded rows, DMA pieces (8 rows each)
  constexpr int PATCH = NROWS * 128;
  constexpr int WBUF = 2 * H8_HALF;
  // LDS: [filter K-tile buffers 0, 1][patch 0][patch 1]; the filter buffers differ in address bit 15 only (toggled by XOR)
  constexpr int WOFF = 0, P0OFF = 2 * WBUF, P1OFF = 2 * WBUF + PATCH;
  static_assert(WBUF == 32768, "the filter buffers are toggled with ^ 0x8000");
  constexpr int MAXP = (NP + 7) / 8;                // patch pieces per wavefront
  constexpr int PPT = (MAXP + NTI - 1) / NTI;       // patch pieces per wavefront behind one K-tile's filter bursts
  static_assert(MAXP <= 6 && NTI <= NT - 2, "a wavefront issues its patch pieces during the first taps of a chunk");
  static_assert((PR - 1) * PC * 128 + PC * 128 < 65536, "tap offsets are ds_read immediates");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;          // pixel half / channel quarter of this wavefront
  const int grp = wave >> 2;                        // waves w and w + 4 share a SIMD: the two ping-pong groups
  unsigned mt = blockIdx.x;
  if ((gridDim.x & 7) == 0) mt = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);      // a contiguous run of tiles per XCD
  const long m0 = (long)mt * 256;
  const int co0 = blockIdx.y * 256;
  const int K = NT * a.Cin;
  const int lrow = lane >> 3, pos = lane & 7;
  // image grid the patch is cut from, and the tile's place in it
  const int HI = PHM ? (a.H >> 1) : a.H, lhi = PHM ? a.lh - 1 : a.lh;
  const long Mph = a.M >> 2;
  const int tph = PHM ? (int)(m0 / Mph) : 0, ph = tph >> 1, pw = tph & 1;
  const long mbase = PHM ? (long)tph * Mph : 0;
  const unsigned ms0 = (unsigned)(m0 - mbase);
  const bf16_t* const wbase = PHM ? a.wph + (long)tph * a.Cout * K : a.wt;
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (a.stamps && tid == 0) a.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  if (a.stamps && tid == 0) {
    a.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_ID
    a.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // XCC_ID
  }

  // ---- patch sources: piece i = wave + 8 j covers patch pixels 8 i .. 8 i + 7; this lane deposits pixel q = 8 i + lrow, slot pos.
  // Halo slots (SAME padding, the padding rows behind the patch) are zeroed once, here, in both buffers, and never written again:
  // their lanes are masked out of every burst.
  const unsigned n_img = ms0 >> (LW + lhi);
  const int oh0 = (int)((ms0 >> LW) & (unsigned)(HI - 1));
  unsigned poff[MAXP];                              // BYTE offset into a.in (without the chunk's channel offset), ~0u = halo
#pragma unroll
  for (int j = 0; j < MAXP; ++j) {
    const int piece = wave + 8 * j;
    const int q = piece * 8 + lrow;
    const int pr = q / PC, pc = q - pr * PC;
    const int ih = oh0 - 1 + pr, iw = pc - 1;
    const bool ok = piece < NP && q < NPX && ih >= 0 && ih < HI && iw >= 0 && iw < W;
    poff[j] = ok ? 2u * (((n_img * (unsigned)HI + (unsigned)ih) * (unsigned)W + (unsigned)iw) * (unsigned)a.Cin + (unsigned)((pos ^ ((q >> 1) & 7)) * 8)) : ~0u;
    if (!ok && piece < NP) {
      *(uint4*)(smem + P0OFF + piece * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
      *(uint4*)(smem + P1OFF + piece * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
  }

  // bit j: piece j of this wavefront has at least one lane to fetch (wave-uniform: the counted waits below count issued bursts)
  unsigned pmask = 0;
#pragma unroll
  for (int j = 0; j < MAXP; ++j) pmask |= (__builtin_amdgcn_ballot_w64(poff[j] != ~0u) != 0 ? 1u : 0u) << j;
  pmask = __builtin_amdgcn_readfirstlane(pmask);

  // ---- filter sources: per half-tile this wavefront deposits rows (wave*2 + j)*8 + lrow, j = 0,1; row r of half h = channel (r>>5)*64 + h*32 + (r&31)
  unsigned woff[2];                                 // BYTE offset into a.wt of half 0 (without the K-tile's column offset); half 1 = + 32 rows, uniform
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 8 + lrow;
    const int co = co0 + (r >> 5) * 64 + (r & 31);
    woff[j] = 2u * ((unsigned)co * (unsigned)K + (unsigned)((pos ^ ((r >> 1) & 7)) * 8));
  }

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // BNIN: the table behind the patches; the eight channels (of a chunk) this thread's deposits hold; its landed pieces of a patch
  float* const bn_tab = (float*)(smem + P1OFF + PATCH);
  const float bn_floor = a.bn_act == RCGAN_ACT_RELU ? 0.f : -INFINITY;
  auto bn_piece = [&](int j, int cnext) __attribute__((always_inline)) {
    // (the thread's channel offset is recomputed from an opaque copy of the lane id where it is used: a loop-invariant of its own would be
    // one more live register in a kernel that sits at its 256-register budget -- a spill's scratch load shares vmcnt with the LDS-DMA)
    int ln;
    asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));
    const int bn_ch = ((ln & 7) ^ ((wave & 1) * 4 + (ln >> 4))) << 3;
    if (poff[j] != ~0u) h8_bn_slot(smem + ((cnext & 1) ? P1OFF : P0OFF) + (wave + 8 * j) * 1024 + ln * 16, bn_tab, a.Cin, cnext * 64 + bn_ch, bn_floor);
  };
  // ... inside an MFMA segment: without a branch where every wavefront's piece j lies inside the patch (8 j + 7 < NP); a piece that was
  // never fetched (all halo) holds zeros and keeps them
  auto bn_piece_seg = [&](int j, int cnext) __attribute__((always_inline)) {
    if (8 * j + 7 < NP) {
      int ln;
      asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));
      const int bn_ch = ((ln & 7) ^ ((wave & 1) * 4 + (ln >> 4))) << 3;
      h8_bn_slot(smem + ((cnext & 1) ? P1OFF : P0OFF) + (wave + 8 * j) * 1024 + ln * 16, bn_tab, a.Cin, cnext * 64 + bn_ch, bn_floor, poff[j] != ~0u);
    } else if ((pmask >> j) & 1) {
      bn_piece(j, cnext);
    }
  };
  auto issue_patch = [&](int j, int cnext) __attribute__((always_inline)) {      // piece j of this wavefront, patch of chunk cnext
    const bf16_t* base = a.in + cnext * 64;
    const unsigned dst = lds0 + ((cnext & 1) ? P1OFF : P0OFF) + (wave + 8 * j) * 1024;
    if (poff[j] != ~0u) glds16_sbase(base, poff[j], dst);
  };
  // K-tile (chunk c, tap): filter columns tap * Cin + c * 64; LDS buffer = parity of t = NT c + tap
  auto issue_w = [&](int h, int c, int tap) __attribute__((always_inline)) {
    const bf16_t* base = wbase + (tap * a.Cin + c * 64) + (long)h * 32 * K;
    const unsigned dst = lds0 + WOFF + ((NT * c + tap) & 1) * WBUF + h * H8_HALF + wave * 2048;
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16_sbase(base, woff[j], dst + j * 1024);
  };

  f32x4_t acc[4][8];       // [co fragment = c-half*2 + g][px fragment = p-half*4 + f]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, kc = lane >> 4;
  // filter fragments: byte address inside the CURRENT K-tile buffer, toggled between the two buffers after every K-tile
  if (lds0 != 0) __builtin_trap();                  // (the fragment addresses below are absolute: the dynamic array is the kernel's only LDS)
  int wad = WOFF + wn * 32 * 128 + frow * 128 + ((kc ^ ((frow >> 1) & 7)) * 16);
  // pixel fragments: patch pixel = pw0 + immP with immP = kh * PC + kw + (first tile pixel of the fragment as a patch offset), a
  // compile-time number once the taps are unrolled; the swizzle key of the slot depends on (pw0 + immP) & 15 only, so sixteen
  // per-lane bases AD[immP & 15] (those that occur stay live) + the immediate immP * 128 address every fragment without any
  // vector ALU work in the loop.  The second K-step's slot is the first's ^ 4: byte address ^ 64.
  const int pw0 = ((wm * 128) >> LW) * PC + frow + (PHM ? ph * PC + pw : 0);
  int AD[16];
#pragma unroll
  for (int sx = 0; sx < 16; ++sx) AD[sx] = P0OFF + pw0 * 128 + ((kc ^ (((pw0 + sx) & 15) >> 1)) << 4);

  bf16x8_t xf[2][4], wfc[2][2][2];                  // [ks][fragment] of the current pixel half; [channel half][ks][fragment]
  auto load_x = [&](int kh, int kw, int h) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int p = h * 64 + f * 16;                // first tile pixel of the fragment inside the wavefront's 128
      const int immP = (H8_ABLATE & 1) ? 0 : kh * PC + kw + (p >> LW) * PC + (p & (W - 1));
      const int ad = AD[immP & 15];
      const int ad1 = xor64(ad);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4_t v = lds_read16(ks ? ad1 : ad, immP * 128);
        if (RELU) { v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w); }
        xf[ks][f] = __builtin_bit_cast(bf16x8_t, v);
      }
    }
  };
  auto load_w = [&](int h) __attribute__((always_inline)) {
    const int wad1 = xor64(wad);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int g = 0; g < 2; ++g) wfc[h][ks][g] = __builtin_bit_cast(bf16x8_t, lds_read16(ks ? wad1 : wad, h * H8_HALF + g * 16 * 128));
  };
  auto mma = [&](int ph, int ch) __attribute__((always_inline)) {
    if (H8_ABLATE & 16) return;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int f = 0; f < 4; ++f)
          acc[ch * 2 + g][ph * 4 + f] = mfma16(wfc[ch][ks][g], xf[ks][f], acc[ch * 2 + g][ph * 4 + f]);
    __builtin_amdgcn_s_setprio(0);
  };
  // the same 16 MFMAs with the batch-norm transform of this thread's pieces [j0, j1) of chunk cnext's patch dealt between their four
  // groups (BNIN).  A piece that may lie behind the patch's end for some wavefronts (8 j + 7 >= NP: the last one) takes the branching form
  // in front of the MFMAs.
  auto mma_bn = [&](int ph, int ch, int j0, int j1, int cnext) __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
    int ln;
    asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));      // (an opaque copy of the lane id: see bn_piece)
    unsigned char* const pbase = smem + ((cnext & 1) ? P1OFF : P0OFF) + wave * 1024 + ln * 16;
    const H8BnTab t = h8_bn_tab_load(bn_tab, a.Cin, cnext * 64 + (((ln & 7) ^ ((wave & 1) * 4 + (ln >> 4))) << 3));
    uint4 v[3];
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      if (8 * j + 7 < NP) v[j - j0] = *(uint4*)(pbase + j * 8192);
      else if ((pmask >> j) & 1) bn_piece(j, cnext);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int f = 0; f < 4; ++f)
        acc[ch * 2 + (q & 1)][ph * 4 + f] = mfma16(wfc[ch][q >> 1][q & 1], xf[q >> 1][f], acc[ch * 2 + (q & 1)][ph * 4 + f]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = j0; j < j1; ++j)
        if (8 * j + 7 < NP) h8_bn_stage(q, v[j - j0], t, bn_floor, poff[j] != ~0u);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = j0; j < j1; ++j)
      if (8 * j + 7 < NP) *(uint4*)(pbase + j * 8192) = v[j - j0];
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  stamp(1);        // (scripts/exp_p8_timeline.py: segment 0 = patch sources, filter sources, fragment addresses)
  // ---- prologue: the first chunk's patch, the filters of K-tiles 0 and 1
  const int nchunks = a.Cin >> 6;
#pragma unroll
  for (int j = 0; j < MAXP; ++j)
    if ((pmask >> j) & 1) issue_patch(j, 0);
  issue_w(0, 0, 0); issue_w(1, 0, 0);
  issue_w(0, 0, 1); issue_w(1, 0, 1);
  // BNIN: the table's global loads go out BEHIND the bursts and land under them (the compiler's own wait for them also covers every
  // older burst: loads retire in order)
  if (BNIN) h8_bn_table(a, bn_tab, n_img, tid);
  wait_vm<4>();
  if (BNIN) {                                       // the first chunk's patch: all of this thread's pieces, now, as one straight-line batch
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (the table is complete; all eight wavefronts: the two groups are not yet a barrier apart)
    int ln;
    asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));
    unsigned char* const pbase = smem + P0OFF + wave * 1024 + ln * 16;
    const H8BnTab t = h8_bn_tab_load(bn_tab, a.Cin, ((ln & 7) ^ ((wave & 1) * 4 + (ln >> 4))) << 3);
    uint4 v[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
      if (8 * j + 7 < NP) v[j] = *(uint4*)(pbase + j * 8192);
      else if ((pmask >> j) & 1) bn_piece(j, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < MAXP; ++j)
        if (8 * j + 7 < NP) h8_bn_stage(q, v[j], t, bn_floor, poff[j] != ~0u);
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
      if (8 * j + 7 < NP) *(uint4*)(pbase + j * 8192) = v[j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (grp) wg_barrier();                            // group 1 runs one barrier behind group 0 from here on
  wg_barrier();
  stamp(2);

  for (int c = 0; c < nchunks; ++c) {
    const bool next_chunk = c + 1 < nchunks;
    const unsigned pmask_c = (next_chunk && !(H8_ABLATE & 2)) ? pmask : 0u;        // the pieces this chunk issues (the next chunk's patch)
    // patch pieces issued behind the filter bursts of tap tp of this chunk: pieces j = tp, tp + NTI, ...
    auto pieces_at = [&](int tp) __attribute__((always_inline)) -> int {
      int n = 0;
      if (tp >= 0 && tp < NTI) {
#pragma unroll
        for (int j = tp; j < MAXP; j += NTI) n += (int)((pmask_c >> j) & 1);
      }
      return n;
    };
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const int kh = PHM ? (tap >> 1) : tap / 3, kw = PHM ? (tap & 1) : tap - 3 * (tap / 3);     // (+ the phase's (ph, pw): in pw0)
      // K-tiles t + 1, t + 2 exist?  (t = NT c + tap)
      const bool more1 = tap < NT - 1 || next_chunk, more2 = tap < NT - 2 || next_chunk;
      const int c2 = tap < NT - 2 ? c : c + 1, tap2 = tap < NT - 2 ? tap + 2 : tap + 2 - NT;      // (chunk, tap) of K-tile t + 2
      // phase 1: (P0, C0)
      if (!(H8_ABLATE & 8) || (c == 0 && tap == 0)) load_x(kh, kw, 0);
      load_w(0);
      // C1 of this tile has landed (newer: C0 and C1 of the next tile and the patch pieces issued behind this tile's and the next
      // tile's C1, i.e. in phase 4 of the previous two K-tiles: taps tap - 2 and tap - 1 of this chunk); read after b_1
      // (the previous chunk's last two taps carry none: NTI <= NT - 2)
      if (more1) wait_vm_plus<4, 2 * PPT>(pieces_at(tap - 2) + pieces_at(tap - 1));
      else wait_vm<0>();
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();                                 // a_1
      if (BNIN && PHM && tap == 3) mma_bn(0, 0, 2, 4, c + 1);      // (see phase 4: the next chunk's pieces 2, 3; 4 and 5 follow below)
      else mma(0, 0);
      raw_barrier();                                 // b_1
      // phase 2: (P0, C1)
      load_w(1);
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();
      if (BNIN && PHM && tap == 3 && MAXP > 4) mma_bn(0, 1, 4, 5, c + 1);
      else mma(0, 1);
      raw_barrier();
      // phase 3: (P1, C1); C0 of tile t + 2 (its slot was last read in phase 1)
      if (!(H8_ABLATE & 8) || (c == 0 && tap == 0)) load_x(kh, kw, 1);
      if (more2 && !(H8_ABLATE & 4)) issue_w(0, c2, tap2);
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();
      if (BNIN && PHM && tap == 3 && MAXP > 5) mma_bn(1, 1, 5, MAXP, c + 1);      // (complete in front of b_3: first read behind b_4)
      else mma(1, 1);
      raw_barrier();
      // phase 4: (P1, C0) -- both operands are still in registers; C1 of tile t + 2 (slot last read in phase 2)
      if (more2 && !(H8_ABLATE & 4)) issue_w(1, c2, tap2);
      // ... and one piece of the next chunk's patch, BEHIND the filter bursts: loads retire in order, and a pixel burst comes from
      // HBM / the Infinity Cache, not from L2 like the filters -- as the newest operation it is never what a counted wait below
      // waits for until two K-tiles later (in front of the filters it stalled every wait of the next K-tile: 0.15 us per K-tile)
      if (tap < NTI) {
#pragma unroll
        for (int j = tap; j < MAXP; j += NTI)
          if ((pmask_c >> j) & 1) issue_patch(j, c + 1);
      }
      // C0 of the next tile has landed (newer: its C1 [, C0 and C1 of the one after], the patch pieces of this and the previous tap); read after b_4
      if (H8_ABLATE & 4) wait_vm<0>(); else
      if (more2) wait_vm_plus<6, 2 * PPT>(pieces_at(tap) + pieces_at(tap - 1));
      else if (more1) wait_vm<2>();
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();
      // BNIN: everything issued before C0 of the next tile has landed (the wait above) -- the next chunk's pieces that went out two taps
      // ago.  They are transformed HERE, in the segment whose 16 MFMAs give the vector ALU work and the LDS round trips something to hide
      // under (in front of the barrier the same work stalled all eight wavefronts: +11 % / +23 % on the plain / sub-pixel launches),
      // complete (lgkmcnt) in front of the segment's closing barrier and first read a chunk -- at least two barriers -- later.  Plain
      // form: the piece of tap - 2; sub-pixel form (all pieces behind tap 0): pieces 0, 1 here at tap 2, then 2, 3 / 4 / 5 under the first
      // three segments of tap 3.  (Also in the last chunk, on the dead patch buffer: no branch in the segment.)
      const bool bn_here = BNIN && (PHM ? tap == 2 : (tap >= 2 && tap - 2 < NTI));
      if (bn_here) mma_bn(1, 0, PHM ? 0 : tap - 2, PHM ? 2 : tap - 1, c + 1);
      else mma(1, 0);
      raw_barrier();
      wad ^= WBUF;                                  // the other filter buffer
    }
    const int pdelta = (c & 1) ? -PATCH : PATCH;    // the patches alternate (a select of two constants: the addresses stay provably 16-byte aligned)
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) AD[sx] += pdelta;
  }
  if (!grp) wg_barrier();                           // group 0 waits for group 1's last barrier: equal counts, everything read

  stamp(3);
  conv_epilogue(acc, a.bias, a.mask, a.resid, a.out, a.accumulate, a.M, a.Cout, m0 + wm * 128, co0 + wn * 64, lane,
                RowPhase{PHM ? 1 : 0, PHM ? a.lw - 1 : a.lw, PHM ? a.lh - 1 : a.lh, ph, pw, mbase}, a.resid_up ? a.lw : -1, a.lh);
  if (a.stamps) {
    stamp(4);
    wait_vm<0>();
    stamp(5);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 256 x 128-tile sibling (round 5): the same patch design for layers with Cout % 128 == 0 whose 256 x 256 grid would leave the
// chip half empty or whose Cout is 128 -- G.Block.2.Conv2 forward / data gradient (16 x 16, 256 -> 256: 128 pixel tiles at n = 128)
// and the four-phase data gradient of D.Block.1.Conv2 + mean pool (low-resolution 16 x 16, 128 -> 128).  The tile-per-tap kernel
// these ran on (conv_mfma_p8n_kernel) re-fetches the pixel tile for every tap: 21-36 % MFMA busy (profiles/r04_pmc_mfma_busy.txt).
//
//   wavefronts : 8 = 4 (pixels, 64 each) x 2 (channels, 64 each); 4 x 4 accumulator tiles; the channel half IS the ping-pong group
//   K-tile     : one tap x 64 channels = two phases of 16 MFMAs, split along K (the first / second 32 reduction elements: bytes 0-63 /
//                64-127 of every staged row), so both phases read 4 pixel + 4 filter fragments and run the same 16 accumulators
//   LDS        : ring of 4 filter K-tiles (128 rows x 128 B = 16 KiB each) + 2 patches = 64 + 2 x 43 KiB = 150 KiB
//   LDS-DMA    : the filters of K-tile t + 2 (two pieces per wavefront) go out in phase B of tile t, one piece of the next chunk's
//                patch per wavefront behind them during the chunk's first taps; ONE counted wait per K-tile in front of phase B's
//                first barrier: vmcnt(2 + patch pieces issued behind the filters of tiles t + 1 and t + 2) = tile t + 1 has landed
//                (read two barriers later: the other group runs one barrier behind).  A patch piece issued in tap s has landed by
//                the wait of tap s + 2 and is read from tap s + 3 on: the pieces ride in taps 0 .. NT - 3.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int H8N_WTILE = 128 * 128;                 // one filter K-tile
constexpr int h8n_lds_bytes(int lw) { return 2 * h8_patch_rows(lw) * 128 + 4 * H8N_WTILE; }
}  // namespace

// NPASS = 2 (sub-pixel form, Cin <= 128: both chunks' patches stay resident): one workgroup walks the phases (ph, 0) and (ph, 1) of its
// 256 low-resolution pixels back to back on the SAME staged patches -- a second K loop with the other phase's filters and window offset
// behind the first one's epilogue; its filters are prefetched through the ring as if the two loops were one.  Half the workgroups, one
// prologue and one patch fetch per pair: D.Block.1.Conv2's pooled data gradient (8 K-tiles per phase) was 2 rounds of 512 workgroups
// whose prologue + epilogue outweighed their K loop.
// GATHER (MfmaConvArgs::phase == 2): the stride-2 16-tap forms -- ConvMeanPool forward (one 4x4 stride-2 convolution with summed filters)
// and the data gradient of the upsample-3x3 sub-pixel form -- over their output's LOW-resolution grid.  Output pixel (i, j) reads
// full-resolution pixels (2i + u - 1, 2j + v - 1), u, v = 0..3: split by the parity (pa, pb) of the source pixel these are four 2x2
// convolutions over the four PARITY PLANES of the input (plane pixel (r, q) = full-resolution pixel (2r + pa, 2q + pb); tap u = 2a + 1 - pa
// reads plane row i + a - pa), each exactly the sub-pixel form's geometry with (ph, pw) = (1 - pa, 1 - pb) -- all four accumulate into the
// same tile.  So: the chunk sequence runs over (plane, 64 channels), a patch is the plane's (TR + 2) x (W + 2) window fetched with pixel
// stride 2, the fragment bases move with the plane, and a K-tile's filter columns are (u * 4 + v) * Cin + c * 64 of the gather layout.
// The tile-per-tap kernel these ran on re-fetched the pixel tile for each of the 16 taps (D.Block.1.Conv2 forward: 20 % MFMA busy on 64 x 64
// tiles; G.Block.3.Conv1's data gradient: 35 %).
template <int LW, bool RELU, bool PHM, bool BNIN = false, int NPASS = 1, bool GATHER = false>
__global__ __launch_bounds__(512) void conv_mfma_h8n_kernel(MfmaConvArgs a) {
  static_assert(NPASS == 1 || (NPASS == 2 && PHM && !BNIN), "two passes: the sub-pixel form without a batch norm on the patch");
  static_assert(!GATHER || (!PHM && !BNIN && NPASS == 1), "the gather form is its own mode");
  constexpr bool SUBP = PHM || GATHER;              // 2 x 2 taps per chunk, (low-resolution / plane) grid of W x HI pixels per image
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int W = 1 << LW, TR = 256 >> LW;
  constexpr int NT = SUBP ? 4 : 9;
  constexpr int NTI = SUBP ? (BNIN ? 1 : 2) : 6;    // taps of a chunk that carry patch pieces (<= NT - 2; BNIN: as in conv_mfma_h8_kernel)
  constexpr int PC = W + H8_PC_EXTRA, PR = TR + 2;
  constexpr int NPX = PR * PC, NROWS = h8_patch_rows(LW), NP = NROWS / 8;
  constexpr int PATCH = NROWS * 128;
  constexpr int WOFF = 0, P0OFF = 4 * H8N_WTILE, P1OFF = 4 * H8N_WTILE + PATCH;
  constexpr int MAXP = (NP + 7) / 8;
  constexpr int PPT = (MAXP + NTI - 1) / NTI;
  static_assert(MAXP <= 6 && NTI <= NT - 2, "a wavefront issues its patch pieces during the first taps of a chunk");
  static_assert((PR - 1) * PC * 128 + PC * 128 < 65536, "tap offsets are ds_read immediates");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;          // pixel quarter / channel half
  const int grp = wave >> 2;                        // waves w and w + 4 share a SIMD
  unsigned mt = blockIdx.x;
  if ((gridDim.x & 7) == 0) mt = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int co0 = blockIdx.y * 128;
  const int K = (GATHER ? 16 : NT) * a.Cin;
  const int lrow = lane >> 3, pos = lane & 7;
  const int HI = SUBP ? (a.H >> 1) : a.H, lhi = SUBP ? a.lh - 1 : a.lh;
  const long Mph = a.M >> 2;
  // one pass: tile mt = 256 pixels of phase tph; two passes: tile mt = 256 low-resolution pixels of row phase ph, column phases 0 then 1
  const unsigned tiles_ph = (unsigned)(Mph >> 8);
  const int tph0 = !PHM ? 0 : (NPASS == 2 ? 2 * (int)(mt / tiles_ph) : (int)(((long)mt * 256) / Mph));
  const int ph = tph0 >> 1;
  const unsigned ms0 = !PHM ? mt * 256u : (NPASS == 2 ? (mt % tiles_ph) * 256u : (unsigned)((long)mt * 256 - (long)tph0 * Mph));
  auto wbase_of = [&](int pp) __attribute__((always_inline)) -> const bf16_t* { return PHM ? a.wph + (long)(tph0 + pp) * a.Cout * K : (GATHER ? a.wph : a.wt); };

  // ---- patch sources (as in conv_mfma_h8_kernel)
  const unsigned n_img = ms0 >> (LW + lhi);
  const int oh0 = (int)((ms0 >> LW) & (unsigned)(HI - 1));
  unsigned poff[MAXP];
#pragma unroll
  for (int j = 0; j < MAXP; ++j) {
    const int piece = wave + 8 * j;
    const int q = piece * 8 + lrow;
    const int pr = q / PC, pc = q - pr * PC;
    const int ih = oh0 - 1 + pr, iw = pc - 1;
    const bool ok = piece < NP && q < NPX && ih >= 0 && ih < HI && iw >= 0 && iw < W;
    // (GATHER: plane pixel (ih, iw) of plane (0, 0) = full-resolution pixel (2 ih, 2 iw); the other planes are a uniform offset: issue_patch)
    poff[j] = !ok ? ~0u : GATHER
      ? 2u * (((n_img * (unsigned)a.H + 2u * (unsigned)ih) * (unsigned)a.W + 2u * (unsigned)iw) * (unsigned)a.Cin + (unsigned)((pos ^ ((q >> 1) & 7)) * 8))
      : 2u * (((n_img * (unsigned)HI + (unsigned)ih) * (unsigned)W + (unsigned)iw) * (unsigned)a.Cin + (unsigned)((pos ^ ((q >> 1) & 7)) * 8));
    if (!ok && piece < NP) {
      *(uint4*)(smem + P0OFF + piece * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
      *(uint4*)(smem + P1OFF + piece * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  unsigned pmask = 0;
#pragma unroll
  for (int j = 0; j < MAXP; ++j) pmask |= (__builtin_amdgcn_ballot_w64(poff[j] != ~0u) != 0 ? 1u : 0u) << j;
  pmask = __builtin_amdgcn_readfirstlane(pmask);

  // ---- filter sources: K-tile row r = channel co0 + r; this wavefront deposits pieces wave and wave + 8 (rows (wave + 8 j) * 8 + lrow)
  unsigned woff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave + 8 * j) * 8 + lrow;
    woff[j] = 2u * ((unsigned)(co0 + r) * (unsigned)K + (unsigned)((pos ^ ((r >> 1) & 7)) * 8));
  }

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  float* const bn_tab = (float*)(smem + P1OFF + PATCH);            // BNIN: as in conv_mfma_h8_kernel
  const int bn_ch = (pos ^ ((wave & 1) * 4 + (lrow >> 1))) << 3;
  const float bn_floor = a.bn_act == RCGAN_ACT_RELU ? 0.f : -INFINITY;
  auto bn_piece = [&](int j, int cnext) __attribute__((always_inline)) {
    if (poff[j] != ~0u) h8_bn_slot(smem + ((cnext & 1) ? P1OFF : P0OFF) + (wave + 8 * j) * 1024 + lane * 16, bn_tab, a.Cin, cnext * 64 + bn_ch, bn_floor);
  };
  auto bn_piece_seg = [&](int j, int cnext) __attribute__((always_inline)) {      // inside an MFMA segment (conv_mfma_h8_kernel)
    if (8 * j + 7 < NP)
      h8_bn_slot(smem + ((cnext & 1) ? P1OFF : P0OFF) + (wave + 8 * j) * 1024 + lane * 16, bn_tab, a.Cin, cnext * 64 + bn_ch, bn_floor, poff[j] != ~0u);
    else if ((pmask >> j) & 1)
      bn_piece(j, cnext);
  };
  const int nchunks = a.Cin >> 6;                   // 64-channel chunks; GATHER: the chunk loop runs over 4 planes x nchunks
  // (GATHER: chunk index -> (plane, channel chunk) by shifts: the plane-patch form takes power-of-two channel-chunk counts only -- a
  // division by a run-time count is ~40 vector-ALU instructions, per patch piece and per filter burst)
  const int lnc = GATHER ? 31 - __builtin_clz((unsigned)nchunks) : 0;
  auto issue_patch = [&](int j, int cnext) __attribute__((always_inline)) {
    const int plane = GATHER ? cnext >> lnc : 0, cc = GATHER ? cnext & (nchunks - 1) : cnext;
    const bf16_t* base = a.in + cc * 64 + (GATHER ? ((plane >> 1) * a.W + (plane & 1)) * a.Cin : 0);
    const unsigned dst = lds0 + ((cnext & 1) ? P1OFF : P0OFF) + (wave + 8 * j) * 1024;
    if (poff[j] != ~0u) glds16_sbase(base, poff[j], dst);
  };
  const int nvc = GATHER ? 4 * nchunks : nchunks;   // chunks the K loop walks
  auto issue_w = [&](int pp, int c, int tap) __attribute__((always_inline)) {      // K-tile t = NT (pp nvc + c) + tap -> ring slot t & 3
    int col = tap * a.Cin + c * 64;
    if (GATHER) {       // plane (pa, pb), tap (a, b) -> filter tap (u, v) = (2a + 1 - pa, 2b + 1 - pb)
      const int plane = c >> lnc, cc = c & (nchunks - 1);
      const int u = 2 * (tap >> 1) + 1 - (plane >> 1), v = 2 * (tap & 1) + 1 - (plane & 1);
      col = (u * 4 + v) * a.Cin + cc * 64;
    }
    const bf16_t* base = wbase_of(pp) + col;
    const unsigned dst = lds0 + WOFF + ((NT * (pp * nvc + c) + tap) & 3) * H8N_WTILE + wave * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16_sbase(base, woff[j], dst + j * 8192);
  };

  f32x4_t acc[4][4];       // [co fragment][px fragment]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, kc = lane >> 4;
  if (lds0 != 0) __builtin_trap();
  const int wad = WOFF + wn * 64 * 128 + frow * 128 + ((kc ^ ((frow >> 1) & 7)) * 16);      // + ring slot * H8N_WTILE
  int AD[16];

  bf16x8_t xf[4], wf[4];
  auto load_x = [&](int kh, int kw, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int p = f * 16;
      const int immP = kh * PC + kw + (p >> LW) * PC + (p & (W - 1));
      const int ad = AD[immP & 15];
      u32x4_t v = lds_read16(ks ? xor64(ad) : ad, immP * 128);
      if (RELU) { v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w); }
      xf[f] = __builtin_bit_cast(bf16x8_t, v);
    }
  };
  auto load_w = [&](int wcur, int ks) __attribute__((always_inline)) {
    const int ad = ks ? xor64(wcur) : wcur;
#pragma unroll
    for (int g = 0; g < 4; ++g) wf[g] = __builtin_bit_cast(bf16x8_t, lds_read16(ad, g * 16 * 128));
  };
  auto mma = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[g][f] = mfma16(wf[g], xf[f], acc[g][f]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto mma_bn = [&](int j0, int j1, int cnext) __attribute__((always_inline)) {      // (conv_mfma_h8_kernel's mma_bn)
    __builtin_amdgcn_s_setprio(1);
    unsigned char* const pbase = smem + ((cnext & 1) ? P1OFF : P0OFF) + wave * 1024 + lane * 16;
    const H8BnTab t = h8_bn_tab_load(bn_tab, a.Cin, cnext * 64 + bn_ch);
    uint4 v[3];
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      if (8 * j + 7 < NP) v[j - j0] = *(uint4*)(pbase + j * 8192);
      else if ((pmask >> j) & 1) bn_piece(j, cnext);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[g][f] = mfma16(wf[g], xf[f], acc[g][f]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = j0; j < j1; ++j)
        if (8 * j + 7 < NP) h8_bn_stage(g, v[j - j0], t, bn_floor, poff[j] != ~0u);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = j0; j < j1; ++j)
      if (8 * j + 7 < NP) *(uint4*)(pbase + j * 8192) = v[j - j0];
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  // ---- prologue: the first chunk's patch, the filters of K-tiles 0 and 1
#pragma unroll
  for (int j = 0; j < MAXP; ++j)
    if ((pmask >> j) & 1) issue_patch(j, 0);
  issue_w(0, 0, 0);
  issue_w(0, 0, 1);
  if (BNIN) h8_bn_table(a, bn_tab, n_img, tid);      // (behind the bursts: conv_mfma_h8_kernel)
  wait_vm<2>();                                     // the patch and K-tile 0
  if (BNIN) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    unsigned char* const pbase = smem + P0OFF + wave * 1024 + lane * 16;
    const H8BnTab t = h8_bn_tab_load(bn_tab, a.Cin, bn_ch);
    uint4 v[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
      if (8 * j + 7 < NP) v[j] = *(uint4*)(pbase + j * 8192);
      else if ((pmask >> j) & 1) bn_piece(j, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < MAXP; ++j)
        if (8 * j + 7 < NP) h8_bn_stage(q, v[j], t, bn_floor, poff[j] != ~0u);
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
      if (8 * j + 7 < NP) *(uint4*)(pbase + j * 8192) = v[j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (grp) wg_barrier();                            // group 1 runs one barrier behind group 0 from here on
  wg_barrier();

  int slot = 0;                                     // ring slot of the current K-tile (wave-uniform)
#pragma unroll 1
  for (int pp = 0; pp < NPASS; ++pp) {
  const int pw = PHM ? ((tph0 + pp) & 1) : 0;
  auto set_window = [&](int dph, int dpw, int buf) __attribute__((always_inline)) {      // fragment bases: window offset (dph, dpw) into patch `buf`
    const int pw0 = ((wm * 64) >> LW) * PC + frow + dph * PC + dpw;
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) AD[sx] = (buf ? P1OFF : P0OFF) + pw0 * 128 + ((kc ^ (((pw0 + sx) & 15) >> 1)) << 4);
  };
  // this pass's window into patch 0 (the chunk loop below toggles the bases and may leave them on patch 1)
  if (!GATHER) set_window(PHM ? ph : 0, PHM ? pw : 0, 0);
  const bool next_pass = pp + 1 < NPASS;
  for (int c = 0; c < nvc; ++c) {
    const bool next_chunk = c + 1 < nvc;
    if (GATHER) { const int plane = c >> lnc; set_window(1 - (plane >> 1), 1 - (plane & 1), c & 1); }      // the plane's window, this chunk's patch
    const unsigned pmask_c = (next_chunk && pp == 0) ? pmask : 0u;      // (a second pass finds every chunk's patch where the first left it)
    auto pieces_at = [&](int tp) __attribute__((always_inline)) -> int {
      int n = 0;
      if (tp >= 0 && tp < NTI) {
#pragma unroll
        for (int j = tp; j < MAXP; j += NTI) n += (int)((pmask_c >> j) & 1);
      }
      return n;
    };
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const int kh = SUBP ? (tap >> 1) : tap / 3, kw = SUBP ? (tap & 1) : tap - 3 * (tap / 3);
      const bool more1 = tap < NT - 1 || next_chunk || next_pass, more2 = tap < NT - 2 || next_chunk || next_pass;
      const int tap2 = tap < NT - 2 ? tap + 2 : tap + 2 - NT;
      const int c2 = tap < NT - 2 ? c : (next_chunk ? c + 1 : 0), pp2 = (tap < NT - 2 || next_chunk) ? pp : pp + 1;
      const int wcur = wad + slot * H8N_WTILE;
      // phase A: the first 32 reduction elements
      load_x(kh, kw, 0);
      load_w(wcur, 0);
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();
      if (BNIN && SUBP && tap == 3) mma_bn(3, MAXP, c + 1);      // (see phase B: the second half of the next chunk's pieces; complete in front of b_A, first read behind b_B)
      else mma();
      raw_barrier();
      // phase B: the second 32; the filters of K-tile t + 2 (its slot was last read in K-tile t - 2) and one patch piece behind them
      load_x(kh, kw, 1);
      load_w(wcur, 1);
      if (more2) issue_w(pp2, c2, tap2);
      if (tap < NTI) {
#pragma unroll
        for (int j = tap; j < MAXP; j += NTI)
          if ((pmask_c >> j) & 1) issue_patch(j, c + 1);
      }
      // K-tile t + 1 has landed (newer: the patch pieces issued behind it, K-tile t + 2 and the patch pieces behind that)
      if (more2) wait_vm_plus<2, 2 * PPT>(pieces_at(tap - 1) + pieces_at(tap));
      else if (more1) wait_vm<0>();
      __builtin_amdgcn_sched_barrier(0);
      raw_barrier();
      // BNIN: the next chunk's pieces that went out two taps ago have landed (the wait above): transformed under this segment's MFMAs
      // (conv_mfma_h8_kernel, phase 4)
      const bool bn_here = BNIN && (SUBP ? tap == 2 : (tap >= 2 && tap - 2 < NTI));
      if (bn_here) mma_bn(SUBP ? 0 : tap - 2, SUBP ? 3 : tap - 1, c + 1);      // (also in the last chunk, on the dead patch buffer: no branch in the segment)
      else mma();
      raw_barrier();
      slot = (slot + 1) & 3;
    }
    if (!GATHER) {
      const int pdelta = (c & 1) ? -PATCH : PATCH;
#pragma unroll
      for (int sx = 0; sx < 16; ++sx) AD[sx] += pdelta;
    }
  }
  // the end of a pass: group 0 waits for group 1's last barrier -- both run the epilogue side by side (a barrier apart they would take
  // turns: each group's next barrier needs the other one past its epilogue); group 1 falls behind again in front of the next pass
  if (!grp) wg_barrier();
  {
    const long mbase = PHM ? (long)(tph0 + pp) * Mph : 0;
    // (measured and dropped: requesting ALL pixel rows' ReLU-mask operands in front of the first store instead of one row ahead -- D.Block.1.Conv2's
    // pooled data gradient 31.1 -> 32.9 us: the launch is throughput-, not latency-bound on its 75 MB)
    conv_epilogue(acc, a.bias, a.mask, a.resid, a.out, a.accumulate, a.M, a.Cout, mbase + ms0 + wm * 64, co0 + wn * 64, lane,
                  RowPhase{PHM ? 1 : 0, PHM ? a.lw - 1 : a.lw, PHM ? a.lh - 1 : a.lh, ph, pw, mbase}, a.resid_up ? a.lw : -1, a.lh);
  }
  if (next_pass) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (grp) wg_barrier();
  }
  }
}

// plain 3x3 stride-1 SAME convolution (forward, or the data gradient with the rotated filters) on 16- or 32-pixel-wide power-of-two
// images whose 256-pixel tiles are whole image rows -- or (phase == 1) the sub-pixel form of an upsample-3x3 convolution whose
// LOW-resolution grid is such an image
bool mfma_conv8_halo_takes(const MfmaConvArgs& a) {
  if (a.KH != 3 || a.KW != 3 || a.stats || a.lw < 0 || a.lh < 0 || a.M % 256 || a.Cin % 64 || a.Cout % 256) return false;
  if ((long)a.N * a.H * a.W * a.Cin >= (1L << 31) || (long)a.Cout * 9 * a.Cin >= (1L << 31)) return false;      // 32-bit byte offsets
  if (a.phase == 0) return a.PT == 1 && a.PL == 1 && !a.up && (a.lw == 4 || a.lw == 5) && (a.H << a.lw) % 256 == 0;
  if (a.phase == 1) return a.up && a.wph != nullptr && (a.lw == 5 || a.lw == 6) && a.lh >= 1 && ((a.H >> 1) << (a.lw - 1)) % 256 == 0 && (a.M >> 2) % 256 == 0;
  return false;
}

#define H8_BN_MAX_CIN 1024       /* the batch-norm table [2][Cin] fp32 sits behind the patches: 150 KiB + 8 KiB <= 160 KiB */
template <int LW, bool RELU, bool PHM, bool BNIN = false>
static int launch8h(rcgan_ctx* ctx, const MfmaConvArgs& a) {
  static bool attr_set = false;
  const size_t lds = h8_lds_bytes(LW) + (BNIN ? 2 * H8_BN_MAX_CIN * sizeof(float) : 0);
  if (!attr_set) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_h8_kernel<LW, RELU, PHM, BNIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, 256), a.Cout / 256);
  {
    ProfScope ps(ctx, RCGAN_PROF_CONV_P8, 2.0 * (double)a.M * 9 * a.Cin * a.Cout, 2.0 * (double)a.M * (PHM ? 4 : 9) * a.Cin * a.Cout);
    if (BNIN && ps.on) ctx->prof_bn_in++;
    hipLaunchKernelGGL((conv_mfma_h8_kernel<LW, RELU, PHM, BNIN>), grid, dim3(512), lds, ctx->stream, a);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int mfma_conv8_halo_launch(rcgan_ctx* ctx, const MfmaConvArgs& a) {
  MfmaConvArgs b = a;
  b.stamps = (unsigned long long*)ctx->dbg_stamps;
  if (a.bn_mean) {        // batch norm + activation on the staged patch (its own activation: no input-ReLU flavour)
    if (a.relu_in || a.Cin > H8_BN_MAX_CIN || (a.bn_act != RCGAN_ACT_NONE && a.bn_act != RCGAN_ACT_RELU))
      RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "batch norm on the staged patch: no input ReLU, Cin <= %d, activation none / ReLU", H8_BN_MAX_CIN);
    if (a.phase == 1) return a.lw == 6 ? launch8h<5, false, true, true>(ctx, b) : launch8h<4, false, true, true>(ctx, b);
    return a.lw == 5 ? launch8h<5, false, false, true>(ctx, b) : launch8h<4, false, false, true>(ctx, b);
  }
  if (a.phase == 1) {
    if (a.lw == 6) return a.relu_in ? launch8h<5, true, true>(ctx, b) : launch8h<5, false, true>(ctx, b);
    return a.relu_in ? launch8h<4, true, true>(ctx, b) : launch8h<4, false, true>(ctx, b);
  }
  if (a.lw == 5) return a.relu_in ? launch8h<5, true, false>(ctx, b) : launch8h<5, false, false>(ctx, b);
  return a.relu_in ? launch8h<4, true, false>(ctx, b) : launch8h<4, false, false>(ctx, b);
}

// the 256 x 128 sibling: the same shapes at Cout % 128 == 0
bool mfma_conv8n_halo_takes(const MfmaConvArgs& a) {
  if (a.KH != 3 || a.KW != 3 || a.stats || a.lw < 0 || a.lh < 0 || a.M % 256 || a.Cin % 64 || a.Cout % 128) return false;
  if ((long)a.N * a.H * a.W * a.Cin >= (1L << 31) || (long)a.Cout * 16 * a.Cin >= (1L << 31)) return false;
  // the stride-2 16-tap forms over their low-resolution output grid (a.H, a.W = the full-resolution source grid, a.M = output pixels)
  if (a.phase == 2) return !a.up && a.wph != nullptr && !a.bn_mean && (a.lw == 5 || a.lw == 6) && a.lh >= 1 && ((a.H >> 1) << (a.lw - 1)) % 256 == 0 &&
                           ((a.Cin >> 6) & ((a.Cin >> 6) - 1)) == 0;      // (power-of-two channel-chunk count: the chunk -> plane map is a shift)
  if (a.phase == 0) return a.PT == 1 && a.PL == 1 && !a.up && (a.lw == 4 || a.lw == 5) && (a.H << a.lw) % 256 == 0;
  if (a.phase == 1) return a.up && a.wph != nullptr && (a.lw == 5 || a.lw == 6) && a.lh >= 1 && ((a.H >> 1) << (a.lw - 1)) % 256 == 0 && (a.M >> 2) % 256 == 0;
  return false;
}

template <int LW, bool RELU, bool PHM, bool BNIN = false, int NPASS = 1, bool GATHER = false>
static int launch8hn(rcgan_ctx* ctx, const MfmaConvArgs& a) {
  static bool attr_set = false;
  const size_t lds = h8n_lds_bytes(LW) + (BNIN ? 2 * H8_BN_MAX_CIN * sizeof(float) : 0);
  if (!attr_set) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_h8n_kernel<LW, RELU, PHM, BNIN, NPASS, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, 256 * NPASS), a.Cout / 128);
  {
    // (GATHER: a.M counts low-resolution pixels; the reference's formulation is the 3x3 layer over the full-resolution grid)
    ProfScope ps(ctx, RCGAN_PROF_CONV_P8N, 2.0 * (double)a.M * (GATHER ? 36 : 9) * a.Cin * a.Cout,
                 2.0 * (double)a.M * (GATHER ? 16 : PHM ? 4 : 9) * a.Cin * a.Cout);
    if (BNIN && ps.on) ctx->prof_bn_in++;
    hipLaunchKernelGGL((conv_mfma_h8n_kernel<LW, RELU, PHM, BNIN, NPASS, GATHER>), grid, dim3(512), lds, ctx->stream, a);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int mfma_conv8n_halo_launch(rcgan_ctx* ctx, const MfmaConvArgs& a) {
  if (a.bn_mean) {
    if (a.relu_in || a.Cin > H8_BN_MAX_CIN || (a.bn_act != RCGAN_ACT_NONE && a.bn_act != RCGAN_ACT_RELU))
      RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "batch norm on the staged patch: no input ReLU, Cin <= %d, activation none / ReLU", H8_BN_MAX_CIN);
    if (a.phase == 1) return a.lw == 6 ? launch8hn<5, false, true, true>(ctx, a) : launch8hn<4, false, true, true>(ctx, a);
    return a.lw == 5 ? launch8hn<5, false, false, true>(ctx, a) : launch8hn<4, false, false, true>(ctx, a);
  }
  if (a.phase == 2) {
    if (a.lw == 6) return a.relu_in ? launch8hn<5, true, false, false, 1, true>(ctx, a) : launch8hn<5, false, false, false, 1, true>(ctx, a);
    return a.relu_in ? launch8hn<4, true, false, false, 1, true>(ctx, a) : launch8hn<4, false, false, false, 1, true>(ctx, a);
  }
  if (a.phase == 1) {
    // both column phases of a row phase in one workgroup where both chunks' patches fit (Cin <= 128) and halving the grid still fills the
    // chip's 256 CUs; RCGAN_H8N_TWO_PASS=0 keeps one phase per workgroup
    static int two = -1;
    if (two < 0) { const char* e = getenv("RCGAN_H8N_TWO_PASS"); two = e ? atoi(e) : 1; }
    if (two && a.Cin <= 128 && (a.M / 512) * (a.Cout / 128) >= 190) {
      if (a.lw == 6) return a.relu_in ? launch8hn<5, true, true, false, 2>(ctx, a) : launch8hn<5, false, true, false, 2>(ctx, a);
      return a.relu_in ? launch8hn<4, true, true, false, 2>(ctx, a) : launch8hn<4, false, true, false, 2>(ctx, a);
    }
    if (a.lw == 6) return a.relu_in ? launch8hn<5, true, true>(ctx, a) : launch8hn<5, false, true>(ctx, a);
    return a.relu_in ? launch8hn<4, true, true>(ctx, a) : launch8hn<4, false, true>(ctx, a);
  }
  if (a.lw == 5) return a.relu_in ? launch8hn<5, true, false>(ctx, a) : launch8hn<5, false, false>(ctx, a);
  return a.relu_in ? launch8hn<4, true, false>(ctx, a) : launch8hn<4, false, false>(ctx, a);
}
