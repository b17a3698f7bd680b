// 256 x 256-tile, 8-wavefront, four-phases-per-K-tile schedule of the implicit-GEMM convolution (forward and data
// gradient of the 3x3 / 1x1 stride-1 SAME convs with Cout % 256 == 0: the 256-channel generator blocks).
//
// The 128 x 128 kernel (conv_mfma.hip) drains its one in-flight K-tile before every barrier; that structure tops
// out near 900 TFLOP/s.  Here a K-tile (64 reduction elements) is cut into four 16-KiB half-tiles, each fetched by
// one LDS-DMA burst (2 x global_load_lds_dwordx4 per wavefront), and the work on it into four phases of 16 MFMAs
// (one quadrant of the wavefront's 128 px x 64 co tile over the whole K-tile).  Every phase issues the burst of one
// half-tile of the NEXT K-tile, so loads, LDS fragment reads and MFMAs of different wavefronts interleave, and the
// s_waitcnt before each barrier is COUNTED (vmcnt(4): the two newest bursts stay in flight across it).
//
//   workgroup tile : 256 pixels x 256 output channels, K-tile 64 = one filter tap x 64 input channels
//   wavefronts     : 8 = 2 (pixels, 128 each) x 4 (channels, 64 each); accumulators 4 x 8 MFMA tiles = 128 AGPRs
//   half-tiles     : P0 / P1 = the first / second 64 pixels of BOTH wavefront rows, C0 / C1 = the first / second
//                    32 channels of ALL FOUR wavefront columns -- i.e. exactly what phase 1 (P0,C0), 2 (+C1),
//                    3 (+P1) read, whatever the wavefront; phase 4 (P1,C0) runs out of registers
//   schedule       : tile t, phase p issues half-tile p of tile t+1 in the order P0, C0, C1, P1; needed at
//                    (t+1).1, (t+1).1, (t+1).2, (t+1).3 -> at most the two newest bursts may be outstanding at
//                    each wait.  LDS: 2 K-tile buffers x (256 + 256) rows x 128 B = 128 KiB.
//   hazards        : a burst is read only after (own vmcnt wait) + (workgroup barrier); a half-tile region is
//                    re-filled at the earliest three phases after its last read, with >= 1 barrier in between.
#include "conv_mfma.h"
#include "mfma_util.h"

// timing-only ablations of the ping-pong K loop (scripts/build_p8_ablate.sh: results are wrong by construction):
// 1 = no LDS-DMA issue, 2 = no fragment reads after the first K-tile, 4 = no MFMAs, 8 = no barriers, 16 = no pixel (X) LDS-DMA, filters only
#ifndef P8_ABLATE
#define P8_ABLATE 0
#endif

namespace {

template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else static_assert(N == 0, "unsupported vmcnt immediate");
}

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// without the LDS wait: fragment reads stay in flight across it (the compiler's counted lgkmcnt waits sit in front of the MFMAs)
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

}  // namespace

// PP ("ping-pong", round 4): the two wavefronts of a SIMD (waves w and w + 4) run ONE BARRIER APART, and every phase gets a second
// barrier between its fragment reads and its MFMAs -- while one group of four wavefronts (one per SIMD) multiplies, the other reads
// its fragments and issues its bursts, instead of all eight reading at once and then queueing for the matrix pipe (the schedule of
// the guide's 8-phase GEMM template).  The counted waits move in front of the phase's FIRST barrier: a burst is read one barrier
// after the wait that retires it PLUS the one-barrier lag of the other group.
template <bool RELU, bool XCDSWZ, bool STATS = false, bool PP = false>
__global__ __launch_bounds__(512) void conv_mfma_p8_kernel(MfmaConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HALF = 128 * 128;                 // bytes of one half-tile (128 rows x 128 B)
  constexpr int XOFF = 0, WOFF = 2 * HALF, BUF = 4 * HALF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;        // pixel half / channel quarter of this wavefront
  // workgroups are dealt round-robin to the 8 XCDs (8 private L2s): give each XCD a CONTIGUOUS run of pixel tiles so the
  // SAME-padding halo rows shared by neighbouring tiles are L2 hits (bijective when the grid is a multiple of 8)
  unsigned mt = blockIdx.x;
  if (XCDSWZ && (gridDim.x & 7) == 0) mt = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const long m0 = (long)mt * 256;
  const int co0 = blockIdx.y * 256;
  const int Hs = a.up ? (a.H >> 1) : a.H, Ws = a.up ? (a.W >> 1) : a.W;
  // sub-pixel form (a.phase): the tile's pixel index runs over (phase, n, i, j) of the low-resolution grid; all 256 pixels of a
  // tile share the phase (pixels per phase % 256 == 0, checked by the launcher)
  // a.phase == 2: the data gradient of that form -- dx over the low-resolution grid gathers the 4 x 4 neighbourhood
  // (2p + u - 1, 2q + v - 1) of the full-resolution dy, each position with the transposed summed filter that reaches it
  // (16 taps, source stride 2; a.H, a.W = the dy grid, a.M = low-resolution pixels)
  const bool phm = a.phase == 1, dgm = a.phase == 2;
  const long Mph = a.M >> 2;
  const int tph = phm ? (int)(m0 / Mph) : 0, ph = tph >> 1, pw = tph & 1;
  const long mbase = phm ? (long)tph * Mph : 0;
  const int glw = (phm || dgm) ? a.lw - 1 : a.lw, glh = (phm || dgm) ? a.lh - 1 : a.lh;
  const int GH = phm ? Hs : (dgm ? (a.H >> 1) : a.H), GW = phm ? Ws : (dgm ? (a.W >> 1) : a.W);     // grid of the tile's pixel index
  const int BH = phm ? Hs : a.H, BW = phm ? Ws : a.W;                                              // grid of the source pixels
  const int sstr = dgm ? 2 : 1;
  const int kwn = phm ? 2 : (dgm ? 4 : a.KW), ntaps = phm ? 4 : (dgm ? 16 : a.KH * a.KW);
  const int pt = phm ? 1 - ph : (dgm ? 1 : a.PT), pl = phm ? 1 - pw : (dgm ? 1 : a.PL);
  const bool up = a.up && !phm && !dgm;
  const int K = ntaps * a.Cin;
  const int KT = K / 64;
  const bf16_t* const wbase = phm ? a.wph + (long)tph * a.Cout * K : (dgm ? a.wph : a.wt);
  const int lrow = lane >> 3, pos = lane & 7;

  // ---- tap-source table: T[tap][p] = element offset (into a.in) of the pixel that tile pixel p reads for that tap, or
  //      ~0u where SAME padding applies.  Built once per workgroup behind the tile buffers (9 x 256 x 4 B); the K loop
  //      then needs one ds_read_b32 + a select per DMA row when the tap changes instead of a bounds test and a pixel
  //      address -- which is what makes a tap change per K-tile (channel-major order, a.cm) affordable.
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (a.stamps && tid == 0) a.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  if (a.stamps && tid == 0) {
    a.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_ID
    a.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // XCC_ID
  }
  unsigned* const tapt = (unsigned*)(smem + 2 * BUF);
  for (int e = tid; e < ntaps * 256; e += 512) {
    const int tap = e >> 8;
    const long m = m0 + (e & 255);
    unsigned off = ~0u;
    if (m < a.M) {
      int n, oh, ow;
      const unsigned mm = (unsigned)(m - mbase);
      if (a.lw >= 0) {
        ow = (int)(mm & (unsigned)(GW - 1));
        oh = (int)((mm >> glw) & (unsigned)(GH - 1));
        n = (int)(mm >> (glw + glh));
      } else {
        ow = (int)(m % a.W);
        oh = (int)((m / a.W) % a.H);
        n = (int)(m / ((long)a.W * a.H));
      }
      const int kh = tap / kwn, kw = tap - kh * kwn;
      int ih = oh * sstr + kh - pt, iw = ow * sstr + kw - pl;
      if (ih >= 0 && ih < BH && iw >= 0 && iw < BW) {
        if (up) { ih >>= 1; iw >>= 1; }
        off = (((unsigned)n * Hs + ih) * Ws + iw) * a.Cin;
      }
    }
    tapt[e] = off;
  }
  __syncthreads();
  stamp(1);

  // ---- DMA roles: per half-tile this wavefront deposits rows (wave*2 + j)*8 + lrow, j = 0,1 (of 128) ----------------
  // LDS row r of X half h  <->  tile pixel (r>>6)*128 + h*64 + (r&63);  W half h: channel (r>>5)*64 + h*32 + (r&31)
  int pix[4], a_coff[4];                          // index = h*2 + j
  const bf16_t* wsrc[4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (wave * 2 + j) * 8 + lrow;
      const int swz = (pos ^ ((r >> 1) & 7)) * 8;
      pix[h * 2 + j] = (r >> 6) * 128 + h * 64 + (r & 63);
      a_coff[h * 2 + j] = swz;
      const int co = co0 + (r >> 5) * 64 + h * 32 + (r & 31);
      wsrc[h * 2 + j] = wbase + (long)co * K + swz;
    }

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const bf16_t* rp[4];
  int rstep[4];
  unsigned noff[4];                               // offsets of the NEXT tap, read ahead of their use
  bool pend = false;
  int i_c0 = 0, i_tap = 0, i_k0 = 0;
  auto read_tap = [&](int tap) {
#pragma unroll
    for (int i = 0; i < 4; ++i) noff[i] = tapt[tap * 256 + pix[i]];
  };
  auto use_tap = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = noff[i] != ~0u;
      rp[i] = ok ? a.in + noff[i] + a_coff[i] : a.zero;
      rstep[i] = ok ? 1 : 0;
    }
  };
  read_tap(0);
  use_tap();
  // half-tile ids: 0 = P0, 1 = C0, 2 = C1, 3 = P1 (issue order); the cursor moves on after P1
  auto issue = [&](int which, int buf) {
    const unsigned base = lds0 + buf * BUF;
    if (which == 0 || which == 3) {
      const int h = which == 0 ? 0 : 1;
      if (which == 0 && pend) { use_tap(); pend = false; }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (!(P8_ABLATE & 16)) glds16_asm(rp[h * 2 + j] + i_c0 * rstep[h * 2 + j], base + XOFF + h * HALF + (wave * 2 + j) * 1024);
      if (which == 3) {
        if (a.cm) {
          // channel-major: the taps of one 64-channel chunk back to back.  They read the same 128-byte line of every
          // pixel (shifted by a pixel or a row), so eight of the nine passes over the tile are L2 hits instead of
          // coming back from the fabric after the other workgroups of the XCD have swept the 4 MB L2.
          if (++i_tap == ntaps) { i_tap = 0; i_c0 += 64; }
          i_k0 = i_tap * a.Cin + i_c0;
          read_tap(i_tap);
          pend = true;
        } else {
          i_k0 += 64;
          i_c0 += 64;
          if (i_c0 == a.Cin) {
            i_c0 = 0;
            ++i_tap;
            read_tap(i_tap < ntaps ? i_tap : 0);
            pend = true;
          }
        }
      }
    } else {
      const int h = which - 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16_asm(wsrc[h * 2 + j] + i_k0, base + WOFF + h * HALF + (wave * 2 + j) * 1024);
    }
  };

  f32x4_t acc[4][8];       // [co fragment = c-half*2 + g][px fragment = p-half*4 + f]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, kc = lane >> 4;
  const int foff0 = frow * 128 + ((kc ^ ((frow >> 1) & 7)) * 16);
  const int foff1 = frow * 128 + (((4 + kc) ^ ((frow >> 1) & 7)) * 16);
  const int xrow0 = wm * 64 * 128;                // byte offset of this wavefront's rows inside an X half-tile
  const int wrow0 = wn * 32 * 128;                // ... inside a W half-tile

  bf16x8_t xf[2][4], wfc[2][2][2];                // [ks][fragment] of the current pixel half; [channel half][ks][fragment]
  auto load_x = [&](const unsigned char* bufp, int h) {
    const unsigned char* p = bufp + XOFF + h * HALF + xrow0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        uint4 v = *(const uint4*)(p + f * 16 * 128 + (ks ? foff1 : foff0));
        if (RELU) { v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w); }
        xf[ks][f] = __builtin_bit_cast(bf16x8_t, v);
      }
  };
  auto load_w = [&](const unsigned char* bufp, int h) {
    const unsigned char* p = bufp + WOFF + h * HALF + wrow0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int g = 0; g < 2; ++g) wfc[h][ks][g] = *(const bf16x8_t*)(p + g * 16 * 128 + (ks ? foff1 : foff0));
  };
  auto mma = [&](int ph, int ch) {
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

  // prologue: the whole first K-tile
#pragma unroll
  for (int w = 0; w < 4; ++w) issue(w, 0);
  wait_vm<0>();
  const int grp = wave >> 2;                       // waves w and w + 4 share a SIMD
  if (PP && grp) wg_barrier();                     // group 1 runs one barrier behind group 0 from here on
  wg_barrier();
  stamp(2);

  for (int t = 0; PP && t < KT; ++t) {
    const unsigned char* bufp = smem + (t & 1) * BUF;
    const int nb = (t + 1) & 1;
    const bool more = t + 1 < KT;
    // phase 1: (P0, C0)
    if (!(P8_ABLATE & 2) || t == 0) { load_x(bufp, 0); load_w(bufp, 0); }
    if (more && !(P8_ABLATE & 1)) issue(0, nb);
    if (P8_ABLATE & 16) wait_vm<0>(); else
    if (more) wait_vm<4>(); else wait_vm<2>();    // C1 of this tile has landed (newer: P1 [, P0 of the next tile]); read after b_1
    __builtin_amdgcn_sched_barrier(0);
    if (!(P8_ABLATE & 8)) raw_barrier();                                 // a_1
    if (!(P8_ABLATE & 4)) mma(0, 0);
    if (!(P8_ABLATE & 8)) raw_barrier();                                 // b_1
    // phase 2: (P0, C1)
    if (!(P8_ABLATE & 2) || t == 0) load_w(bufp, 1);
    if (more && !(P8_ABLATE & 1)) issue(1, nb);
    if (P8_ABLATE & 16) {} else
    if (more) wait_vm<4>(); else wait_vm<0>();    // P1 of this tile has landed; read after b_2
    __builtin_amdgcn_sched_barrier(0);
    if (!(P8_ABLATE & 8)) raw_barrier();
    if (!(P8_ABLATE & 4)) mma(0, 1);
    if (!(P8_ABLATE & 8)) raw_barrier();
    // phase 3: (P1, C1)
    if (!(P8_ABLATE & 2) || t == 0) load_x(bufp, 1);
    if (more && !(P8_ABLATE & 1)) issue(2, nb);
    __builtin_amdgcn_sched_barrier(0);
    if (!(P8_ABLATE & 8)) raw_barrier();
    if (!(P8_ABLATE & 4)) mma(1, 1);
    if (!(P8_ABLATE & 8)) raw_barrier();
    // phase 4: (P1, C0) -- both operands are still in registers
    if (more && !(P8_ABLATE & 1)) issue(3, nb);
    if (P8_ABLATE & 16) wait_vm<2>(); else
    if (more) wait_vm<4>();                       // P0 and C0 of the next tile have landed (newer: its C1, P1); read after b_4
    __builtin_amdgcn_sched_barrier(0);
    if (!(P8_ABLATE & 8)) raw_barrier();
    if (!(P8_ABLATE & 4)) mma(1, 0);
    if (!(P8_ABLATE & 8)) raw_barrier();
  }
  if (PP && !grp) wg_barrier();                    // group 0 waits for group 1's last barrier: equal counts, everything read

  for (int t = 0; !PP && t < KT; ++t) {
    const unsigned char* bufp = smem + (t & 1) * BUF;
    const int nb = (t + 1) & 1;
    const bool more = t + 1 < KT;
    // phase 1: (P0, C0)
    load_x(bufp, 0);
    load_w(bufp, 0);
    if (more) issue(0, nb);
    __builtin_amdgcn_sched_barrier(0);
    mma(0, 0);
    if (more) wait_vm<4>(); else wait_vm<2>();    // C1 of this tile has landed (newer: P1 [, P0 of the next tile])
    wg_barrier();
    // phase 2: (P0, C1)
    load_w(bufp, 1);
    if (more) issue(1, nb);
    __builtin_amdgcn_sched_barrier(0);
    mma(0, 1);
    if (more) wait_vm<4>(); else wait_vm<0>();    // P1 of this tile has landed
    wg_barrier();
    // phase 3: (P1, C1)
    load_x(bufp, 1);
    if (more) issue(2, nb);
    __builtin_amdgcn_sched_barrier(0);
    mma(1, 1);
    wg_barrier();                                  // (C0 of this tile landed long ago)
    // phase 4: (P1, C0) -- both operands are still in registers
    if (more) issue(3, nb);
    __builtin_amdgcn_sched_barrier(0);
    mma(1, 0);
    if (more) wait_vm<4>();                       // P0 and C0 of the next tile have landed (newer: its C1, P1)
    wg_barrier();
  }

  // epilogue: lane holds out[pixel (lane&15)][co .. co+3], co = 4*(lane>>4)
  stamp(3);
  if constexpr (STATS) {
    // batch-norm statistics of the layer behind this convolution: column sums of the stored tile (bias, residual, rounding
    // included) and of its squares.  Lane partials -> 16 pixel lanes (DPP) -> the two pixel halves (LDS) -> [tile][Cout][2].
    float st[2][16];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int k = 0; k < 16; ++k) st[p][k] = 0.f;
    conv_epilogue<4, 8, RowPhase, true>(acc, a.bias, a.mask, a.resid, a.out, a.accumulate, a.M, a.Cout, m0 + wm * 128, co0 + wn * 64, lane,
                                        RowPhase{phm ? 1 : 0, glw, glh, ph, pw, mbase}, a.resid_up ? a.lw : -1, a.lh, st);
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int k = 0; k < 16; ++k) st[p][k] = row16_sum(st[p][k]);
    float* const red = (float*)smem;            // [pixel half][256 channels][2]: the tile buffers are dead (last barrier of the K loop)
    if ((lane & 15) == 0) {
      const int row = lane >> 4;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = wn * 64 + (2 * p + (row & 1)) * 16 + (row >> 1) * 8 + e;
          red[(wm * 256 + c) * 2 + 0] = st[p][e];
          red[(wm * 256 + c) * 2 + 1] = st[p][8 + e];
        }
    }
    __syncthreads();
    a.stats[((long)mt * a.Cout + co0) * 2 + tid] = red[tid] + red[512 + tid];
  } else {
    conv_epilogue(acc, a.bias, a.mask, a.resid, a.out, a.accumulate, a.M, a.Cout, m0 + wm * 128, co0 + wn * 64, lane,
                  RowPhase{phm ? 1 : 0, glw, glh, ph, pw, mbase}, a.resid_up ? a.lw : -1, a.lh);
  }
  if (a.stamps) {
    stamp(4);
    wait_vm<0>();
    stamp(5);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Persistent form of the 256 x 256 kernel: one workgroup per CU walks its tiles in ONE continuous K-tile stream.
//
// Measured on the one-tile-per-workgroup kernel above (scripts/exp_p8_fixed_cost.py, 32x32 convs, Cout 256):
// time per round of 256 workgroups = 12-16 us + 1.59 us per K-tile.  The K loop itself runs at 1.35 PFLOP/s; the fixed
// part -- tap table, the first K-tile's round trip, and 256 CUs storing their 128 KB of output at the same moment with
// nothing left to multiply (one workgroup per CU: LDS) -- is a fifth of a 36-K-tile tile and half of a 1x1 shortcut's.
// Here a workgroup takes tiles b, b+P, b+2P, ...: the LDS-DMA cursor runs one K-tile ahead of the MFMAs ACROSS tile
// boundaries (the last K-tile of a tile issues the first K-tile of the next), the epilogue's stores are issued and left
// to drain under the next tile's MFMAs (stores only count in vmcnt: the counted waits stay sufficient, they then also
// cover the older stores), and the tap-source table of tile i+2 is built in the shadow of the tile i -> i+1 boundary
// (two tables, alternating).  Cost of a tile boundary: the epilogue's own issue time.
// ---------------------------------------------------------------------------------------------------------
template <bool RELU>
__global__ __launch_bounds__(512) void conv_mfma_p8p_kernel(MfmaConvArgs a, int tiles_m, int xcd_swz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HALF = 128 * 128;
  constexpr int XOFF = 0, WOFF = 2 * HALF, BUF = 4 * HALF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int co0 = blockIdx.y * 256;
  const int K = a.KH * a.KW * a.Cin;
  const int KT = K / 64;
  const int Hs = a.up ? (a.H >> 1) : a.H, Ws = a.up ? (a.W >> 1) : a.W;
  const int lrow = lane >> 3, pos = lane & 7;
  const int ntaps = a.KH * a.KW;
  const int P = gridDim.x;
  const int my_tiles = (tiles_m - (int)blockIdx.x + P - 1) / P;          // >= 1 (grid <= tiles)
  // virtual block id v -> pixel tile: each XCD (v % 8 == blockIdx.x % 8 as P % 8 == 0) owns a contiguous run of tiles
  auto tile_of = [&](int i) __attribute__((always_inline)) -> int {
    const int v = (int)blockIdx.x + i * P;
    return xcd_swz ? (v & 7) * (tiles_m >> 3) + (v >> 3) : v;
  };

  unsigned* const tapt = (unsigned*)(smem + 2 * BUF);                   // two tables of ntaps x 256 offsets
  const int tab_words = ntaps * 256;
  auto build_table = [&](int slot, long m0) __attribute__((always_inline)) {
    unsigned* const T = tapt + slot * tab_words;
    for (int e = tid; e < tab_words; e += 512) {
      const int tap = e >> 8;
      const long m = m0 + (e & 255);
      unsigned off = ~0u;
      if (m < a.M) {
        int n, oh, ow;
        const unsigned mm = (unsigned)m;
        if (a.lw >= 0) {
          ow = (int)(mm & (unsigned)(a.W - 1));
          oh = (int)((mm >> a.lw) & (unsigned)(a.H - 1));
          n = (int)(mm >> (a.lw + a.lh));
        } else {
          ow = (int)(m % a.W);
          oh = (int)((m / a.W) % a.H);
          n = (int)(m / ((long)a.W * a.H));
        }
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        int ih = oh + kh - a.PT, iw = ow + kw - a.PL;
        if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) {
          if (a.up) { ih >>= 1; iw >>= 1; }
          off = (((unsigned)n * Hs + ih) * Ws + iw) * a.Cin;
        }
      }
      T[e] = off;
    }
  };
  build_table(0, (long)tile_of(0) * 256);
  if (my_tiles > 1) build_table(1, (long)tile_of(1) * 256);
  __syncthreads();

  int pix[4], a_coff[4];
  const bf16_t* wsrc[4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (wave * 2 + j) * 8 + lrow;
      const int swz = (pos ^ ((r >> 1) & 7)) * 8;
      pix[h * 2 + j] = (r >> 6) * 128 + h * 64 + (r & 63);
      a_coff[h * 2 + j] = swz;
      const int co = co0 + (r >> 5) * 64 + h * 32 + (r & 31);
      wsrc[h * 2 + j] = a.wt + (long)co * K + swz;
    }

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const bf16_t* rp[4];
  int rstep[4];
  unsigned noff[4];
  bool pend = false;
  int i_c0 = 0, i_tap = 0, i_k0 = 0, i_slot = 0;          // DMA cursor: (tile parity, tap, channel chunk) of the NEXT K-tile to issue
  auto read_tap = [&](int slot, int tap) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) noff[i] = tapt[slot * tab_words + tap * 256 + pix[i]];
  };
  auto use_tap = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = noff[i] != ~0u;
      rp[i] = ok ? a.in + noff[i] + a_coff[i] : a.zero;
      rstep[i] = ok ? 1 : 0;
    }
  };
  read_tap(0, 0);
  use_tap();
  auto issue = [&](int which, int buf) __attribute__((always_inline)) {
    const unsigned base = lds0 + buf * BUF;
    if (which == 0 || which == 3) {
      const int h = which == 0 ? 0 : 1;
      if (which == 0 && pend) { use_tap(); pend = false; }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        glds16_asm(rp[h * 2 + j] + i_c0 * rstep[h * 2 + j], base + XOFF + h * HALF + (wave * 2 + j) * 1024);
      if (which == 3) {
        // advance the cursor by one K-tile; past a tile's last K-tile it moves on to the next tile's table
        // (computed on copies and written back unconditionally: stores to i_tap / i_c0 in both arms of a branch are merged
        // by the optimiser into a store through a selected POINTER, which pins both cursors in scratch memory)
        const int cin = a.Cin, cmaj = a.cm;
        int nt = i_tap + (cmaj ? 1 : 0), nc = i_c0 + (cmaj ? 0 : 64);
        const bool tap_wrap = nt == ntaps, c_wrap = nc == cin;
        if (cmaj) { nc += tap_wrap ? 64 : 0; nt = tap_wrap ? 0 : nt; }
        else { nt += c_wrap ? 1 : 0; nc = c_wrap ? 0 : nc; }
        const bool tile_wrap = cmaj ? (nc == cin) : (nt == ntaps);
        const bool tapchg = cmaj || c_wrap || tile_wrap;
        nt = tile_wrap ? 0 : nt;
        nc = tile_wrap ? 0 : nc;
        i_tap = nt;
        i_c0 = nc;
        i_slot ^= tile_wrap ? 1 : 0;
        i_k0 = nt * cin + nc;
        if (tapchg) { read_tap(i_slot, nt); pend = true; }
      }
    } else {
      const int h = which - 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16_asm(wsrc[h * 2 + j] + i_k0, base + WOFF + h * HALF + (wave * 2 + j) * 1024);
    }
  };

  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, kc = lane >> 4;
  const int foff0 = frow * 128 + ((kc ^ ((frow >> 1) & 7)) * 16);
  const int foff1 = frow * 128 + (((4 + kc) ^ ((frow >> 1) & 7)) * 16);
  const int xrow0 = wm * 64 * 128;
  const int wrow0 = wn * 32 * 128;

  bf16x8_t xf[2][4], wfc[2][2][2];
  auto load_x = [&](const unsigned char* bufp, int h) __attribute__((always_inline)) {
    const unsigned char* p = bufp + XOFF + h * HALF + xrow0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        uint4 v = *(const uint4*)(p + f * 16 * 128 + (ks ? foff1 : foff0));
        if (RELU) { v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w); }
        xf[ks][f] = __builtin_bit_cast(bf16x8_t, v);
      }
  };
  auto load_w = [&](const unsigned char* bufp, int h) __attribute__((always_inline)) {
    const unsigned char* p = bufp + WOFF + h * HALF + wrow0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int g = 0; g < 2; ++g) wfc[h][ks][g] = *(const bf16x8_t*)(p + g * 16 * 128 + (ks ? foff1 : foff0));
  };
  auto mma = [&](int ph, int ch) __attribute__((always_inline)) {
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

  // prologue: the whole first K-tile of the first tile
#pragma unroll
  for (int w = 0; w < 4; ++w) issue(w, 0);
  wait_vm<0>();
  wg_barrier();

  int gk = 0;                                     // K-tiles multiplied so far (LDS buffer parity)
  for (int ti = 0; ti < my_tiles; ++ti) {
    const bool last_tile = ti + 1 == my_tiles;
    for (int t = 0; t < KT; ++t, ++gk) {
      const unsigned char* bufp = smem + (gk & 1) * BUF;
      const int nb = (gk + 1) & 1;
      const bool more = !(last_tile && t + 1 == KT);
      // phase 1: (P0, C0)
      load_x(bufp, 0);
      load_w(bufp, 0);
      if (more) issue(0, nb);
      __builtin_amdgcn_sched_barrier(0);
      mma(0, 0);
      if (more) wait_vm<4>(); else wait_vm<2>();
      wg_barrier();
      // phase 2: (P0, C1)
      load_w(bufp, 1);
      if (more) issue(1, nb);
      __builtin_amdgcn_sched_barrier(0);
      mma(0, 1);
      if (more) wait_vm<4>(); else wait_vm<0>();
      wg_barrier();
      // phase 3: (P1, C1)
      load_x(bufp, 1);
      if (more) issue(2, nb);
      __builtin_amdgcn_sched_barrier(0);
      mma(1, 1);
      wg_barrier();
      // phase 4: (P1, C0)
      if (more) issue(3, nb);
      __builtin_amdgcn_sched_barrier(0);
      mma(1, 0);
      if (more) wait_vm<4>();
      wg_barrier();
    }

    // ---- tile boundary: store this tile (the next tile's first K-tile is already landing), clear the accumulators,
    //      build the table of the tile after next in the slot this tile's table occupied (dead since its last K-tile
    //      was issued, one K-tile ago; first read a whole tile from now, with >= 4 barriers in between)
    const long m0 = (long)tile_of(ti) * 256;
    conv_epilogue(acc, a.bias, a.mask, a.resid, a.out, a.accumulate, a.M, a.Cout, m0 + wm * 128, co0 + wn * 64, lane, RowIdent(), a.resid_up ? a.lw : -1, a.lh);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (ti + 2 < my_tiles) build_table(ti & 1, (long)tile_of(ti + 2) * 256);
  }
}

// ---------------------------------------------------------------------------------------------------------
// 256 x 128-tile variant (Cout % 128 == 0: D.Block.1.Conv2 and its data gradient).  8 wavefronts = 4 (pixels, 64
// each) x 2 (channels, 64 each); two phases of 16 MFMAs per K-tile: phase A multiplies the first 32 pixels of every
// wavefront row (half-tile P0) with the whole filter tile W, phase B the second 32 (P1) with W kept in registers.
// Bursts: t.A issues P0 and W of tile t+1 (4 loads), t.B issues P1 (2 loads); waits vmcnt(4) / vmcnt(2).
// LDS: 2 x (256 + 128) rows x 128 B = 96 KiB.
// ---------------------------------------------------------------------------------------------------------
template <bool RELU>
__global__ __launch_bounds__(512) void conv_mfma_p8n_kernel(MfmaConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HALF = 128 * 128;
  constexpr int XOFF = 0, WOFF = 2 * HALF, BUF = 3 * HALF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;        // pixel quarter / channel half of this wavefront
  const long m0 = (long)blockIdx.x * 256;
  const int co0 = blockIdx.y * 128;
  const int Hs = a.up ? (a.H >> 1) : a.H, Ws = a.up ? (a.W >> 1) : a.W;
  // sub-pixel form (a.phase), as in the 256 x 256 kernel above
  // a.phase == 2: the data gradient of that form -- dx over the low-resolution grid gathers the 4 x 4 neighbourhood
  // (2p + u - 1, 2q + v - 1) of the full-resolution dy, each position with the transposed summed filter that reaches it
  // (16 taps, source stride 2; a.H, a.W = the dy grid, a.M = low-resolution pixels)
  const bool phm = a.phase == 1, dgm = a.phase == 2;
  const long Mph = a.M >> 2;
  const int tph = phm ? (int)(m0 / Mph) : 0, ph = tph >> 1, pw = tph & 1;
  const long mbase = phm ? (long)tph * Mph : 0;
  const int glw = (phm || dgm) ? a.lw - 1 : a.lw, glh = (phm || dgm) ? a.lh - 1 : a.lh;
  const int GH = phm ? Hs : (dgm ? (a.H >> 1) : a.H), GW = phm ? Ws : (dgm ? (a.W >> 1) : a.W);     // grid of the tile's pixel index
  const int BH = phm ? Hs : a.H, BW = phm ? Ws : a.W;                                              // grid of the source pixels
  const int sstr = dgm ? 2 : 1;
  const int kwn = phm ? 2 : (dgm ? 4 : a.KW), ntaps = phm ? 4 : (dgm ? 16 : a.KH * a.KW);
  const int pt = phm ? 1 - ph : (dgm ? 1 : a.PT), pl = phm ? 1 - pw : (dgm ? 1 : a.PL);
  const bool up = a.up && !phm && !dgm;
  const int K = ntaps * a.Cin;
  const int KT = K / 64;
  const bf16_t* const wbase = phm ? a.wph + (long)tph * a.Cout * K : (dgm ? a.wph : a.wt);
  const int lrow = lane >> 3, pos = lane & 7;

  // X half h, LDS row r <-> tile pixel (r>>5)*64 + h*32 + (r&31);  W row r <-> channel r
  int p_n[4], p_oh[4], p_ow[4], a_coff[4];       // index = h*2 + j
  const bf16_t* wsrc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 8 + lrow;
    const int swz = (pos ^ ((r >> 1) & 7)) * 8;
    wsrc[j] = wbase + (long)(co0 + r) * K + swz;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long m = m0 + (r >> 5) * 64 + h * 32 + (r & 31);
      a_coff[h * 2 + j] = swz;
      if (m < a.M) {
        const unsigned mm = (unsigned)(m - mbase);
        if (a.lw >= 0) {
          p_ow[h * 2 + j] = (int)(mm & (unsigned)(GW - 1));
          p_oh[h * 2 + j] = (int)((mm >> glw) & (unsigned)(GH - 1));
          p_n[h * 2 + j] = (int)(mm >> (glw + glh));
        } else {
          p_ow[h * 2 + j] = (int)(m % a.W);
          p_oh[h * 2 + j] = (int)((m / a.W) % a.H);
          p_n[h * 2 + j] = (int)(m / ((long)a.W * a.H));
        }
      } else {
        p_n[h * 2 + j] = 0; p_oh[h * 2 + j] = -100000; p_ow[h * 2 + j] = 0;
      }
    }
  }

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const bf16_t* rp[4];
  int rstep[4];
  int i_c0 = 0, i_kh = 0, i_kw = 0, i_k0 = 0;
  auto set_tap = [&](int kh, int kw) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ih = p_oh[i] * sstr + kh - pt, iw = p_ow[i] * sstr + kw - pl;
      const bool ok = ih >= 0 && ih < BH && iw >= 0 && iw < BW;
      if (up) { ih >>= 1; iw >>= 1; }
      rp[i] = ok ? a.in + (unsigned)((((unsigned)p_n[i] * Hs + ih) * Ws + iw) * a.Cin + a_coff[i]) : a.zero;
      rstep[i] = ok ? 1 : 0;
    }
  };
  set_tap(0, 0);
  auto issue_x = [&](int h, int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      glds16_asm(rp[h * 2 + j] + i_c0 * rstep[h * 2 + j], lds0 + buf * BUF + XOFF + h * HALF + (wave * 2 + j) * 1024);
  };
  auto issue_w = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16_asm(wsrc[j] + i_k0, lds0 + buf * BUF + WOFF + (wave * 2 + j) * 1024);
  };
  auto advance = [&]() {
    i_k0 += 64;
    i_c0 += 64;
    if (i_c0 == a.Cin) {
      i_c0 = 0;
      if (++i_kw == kwn) { i_kw = 0; ++i_kh; }
      set_tap(i_kh, i_kw);
    }
  };

  f32x4_t acc[4][4];       // [co fragment][px fragment = half*2 + f]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, kc = lane >> 4;
  const int foff0 = frow * 128 + ((kc ^ ((frow >> 1) & 7)) * 16);
  const int foff1 = frow * 128 + (((4 + kc) ^ ((frow >> 1) & 7)) * 16);
  const int xrow0 = wm * 32 * 128, wrow0 = wn * 64 * 128;

  bf16x8_t xf[2][2], wf[2][4];
  auto load_x = [&](const unsigned char* bufp, int h) {
    const unsigned char* p = bufp + XOFF + h * HALF + xrow0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        uint4 v = *(const uint4*)(p + f * 16 * 128 + (ks ? foff1 : foff0));
        if (RELU) { v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w); }
        xf[ks][f] = __builtin_bit_cast(bf16x8_t, v);
      }
  };
  auto load_w = [&](const unsigned char* bufp) {
    const unsigned char* p = bufp + WOFF + wrow0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int g = 0; g < 4; ++g) wf[ks][g] = *(const bf16x8_t*)(p + g * 16 * 128 + (ks ? foff1 : foff0));
  };
  auto mma = [&](int ph) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int f = 0; f < 2; ++f)
          acc[g][ph * 2 + f] = mfma16(wf[ks][g], xf[ks][f], acc[g][ph * 2 + f]);
    __builtin_amdgcn_s_setprio(0);
  };

  issue_x(0, 0); issue_w(0); issue_x(1, 0); advance();
  wait_vm<0>();
  wg_barrier();
  for (int t = 0; t < KT; ++t) {
    const unsigned char* bufp = smem + (t & 1) * BUF;
    const int nb = (t + 1) & 1;
    const bool more = t + 1 < KT;
    // phase A: P0 x W
    load_x(bufp, 0);
    load_w(bufp);
    if (more) { issue_x(0, nb); issue_w(nb); }
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    if (more) wait_vm<4>(); else wait_vm<0>();    // P1 of this tile has landed (newer: P0 and W of the next)
    wg_barrier();
    // phase B: P1 x W (registers)
    load_x(bufp, 1);
    if (more) { issue_x(1, nb); advance(); }
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    if (more) wait_vm<2>();                       // P0 and W of the next tile have landed (newer: its P1)
    wg_barrier();
  }

  conv_epilogue(acc, a.bias, a.mask, a.resid, a.out, a.accumulate, a.M, a.Cout, m0 + wm * 64, co0 + wn * 64, lane,
                RowPhase{phm ? 1 : 0, glw, glh, ph, pw, mbase}, a.resid_up ? a.lw : -1, a.lh);
}

template <bool RELU, bool XCDSWZ, bool STATS = false, bool PP = false>
static int launch8(rcgan_ctx* ctx, const MfmaConvArgs& a) {
  static bool attr_set = false;
  const size_t lds = (size_t)2 * 4 * 128 * 128 + 16 * 256 * sizeof(unsigned);     // tile buffers + tap-source table (<= 16 taps)
  if (!attr_set) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_p8_kernel<RELU, XCDSWZ, STATS, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, 256), a.Cout / 256);
  {
    ProfScope ps(ctx, RCGAN_PROF_CONV_P8, 2.0 * (double)a.M * (a.phase == 2 ? 36 : a.KH * a.KW) * a.Cin * a.Cout,
                 2.0 * (double)a.M * (a.phase == 2 ? 16 : a.phase == 1 ? 4 : a.KH * a.KW) * a.Cin * a.Cout);
    hipLaunchKernelGGL((conv_mfma_p8_kernel<RELU, XCDSWZ, STATS, PP>), grid, dim3(512), lds, ctx->stream, a);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

template <bool RELU>
static int launch8p(rcgan_ctx* ctx, const MfmaConvArgs& a, int swz) {
  static bool attr_set = false;
  const size_t lds = (size_t)2 * 4 * 128 * 128 + 2 * 9 * 256 * sizeof(unsigned);  // tile buffers + two tap-source tables
  if (!attr_set) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_p8p_kernel<RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int tiles = (int)cdiv(a.M, 256), ny = a.Cout / 256;
  // one workgroup per CU (LDS): P workgroups per channel column walk the pixel tiles; P a multiple of 8 keeps a
  // workgroup's tiles on its own XCD's run of the swizzled order
  int P = ctx->num_cus / ny;
  if (P > tiles) P = tiles;
  if (P >= 8) P &= ~7;
  const int xs = (swz && (tiles & 7) == 0 && (P & 7) == 0) ? 1 : 0;
  dim3 grid(P, ny);
  {
    ProfScope ps(ctx, RCGAN_PROF_CONV_P8, 2.0 * (double)a.M * a.KH * a.KW * a.Cin * a.Cout);
    hipLaunchKernelGGL(conv_mfma_p8p_kernel<RELU>, grid, dim3(512), lds, ctx->stream, a, tiles, xs);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

template <bool RELU>
static int launch8n(rcgan_ctx* ctx, const MfmaConvArgs& a) {
  static bool attr_set = false;
  const size_t lds = (size_t)2 * 3 * 128 * 128;
  if (!attr_set) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_p8n_kernel<RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, 256), a.Cout / 128);
  {
    ProfScope ps(ctx, RCGAN_PROF_CONV_P8N, 2.0 * (double)a.M * (a.phase == 2 ? 36 : a.KH * a.KW) * a.Cin * a.Cout,
                 2.0 * (double)a.M * (a.phase == 2 ? 16 : a.phase == 1 ? 4 : a.KH * a.KW) * a.Cin * a.Cout);
    hipLaunchKernelGGL(conv_mfma_p8n_kernel<RELU>, grid, dim3(512), lds, ctx->stream, a);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

static int up_phase_enabled() {
  static const int v = [] { const char* e = getenv("RCGAN_UP_PHASE"); return e ? atoi(e) : 1; }();
  return v;
}

// sub-pixel form of an upsample-3x3 convolution (MfmaConvArgs::wph): power-of-two images, whole tiles per phase
bool mfma_conv8_phase_form(const MfmaConvArgs& a) {
  return up_phase_enabled() && a.up && a.KH == 3 && a.KW == 3 && a.wph != nullptr && a.lw >= 1 && a.lh >= 1 && ((a.M >> 2) % 256) == 0;
}

// wide = 256 output channels per workgroup (Cout % 256 == 0), else 128; halo_patch = the halo-patch kernel of that width (conv_mfma8h.hip).
// Both are mfma_conv_route's decision (conv_mfma.hip) -- nothing here re-derives the routing.
int mfma_conv8_launch(rcgan_ctx* ctx, const MfmaConvArgs& a, bool wide, bool halo_patch) {
  static int swz = -1, cm = -1;
  if (swz < 0) { const char* e = getenv("RCGAN_P8_XCD"); swz = e ? atoi(e) : 1; }
  if (cm < 0) { const char* e = getenv("RCGAN_P8_CM"); cm = e ? atoi(e) : 1; }
  static int persist = -1;
  if (persist < 0) { const char* e = getenv("RCGAN_P8_PERSIST"); persist = e ? atoi(e) : 0; }
  const bool phase = mfma_conv8_phase_form(a);
  if (wide) {
    MfmaConvArgs b = a;
    b.phase = a.phase ? a.phase : (phase ? 1 : 0);
    b.stamps = (unsigned long long*)ctx->dbg_stamps;
    b.cm = (cm && a.KH * a.KW > 1) ? 1 : 0;
    // plain 3x3 layers on 16- / 32-wide images: the pixel operand as a zero-padded patch in LDS, fetched once per 64-channel chunk
    // instead of once per tap (conv_mfma8h.hip); RCGAN_P8_HALO=0 keeps the tile-per-tap kernel
    if (halo_patch) return mfma_conv8_halo_launch(ctx, b);
    // the persistent form needs >= 2 K-tiles per tile (table hand-over) and 3x3 / 1x1 filters (two 9-tap tables in LDS)
    if (persist && !b.phase && a.KH * a.KW * a.Cin >= 128 && a.KH * a.KW <= 9)
      return b.relu_in ? launch8p<true>(ctx, b, swz) : launch8p<false>(ctx, b, swz);
    if (b.stats) {        // tile statistics in the epilogue (the batch-normed layers read a batch-normed, ReLU-ed tensor: no input ReLU form)
      if (b.relu_in || b.Cout != 256 || b.M % 256 != 0 || b.accumulate || b.mask) RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "tile statistics: unsupported form");
      return swz ? launch8<false, true, true>(ctx, b) : launch8<false, false, true>(ctx, b);
    }
    static int pp = -1;
    if (pp < 0) { const char* e = getenv("RCGAN_P8_PP"); pp = e ? atoi(e) : 0; }
    if (swz && pp) return b.relu_in ? launch8<true, true, false, true>(ctx, b) : launch8<false, true, false, true>(ctx, b);
    if (swz) return b.relu_in ? launch8<true, true>(ctx, b) : launch8<false, true>(ctx, b);
    return b.relu_in ? launch8<true, false>(ctx, b) : launch8<false, false>(ctx, b);
  }
  if (a.stats) RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "tile statistics need the 256 x 256 kernel");
  MfmaConvArgs b = a;
  b.phase = a.phase ? a.phase : (phase ? 1 : 0);
  // plain 3x3 layers / their sub-pixel forms on 16- / 32-wide (low-resolution) images: the patch kernel's 256 x 128 sibling
  // (conv_mfma8h.hip); RCGAN_P8N_HALO=0 keeps the tile-per-tap kernel
  if (halo_patch) {
    b.stamps = (unsigned long long*)ctx->dbg_stamps;
    return mfma_conv8n_halo_launch(ctx, b);
  }
  return b.relu_in ? launch8n<true>(ctx, b) : launch8n<false>(ctx, b);
}
