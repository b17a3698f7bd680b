// bf16 MFMA implicit-GEMM convolution for gfx950 (CDNA4): the hot path of the CIFAR ResNet G/D.
//
// Forward / data-gradient kernel  (3x3 or 1x1, stride 1, SAME, NHWC, Cin % 64 == 0, Cout % 64 == 0):
//   Out[m][co] = sum_{tap,ci} In[pix(m,tap)][ci] * Wt[co][tap*Cin+ci]
//   * 256 threads = 4 wavefronts (2 pixel-halves x 2 cout-halves), block tile BM pixels x BN couts,
//     K-step 64 (one filter tap, 64 contiguous NHWC channels = 128 B per pixel row -> coalesced).
//   * register-staged double-buffered LDS pipeline: the global loads of K-tile t+1 are issued before
//     the MFMAs of tile t and written to the other LDS buffer after them (one barrier per tile).
//     Staging through registers lets the loader zero-fill the SAME-padding halo, read through a
//     nearest-2x upsample and apply the input ReLU for free.
//   * LDS rows are [row][64 + 8 pad] bf16 (144 B pitch): 16 consecutive rows land on 16 distinct
//     16-B bank slots, so the ds_read_b128 fragment reads of one 16-lane group do not collide.
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand and the pixels as B, so every lane
//     ends up with 4 consecutive output channels of one pixel: the epilogue (bias, residual
//     accumulate, ReLU-mask of the data gradient) writes 8-byte packed bf16 stores.
//   The data gradient is the same kernel run on dy with the 180-degree-rotated, in/out-swapped filter
//   prepared by conv_prepare_kernel.
//
// Filter-gradient kernel: dW[tap][ci][co] = sum_m X[pix(m,tap)][ci] * dY[m][co], a GEMM whose
//   reduction runs over pixels.  Both operands are channel-contiguous in HBM but the MFMA wants
//   them pixel(k)-contiguous per lane; tiles are staged as [pixel][channel] and the fragments are
//   read with ds_read_b64_tr_b16 (the gfx950 LDS transpose read).  Pixel ranges are split across
//   blocks into fp32 slabs that a second kernel reduces (deterministic, no atomics).
#include <vector>
#include "conv_mfma.h"
#include "small_gemm.h"
#include "step_inputs.h"
#include "mfma_util.h"
#include "conv_image.h"
#include "head_rider.h"

#define LDS_PITCH 72          // elements per LDS row in the fwd kernel (64 + 8 pad)
#define WG_PITCH 144          // elements per LDS row in the wgrad kernel (128 + 16 pad = 288 B)



// pixel index -> (n, oh, ow); shifts when H and W are powers of two (every layer of both nets), else divisions
__device__ __forceinline__ void decode_pix(long m, int H, int W, int lh, int lw, int& n, int& oh, int& ow) {
  if (lw >= 0 && lh >= 0) {
    const unsigned mm = (unsigned)m;
    ow = (int)(mm & (unsigned)(W - 1));
    oh = (int)((mm >> lw) & (unsigned)(H - 1));
    n = (int)(mm >> (lw + lh));
  } else {
    ow = (int)(m % W);
    long t = m / W;
    oh = (int)(t % H);
    n = (int)(t / H);
  }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_mfma_kernel(MfmaConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* As = (bf16_t*)smem;                                  // [2][BM][LDS_PITCH]
  bf16_t* Bs = As + 2 * BM * LDS_PITCH;                        // [2][BN][LDS_PITCH]
  constexpr int AR = BM / 32, BR = BN / 32;                    // 16-B chunks per thread per tile
  constexpr int TM = BM / 2, TN = BN / 2;                      // wave tile
  constexpr int NI = TN / 16, NJ = TM / 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const long m0 = (long)blockIdx.x * BM;
  const int co0 = blockIdx.y * BN;
  const int chunk = tid & 7, lrow = tid >> 3;
  const int K = a.KH * a.KW * a.Cin;
  const int KT = K / 64;
  const int Hs = a.up ? (a.H >> 1) : a.H, Ws = a.up ? (a.W >> 1) : a.W;

  // per-thread pixel decode for the A rows it stages
  int p_n[AR], p_oh[AR], p_ow[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    long m = m0 + lrow + 32 * i;
    if (m < a.M) {
      p_ow[i] = (int)(m % a.W);
      long t = m / a.W;
      p_oh[i] = (int)(t % a.H);
      p_n[i] = (int)(t / a.H);
    } else {
      p_n[i] = 0; p_oh[i] = -100000; p_ow[i] = 0;
    }
  }
  const bf16_t* wrow[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) wrow[i] = a.wt + (long)(co0 + lrow + 32 * i) * K + chunk * 8;

  uint4 ra[AR], rb[BR];
  auto load_tile = [&](int kt) {
    const int k0 = kt * 64;
    const int tap = k0 / a.Cin, c0 = k0 - tap * a.Cin;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int ih = p_oh[i] + kh - a.PT, iw = p_ow[i] + kw - a.PL;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) {
        if (a.up) { ih >>= 1; iw >>= 1; }
        const bf16_t* p = a.in + (((long)p_n[i] * Hs + ih) * Ws + iw) * a.Cin + c0 + chunk * 8;
        v = *(const uint4*)p;
        if (a.relu_in) { v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w); }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) rb[i] = *(const uint4*)(wrow[i] + k0);
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AR; ++i)
      *(uint4*)(As + ((long)buf * BM + lrow + 32 * i) * LDS_PITCH + chunk * 8) = ra[i];
#pragma unroll
    for (int i = 0; i < BR; ++i)
      *(uint4*)(Bs + ((long)buf * BN + lrow + 32 * i) * LDS_PITCH + chunk * 8) = rb[i];
  };

  f32x4_t acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  load_tile(0);
  store_tile(0);
  __syncthreads();
  const int frow = lane & 15, fk = (lane >> 4) * 8;
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) load_tile(kt + 1);
    const bf16_t* Ab = As + (long)buf * BM * LDS_PITCH + (wm * TM + frow) * LDS_PITCH + fk;
    const bf16_t* Bb = Bs + (long)buf * BN * LDS_PITCH + (wn * TN + frow) * LDS_PITCH + fk;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t wf[NI], xf[NJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[i] = *(const bf16x8_t*)(Bb + i * 16 * LDS_PITCH + ks * 32);
#pragma unroll
      for (int j = 0; j < NJ; ++j) xf[j] = *(const bf16x8_t*)(Ab + j * 16 * LDS_PITCH + ks * 32);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = mfma16(wf[i], xf[j], acc[i][j]);
    }
    if (kt + 1 < KT) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds out[m][co..co+3], m = pixel (lane&15), co = 4*(lane>>4)
  conv_epilogue(acc, a.bias, a.mask, a.resid, a.out, a.accumulate, a.M, a.Cout, m0 + wm * TM, co0 + wn * TN, lane, RowIdent(), a.resid_up ? a.lw : -1, a.lh);
}


// ---------------------------------------------------------------------------------------------
// Direct-to-LDS variant (global_load_lds_dwordx4): no staging registers, no ds_write pass.
//   * LDS rows are exactly 128 B (64 bf16); one wavefront instruction deposits 8 rows (lane l -> row l/8,
//     16-B slot l%8).  Bank conflicts are avoided by an XOR swizzle applied on the SOURCE side: slot p of
//     row r receives K-chunk p ^ ((r >> 1) & 7), and the fragment reads apply the same XOR.  (A ds_read_b128
//     serves 16 lanes = 16 consecutive rows per pass; two 128-B rows share one 256-B bank row, so the row
//     PAIR index must pick the slot: rows r and r+8 would collide with p ^ (r & 7).)
//   * the SAME-padding halo is read from a 16-byte zero page (each lane supplies its own global address).
//   * the input ReLU, when requested, is applied to the pixel fragments after the LDS read.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// KS > 1: intra-workgroup split of the reduction.  Layers whose grid is smaller than the chip (the 8x8 / 4x4 stages:
// <= 1 workgroup per CU) are bound by the serial chain of K-tiles of one workgroup, with three quarters of each
// SIMD's wave slots empty.  KS groups of 4 wavefronts then take K-tiles g, g+KS, g+2KS, ... of the SAME output tile,
// each group with its own LDS pipeline (the barriers stay workgroup-wide: all groups run the same iteration count),
// and the partial accumulators are summed through LDS in a fixed order before the epilogue.
// PHASE 1: the sub-pixel form of an upsample-3x3 convolution (MfmaConvArgs::wph; conv_mfma8.hip has the description) -- the tile's
// pixel index runs over (phase, n, i, j) of the low-resolution grid, four taps with the phase's summed filters.  PHASE 2: its data
// gradient (16 taps at source stride 2 over the full-resolution dy).  A template flag: the ordinary instantiations stay
// instruction-for-instruction what they were.
template <int BM, int BN, int NS, int KS, int PHASE = 0>
__global__ __launch_bounds__(256 * KS) void conv_mfma_glds_kernel(MfmaConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  constexpr int ABYTES = BM * 128, BBYTES = BN * 128, STAGE = ABYTES + BBYTES;
  constexpr int AI = BM / 32, BI = BN / 32;          // 1-KiB deposits per wave per tile
  constexpr int TM = BM / 2, TN = BN / 2;
  constexpr int NI = TN / 16, NJ = TM / 16;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = KS > 1 ? wave_all >> 2 : 0;       // K-split group of this wavefront
  const int wave = wave_all & 3;
  unsigned char* smem = smem_all + grp * (NS * STAGE);
  const int wm = wave & 1, wn = wave >> 1;
  const long m0 = (long)blockIdx.x * BM;
  const int co0 = blockIdx.y * BN;
  const int Hs = a.up ? (a.H >> 1) : a.H, Ws = a.up ? (a.W >> 1) : a.W;
  constexpr bool PHM = PHASE == 1, DGM = PHASE == 2;
  const long Mph = a.M >> 2;
  const int tph = PHM ? (int)(m0 / Mph) : 0, ph = tph >> 1, pw = tph & 1;      // a tile lies inside one phase (launcher)
  const long mbase = PHM ? (long)tph * Mph : 0;
  const int glw = PHASE ? a.lw - 1 : a.lw, glh = PHASE ? a.lh - 1 : a.lh;
  const int GH = PHM ? Hs : (DGM ? (a.H >> 1) : a.H), GW = PHM ? Ws : (DGM ? (a.W >> 1) : a.W);
  const int BH = PHM ? Hs : a.H, BW = PHM ? Ws : a.W;
  constexpr int sstr = DGM ? 2 : 1;
  const int kwn = PHM ? 2 : (DGM ? 4 : a.KW);
  const int pt = PHM ? 1 - ph : (DGM ? 1 : a.PT), pl = PHM ? 1 - pw : (DGM ? 1 : a.PL);
  const bool up = PHASE ? false : (a.up != 0);
  const int K = (PHM ? 4 : (DGM ? 16 : a.KH * a.KW)) * a.Cin;
  const int KT_all = K / 64;
  const int KT = KS > 1 ? (KT_all - grp + KS - 1) / KS : KT_all;      // K-tiles of this group
  const int KT_max = (KT_all + KS - 1) / KS;                          // iterations every group runs (barrier count)
  const int lrow = lane >> 3, pos = lane & 7;

  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (a.stamps && tid == 0) a.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  int p_n[AI], p_oh[AI], p_ow[AI], a_coff[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = (wave * AI + i) * 8 + lrow;
    const long m = m0 + row;
    a_coff[i] = (pos ^ ((row >> 1) & 7)) * 8;
    if (m < a.M) {
      decode_pix(m - mbase, GH, GW, glh, glw, p_n[i], p_oh[i], p_ow[i]);
    } else {
      p_n[i] = 0; p_oh[i] = -100000; p_ow[i] = 0;
    }
  }
  const bf16_t* const wbase = PHM ? a.wph + (long)tph * a.Cout * K : (DGM ? a.wph : a.wt);
  const bf16_t* wsrc[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = (wave * BI + i) * 8 + lrow;
    wsrc[i] = wbase + (long)(co0 + row) * K + (pos ^ ((row >> 1) & 7)) * 8;
  }

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // K-tiles are requested in order, so the (tap, channel) position of the next request is carried
  // incrementally: the per-row halo test and address are recomputed once per tap, not once per K-tile
  // (small grids run 1-2 wavefronts per SIMD and are bound by the instruction count of this loop).
  const bf16_t* rp[AI];
  int rstep[AI];                       // 1 if the row's source pixel exists for the current tap, else 0 (zero page)
  int i_c0 = 0, i_kh = 0, i_kw = 0, i_k0 = 0;
  auto set_tap = [&](int kh, int kw) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int ih = p_oh[i] * sstr + kh - pt, iw = p_ow[i] * sstr + kw - pl;
      const bool ok = ih >= 0 && ih < BH && iw >= 0 && iw < BW;
      if (up) { ih >>= 1; iw >>= 1; }
      rp[i] = ok ? a.in + (unsigned)((((unsigned)p_n[i] * Hs + ih) * Ws + iw) * a.Cin + a_coff[i]) : a.zero;
      rstep[i] = ok ? 1 : 0;
    }
  };
  auto advance = [&](int steps) {        // move the (tap, channel) cursor `steps` K-tiles on
    bool moved = false;
    i_k0 += 64 * steps;
    i_c0 += 64 * steps;
    while (i_c0 >= a.Cin) {
      i_c0 -= a.Cin;
      if (++i_kw == kwn) { i_kw = 0; ++i_kh; }
      moved = true;
    }
    if (moved) set_tap(i_kh, i_kw);
  };
  set_tap(0, 0);
  if (KS > 1 && grp > 0) advance(grp);
  auto issue = [&](int buf) {
    const unsigned stage = lds0 + buf * STAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i) glds16_asm(rp[i] + i_c0 * rstep[i], stage + (wave * AI + i) * 1024);
#pragma unroll
    for (int i = 0; i < BI; ++i) glds16_asm(wsrc[i] + i_k0, stage + ABYTES + (wave * BI + i) * 1024);
    advance(KS);
  };

  f32x4_t acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, kc = lane >> 4;
  // byte offset of this lane's fragment inside a 16-row slab, for the two 32-deep K steps of a tile
  const int foff0 = frow * 128 + ((kc ^ ((frow >> 1) & 7)) * 16);
  const int foff1 = frow * 128 + (((4 + kc) ^ ((frow >> 1) & 7)) * 16);

  const uint32_t relu_lb = a.relu_in ? 0u : 0x80008000u;      // branch-free optional input ReLU
  constexpr int PER_TILE = AI + BI;                 // glds instructions per wave per tile
  constexpr int INFLIGHT = (NS - 2) * PER_TILE;     // newest tiles allowed to stay in flight at the wait
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < KT) issue(t);
  int buf = 0;
  stamp(1);
  for (int kt = 0; kt < KT_max; ++kt) {
    // tile kt landed (this wave's part), then rendezvous: everyone's part landed and everyone left tile kt-1
    if (kt + NS - 2 < KT) wait_vmcnt<INFLIGHT>(); else wait_vmcnt<0>();
    // the barrier is issued from asm with a memory clobber: the s_barrier builtin is IntrNoMem, so the compiler
    // could otherwise hoist the LDS reads of this tile above it in iterations that issue no further DMA
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + NS - 1 < KT) {
      int nb = buf + NS - 1;
      if (nb >= NS) nb -= NS;
      issue(nb);
    }
    if (KS == 1 || kt < KT) {
      const unsigned char* Ab = smem + buf * STAGE + (wm * TM) * 128;
      const unsigned char* Bb = smem + buf * STAGE + ABYTES + (wn * TN) * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int fo = ks ? foff1 : foff0;
        bf16x8_t wf[NI], xf[NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i) wf[i] = *(const bf16x8_t*)(Bb + i * 16 * 128 + fo);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          uint4 v = *(const uint4*)(Ab + j * 16 * 128 + fo);
          v.x = pk_max_i16(v.x, relu_lb); v.y = pk_max_i16(v.y, relu_lb); v.z = pk_max_i16(v.z, relu_lb); v.w = pk_max_i16(v.w, relu_lb);
          xf[j] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = mfma16(wf[i], xf[j], acc[i][j]);
      }
    }
    if (++buf == NS) buf = 0;
    if (kt == 0) stamp(2);
  }
  stamp(3);

  if (KS > 1) {
    // sum the groups' partial tiles in group order: groups 1.. park their accumulators in LDS (the pipelines are idle)
    __syncthreads();
    float* part = (float*)smem_all;                 // [KS-1][256 threads][NI*NJ*4]
    constexpr int PER = NI * NJ * 4;
    if (grp > 0) {
      float* dst = part + ((grp - 1) * 256 + (tid & 255)) * PER;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) *(f32x4_t*)(dst + (i * NJ + j) * 4) = acc[i][j];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < KS; ++g) {
      const float* src = part + ((g - 1) * 256 + tid) * PER;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] += *(const f32x4_t*)(src + (i * NJ + j) * 4);
    }
  }

  stamp(4);
  if constexpr (PHASE == 1)
    conv_epilogue(acc, a.bias, a.mask, a.resid, a.out, a.accumulate, a.M, a.Cout, m0 + wm * TM, co0 + wn * TN, lane, RowPhase{1, glw, glh, ph, pw, mbase}, a.resid_up ? a.lw : -1, a.lh);
  else
    conv_epilogue(acc, a.bias, a.mask, a.resid, a.out, a.accumulate, a.M, a.Cout, m0 + wm * TM, co0 + wn * TN, lane, RowIdent(), a.resid_up ? a.lw : -1, a.lh);
  if (a.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(5);
    if (tid == 0) {
      a.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
      a.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Halo-staged variant of the small-tile kernel (3x3 filters, tiles = whole image rows).
//
// Measured on the kernel above (rcgan_debug_stamps, scripts/exp_p8_timeline.py 128 128 3 8 128): a workgroup's K loop
// moves 294 KB through the LDS-DMA path -- 147 KB of filters and 147 KB of pixels, the SAME 16 KB of pixels once per
// tap -- at the 60-75 GB/s a CU ingests by LDS-DMA however many tiles are in flight (2 x 4 stages: 58 GB/s, 4 x 2: 72)
// and however few workgroups run (32 or 256: the same 7.65 us per workgroup).  The 8x8 layers (one workgroup per CU) and
// the 16x16 layers (four per CU: 1.18 MB per CU = 16 us of their 19) are bound by that rate, not by latency or L2.
// Here the tile's pixels are fetched ONCE, as a zero-padded patch of (rows + 2) x (W + 2) pixels per 64-channel chunk
// (all chunks resident: <= 70 KB at 256 channels), and the nine taps are row offsets into it, so the K loop streams only
// the filter tiles (8 KB per K-tile): 163 KB instead of 294 KB per 64 x 64 tile.
//   patch pixel pp = pr * (W + 2) + pc  <->  input (oh0 + pr - 1, pc - 1); 128-byte rows, 16-byte slots XOR-swizzled with
//   (pp >> 1) & 7 on the DMA's source side and in the fragment reads, exactly like the tile rows of the kernel above.
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int NS, int KS>
__global__ __launch_bounds__(256 * KS) void conv_mfma_halo_kernel(MfmaConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  constexpr int BBYTES = BN * 128;
  constexpr int BI = BN / 32;
  constexpr int TM = BM / 2, TN = BN / 2;
  constexpr int NI = TN / 16, NJ = TM / 16;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = KS > 1 ? wave_all >> 2 : 0;
  const int wave = wave_all & 3;
  const int wm = wave & 1, wn = wave >> 1;
  const long m0 = (long)blockIdx.x * BM;
  const int co0 = blockIdx.y * BN;
  const int K = 9 * a.Cin;
  const int KT_all = K / 64;
  const int KT = KS > 1 ? (KT_all - grp + KS - 1) / KS : KT_all;
  const int KT_max = (KT_all + KS - 1) / KS;
  const int nchunk = a.Cin >> 6;
  const int W = a.W, PW = W + 2;
  const int TR = BM >> a.lw;                       // image rows of the tile
  const int PP = (TR + 2) * PW;                    // patch pixels
  const int PD = (PP + 7) >> 3;                    // 1-KiB deposits (8 patch pixels x 128 B) per chunk
  const int a_chunk = PD * 1024;
  const int lrow = lane >> 3, pos = lane & 7;
  unsigned char* const Abase = smem_all;
  unsigned char* const Bbase = smem_all + nchunk * a_chunk + grp * (NS * BBYTES);
  const unsigned lds_all = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_all;
  const unsigned ldsB = lds_all + nchunk * a_chunk + grp * (NS * BBYTES);

  // ---- the patch, one 64-channel chunk at a time: every wavefront of the workgroup deposits its share.  The K loop runs
  //      chunk-major (the nine taps of chunk 0, then chunk 1, ...), so only chunk 0 has to land before the first MFMA; chunk
  //      c + 1 is requested when chunk c's first K-tile starts and is older than every filter tile waited for after that
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (a.stamps && tid == 0) a.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  const unsigned mm0 = (unsigned)m0;
  const int n_img = (int)(mm0 >> (a.lw + a.lh));
  const int oh0 = (int)((mm0 >> a.lw) & (unsigned)(a.H - 1));
  const int Hs = a.up ? (a.H >> 1) : a.H, Ws = a.up ? (a.W >> 1) : a.W;
  auto load_patch = [&](int chunk) __attribute__((always_inline)) {
    for (int d = wave_all; d < PD; d += 4 * KS) {
      const int pp = d * 8 + lrow;
      const int pr = pp / PW, pc = pp - pr * PW;
      int ih = oh0 + pr - 1, iw = pc - 1;
      const bool ok = pp < PP && ih >= 0 && ih < a.H && iw >= 0 && iw < W;
      if (a.up) { ih >>= 1; iw >>= 1; }
      const bf16_t* src = ok ? a.in + (unsigned)((((unsigned)n_img * Hs + ih) * Ws + iw) * a.Cin + chunk * 64 + ((pos ^ ((pp >> 1) & 7)) * 8))
                             : a.zero;
      glds16_asm(src, lds_all + chunk * a_chunk + d * 1024);
    }
  };
  load_patch(0);

  // ---- filter tiles: as in the kernel above ------------------------------------------------------------------------------
  const bf16_t* wsrc[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = (wave * BI + i) * 8 + lrow;
    wsrc[i] = a.wt + (long)(co0 + row) * K + (pos ^ ((row >> 1) & 7)) * 8;
  }
  // K-tile kg of the chunk-major order = (chunk kg / 9, tap kg % 9); its filter columns start at tap * Cin + chunk * 64
  int i_kg = grp;
  auto issue = [&](int buf) __attribute__((always_inline)) {
    const int chunk = i_kg / 9, tap = i_kg - chunk * 9;
    const int k0 = tap * a.Cin + chunk * 64;
#pragma unroll
    for (int i = 0; i < BI; ++i) glds16_asm(wsrc[i] + k0, ldsB + buf * BBYTES + (wave * BI + i) * 1024);
    i_kg += KS;
  };

  f32x4_t acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, kc = lane >> 4;
  const int foff0 = frow * 128 + ((kc ^ ((frow >> 1) & 7)) * 16);
  const int foff1 = frow * 128 + (((4 + kc) ^ ((frow >> 1) & 7)) * 16);
  // patch index of this lane's pixel of fragment j for tap (0, 0)
  int pp00[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int p = wm * TM + j * 16 + frow;
    pp00[j] = (p >> a.lw) * PW + (p & (W - 1));
  }

  const uint32_t relu_lb = a.relu_in ? 0u : 0x80008000u;
  constexpr int INFLIGHT = (NS - 2) * BI;
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < KT) issue(t);
  int buf = 0;
  bool drain = false;                    // the previous iteration requested a patch chunk: wait for everything once
  stamp(1);
  for (int kt = 0; kt < KT_max; ++kt) {
    // (chunk 0 of the patch is older than every filter tile; later chunks are covered by the full wait that follows them)
    if (!drain && kt + NS - 2 < KT) wait_vmcnt<INFLIGHT>(); else wait_vmcnt<0>();
    drain = false;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + NS - 1 < KT) {
      int nb = buf + NS - 1;
      if (nb >= NS) nb -= NS;
      issue(nb);
    }
    {
      // the next chunk's patch: requested by every wavefront once, when the workgroup's first K-group enters the chunk before
      const int kg0 = kt * KS;                       // K-tile of group 0 in this iteration
      const int c0 = kg0 / 9;
      if (kg0 - c0 * 9 < KS && c0 + 1 < nchunk) { load_patch(c0 + 1); drain = true; }
    }
    if (KS == 1 || kt < KT) {
      const int kg = grp + kt * KS;                  // global K-tile, chunk-major
      const int chunk = kg / 9, tap = kg - chunk * 9;
      const int kh = tap / 3, kw = tap - kh * 3;
      const unsigned char* Ac = Abase + chunk * a_chunk;
      const unsigned char* Bb = Bbase + buf * BBYTES + (wn * TN) * 128;
      int prow[NJ], psw[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) { const int pp = pp00[j] + kh * PW + kw; prow[j] = pp * 128; psw[j] = (pp >> 1) & 7; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int fo = ks ? foff1 : foff0;
        bf16x8_t wf[NI], xf[NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i) wf[i] = *(const bf16x8_t*)(Bb + i * 16 * 128 + fo);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          uint4 v = *(const uint4*)(Ac + prow[j] + (((ks * 4 + kc) ^ psw[j]) << 4));
          v.x = pk_max_i16(v.x, relu_lb); v.y = pk_max_i16(v.y, relu_lb); v.z = pk_max_i16(v.z, relu_lb); v.w = pk_max_i16(v.w, relu_lb);
          xf[j] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = mfma16(wf[i], xf[j], acc[i][j]);
      }
    }
    if (++buf == NS) buf = 0;
    if (kt == 0) stamp(2);
  }
  stamp(3);

  if (KS > 1) {
    __syncthreads();
    float* part = (float*)smem_all;
    constexpr int PER = NI * NJ * 4;
    if (grp > 0) {
      float* dst = part + ((grp - 1) * 256 + (tid & 255)) * PER;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) *(f32x4_t*)(dst + (i * NJ + j) * 4) = acc[i][j];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < KS; ++g) {
      const float* src = part + ((g - 1) * 256 + tid) * PER;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] += *(const f32x4_t*)(src + (i * NJ + j) * 4);
    }
  }
  stamp(4);
  conv_epilogue(acc, a.bias, a.mask, a.resid, a.out, a.accumulate, a.M, a.Cout, m0 + wm * TM, co0 + wn * TN, lane, RowIdent(), a.resid_up ? a.lw : -1, a.lh);
  if (a.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(5);
    if (tid == 0) {
      a.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
      a.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// filter gradient
// ---------------------------------------------------------------------------------------------


__device__ __forceinline__ bf16x8_t frag_tr(const bf16_t* base /* &tile[row0][ch0] */, int lane, int use_tr) {
  // returns, for lane (g = lane>>4, i = lane&15), channel ch0+i at k-slots {g*4+e} U {16+g*4+e}, e=0..3
  const int g = lane >> 4, i = lane & 15;
  s16x8_t r;
  if (use_tr) {
    const bf16_t* p = base + (g * 4 + (i >> 2)) * WG_PITCH + (i & 3) * 4;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 16 * WG_PITCH));
    r = (s16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = (e < 4) ? (g * 4 + e) : (16 + g * 4 + (e - 4));
      r[e] = (short)base[row * WG_PITCH + i];
    }
  }
  return __builtin_bit_cast(bf16x8_t, r);
}

// block tile: 128 ci x 128 co for one tap; 4 waves as 2 (ci) x 2 (co), wave tile 64 x 64
__global__ __launch_bounds__(256) void conv_mfma_wgrad_kernel(MfmaWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Xs = (bf16_t*)smem;                    // [2][64][WG_PITCH]
  bf16_t* Ys = Xs + 2 * 64 * WG_PITCH;           // [2][64][WG_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave & 1, wo = wave >> 1;
  const int nci = a.Cin / 128, nco = a.Cout / 128;
  int b = blockIdx.x;
  const int cot = b % nco; b /= nco;
  const int cit = b % nci; b /= nci;
  const int tap = b;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int ci0 = cit * 128, co0 = cot * 128;
  const long mb = (long)blockIdx.y * a.m_chunk;
  long me = mb + a.m_chunk;
  if (me > a.M) me = a.M;
  const int Hs = a.up ? (a.H >> 1) : a.H, Ws = a.up ? (a.W >> 1) : a.W;
  const int chunk = tid & 15, lrow = tid >> 4;   // 16 chunks of 8 channels, 16 rows per pass, 4 passes

  uint4 rx[4], ry[4];
  auto load_tile = [&](long p0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = p0 + lrow + 16 * i;
      uint4 vx = make_uint4(0, 0, 0, 0), vy = make_uint4(0, 0, 0, 0);
      if (m < me) {
        int ow, oh, n;
        decode_pix(m, a.H, a.W, a.lh, a.lw, n, oh, ow);
        int ih = oh + kh - a.PT, iw = ow + kw - a.PL;
        if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) {
          if (a.up) { ih >>= 1; iw >>= 1; }
          vx = *(const uint4*)(a.x + (unsigned)((((unsigned)n * Hs + ih) * Ws + iw) * a.Cin + ci0 + chunk * 8));
          if (a.relu_in) { vx.x = relu_bf16x2(vx.x); vx.y = relu_bf16x2(vx.y); vx.z = relu_bf16x2(vx.z); vx.w = relu_bf16x2(vx.w); }
        }
        vy = *(const uint4*)(a.dy + (unsigned)((unsigned)m * a.Cout + co0 + chunk * 8));
      }
      rx[i] = vx; ry[i] = vy;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(uint4*)(Xs + ((long)buf * 64 + lrow + 16 * i) * WG_PITCH + chunk * 8) = rx[i];
      *(uint4*)(Ys + ((long)buf * 64 + lrow + 16 * i) * WG_PITCH + chunk * 8) = ry[i];
    }
  };

  f32x4_t acc[4][4];   // [co tile][ci tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const long ntile = (me - mb + 63) / 64;
  if (ntile > 0) {
    load_tile(mb);
    store_tile(0);
  }
  __syncthreads();
  for (long t = 0; t < ntile; ++t) {
    const int buf = (int)(t & 1);
    if (t + 1 < ntile) load_tile(mb + (t + 1) * 64);
    const bf16_t* Xb = Xs + (long)buf * 64 * WG_PITCH + wi * 64;
    const bf16_t* Yb = Ys + (long)buf * 64 * WG_PITCH + wo * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t yf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) yf[i] = frag_tr(Yb + ks * 32 * WG_PITCH + i * 16, lane, a.use_tr);
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = frag_tr(Xb + ks * 32 * WG_PITCH + j * 16, lane, a.use_tr);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma16(yf[i], xf[j], acc[i][j]);
    }
    if (t + 1 < ntile) store_tile(buf ^ 1);
    __syncthreads();
  }
  // D[row = co (4*(lane>>4)+r)][col = ci (lane&15)]
  float* slab = a.slab + (long)blockIdx.y * a.slab_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + wo * 64 + i * 16 + (lane >> 4) * 4;
      const int ci = ci0 + wi * 64 + j * 16 + (lane & 15);
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      *(float4*)(slab + ((long)tap * a.Cin + ci) * a.Cout + co) = v;
    }
}


// ---------------------------------------------------------------------------------------------
// filter gradient, direct-to-LDS variant.  Tiles are [64 pixels][128 channels] = 256-B rows, deposited
// 4 rows per wavefront instruction.  16-B slot s of row r is stored at slot s ^ ((r & 7) << 1): the eight
// rows a ds_read_b64_tr_b16 half-wave touches then fall into eight distinct 32-B bank segments.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8_t frag_tr_swz(const unsigned char* tile, int row0, int slot0, int lane, int use_tr, int relu) {
  // lane (g = lane>>4, i = lane&15): channel = 8*slot0 + i (slot0 even), k-slots {g*4+e} U {16+g*4+e}
  const int g = lane >> 4, i = lane & 15;
  s16x8_t r;
  if (use_tr) {
    const int row = row0 + g * 4 + (i >> 2);
    const int sw = (row & 7) << 1;
    const unsigned char* p = tile + row * 256 + (((slot0 + ((i & 3) >> 1)) ^ sw) << 4) + (i & 1) * 8;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 16 * 256));
    r = (s16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = row0 + ((e < 4) ? (g * 4 + e) : (16 + g * 4 + (e - 4)));
      const int slot = (slot0 + (i >> 3)) ^ ((row & 7) << 1);
      r[e] = *(const short*)(tile + row * 256 + (slot << 4) + (i & 7) * 2);
    }
  }
  if (relu) {
    uint4 v = __builtin_bit_cast(uint4, r);
    v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w);
    r = __builtin_bit_cast(s16x8_t, v);
  }
  return __builtin_bit_cast(bf16x8_t, r);
}

template <int NS>
__device__ __forceinline__ void wgrad_glds_body(const MfmaWgradArgs& a, const unsigned bx, const unsigned by) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TILE = 32 * 256;                 // bytes per operand per stage: 32 pixels x 128 channels
  constexpr int STAGE = 2 * TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave & 1, wo = wave >> 1;
  const int nci = a.Cin / 128, nco = a.Cout / 128;
  int b = (int)bx;
  const int cot = b % nco; b /= nco;
  const int cit = b % nci; b /= nci;
  const int tap = b;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int ci0 = cit * 128, co0 = cot * 128;
  const long mb = (long)by * a.m_chunk;
  long me = mb + a.m_chunk;
  if (me > a.M) me = a.M;
  const int Hs = a.up ? (a.H >> 1) : a.H, Ws = a.up ? (a.W >> 1) : a.W;
  const int lrow = lane >> 4, pos = lane & 15;
  const int dh = kh - a.PT, dw = kw - a.PL;

  // this lane deposits 16-B slot `pos` of rows r_i = (wave*2+i)*4 + lrow, i = 0,1, of both operand tiles
  int r_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 4 + lrow;
    r_off[i] = (pos ^ ((row & 7) << 1)) * 8;     // source-side swizzle (channel offset)
  }
  // (the geometry pinned in scalar registers: see wgrad3_body)
  const bf16_t* xbase = a.x + ci0;
  const bf16_t* ybase = a.dy + co0;
  const bf16_t* azero = a.zero;
  int aH = a.H, aW = a.W, aCin = a.Cin, aCout = a.Cout, alw = a.lw, alh = a.lh, aup = a.up;
  asm volatile("" : "+s"(xbase), "+s"(ybase), "+s"(azero), "+s"(aH), "+s"(aW), "+s"(aCin), "+s"(aCout), "+s"(alw), "+s"(alh), "+s"(aup));
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  long i_p0 = mb;
  auto issue = [&](int buf) {
    const unsigned stage = lds0 + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long m = i_p0 + (wave * 2 + i) * 4 + lrow;
      const bf16_t* px = azero;
      const bf16_t* py = azero;
      if (m < me) {
        int ow, oh, n;
        decode_pix(m, aH, aW, alh, alw, n, oh, ow);
        int ih = oh + dh, iw = ow + dw;
        if (ih >= 0 && ih < aH && iw >= 0 && iw < aW) {
          if (aup) { ih >>= 1; iw >>= 1; }
          px = xbase + (unsigned)((((unsigned)n * Hs + ih) * Ws + iw) * aCin + r_off[i]);
        }
        py = ybase + (unsigned)((unsigned)m * aCout + r_off[i]);
      }
      glds16_asm(px, stage + (wave * 2 + i) * 1024);
      glds16_asm(py, stage + TILE + (wave * 2 + i) * 1024);
    }
    i_p0 += 32;
  };

  f32x4_t acc[4][4];   // [co tile][ci tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const bool do_bias = a.want_bias && tap == 0 && cit == 0 && wi == 0;     // see conv_mfma_wgrad3_kernel
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(H16_ONE_X2, H16_ONE_X2, H16_ONE_X2, H16_ONE_X2));
  f32x4_t accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int KT = (int)((me - mb + 31) / 32);
  constexpr int INFLIGHT = (NS - 2) * 4;
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < KT) issue(t);
  int buf = 0;
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + NS - 2 < KT) wait_vmcnt<INFLIGHT>(); else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + NS - 1 < KT) {
      int nb = buf + NS - 1;
      if (nb >= NS) nb -= NS;
      issue(nb);
    }
    const unsigned char* Xb = smem + buf * STAGE;
    const unsigned char* Yb = Xb + TILE;
    bf16x8_t yf[4], xf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) yf[i] = frag_tr_swz(Yb, 0, wo * 8 + i * 2, lane, a.use_tr, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) xf[j] = frag_tr_swz(Xb, 0, wi * 8 + j * 2, lane, a.use_tr, a.relu_in);
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) accb[i] = mfma16(yf[i], ones, accb[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = mfma16(yf[i], xf[j], acc[i][j]);
    if (++buf == NS) buf = 0;
  }
  float* slab = a.slab + (long)by * a.slab_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + wo * 64 + i * 16 + (lane >> 4) * 4;
      const int ci = ci0 + wi * 64 + j * 16 + (lane & 15);
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      *(float4*)(slab + ((long)tap * a.Cin + ci) * a.Cout + co) = v;
    }
  if (do_bias && (lane & 15) == 0) {
    float* bs = slab + (long)a.cells * a.Cin * a.Cout;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *(float4*)(bs + co0 + wo * 64 + i * 16 + (lane >> 4) * 4) = make_float4(accb[i][0], accb[i][1], accb[i][2], accb[i][3]);
  }
}

template <int NS>
__global__ __launch_bounds__(256) void conv_mfma_wgrad_glds_kernel(MfmaWgradArgs a) {
  wgrad_glds_body<NS>(a, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------
// filter gradient, three horizontal taps per workgroup (3x3 filters, W in {4,8,16,32}).
// The per-tap kernels above re-read every input and gradient pixel once per tap and channel tile
// (64 FLOP per byte fetched from L2) and sit on the L2->LDS bandwidth.  Here one workgroup owns
// (kh, 64 input channels, 128 output channels) and accumulates kw = 0,1,2 from the SAME staged rows:
// the tap shift is a +-1 row offset of the transposing LDS read, the left/right SAME-padding columns
// are removed by a loop-invariant AND mask on the input fragments (a stage is 32 consecutive pixels
// and W divides 32, so the column of fragment slot k is k mod W in every stage).  128 FLOP per byte.
//   stage: X rows = pixels p0-4 .. p0+35 (40 x 128 B, 8 rows per deposit), Y rows = p0 .. p0+31 (32 x 256 B)
//   waves 2 (ci) x 2 (co): wave tile 32 ci x 64 co x 3 taps = 24 MFMA accumulators (96 registers)
// ---------------------------------------------------------------------------------------------
// bias gradient = column sums of dy.  Extra workgroups (blockIdx.x >= number of filter tiles) stream ONLY the dy
// rows of their pixel chunk and multiply each transposed fragment with an all-ones operand on the matrix core.
// Kept as a separate code path so that its accumulators share registers with the filter path (occupancy).
template <int NS>
__device__ __forceinline__ void wgrad_bias_block(const MfmaWgradArgs& a, unsigned char* smem, int cot, long mb, long me, float* slab) {
  constexpr int YT = 32 * 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int co0 = cot * 128;
  const int y_row0 = (wave * 2) * 4 + (lane >> 4), y_row1 = y_row0 + 4;
  const int y_c0 = co0 + ((lane & 15) ^ ((y_row0 & 7) << 1)) * 8;
  const int y_c1 = co0 + ((lane & 15) ^ ((y_row1 & 7) << 1)) * 8;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  long i_p0 = mb;
  const bf16_t* ady = a.dy; const bf16_t* azero = a.zero;
  int aCout = a.Cout;
  asm volatile("" : "+s"(ady), "+s"(azero), "+s"(aCout));        // (pinned in scalar registers: see wgrad3_body)
  auto issue = [&](int buf) {
    const unsigned stage = lds0 + buf * YT;
    const long m0 = i_p0 + y_row0, m1 = i_p0 + y_row1;
    const bf16_t* py0 = m0 < me ? ady + (unsigned)((unsigned)m0 * aCout + y_c0) : azero;
    const bf16_t* py1 = m1 < me ? ady + (unsigned)((unsigned)m1 * aCout + y_c1) : azero;
    glds16_asm(py0, stage + (wave * 2) * 1024);
    glds16_asm(py1, stage + (wave * 2 + 1) * 1024);
    i_p0 += 32;
  };
  int offy[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = g * 4 + (li >> 2);
    const int slot0 = wave * 4 + i * 2;
    offy[i] = row * 256 + (((slot0 + ((li & 3) >> 1)) ^ ((row & 7) << 1)) << 4) + (li & 1) * 8;
  }
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(H16_ONE_X2, H16_ONE_X2, H16_ONE_X2, H16_ONE_X2));
  f32x4_t accb[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
  const int KT = (int)((me - mb + 31) / 32);
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < KT) issue(t);
  for (int kt0 = 0; kt0 < KT; kt0 += NS) {
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      const int kt = kt0 + sidx;
      if (kt < KT) {
        if (kt + NS - 2 < KT) wait_vmcnt_any<(NS - 2) * 2>(); else wait_vmcnt_any<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + NS - 1 < KT) issue((sidx + NS - 1) % NS);
        const unsigned char* sb = smem + sidx * YT;
#pragma unroll
        for (int i = 0; i < 2; ++i)
          accb[i] = mfma16(tr_pair(sb + offy[i], 16 * 256), ones, accb[i]);
      }
    }
  }
  if ((lane & 15) == 0) {
    float* bs = slab + (long)a.cells * a.Cin * a.Cout;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *(float4*)(bs + co0 + wave * 32 + i * 16 + (lane >> 4) * 4) = make_float4(accb[i][0], accb[i][1], accb[i][2], accb[i][3]);
  }
}

// timing-only ablations of the three-tap body (scripts/build_p8_ablate.sh w<k>; results are wrong by construction): 1 = no LDS-DMA after the
// prologue, 2 = no MFMAs, 4 = fragment reads of the first stage only, 8 = no ReLU / edge masks on the pixel fragments
#ifndef WG3_ABLATE
#define WG3_ABLATE 0
#endif

// SUB: the sub-pixel form (MfmaWgradArgs::sub != 0, checked by the caller): two column taps per workgroup (dw = tb, tb + 1 with
// tb = -1 or 0 by the column parity) instead of three -- 16 accumulators, the tap rows and the edge mask chosen once per workgroup
template <int NS, bool RELU, bool SUB>
__device__ __forceinline__ void wgrad3_body(const MfmaWgradArgs& a, const unsigned bx, const unsigned by) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int XT = 40 * 128, YT = 32 * 256, STAGE = XT + YT;
  constexpr int NT = SUB ? 2 : 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave & 1, wo = wave >> 1;
  const int nci = a.Cin / 64, nco = a.Cout / 128;
  const long mb = (long)by * a.m_chunk;
  long me = mb + a.m_chunk;
  if (me > a.M) me = a.M;
  const int sub = SUB ? a.sub : 0;
  const int trows = SUB ? (sub == 3 ? 1 : 8) : a.KH;        // tile rows: filter rows kh, or (pa, pb, s) of the sub-pixel form
  if ((int)bx >= trows * nci * nco) {      // bias-gradient workgroup (only launched when want_bias)
    // (sub 1: dy lives on the full-resolution grid, whose pixels [4 mb, 4 me) are as good a quarter-share as any)
    wgrad_bias_block<NS>(a, smem, (int)bx - trows * nci * nco, sub == 1 ? 4 * mb : mb, sub == 1 ? 4 * me : me,
                         a.slab + (long)by * a.slab_stride);
    return;
  }
  int b = (int)bx;
  const int cot = b % nco; b /= nco;
  const int cit = b % nci; b /= nci;
  const int kh = b;
  const int ci0 = cit * 64, co0 = cot * 128;
  // The problem's geometry, PINNED in scalar registers (round 5).  `a` lives in the kernel-argument segment (a grouped launch indexes it with
  // the problem number) and the compiler re-read its fields there on every use: 17 s_load + 9 s_waitcnt lgkmcnt(0) in the address
  // arithmetic of EVERY K-step, each a scalar-cache round trip in front of the LDS-DMA issue.  An empty asm makes the copies opaque
  // (scripts/bench_wgrad_group.py: the critic's 32x32 layer alone 78.7 -> 70.9 us, the critic step's set 166 -> 155 us).
  const bf16_t* ax = a.x; const bf16_t* ady = a.dy; const bf16_t* azero = a.zero;
  int aH = a.H, aW = a.W, aCin = a.Cin, aCout = a.Cout, alw = a.lw, alh = a.lh;
  long aM = a.M;
  asm volatile("" : "+s"(ax), "+s"(ady), "+s"(azero), "+s"(aH), "+s"(aW), "+s"(aCin), "+s"(aCout), "+s"(alw), "+s"(alh), "+s"(aM));
  const int pa = kh >> 2, pb = (kh >> 1) & 1, srow = kh & 1;                 // sub-pixel form only
  const int dh = !SUB ? kh - a.PT : (sub == 3 ? 0 : (sub == 1 ? srow - 1 + pa : srow - pa));
  // first column tap: dw = tap - 1 for the three taps, or dw in {-1, 0} / {0, +1} by the column parity (1x1: dw = 0, second tap idle)
  const int tb = !SUB ? -1 : (sub == 3 ? 0 : ((sub == 1 ? pb == 0 : pb == 1) ? -1 : 0));
  const int ntap = (SUB && sub == 3) ? 1 : NT;
  const bool relu_on = RELU && a.relu_in;      // (a grouped launch is compiled for its 3x3 layers' flavour; a riding 1x1 may differ)
  const int g = lane >> 4, li = lane & 15;

  // ---- DMA roles -------------------------------------------------------------------------------
  // Y: deposits q = 2*wave + {0,1}; lane -> row q*4 + lane/16, 16-B slot lane%16
  const int y_row0 = (wave * 2) * 4 + (lane >> 4), y_row1 = y_row0 + 4;
  const int y_c0 = co0 + ((lane & 15) ^ ((y_row0 & 7) << 1)) * 8;
  const int y_c1 = co0 + ((lane & 15) ^ ((y_row1 & 7) << 1)) * 8;
  // X: deposit q = wave (and q = 4 from wave 0); lane -> row q*8 + lane/8, 16-B slot lane%8
  const int x_rowa = wave * 8 + (lane >> 3), x_rowb = 32 + (lane >> 3);
  const int x_ca = ci0 + ((lane & 7) ^ (((x_rowa >> 1) & 3) << 1)) * 8;        // (rows 32 + r share row r's swizzle key)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // (pixel indices are 32-bit in the loop -- the element counts are bounded by the host -- : half the vector ALU work of 64-bit compares and adds)
  const int M32 = (int)aM;
  int i_p0 = (int)mb;
  // Linear addressing (round 5; the nearest-upsampled input, whose source row is floor(row / 2), goes to the per-tap kernel or, by default,
  // to the sub-pixel form sub = 1).  A K-step advances the reduction index by 32 pixels = whole grid
  // rows (W divides 32), so every lane's source address advances by the SAME amount per step, also in the strided sub-pixel forms (4 W elements
  // per grid row of the full-resolution tensor): dy is fetched through a scalar base that steps + a constant per-lane offset -- no vector ALU
  // work at all (chunks are whole K-steps: mfma_wgrad3_takes) --, x through a per-lane offset that steps, with the in-image test (halo rows
  // read the zero page) as two compares and a select instead of the pixel's decomposition, three multiplies and two divergent branches.
  auto y_lin = [&](int row) -> unsigned {        // element offset of pixel (step base + row), step base = a multiple of 32
    if (sub != 1) return (unsigned)(row * aCout);
    return (unsigned)((4 * aW * (row >> alw) + 2 * (row & (aW - 1)) + pa * 2 * aW + pb) * aCout);
  };
  auto x_lin = [&](int m, int coff) -> unsigned {      // element offset of the pixel tap row dh of reduction pixel m reads (m < 0: the same line, continued)
    if (sub == 2) return (unsigned)((4 * aW * (m >> alw) + 2 * (m & (aW - 1)) + (2 * dh + pa) * 2 * aW + pb) * aCin + coff);
    return (unsigned)((m + dh * aW) * aCin + coff);
  };
  const unsigned yv0 = 2u * (y_lin(y_row0) + (unsigned)y_c0), yv1 = 2u * (y_lin(y_row1) + (unsigned)y_c1);
  const bf16_t* ybase = ady + (sub == 1 ? 4L : 1L) * mb * aCout;
  const long ystep = (sub == 1 ? 128L : 32L) * aCout;
  // (wave 0's second x deposit, rows 32..39 of the stage, reads what its first one reads a K-step later: same lane pattern, same swizzle key)
  unsigned xoa = x_lin((int)mb - 4 + x_rowa, x_ca);
  const unsigned xstep = (sub == 2 ? 128u : 32u) * (unsigned)aCin;
  auto x_ptr = [&](int rowc, unsigned xo) -> const bf16_t* {
    const int m = i_p0 + rowc;
    const int ih = (int)__builtin_amdgcn_ubfe((unsigned)m, (unsigned)alw, (unsigned)alh) + dh;
    const bool ok = (unsigned)m < (unsigned)M32 && (unsigned)ih < (unsigned)aH;
    return ok ? ax + xo : azero;
  };
  auto issue = [&](int buf) {
    const unsigned stage = lds0 + buf * STAGE;
    glds16_sbase(ybase, yv0, stage + XT + (wave * 2) * 1024);
    glds16_sbase(ybase, yv1, stage + XT + (wave * 2 + 1) * 1024);
    ybase += ystep;
    glds16_asm(x_ptr(x_rowa - 4, xoa), stage + wave * 1024);
    xoa += xstep;
    if (wave == 0) glds16_asm(x_ptr(x_rowb - 4, xoa), stage + 4 * 1024);
    i_p0 += 32;
  };

  // ---- fragment addresses (per lane, stage-relative) and the left/right column masks -----------------
  // lane (g, li): channel = subtile*16 + li... transposing read: row = k-group g*4 + li/4, columns 4*(li%4)..+3
  int offx[NT][2], offy[4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 4 + (t + tb) + g * 4 + (li >> 2);
      const int slot0 = wi * 4 + j * 2;
      offx[t][j] = row * 128 + (((slot0 + ((li & 3) >> 1)) ^ (((row >> 1) & 3) << 1)) << 4) + (li & 1) * 8;
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = g * 4 + (li >> 2);
    const int slot0 = wo * 8 + i * 2;
    offy[i] = XT + row * 256 + (((slot0 + ((li & 3) >> 1)) ^ ((row & 7) << 1)) << 4) + (li & 1) * 8;
  }
  // fragment element e <-> pixel k = (e<4 ? g*4+e : 16+g*4+e-4); maskl clears the pixels of image column 0 (tap dw = -1),
  // maskr those of column W-1 (dw = +1).  SUB: only one of the two taps needs a mask -- mask1 belongs to tap 0 when tb = -1
  // (dw = -1: left) and to tap 1 when tb = 0 (dw = +1: right)
  uint32_t maskl[4], maskr[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t ml = 0, mr = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = 2 * d + h;
      const int k = (e < 4) ? (g * 4 + e) : (16 + g * 4 + (e - 4));
      const int ow = k & (a.W - 1);
      if (ow != 0) ml |= 0xFFFFu << (16 * h);
      if (ow != a.W - 1) mr |= 0xFFFFu << (16 * h);
    }
    maskl[d] = ml; maskr[d] = mr;
    if (SUB) { maskl[d] = tb < 0 ? ml : 0xFFFFFFFFu; maskr[d] = tb < 0 ? 0xFFFFFFFFu : mr; }     // = the masks of tap 0 / tap 1
  }

  f32x4_t acc[NT][4][2];   // [kw][co subtile][ci subtile]
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int KT = (int)((me - mb + 31) / 32);
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < KT) issue(t);
  for (int kt0 = 0; kt0 < KT; kt0 += NS) {
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      const int kt = kt0 + sidx;
      if (kt < KT) {
        // this wave's share of stage kt has landed once at most (NS-2) newer stages are still in flight
        if (kt + NS - 2 < KT) {
          if (wave == 0) wait_vmcnt_any<(NS - 2) * 4>(); else wait_vmcnt_any<(NS - 2) * 3>();
        } else {
          wait_vmcnt_any<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + NS - 1 < KT && !(WG3_ABLATE & 1)) issue((sidx + NS - 1) % NS);
        const unsigned char* sb = smem + ((WG3_ABLATE & 4) ? 0 : sidx) * STAGE;
        bf16x8_t yf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) yf[i] = tr_pair(sb + offy[i], 16 * 256);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (SUB && t >= ntap) continue;        // (workgroup-uniform: the 1x1 form)
          bf16x8_t xf[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            uint4 v = __builtin_bit_cast(uint4, tr_pair(sb + offx[t][j], 16 * 128));
            if (!(WG3_ABLATE & 8)) {
            if (SUB ? relu_on : RELU) { v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w); }
            // (column 0 can only be fragment element 0 or 4, column W-1 only element 3 or 7 -- W divides the 4-pixel runs a lane holds:
            // the left mask lives in words 0 and 2, the right one in words 1 and 3)
            if (t == 0) { v.x &= maskl[0]; v.z &= maskl[2]; }
            if (t == NT - 1) { v.y &= maskr[1]; v.w &= maskr[3]; }
            }
            xf[j] = __builtin_bit_cast(bf16x8_t, v);
          }
          if (WG3_ABLATE & 2) {       // keep the fragment reads alive without the matrix pipe
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(xf[j]));
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(yf[i]));
            continue;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[t][i][j] = mfma16(yf[i], xf[j], acc[t][i][j]);
        }
      }
    }
  }
  // D[row = co (4*(lane>>4)+r)][col = ci (lane&15)]
  float* slab = a.slab + (long)by * a.slab_stride;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (SUB && t >= ntap) continue;
    // slab cell: tap (kh, t), or [(pa*2 + pb)*4 + s*2 + d] with d = t, the rank among this parity's two taps
    const int cell = !SUB ? kh * 3 + t : (kh >> 1) * 4 + srow * 2 + t;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int co = co0 + wo * 64 + i * 16 + (lane >> 4) * 4;
        const int ci = ci0 + wi * 32 + j * 16 + (lane & 15);
        float4 v = make_float4(acc[t][i][j][0], acc[t][i][j][1], acc[t][i][j][2], acc[t][i][j][3]);
        *(float4*)(slab + ((long)cell * a.Cin + ci) * a.Cout + co) = v;
      }
  }
}


template <int NS, bool RELU>
__global__ __launch_bounds__(256) void conv_mfma_wgrad3_kernel(MfmaWgradArgs a) {
  if (__builtin_expect(a.sub != 0, 0)) wgrad3_body<NS, RELU, true>(a, blockIdx.x, blockIdx.y);
  else wgrad3_body<NS, RELU, false>(a, blockIdx.x, blockIdx.y);
}

// Several layers' filter gradients in ONE launch: workgroup b belongs to the problem p with first[p] <= b < first[p+1]
// and plays (b - first[p]) % gx[p], (b - first[p]) / gx[p] of that problem's own grid.  The 8x8 / 16x16 discriminator layers
// launch 224..602 workgroups each and are latency-bound alone; together their workgroups share the CUs (3 fit per CU).
#define WGRAD_GROUP_MAX 12
static_assert(WGRAD_GROUP_MAX == WGRAD_GROUP_MAX_HOST, "group size");
// The image-end layers' filter gradients (conv_image.h: D.Block.1.Conv1 / Shortcut) ride in the same launch as its LAST
// img.first[IMG_GROUP_MAX] workgroups.  Alone they are two launches of a few hundred short workgroups plus two slab reductions
// (59 us per critic step, 4 % of the iteration); here ~128 workgroups per layer walk their pixel blocks in the CU slots the
// three-tap workgroups leave free and finish under them: 8.41 -> 8.19 ms per iteration.  Measured: leading instead of trailing
// workgroups +0.11 ms, 512 / 1024 instead of 128 per layer +0.19 / +0.46 ms (slots taken from the three-tap workgroups,
// more slabs), 32 per layer +0.38 ms (they outlast the launch).
struct WgradGroup {
  int n;
  int xcd_runs;             // consecutive three-tap workgroups (a pixel chunk's tiles) on one XCD
  unsigned first[WGRAD_GROUP_MAX + 1];
  unsigned gx[WGRAD_GROUP_MAX];
  MfmaWgradArgs a[WGRAD_GROUP_MAX];
  ImgWGroup img;
  HeadWgradRider head;      // the projection head's deferred parameter sums: head.blocks workgroups behind the image-end ones (three-tap launches only)
};
static_assert(sizeof(WgradGroup) <= 4096, "kernel argument block");

template <int NS, bool RELU>
__global__ __launch_bounds__(256) void conv_mfma_wgrad3_group_kernel(WgradGroup g) {
  // (the branch is marked unlikely so that its code is laid out BEHIND the three-tap body: with the image-end code in front the
  // same, instruction-for-instruction identical hot loop ran 40 % slower -- 198 vs 141 us for the critic step's eleven layers)
  unsigned b = blockIdx.x;
  if (__builtin_expect(b >= g.first[g.n], 0)) {         // the image-end workgroups trail the grid, the head's trail them
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_img[];
    const unsigned bb = b - g.first[g.n];
    if (bb >= g.img.first[IMG_GROUP_MAX]) head_wgrad_body(g.head.a, g.head.dEg, (int)(bb - g.img.first[IMG_GROUP_MAX]), (float*)smem_img);
    else img_wgrad_group_body(g.img, bb, smem_img);
    return;
  }
  // Workgroups are dealt to the 8 XCDs round-robin by blockIdx.  The workgroups of one pixel chunk (consecutive indices: the filter rows and
  // channel tiles of a layer) read the same dy and x pixels: run them on ONE XCD, through one L2 (round 4; RCGAN_WGRAD3_XCD=0 restores the
  // dealt order) -- dealt out, every XCD fetched every chunk: 1.3 GB from beyond L2 per launch of a 256-channel layer at ~6.7 TB/s
  if (g.xcd_runs) {
    const unsigned t8 = g.first[g.n] >> 3;
    if (b < t8 * 8) b = (b & 7) * t8 + (b >> 3);
  }
  int p = 0;
#pragma unroll
  for (int q = 1; q < WGRAD_GROUP_MAX; ++q)
    if (q < g.n && b >= g.first[q]) p = q;
  const unsigned l = b - g.first[p];
  const unsigned gxp = g.gx[p];
  // (the sub-pixel body behind the three-tap one, as the image-end body: see the note on code placement above)
  if (__builtin_expect(g.a[p].sub != 0, 0)) wgrad3_body<NS, RELU, true>(g.a[p], l % gxp, l / gxp);
  else wgrad3_body<NS, RELU, false>(g.a[p], l % gxp, l / gxp);
}

// the same grouping for the per-tap kernel (1x1 shortcuts and the shapes the three-tap kernel does not take)
template <int NS>
__global__ __launch_bounds__(256) void conv_mfma_wgrad_glds_group_kernel(WgradGroup g) {
  const unsigned b = blockIdx.x;
  if (__builtin_expect(b >= g.first[g.n], 0)) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_img[];
    img_wgrad_group_body(g.img, b - g.first[g.n], smem_img);
    return;
  }
  int p = 0;
#pragma unroll
  for (int q = 1; q < WGRAD_GROUP_MAX; ++q)
    if (q < g.n && b >= g.first[q]) p = q;
  const unsigned l = b - g.first[p];
  const unsigned gxp = g.gx[p];
  wgrad_glds_body<NS>(g.a[p], l % gxp, l / gxp);
}

// ---------------------------------------------------------------------------------------------
// filter preparation: fp32 HWIO (optionally / sigma) -> bf16 [Cout][T*Cin] and the rotated
// in/out-swapped [Cin][T*Cout] used by the data gradient.
// ---------------------------------------------------------------------------------------------
__global__ void conv_prepare_mfma_kernel(const float* w, const float* sigma, bf16_t* wt, bf16_t* wd, int T, int Cin, int Cout) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)T * Cin * Cout;
  if (idx >= total) return;
  const float inv = sigma ? 1.f / *sigma : 1.f;
  int co = (int)(idx % Cout);
  long r = idx / Cout;
  int ci = (int)(r % Cin);
  int t = (int)(r / Cin);
  bf16_t v = f32_to_bf16(w[idx] * inv);
  wt[(long)co * T * Cin + (long)t * Cin + ci] = v;
  wd[(long)ci * T * Cout + (long)(T - 1 - t) * Cout + co] = v;
}

// Summed filters of the sub-pixel form of an upsample-3x3 convolution (MfmaConvArgs::wph):
//   wph[phase = ph*2 + pw][co][(a*2 + b)*Cin + ci] = sum of W[kh][kw][ci][co] over the taps (kh, kw) that read source pixel
//   (i + a - 1 + ph, j + b - 1 + pw) for output pixel (2i + ph, 2j + pw): row sets {0},{1,2} for ph = 0 and {0,1},{2} for ph = 1.
// Summed in fp32, rounded to 16 bits once.
struct PhasePrepBatch { struct It { const float* w; const float* sigma; bf16_t* wph; int Cin, Cout, kind; } it[8]; };
// Tap classes of the summed filters.  Upsample family (kind 0): row class a of phase ph covers kh in U(ph, a) = {0},{1,2} (ph = 0) /
// {0,1},{2} (ph = 1); the data gradient's 4 taps u cover {2},{1,2},{0,1},{0}.  Mean-pool family (kind 1, ConvMeanPool): the
// forward's 4 taps u cover P(u) = {0},{0,1},{1,2},{2} (x 1/4), and the data gradient's phase classes are P(u(ph, a)) with
// u(0,0) = 3, u(0,1) = 1, u(1,0) = 2, u(1,1) = 0.
__device__ __forceinline__ void phase_taps2(int kind, int ph, int a, int& k0, int& k1) {       // 2 classes per phase
  if (kind == 0) { k0 = ph == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2); k1 = ph == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2); }
  else {
    const int u = ph == 0 ? (a == 0 ? 3 : 1) : (a == 0 ? 2 : 0);
    k0 = u == 0 ? 0 : (u == 1 ? 0 : (u == 2 ? 1 : 2)); k1 = u == 0 ? 0 : (u == 1 ? 1 : 2);
  }
}
__device__ __forceinline__ void phase_taps4(int kind, int u, int& k0, int& k1) {               // 4 classes of the stride-2 form
  if (kind == 0) { k0 = u == 0 ? 2 : (u == 1 ? 1 : 0); k1 = u == 0 ? 2 : (u == 1 ? 2 : (u == 2 ? 1 : 0)); }
  else { k0 = u == 0 ? 0 : (u == 1 ? 0 : (u == 2 ? 1 : 2)); k1 = u == 0 ? 0 : (u == 1 ? 1 : 2); }
}

// Layouts (16 * Cin * Cout elements each):
//   "phase" layout  [phase = ph*2 + pw][O][(a*2 + b)*R + r]   -- four 2x2 convolutions over the low-resolution grid
//   "gather" layout [O][(u*4 + v)*R + r]                      -- one 4x4 stride-2 convolution over the full-resolution grid
// kind 0 (upsample -> conv): forward = phase layout (O = Cout, R = Cin), data gradient = gather layout (O = Cin, R = Cout)
// kind 1 (conv -> mean pool): forward = gather layout (O = Cout, R = Cin, x 1/4), data gradient = phase layout (O = Cin, R = Cout, x 1/4)
__global__ __launch_bounds__(256) void conv_prepare_phase_kernel(PhasePrepBatch b) {
  const PhasePrepBatch::It it = b.it[blockIdx.y];
  const float inv = (it.sigma ? 1.f / *it.sigma : 1.f) * (it.kind == 1 ? 0.25f : 1.f);
  const long total = 16L * it.Cin * it.Cout;
  bf16_t* const phase_l = it.kind == 0 ? it.wph : it.wph + total;
  bf16_t* const gather_l = it.kind == 0 ? it.wph + total : it.wph;
  const bool fwd_is_phase = it.kind == 0;
  const int Op = fwd_is_phase ? it.Cout : it.Cin, Rp = fwd_is_phase ? it.Cin : it.Cout;     // phase layout: output / reduction channels
  const int Og = fwd_is_phase ? it.Cin : it.Cout, Rg = fwd_is_phase ? it.Cout : it.Cin;     // gather layout
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    {   // phase layout element e
      const long per = (long)Op * 4 * Rp;
      const int phase = (int)(e / per);
      long r = e - phase * per;
      const int o = (int)(r / (4 * Rp));
      r -= (long)o * 4 * Rp;
      const int ab = (int)(r / Rp), rr = (int)(r - (long)ab * Rp);
      int kh0, kh1, kw0, kw1;
      phase_taps2(it.kind, phase >> 1, ab >> 1, kh0, kh1);
      phase_taps2(it.kind, phase & 1, ab & 1, kw0, kw1);
      const int ci = fwd_is_phase ? rr : o, co = fwd_is_phase ? o : rr;
      float s = 0.f;
      for (int kh = kh0; kh <= kh1; ++kh)
        for (int kw = kw0; kw <= kw1; ++kw) s += it.w[(((long)kh * 3 + kw) * it.Cin + ci) * it.Cout + co];
      phase_l[e] = f32_to_bf16(s * inv);
    }
    {   // gather layout element e
      const int o = (int)(e / (16L * Rg));
      long r = e - (long)o * 16 * Rg;
      const int uv = (int)(r / Rg), rr = (int)(r - (long)uv * Rg);
      int kh0, kh1, kw0, kw1;
      phase_taps4(it.kind, uv >> 2, kh0, kh1);
      phase_taps4(it.kind, uv & 3, kw0, kw1);
      const int ci = fwd_is_phase ? o : rr, co = fwd_is_phase ? rr : o;
      float s = 0.f;
      for (int kh = kh0; kh <= kh1; ++kh)
        for (int kw = kw0; kw <= kw1; ++kw) s += it.w[(((long)kh * 3 + kw) * it.Cin + ci) * it.Cout + co];
      gather_l[e] = f32_to_bf16(s * inv);
    }
  }
}

// phase filters of every upsample-3x3 convolution among the items (at most 8 per launch)
int conv_prepare_phase_launch(rcgan_ctx* ctx, int n, const float* const* ws, const float* const* sigmas, bf16_t* const* outs, const int* cins, const int* couts,
                              const int* kinds) {
  for (int base = 0; base < n; base += 8) {
    PhasePrepBatch b;
    const int m = n - base < 8 ? n - base : 8;
    long maxel = 0;
    for (int i = 0; i < m; ++i) {
      b.it[i] = {ws[base + i], sigmas[base + i], outs[base + i], cins[base + i], couts[base + i], kinds[base + i]};
      const long el = 16L * cins[base + i] * couts[base + i];
      if (el > maxel) maxel = el;
    }
    int bx = (int)cdiv(maxel, 256 * 8);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(conv_prepare_phase_kernel, dim3(bx, m), dim3(256), 0, ctx->stream, b);
    RC_LAUNCH_CHECK(ctx);
  }
  return RCGAN_OK;
}

// batched preparation: every filter of a network in ONE launch (blockIdx.y = filter)
struct PrepItem { const float* w; const float* sigma; void* out; void* extra; int T, Cin, Cout, mfma, img; };
// rows = the n filters, then the phase filters; row r owns workgroups [start[r], start[r+1]) of the one-dimensional grid
// ... and, last, an optional small-left GEMM (the label embeddings of the projection head: parameters only, see small_gemm.h)
// ... and the critic step's input work (step_inputs.h): noise, preprocessing, image pool, zero-fill
// (round 6) fragment rows: the fragment-major copies the fused 8x8 stage and the register-filter kernel read (conv_trunk.hip, conv_rf.hip)
// written by THIS launch straight from the fp32 weights -- they used to be a launch of their own behind this one (rf_fragments_kernel, 5 us on
// every step's dependency chain) that re-read the 16-bit rows this launch had just written.  Same values: the same fp32 product rounded once.
struct FragRow { int item; int ctn, ss; bf16_t* fwd; bf16_t* bwd; };
struct PrepBatch { PrepItem it[48]; PhasePrepBatch::It ph[8]; FragRow fr[12]; int start[72]; int n, rows, frag_row0, nfrag, gemm_row, inputs_row; SmallGemmArgs gemm; StepInputsArgs inputs; };

static_assert(sizeof(PrepBatch) <= 4096, "kernel arguments are limited to 4 KiB");

// rows [R][K] -> fragment-major [block of 16*ctn rows][slice][step ss][tile ctn][lane 64][8] (rf_fragments_kernel's layout): the element index of (row, k)
__device__ __forceinline__ long frag_index(int row, int k, int ctn, int ss, int nsl) {
  const int blk = row / (16 * ctn), ct = (row - blk * 16 * ctn) >> 4, r = row & 15;
  const int ks = k >> 5, sl = ks / ss, s = ks - sl * ss, kc = (k >> 3) & 3, e = k & 7;
  return ((((long)(blk * nsl + sl) * ss + s) * ctn + ct) * 64 + kc * 16 + r) * 8 + e;
}

__device__ __forceinline__ void prepare_fragment_units(const PrepItem& it, const FragRow& fr, bf16_t (*tile)[66], int bid, int nb) {
  const float inv = it.sigma ? 1.f / *it.sigma : 1.f;
  const int nci = it.Cin / 64, nco = it.Cout / 64;
  const int ntiles = it.T * nci * nco;
  const int nsl_f = it.T * it.Cin / 32 / fr.ss, nsl_d = it.T * it.Cout / 32 / fr.ss;
  const int lane64 = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (int tl = bid; tl < ntiles; tl += nb) {
    const int cot = tl % nco, cit = (tl / nco) % nci, t = tl / (nco * nci);
    const int ci0 = cit * 64, co0 = cot * 64;
    __syncthreads();
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = it.w[((long)t * it.Cin + ci0 + i * 4 + grp) * it.Cout + co0 + lane64];     // one round trip
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ci = i * 4 + grp;
      const bf16_t h = f32_to_bf16(v[i] * inv);
      // data-gradient rows: row = ci, reduction index (T - 1 - t) * Cout + co (the rotated filter): eight lanes = one 16-byte piece
      fr.bwd[frag_index(ci0 + ci, (it.T - 1 - t) * it.Cout + co0 + lane64, fr.ctn, fr.ss, nsl_d)] = h;
      tile[ci][lane64] = h;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int co = i * 4 + grp;
      fr.fwd[frag_index(co0 + co, t * it.Cin + ci0 + lane64, fr.ctn, fr.ss, nsl_f)] = tile[lane64][co];
    }
  }
}

// One (64 ci x 64 co tile, tap class) unit of the summed filters of the sub-pixel forms (layouts and tap classes: see
// conv_prepare_phase_kernel, whose values these are bit for bit -- same fp32 summation order).  The filter is read along co;
// the layout whose reduction index is co is written straight from registers, the one whose reduction index is ci goes through
// an LDS transpose, so all three streams are coalesced (the element-per-thread kernel read one of its two layouts with a stride
// of Cout floats per lane: 20 us for two 128 x 128 filters).
__device__ __forceinline__ void prepare_phase_units(const PhasePrepBatch::It& p, bf16_t (*tile)[66], int bid, int nb) {
  const float inv = (p.sigma ? 1.f / *p.sigma : 1.f) * (p.kind == 1 ? 0.25f : 1.f);
  const long total = 16L * p.Cin * p.Cout;
  bf16_t* const phase_l = p.kind == 0 ? p.wph : p.wph + total;
  bf16_t* const gather_l = p.kind == 0 ? p.wph + total : p.wph;
  const bool fwd_is_phase = p.kind == 0;
  const int nco = p.Cout / 64, units = (p.Cin / 64) * nco * 32;
  const int lane64 = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (int unit = bid; unit < units; unit += nb) {
    const int cls = unit & 31, tl = unit >> 5, c = cls & 15;
    const int co0 = (tl % nco) * 64, ci0 = (tl / nco) * 64;
    const bool is_phase = cls < 16;
    int kh0, kh1, kw0, kw1;
    if (is_phase) { phase_taps2(p.kind, c >> 3, (c >> 1) & 1, kh0, kh1); phase_taps2(p.kind, (c >> 2) & 1, c & 1, kw0, kw1); }
    else { phase_taps4(p.kind, c >> 2, kh0, kh1); phase_taps4(p.kind, c & 3, kw0, kw1); }
    const bool o_is_co = is_phase ? fwd_is_phase : !fwd_is_phase;
    const int O = o_is_co ? p.Cout : p.Cin, R = o_is_co ? p.Cin : p.Cout;
    bf16_t* const dst = is_phase ? phase_l + (long)(c >> 2) * O * 4 * R + (long)(c & 3) * R : gather_l + (long)c * R;
    const long ostride = is_phase ? 4L * R : 16L * R;
    __syncthreads();
    // all 16 rows x (up to) 4 taps requested before the first sum: one memory round trip per unit instead of a chain of 64
    const float* src[4];
    bool on[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kh = kh0 + (j >> 1), kw = kw0 + (j & 1);
      on[j] = kh <= kh1 && kw <= kw1;
      src[j] = p.w + (((long)(on[j] ? kh * 3 + kw : 0)) * p.Cin + ci0 + grp) * p.Cout + co0 + lane64;
    }
    float v[16][4];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i][j] = on[j] ? src[j][(long)i * 4 * p.Cout] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ci = i * 4 + grp;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s += v[i][j];        // (kh, kw) ascending, absent taps add +0: the sum of the nested tap loops
      const bf16_t h = f32_to_bf16(s * inv);
      if (o_is_co) tile[ci][lane64] = h;
      else dst[(long)(ci0 + ci) * ostride + co0 + lane64] = h;
    }
    if (o_is_co) {
      __syncthreads();
#pragma unroll 4
      for (int i = 0; i < 16; ++i) {
        const int co = i * 4 + grp;
        dst[(long)(co0 + co) * ostride + ci0 + lane64] = tile[lane64][co];
      }
    }
  }
}

__global__ __launch_bounds__(256) void conv_prepare_batch_kernel(PrepBatch b) {
  // (one buffer for the three row kinds: the 64 x 66 transpose tile, or the small GEMM's A chunk and reduction scratch)
  __shared__ __attribute__((aligned(16))) float lds_u[SG_AS_FLOATS + SG_RED_FLOATS];
  static_assert(sizeof(bf16_t) * 64 * 66 <= sizeof(float) * (SG_AS_FLOATS + SG_RED_FLOATS), "transpose tile fits");
  bf16_t (*tile)[66] = (bf16_t (*)[66])lds_u;
  int row = 0;                                   // largest r with start[r] <= blockIdx.x (six dependent scalar loads, not 57)
#pragma unroll
  for (int step = 32; step > 0; step >>= 1) {
    const int r = row + step;
    if (r < b.rows && (int)blockIdx.x >= b.start[r]) row = r;
  }
  const int bid = (int)blockIdx.x - b.start[row], nb = b.start[row + 1] - b.start[row];
  if (row >= b.frag_row0 && row < b.frag_row0 + b.nfrag) {
    const FragRow fr = b.fr[row - b.frag_row0];
    prepare_fragment_units(b.it[fr.item], fr, tile, bid, nb);
    return;
  }
  if (row == b.gemm_row) { small_gemm_body(b.gemm, bid, lds_u, lds_u + SG_AS_FLOATS); return; }
  if (row == b.inputs_row) { step_inputs_body(b.inputs, bid, nb); return; }
  if (row >= b.n) { prepare_phase_units(b.ph[row - b.n], tile, bid, nb); return; }
  const PrepItem it = b.it[row];
  const long total = (long)it.T * it.Cin * it.Cout;
  const float inv = it.sigma ? 1.f / *it.sigma : 1.f;
  if (it.mfma) {
    // 64(ci) x 64(co) tiles of one tap: read along co, write the straight (rotated-tap) copy along co and the
    // transposed copy along ci through LDS -- all three streams coalesced
    bf16_t* wt = (bf16_t*)it.out;
    bf16_t* wd = wt + total;
    const int nci = it.Cin / 64, nco = it.Cout / 64;
    const int ntiles = it.T * nci * nco;
    const int lane64 = threadIdx.x & 63, grp = threadIdx.x >> 6;
    for (int tl = bid; tl < ntiles; tl += nb) {
      const int cot = tl % nco, cit = (tl / nco) % nci, t = tl / (nco * nci);
      const int ci0 = cit * 64, co0 = cot * 64;
      __syncthreads();
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = it.w[((long)t * it.Cin + ci0 + i * 4 + grp) * it.Cout + co0 + lane64];     // one round trip
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = i * 4 + grp;
        const bf16_t h = f32_to_bf16(v[i] * inv);
        wd[(long)(ci0 + ci) * it.T * it.Cout + (long)(it.T - 1 - t) * it.Cout + co0 + lane64] = h;
        tile[ci][lane64] = h;
      }
      __syncthreads();
#pragma unroll 4
      for (int i = 0; i < 16; ++i) {
        const int co = i * 4 + grp;
        wt[(long)(co0 + co) * it.T * it.Cin + (long)t * it.Cin + ci0 + lane64] = tile[lane64][co];
      }
    }
    return;
  }
  for (long idx = (long)bid * blockDim.x + threadIdx.x; idx < total; idx += (long)nb * blockDim.x)
    ((float*)it.out)[idx] = it.w[idx] * inv;
  if (it.img) {      // bf16 layouts of the image-end kernels (conv_image.hip)
    const int cb = it.img == 1 ? it.Cout : it.Cin;
    const long extra = (long)cb * 32 + (long)it.T * 16 * cb;
    for (long e = (long)bid * blockDim.x + threadIdx.x; e < extra; e += (long)nb * blockDim.x)
      ((bf16_t*)it.extra)[e] = img_prepare_elem(e, it.img, it.T, it.Cin, it.Cout, it.w, inv);
  }
}

static int env_int(const char* name, int dflt);
int conv_prepare_batch_launch(rcgan_ctx* ctx, const rcgan_prepare_item* items, int n, const SmallGemmArgs* gemm, const StepInputsArgs* inputs,
                              const rcgan_frag_item* frags, int n_frags) {
  // the summed phase filters of the sub-pixel forms (upsample-3x3, ConvMeanPool): up to 8 of them ride as extra rows of the
  // first launch's grid, the rest (none in these networks) take the stand-alone kernel
  static const int ride = env_int("RCGAN_PREP_PHASE_RIDE", 1);
  std::vector<const float*> pw_, ps_;
  std::vector<bf16_t*> po_;
  std::vector<int> pci, pco, pk;
  for (int i = 0; i < n; ++i) {
    const rcgan_conv_desc& d = items[i].desc;
    if (!mfma_phase_filters(&d)) continue;
    pw_.push_back(items[i].w); ps_.push_back(items[i].sigma);
    po_.push_back((bf16_t*)items[i].prepared + 2 * (size_t)d.kh * d.kw * d.cin * d.cout);
    pci.push_back(d.cin); pco.push_back(d.cout); pk.push_back((d.flags & RCGAN_CONV_OUT_MEANPOOL2) ? 1 : 0);
  }
  int nride = ride ? (int)pw_.size() : 0;
  if (nride > 8) nride = 8;
  for (int base = 0; base < n; base += 48) {
    PrepBatch b;
    int m = n - base < 48 ? n - base : 48;
    b.n = m;
    const int np = base == 0 ? nride : 0;
    for (int i = 0; i < np; ++i) b.ph[i] = {pw_[i], ps_[i], po_[i], pci[i], pco[i], pk[i]};
    // workgroups per row: one 64 x 64 tile (MFMA layouts) or 256 elements each, at most 512 per filter; a phase filter has
    // (Cin/64)(Cout/64) * 32 units (tile x tap class), one per workgroup up to 1024
    int at = 0;
    for (int i = 0; i < m; ++i) {
      const rcgan_prepare_item& s = items[base + i];
      rcgan_conv_desc d = s.desc;
      b.it[i].w = s.w; b.it[i].sigma = s.sigma; b.it[i].out = s.prepared;
      b.it[i].T = d.kh * d.kw; b.it[i].Cin = d.cin; b.it[i].Cout = d.cout; b.it[i].mfma = mfma_eligible(&d) ? 1 : 0;
      b.it[i].img = b.it[i].mfma ? 0 : img_side(&d);
      b.it[i].extra = b.it[i].img ? (char*)s.prepared + img_extra_offset(&d) : nullptr;
      const long el = (long)d.kh * d.kw * d.cin * d.cout;
      // (the image-end layouts gather up to 27 taps per element: one element per thread)
      const long img_el = b.it[i].img ? (long)(b.it[i].img == 1 ? d.cout : d.cin) * (32 + b.it[i].T * 16) : 0;
      long want = b.it[i].mfma ? el / 4096 : cdiv(el > img_el ? el : img_el, 256);
      if (want < 1) want = 1;
      if (want > 512) want = 512;
      b.start[i] = at;
      at += (int)want;
    }
    for (int i = 0; i < np; ++i) {
      const long units = (long)(pci[i] / 64) * (pco[i] / 64) * 32;
      b.start[m + i] = at;
      at += units > 1024 ? 1024 : (int)units;
    }
    b.rows = m + np;
    // fragment rows: one workgroup per 64 x 64 tile of a tap, for the filters of THIS launch's items
    b.frag_row0 = b.rows;
    b.nfrag = 0;
    for (int f = 0; f < n_frags; ++f) {
      const int idx = frags[f].item - base;
      if (idx < 0 || idx >= m) continue;
      const rcgan_conv_desc& d = items[frags[f].item].desc;
      b.fr[b.nfrag] = FragRow{idx, frags[f].ctn, frags[f].ss, (bf16_t*)frags[f].fwd, (bf16_t*)frags[f].bwd};
      b.start[b.rows] = at;
      at += d.kh * d.kw * (d.cin / 64) * (d.cout / 64);
      ++b.nfrag;
      ++b.rows;
    }
    b.start[b.rows] = at;
    b.gemm_row = -1;
    if (gemm && base == 0) {             // the riding product: cdiv(d, 16) workgroups behind everything else
      b.gemm = *gemm;
      b.gemm_row = b.rows++;
      at += (gemm->d + 15) / 16;
      b.start[b.rows] = at;
    }
    b.inputs_row = -1;
    if (inputs && base == 0) {
      b.inputs = *inputs;
      b.inputs_row = b.rows++;
      // a few elements per thread: the image units, the fake half's pooled outputs and the fill share the workgroups
      const size_t work = (size_t)inputs->n * 3 * 16 * 8 + (inputs->pooled ? (size_t)inputs->n * 768 : 0) + inputs->fill4;
      // (measured, round 2: 256 workgroups 6.37 ms, 512: 6.42, 2048: 6.48, 64: 6.39 per iteration; round 3: 128: 5.770, 256: 5.784, 512: 5.785)
      static const int per = env_int("RCGAN_RIDE_PER_THREAD", 16), cap = env_int("RCGAN_RIDE_MAXWG", 128);
      size_t wgs = (work + 256 * per - 1) / (256 * per);
      if (wgs < 1) wgs = 1;
      if (wgs > (size_t)cap) wgs = cap;
      at += (int)wgs;
      b.start[b.rows] = at;
    }
    hipLaunchKernelGGL(conv_prepare_batch_kernel, dim3(at), dim3(256), 0, ctx->stream, b);
    RC_LAUNCH_CHECK(ctx);
  }
  const int rest = (int)pw_.size() - nride;
  if (rest > 0)
    return conv_prepare_phase_launch(ctx, rest, pw_.data() + nride, ps_.data() + nride, po_.data() + nride, pci.data() + nride, pco.data() + nride,
                                     pk.data() + nride);
  return RCGAN_OK;
}

__global__ void conv_prepare_direct_kernel(const float* w, const float* sigma, float* out, long total) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const float inv = sigma ? 1.f / *sigma : 1.f;
  out[idx] = w[idx] * inv;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

bool mfma_eligible(const rcgan_conv_desc* d) {
  if (d->dtype != RCGAN_H16) return false;
  if (d->flags & RCGAN_CONV_FORCE_DIRECT) return false;
  if (d->stride != 1) return false;
  if (!((d->kh == 3 && d->kw == 3) || (d->kh == 1 && d->kw == 1))) return false;
  if (d->cin % 64 || d->cout % 64) return false;
  return true;
}

// an upsample-3x3 convolution on the matrix-core path: its prepared buffer carries the four summed phase filters behind the two
// ordinary layouts (16 * cin * cout elements)
bool mfma_phase_filters(const rcgan_conv_desc* d) {
  return mfma_eligible(d) && (d->flags & (RCGAN_CONV_IN_UPSAMPLE2X | RCGAN_CONV_OUT_MEANPOOL2)) && d->kh == 3 && d->kw == 3;
}

// ConvMeanPool with the pool folded in (RCGAN_CONV_OUT_MEANPOOL2): power-of-two images, whole 64-pixel tiles of pooled pixels
bool mfma_pool_ok(const rcgan_conv_desc* d) {
  static const int on = env_int("RCGAN_FUSED_POOL", 1);
  if (!on || !mfma_phase_filters(d) || !(d->flags & RCGAN_CONV_OUT_MEANPOOL2) || (d->flags & RCGAN_CONV_IN_UPSAMPLE2X)) return false;
  const int lw = ilog2_exact(d->w), lh = ilog2_exact(d->h);
  return lw >= 1 && lh >= 1 && ((long)d->n * (d->h / 2) * (d->w / 2)) % 64 == 0;
}

// the data gradient of such a layer in the sub-pixel form: power-of-two images, whole 64-pixel tiles of low-resolution pixels
bool mfma_phase_dgrad_ok(const rcgan_conv_desc* d) {
  static const int on = env_int("RCGAN_UP_PHASE_DGRAD", 1), on1 = env_int("RCGAN_UP_PHASE", 1);
  if (!on || !on1 || !mfma_phase_filters(d)) return false;
  const int lw = ilog2_exact(d->w), lh = ilog2_exact(d->h);
  return lw >= 1 && lh >= 1 && ((long)d->n * (d->h / 2) * (d->w / 2)) % 64 == 0;
}

bool mfma_wgrad_eligible(const rcgan_conv_desc* d) {
  return mfma_eligible(d) && d->cin % 128 == 0 && d->cout % 128 == 0;
}

static int conv_impl() {      // 1 = direct-to-LDS (default), 0 = register-staged (RCGAN_CONV_IMPL=reg)
  static int v = -1;
  if (v < 0) { const char* e = getenv("RCGAN_CONV_IMPL"); v = (e && e[0] == 'r') ? 0 : 1; }
  return v;
}


// the sub-pixel instantiation of the 64 x 64 kernel (G.Block.1.Conv1: 8x8 from 4x4, 1024 -> 256): whole tiles per phase
template <int BM, int BN, int NS, int KS, int PHASE = 1>
static int launch_conv_glds_phase(rcgan_ctx* ctx, const MfmaConvArgs& a_in) {
  MfmaConvArgs a = a_in;
  a.phase = PHASE;
  static bool attr_set = false;
  size_t lds = (size_t)KS * NS * (BM + BN) * 128;
  const size_t part = (size_t)(KS - 1) * 256 * (BM / 32) * (BN / 32) * 4 * sizeof(float);
  if (part > lds) lds = part;
  if (!attr_set) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_glds_kernel<BM, BN, NS, KS, PHASE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, BM), a.Cout / BN);
  {
    ProfScope ps(ctx, RCGAN_PROF_CONV_MFMA_64, 2.0 * (double)a.M * (PHASE == 2 ? 36 : 9) * a.Cin * a.Cout, 2.0 * (double)a.M * (PHASE == 2 ? 16 : 4) * a.Cin * a.Cout);
    hipLaunchKernelGGL((conv_mfma_glds_kernel<BM, BN, NS, KS, PHASE>), grid, dim3(256 * KS), lds, ctx->stream, a);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

template <int BM, int BN, int NS, int KS = 1>
static int launch_conv_glds(rcgan_ctx* ctx, const MfmaConvArgs& a) {
  static bool attr_set = false;
  size_t lds = (size_t)KS * NS * (BM + BN) * 128;
  if (!attr_set) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_glds_kernel<BM, BN, NS, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, BM), a.Cout / BN);
  {
    ProfScope ps(ctx, BM >= 128 ? RCGAN_PROF_CONV_MFMA_128 : RCGAN_PROF_CONV_MFMA_64,
                 2.0 * (double)a.M * a.KH * a.KW * a.Cin * a.Cout);
    hipLaunchKernelGGL((conv_mfma_glds_kernel<BM, BN, NS, KS>), grid, dim3(256 * KS), lds, ctx->stream, a);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// halo-staged kernel: 3x3 filters on power-of-two images whose rows fit the tile (tile = whole rows of one image), at most 256
// input channels (all 64-channel patches resident in LDS)
template <int BM>
static bool halo_ok(const MfmaConvArgs& a) {
  static const int on = env_int("RCGAN_CONV_HALO", 1);
  return on && a.KH == 3 && a.KW == 3 && a.lw >= 0 && a.lh >= 0 && a.W <= BM && (a.H * a.W) % BM == 0 && a.Cin <= 256 && a.zero != nullptr &&
         (long)a.N * a.H * a.W * a.Cin < (1L << 32);
}

template <int BM, int BN, int NS, int KS>
static int launch_conv_halo(rcgan_ctx* ctx, const MfmaConvArgs& a) {
  const int TR = BM / a.W, PD = ((TR + 2) * (a.W + 2) + 7) / 8;
  size_t lds = (size_t)(a.Cin / 64) * PD * 1024 + (size_t)KS * NS * BN * 128;
  const size_t part = (size_t)(KS - 1) * 256 * (BM / 32) * (BN / 32) * 4 * sizeof(float);       // K-group partial sums
  if (part > lds) lds = part;
  static size_t attr = 0;
  if (lds > attr) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_halo_kernel<BM, BN, NS, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = lds;
  }
  dim3 grid(cdiv(a.M, BM), a.Cout / BN);
  {
    ProfScope ps(ctx, RCGAN_PROF_CONV_MFMA_64, 2.0 * (double)a.M * a.KH * a.KW * a.Cin * a.Cout);
    hipLaunchKernelGGL((conv_mfma_halo_kernel<BM, BN, NS, KS>), grid, dim3(256 * KS), lds, ctx->stream, a);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

template <int BM, int BN>
static int launch_conv_mfma(rcgan_ctx* ctx, const MfmaConvArgs& a) {
  if (conv_impl() == 1 && a.zero != nullptr) {
    if (BM == 128) return launch_conv_glds<BM, BN, 2>(ctx, a);
    // small grids are latency-bound (one HBM/L2 round trip per K-tile): deepen the pipeline; large grids
    // prefer the smaller LDS footprint (more resident workgroups)
    static const int ns4_max = env_int("RCGAN_NS4_MAXBLK", 0);
    static const int ks4_max = env_int("RCGAN_KS4_MAXBLK", 288), ks2_max = env_int("RCGAN_KS2_MAXBLK", 576);
    const long blocks = (long)cdiv(a.M, BM) * (a.Cout / BN);
    const int ktiles = a.KH * a.KW * a.Cin / 64;
    if constexpr (BM == 64) {
      if (halo_ok<64>(a)) {
        // Measured (scripts/bench_conv.py): the halo form pays where ONE workgroup per CU runs a short K loop -- the 8x8
        // 128-channel layers, 9.7 -> 8.4 us.  On the 16x16 layers (four resident workgroups per CU hide each other's DMA
        // round trips; the patch costs LDS = residency) and at 256 input channels (53 KB of patches: one workgroup per CU
        // instead of two) it is 5-30 % slower, with 64- or 128-pixel tiles, two or four filter stages alike: those keep the
        // tile-per-tap kernel.
        if (blocks <= ks4_max && ktiles >= 8 && a.Cin <= 128) return launch_conv_halo<64, 64, 2, 4>(ctx, a);
      }
      static const int deep = env_int("RCGAN_KS_DEEP", 0);      // experiment: 1 = 2 K-groups x 4 stages, 2 = 3 K-groups x 3 stages
      if (deep == 1 && blocks <= ks4_max && ktiles >= 8) return launch_conv_glds<BM, BN, 4, 2>(ctx, a);
      if (deep == 2 && blocks <= ks4_max && ktiles >= 9) return launch_conv_glds<BM, BN, 3, 3>(ctx, a);
      {
        static const int up_phase = env_int("RCGAN_UP_PHASE", 1);
        if (up_phase && a.up && a.KH == 3 && a.KW == 3 && a.wph != nullptr && a.lw >= 1 && a.lh >= 1 && ((a.M >> 2) % BM) == 0) {
          if (blocks <= ks2_max) return launch_conv_glds_phase<BM, BN, 2, 2>(ctx, a);
          return launch_conv_glds_phase<BM, BN, 2, 1>(ctx, a);
        }
      }
      if (blocks <= ks4_max && ktiles >= 8) return launch_conv_glds<BM, BN, 2, 4>(ctx, a);
      if (blocks <= ks2_max && ktiles >= 4) return launch_conv_glds<BM, BN, 2, 2>(ctx, a);
    }
    if (blocks <= ns4_max) return launch_conv_glds<BM, BN, 4>(ctx, a);
    return launch_conv_glds<BM, BN, 2>(ctx, a);
  }
  static bool attr_set = false;
  size_t lds = (size_t)2 * (BM + BN) * LDS_PITCH * sizeof(bf16_t);
  if (!attr_set) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, BM), a.Cout / BN);
  {
    ProfScope ps(ctx, BM == 128 ? RCGAN_PROF_CONV_MFMA_128 : RCGAN_PROF_CONV_MFMA_64,
                 2.0 * (double)a.M * a.KH * a.KW * a.Cin * a.Cout);
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN>), grid, dim3(256), lds, ctx->stream, a);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// THE routing decision of a forward / data-gradient launch (round 6: one function instead of three hand-kept mirrors).  It is the only
// place that reads the routing thresholds; mfma_conv_launch launches what it returns, and the questions the host asks before it builds a
// fused call (mfma_conv_is_p8: tile statistics; mfma_conv_bn_route: batch norm on the staged input) read the same answer.
enum ConvRoute {
  CR_H8,            // 256 x 256 halo-patch kernel (conv_mfma8h.hip)
  CR_P8,            // 256 x 256 tile-per-tap kernel (conv_mfma8.hip)
  CR_H8N,           // 256 x 128 halo-patch kernel; on a phase-2 launch: its parity-plane gather form
  CR_P8N,           // 256 x 128 tile-per-tap kernel
  CR_PHASE_KS2,     // 64 x 64 sub-pixel forms, two K-groups
  CR_PHASE,         // 64 x 64 sub-pixel forms
  CR_T256, CR_T256N, CR_T128, CR_64
};

static ConvRoute mfma_conv_route(const MfmaConvArgs& a) {
  static const int p8_min = env_int("RCGAN_P8_MINBLK", 200);          // 256 x 256 eight-wave kernels
  static const int p8n_min = env_int("RCGAN_P8N_MINBLK", 190);        // 256 x 128 eight-wave kernels
  static const int ks2_max = env_int("RCGAN_KS2_MAXBLK", 576);
  static const int halo = env_int("RCGAN_P8_HALO", 1), halo_n = env_int("RCGAN_P8N_HALO", 1);
  static const int t128_min = env_int("RCGAN_T128_MINBLK", 384);
  static const int t256_min = env_int("RCGAN_T256_MINBLK", 1 << 30);     // experiment: 256-pixel tiles (1 wave/SIMD)
  if (a.phase == 1 || a.phase == 2) {
    // forced sub-pixel forms.  1: the data gradient of a ConvMeanPool (a.wt is not a usable fallback); 2: the data gradient of the sub-pixel
    // form (the caller checked mfma_phase_dgrad_ok).  256-pixel tiles where the grid fills the chip, else the 64 x 64 kernel
    const bool whole = a.phase == 1 ? ((a.M >> 2) % 256 == 0) : (a.M % 256 == 0);
    const long b8 = (a.M / 256) * (a.Cout / 256), b8n = (a.M / 256) * (a.Cout / 128);
    if (a.phase == 2) {
      // (round 5) the parity-plane patch form of the 256 x 128 halo kernel where its tiles fill the chip: G.Block.3.Conv1's data gradient
      // (128 x 2 tiles) 78 -> 61 us.  Measured and left to the 64 x 64 kernel: D.Block.1.Conv2's forward (128 tiles = half the CUs) 33 -> 37 us
      // (RCGAN_H8N_GATHER_MINBLK=120 takes it too; 100000 switches the form off -- since round 6 really off: the 256 x 128 tile-per-tap
      // kernel gets the launch, not the gather form by the back door of mfma_conv8_launch)
      const int gather_min = env_int("RCGAN_H8N_GATHER_MINBLK", 190);        // (read per call: the tests force the form on smaller grids)
      if (halo_n && whole && a.Cout % 128 == 0 && b8n >= gather_min && mfma_conv8n_halo_takes(a)) return CR_H8N;
    }
    if (whole && a.Cout % 256 == 0 && b8 >= p8_min && 4 * b8 >= 3 * (long)cdiv(b8, 256) * 256)
      return (halo && !a.stats && mfma_conv8_halo_takes(a)) ? CR_H8 : CR_P8;
    if (whole && a.Cout % 128 == 0 && b8n >= p8n_min)
      return (a.phase == 1 && halo_n && mfma_conv8n_halo_takes(a)) ? CR_H8N : CR_P8N;
    // (measured in round 5 and not kept: 128 x 128 tiles of the 64 x 64 kernel for D.Block.1.Conv2's forward -- 256 tiles, one per CU, against
    // 1024 of 64 x 64 -- same-box 5.20 -> 5.25 ms; for every 16-tap layer 5.40)
    return (a.M / 64) * (a.Cout / 64) <= ks2_max ? CR_PHASE_KS2 : CR_PHASE;
  }
  MfmaConvArgs b = a;
  b.phase = mfma_conv8_phase_form(a) ? 1 : 0;                         // what mfma_conv8_launch will run it as
  const bool off32 = (long)a.N * a.H * a.W * a.Cin < (1L << 32);      // the tap-source table holds 32-bit element offsets
  // one 256 x 256 workgroup per CU: a grid of 320 runs two rounds for 1.25 rounds of work -- such grids go to the
  // 256 x 128 kernel (twice the workgroups, finer rounds) when less than 3/4 of the last round would be busy
  const long b8 = (long)cdiv(a.M, 256) * (a.Cout / 256), b8n = (long)cdiv(a.M, 256) * (a.Cout / 128);
  const bool fit8 = 4 * b8 >= 3 * (long)cdiv(b8, 256) * 256;
  if (a.Cout % 256 == 0 && a.zero != nullptr && off32 && b8 >= p8_min && fit8) return (halo && !a.stats && mfma_conv8_halo_takes(b)) ? CR_H8 : CR_P8;
  if (a.Cout % 128 == 0 && a.zero != nullptr && b8n >= p8n_min) return (halo_n && mfma_conv8n_halo_takes(b)) ? CR_H8N : CR_P8N;
  if (a.Cout % 256 == 0 && b8 >= t256_min && a.zero != nullptr) return CR_T256;
  if (a.Cout % 128 == 0 && b8n >= t256_min && a.zero != nullptr) return CR_T256N;
  if (a.Cout % 128 == 0 && (long)cdiv(a.M, 128) * (a.Cout / 128) >= t128_min) return CR_T128;
  return CR_64;
}

// an ordinary (phase 0 on entry) forward / data-gradient launch that runs on a 256 x 256 eight-wave kernel
bool mfma_conv_is_p8(const MfmaConvArgs& a) {
  if (a.phase != 0 || a.Cin % 64 || a.Cout % 64) return false;
  const ConvRoute r = mfma_conv_route(a);
  return r == CR_H8 || r == CR_P8;
}

// Does this (phase 0 on entry) forward launch run on one of the halo-patch kernels -- the ones that can apply a batch norm to their
// staged input (MfmaConvArgs::bn_*)?  1: 256 x 256 tile, 2: 256 x 128 tile, 0: no.
int mfma_conv_bn_route(const MfmaConvArgs& a) {
  if (a.phase != 0 || a.Cin % 64 || a.Cout % 64 || a.zero == nullptr || a.stats || a.relu_in || a.Cin > 1024) return 0;
  const ConvRoute r = mfma_conv_route(a);
  return r == CR_H8 ? 1 : r == CR_H8N ? 2 : 0;
}

int mfma_conv_launch(rcgan_ctx* ctx, const MfmaConvArgs& a_in) {
  MfmaConvArgs a = a_in;
  a.stamps = (unsigned long long*)ctx->dbg_stamps;       // diagnostics (rcgan_debug_stamps), normally null
  if (a.Cin % 64 || a.Cout % 64) RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "channels %d -> %d", a.Cin, a.Cout);
  const ConvRoute r = mfma_conv_route(a);
  if (a.bn_mean && !(a.phase == 0 && !a.stats && !a.relu_in && a.Cin <= 1024 && a.zero != nullptr && (r == CR_H8 || r == CR_H8N)))
    RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "batch norm on the staged input needs a halo-patch kernel (rcgan_conv_bn_in_ok)");
  if (a.stats && !(a.phase == 0 && a.zero != nullptr && (r == CR_H8 || r == CR_P8)))
    RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "tile statistics need the 256 x 256 kernel (rcgan_conv_stats_ok)");
  switch (r) {
    case CR_H8: case CR_P8: return mfma_conv8_launch(ctx, a, true, r == CR_H8);
    case CR_H8N: case CR_P8N: return mfma_conv8_launch(ctx, a, false, r == CR_H8N);
    case CR_PHASE_KS2: return a.phase == 1 ? launch_conv_glds_phase<64, 64, 2, 2, 1>(ctx, a) : launch_conv_glds_phase<64, 64, 2, 2, 2>(ctx, a);
    case CR_PHASE: return a.phase == 1 ? launch_conv_glds_phase<64, 64, 2, 1, 1>(ctx, a) : launch_conv_glds_phase<64, 64, 2, 1, 2>(ctx, a);
    case CR_T256: return launch_conv_glds<256, 256, 2>(ctx, a);
    case CR_T256N: return launch_conv_glds<256, 128, 2>(ctx, a);
    case CR_T128: return launch_conv_mfma<128, 128>(ctx, a);
    default: return launch_conv_mfma<64, 64>(ctx, a);
  }
}

static bool wgrad3_shape(int kh, int kw, int h, int w) {
  // three-taps-per-workgroup kernel: 3x3 filters, W a power of two dividing the 32-pixel stage, H a power of two
  return kh == 3 && kw == 3 && ilog2_exact(w) >= 0 && ilog2_exact(h) >= 0 && w >= 4 && w <= 32;
}

static bool wgrad3_geom(const MfmaWgradArgs& a) {     // (sub-pixel and 1x1 forms: the grid of the reduction index decides)
  if (a.M % 32 != 0 || a.up) return false;       // whole K-steps: the dy stream of the linear addressing has no tail test; no floor(row / 2) sources
  return a.sub ? wgrad3_shape(3, 3, a.H, a.W) : (wgrad3_shape(a.KH, a.KW, a.H, a.W) && a.PL == 1);
}

static int wgrad3_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("RCGAN_WGRAD_IMPL"); v = (e && (e[0] == 'r' || e[0] == 'g')) ? 0 : 1; }
  return v;
}

static long wgrad_clamp_splits(long want, long M) {
  static const int min_px = env_int("RCGAN_WGRAD_MINPX", 256);
  long maxs = (M + min_px - 1) / min_px;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 128) want = 128;
  return want;
}

int mfma_wgrad_splits(const rcgan_conv_desc* d, long M) {
  long tiles = (long)d->kh * d->kw * (d->cin / 128) * (d->cout / 128);
  long want = wgrad_clamp_splits((768 + tiles - 1) / tiles, M);
  if (wgrad3_shape(d->kh, d->kw, d->h, d->w)) {      // workspace must cover whichever kernel the launch picks
    long tiles3 = (long)d->kh * (d->cin / 64) * (d->cout / 128);
    long want3 = wgrad_clamp_splits((512 + tiles3 - 1) / tiles3, M);
    if (want3 > want) want = want3;
    // ... or the nine-tap kernel: one round of 256 workgroups (a 256-channel layer: 8 tiles x 32 chunks, not the 22 of the rule above)
    if (d->cin % 64 == 0 && d->cout % 128 == 0) {
      long tiles9 = (long)(d->cin / 64) * (d->cout / 128);
      long want9 = wgrad_clamp_splits((256 + tiles9 - 1) / tiles9, M);
      if (want9 > want) want = want9;
    }
  }
  return (int)want;
}

// Sub-pixel filter gradient (MfmaWgradArgs::sub) of an upsample-3x3 (1) or ConvMeanPool (2) layer: 0 where the layer is posed
// as an ordinary 3x3 filter gradient (shape the three-tap kernel does not take, RCGAN_WGRAD_SUBPIXEL=0)
int mfma_wgrad_sub_kind(const rcgan_conv_desc* d, int use_tr) {
  static const int on = env_int("RCGAN_WGRAD_SUBPIXEL", 1);
  if (!use_tr || !wgrad3_enabled() || !mfma_wgrad_eligible(d)) return 0;
  if (d->kh == 1 && d->kw == 1) {
    // a SMALL 1x1 (the down blocks' shortcuts: 0.27 GFLOP at the bench batch) rides in the grouped launch on the two-tap body; a big
    // one keeps the per-tap kernel's 128 x 128 tiles (a staged pixel feeds one tap's MFMAs here: a third of the arithmetic intensity)
    static const double max_gflop = env_int("RCGAN_WGRAD_1X1_GROUP_MAXMFLOP", 600) * 1e6;
    const double fl = 2.0 * d->n * d->h * d->w * (double)d->cin * d->cout;
    // (the reduction index runs in whole 32-pixel K-steps: wgrad3_geom)
    return fl <= max_gflop && !(d->flags & (RCGAN_CONV_OUT_MEANPOOL2 | RCGAN_CONV_IN_UPSAMPLE2X)) && wgrad3_shape(3, 3, d->h, d->w) &&
           ((long)d->n * d->h * d->w) % 32 == 0 ? 3 : 0;
  }
  if (!on || d->kh != 3 || d->kw != 3) return 0;
  const bool up = (d->flags & RCGAN_CONV_IN_UPSAMPLE2X) != 0, pool = (d->flags & RCGAN_CONV_OUT_MEANPOOL2) != 0;
  if (up == pool || (d->h & 1) || (d->w & 1)) return 0;
  if (!wgrad3_shape(3, 3, d->h / 2, d->w / 2) || ((long)d->n * (d->h / 2) * (d->w / 2)) % 32 != 0) return 0;
  return up ? 1 : 2;
}

int mfma_wgrad_sub_splits(const rcgan_conv_desc* d, long M) {
  const long tiles3 = (d->kh == 1 ? 1L : 8L) * (d->cin / 64) * (d->cout / 128);
  return (int)wgrad_clamp_splits((512 + tiles3 - 1) / tiles3, M);
}

template <bool RELU>
static int launch_wgrad3(rcgan_ctx* ctx, MfmaWgradArgs& a, dim3 grid) {
  constexpr int NS = 4;
  static bool attr = false;
  size_t lds = (size_t)NS * (40 * 128 + 32 * 256);
  if (!attr) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_wgrad3_kernel<NS, RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  {
    ProfScope ps(ctx, RCGAN_PROF_WGRAD_MFMA, 2.0 * (double)a.M * wgrad3_alg_taps(a) * a.Cin * a.Cout,
                 2.0 * (double)a.M * wgrad3_exec_taps(a) * a.Cin * a.Cout);
    hipLaunchKernelGGL((conv_mfma_wgrad3_kernel<NS, RELU>), grid, dim3(256), lds, ctx->stream, a);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int mfma_wgrad_launch(rcgan_ctx* ctx, MfmaWgradArgs& a, int nz, bool* bias_done, size_t ws_bytes) {
  *bias_done = false;
  if (a.sub && !mfma_wgrad3_takes(a)) RC_FAIL(ctx, RCGAN_EUNSUPPORTED_SHAPE, "sub-pixel filter gradient needs the three-tap kernel");
  {
    // the nine-tap kernel where a workgroup's share of the pixels amortises its nine-cell slab (scripts/bench_conv.py, n = 128, layers alone:
    // 32x32 256-channel 175 -> 142 us, 32x32 128-channel 72 -> 67, 16x16 256-channel 60 -> 60; but 8x8 256-channel 32 -> 37, 8x8 128-channel
    // 21 -> 28, the sub-pixel forms 43 -> 83 / 122 -> 128: those stay on the three-tap kernel)
    const long min_work = env_int("RCGAN_WGRAD9_MINWORK", 200000);        // (read per call: the tests force the kernel on small shapes)
    const long work9 = a.M * (long)(a.Cin / 64) * (a.Cout / 128);
    unsigned gx9 = 0, gy9 = 0;
    // (planned on a copy: the plan rewrites slab_stride -- the upsample form carries two bias tails per slab -- AFTER the caller sized
    // the workspace for the stride it knew; the nine-tap kernel only runs if its slabs fit what the caller really handed over)
    MfmaWgradArgs a9 = a;
    if ((a.sub ? min_work <= 0 : work9 >= min_work) && mfma_wgrad9_plan(a9, nz, &gx9, &gy9, 0) &&
        (size_t)gy9 * (size_t)a9.slab_stride * sizeof(float) <= ws_bytes) {
      a = a9;
      int rc = mfma_wgrad9_group_launch(ctx, 1, &a, &gx9, &gy9);
      *bias_done = a.want_bias != 0;
      return rc ? -1 : (int)gy9;
    }
  }
  if (wgrad3_enabled() && a.use_tr && a.zero != nullptr && wgrad3_geom(a)) {
    long tiles3 = (long)wgrad3_rows(a) * (a.Cin / 64) * (a.Cout / 128);
    int want = (int)wgrad_clamp_splits((512 + tiles3 - 1) / tiles3, a.M);
    if (want > nz) want = nz;
    a.m_chunk = ((a.M + want - 1) / want + 63) / 64 * 64;
    int nzz = cdiv(a.M, a.m_chunk);
    dim3 grid((unsigned)(tiles3 + (a.want_bias ? a.Cout / 128 : 0)), nzz);
    int rc = a.relu_in ? launch_wgrad3<true>(ctx, a, grid) : launch_wgrad3<false>(ctx, a, grid);
    *bias_done = a.want_bias != 0;
    return rc ? -1 : nzz;
  }
  static bool attr_set = false;
  size_t lds = (size_t)4 * 64 * WG_PITCH * sizeof(bf16_t);
  if (!attr_set) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  a.m_chunk = ((a.M + nz - 1) / nz + 63) / 64 * 64;
  int nzz = cdiv(a.M, a.m_chunk);
  dim3 grid(a.KH * a.KW * (a.Cin / 128) * (a.Cout / 128), nzz);
  static int wg_glds = -1;     // per-tap kernels: direct-to-LDS (default) or register-staged (RCGAN_WGRAD_IMPL=reg)
  if (wg_glds < 0) { const char* e = getenv("RCGAN_WGRAD_IMPL"); wg_glds = (e && e[0] == 'r') ? 0 : 1; }
  if (wg_glds == 1 && a.zero != nullptr) {
    constexpr int NS = 4;
    static bool attr2 = false;
    size_t lds2 = (size_t)NS * 2 * 32 * 256;
    if (!attr2) {
      RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_wgrad_glds_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      attr2 = true;
    }
    ProfScope ps(ctx, RCGAN_PROF_WGRAD_MFMA, 2.0 * (double)a.M * a.KH * a.KW * a.Cin * a.Cout);
    hipLaunchKernelGGL(conv_mfma_wgrad_glds_kernel<NS>, grid, dim3(256), lds2, ctx->stream, a);
    *bias_done = a.want_bias != 0;
  } else {
    ProfScope ps(ctx, RCGAN_PROF_WGRAD_MFMA, 2.0 * (double)a.M * a.KH * a.KW * a.Cin * a.Cout);
    hipLaunchKernelGGL(conv_mfma_wgrad_kernel, grid, dim3(256), lds, ctx->stream, a);
  }
  RC_LAUNCH_CHECK(ctx);
  return nzz;
}

// the three-tap kernel's plan for one problem (also what mfma_wgrad_launch does): grid and pixel chunk, or false
bool mfma_wgrad3_takes(const MfmaWgradArgs& a) {
  return wgrad3_enabled() && a.use_tr && a.zero != nullptr && wgrad3_geom(a);
}

bool mfma_wgrad3_plan(MfmaWgradArgs& a, int nz, unsigned* gx, unsigned* gy, long px_per_block) {
  if (!(wgrad3_enabled() && a.use_tr && a.zero != nullptr && wgrad3_geom(a))) return false;
  long tiles3 = (long)wgrad3_rows(a) * (a.Cin / 64) * (a.Cout / 128);
  // pixel chunks: alone, enough of them for ~512 workgroups; in a group the caller fixes the pixels per workgroup for all
  // layers (equal workgroup run times, and far fewer fp32 slabs to write and reduce than 512 workgroups per layer)
  int want = px_per_block > 0 ? (int)wgrad_clamp_splits(cdiv(a.M, px_per_block), a.M)
                              : (int)wgrad_clamp_splits((512 + tiles3 - 1) / tiles3, a.M);
  if (want > nz) want = nz;
  if (want < 1) want = 1;
  a.m_chunk = ((a.M + want - 1) / want + 63) / 64 * 64;
  *gx = (unsigned)(tiles3 + (a.want_bias ? a.Cout / 128 : 0));
  *gy = (unsigned)cdiv(a.M, a.m_chunk);
  return true;
}

template <bool RELU>
static int launch_wgrad3_group(rcgan_ctx* ctx, const WgradGroup& g) {
  constexpr int NS = 4;
  static bool attr = false;
  size_t lds = (size_t)NS * (40 * 128 + 32 * 256);
  if (!attr) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_wgrad3_group_kernel<NS, RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  double fl = 0, fx = 0;
  for (int p = 0; p < g.n; ++p) {
    fl += 2.0 * (double)g.a[p].M * wgrad3_alg_taps(g.a[p]) * g.a[p].Cin * g.a[p].Cout;
    fx += 2.0 * (double)g.a[p].M * wgrad3_exec_taps(g.a[p]) * g.a[p].Cin * g.a[p].Cout;
  }
  {
    ProfScope ps(ctx, RCGAN_PROF_WGRAD_MFMA, fl, fx);
    hipLaunchKernelGGL((conv_mfma_wgrad3_group_kernel<NS, RELU>), dim3(g.img.first[IMG_GROUP_MAX] + g.first[g.n] + g.head.blocks), dim3(256), lds, ctx->stream, g);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

static int launch_wgrad_glds_group(rcgan_ctx* ctx, const WgradGroup& g) {
  constexpr int NS = 4;
  static bool attr = false;
  size_t lds = (size_t)NS * 2 * 32 * 256;
  if (!attr) {
    RC_HIP(ctx, hipFuncSetAttribute((const void*)conv_mfma_wgrad_glds_group_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  double fl = 0;
  for (int p = 0; p < g.n; ++p) fl += 2.0 * (double)g.a[p].M * g.a[p].KH * g.a[p].KW * g.a[p].Cin * g.a[p].Cout;
  {
    ProfScope ps(ctx, RCGAN_PROF_WGRAD_MFMA, fl);
    hipLaunchKernelGGL(conv_mfma_wgrad_glds_group_kernel<NS>, dim3(g.img.first[IMG_GROUP_MAX] + g.first[g.n]), dim3(256), lds, ctx->stream, g);
  }
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// the per-tap kernel's plan for one problem inside a group (its own pixel chunking rule, as mfma_wgrad_launch)
bool mfma_wgrad_tap_plan(MfmaWgradArgs& a, int nz, unsigned* gx, unsigned* gy) {
  static int wg_glds = -1;
  if (wg_glds < 0) { const char* e = getenv("RCGAN_WGRAD_IMPL"); wg_glds = (e && e[0] == 'r') ? 0 : 1; }
  if (!(wg_glds == 1 && a.zero != nullptr && a.Cin % 128 == 0 && a.Cout % 128 == 0)) return false;
  a.m_chunk = ((a.M + nz - 1) / nz + 63) / 64 * 64;
  *gx = (unsigned)(a.KH * a.KW * (a.Cin / 128) * (a.Cout / 128));
  *gy = (unsigned)cdiv(a.M, a.m_chunk);
  return true;
}

// args[i] planned by mfma_wgrad3_plan (family 0; all with the same relu_in) or mfma_wgrad_tap_plan (family 1)
// img (optional): image-end problems that ride in the FIRST launch
int mfma_wgrad3_group_launch(rcgan_ctx* ctx, int n, const MfmaWgradArgs* args, const unsigned* gx, const unsigned* gy, int family,
                             const ImgWGroup* img, bool carry_head) {
  static_assert(ImgWGeom<128>::LDS <= 4 * (40 * 128 + 32 * 256), "image-end body needs more LDS than the three-tap kernel");
  static_assert((HEAD_MAX_V + 1) * HEAD_MAX_D * 4 <= 4 * (40 * 128 + 32 * 256), "head body needs more LDS than the three-tap kernel");
  for (int i0 = 0; i0 < n; i0 += WGRAD_GROUP_MAX) {
    WgradGroup g;
    static const int xcd_runs = env_int("RCGAN_WGRAD3_XCD", 1);
    g.xcd_runs = xcd_runs;
    g.head.blocks = 0;
    if (carry_head && family != 1 && i0 == 0) (void)head_take_wgrad(ctx, &g.head);
    if (img && i0 == 0) g.img = *img;
    else { g.img.n = 0; for (int q = 0; q <= IMG_GROUP_MAX; ++q) g.img.first[q] = 0; }
    g.n = (n - i0 < WGRAD_GROUP_MAX) ? n - i0 : WGRAD_GROUP_MAX;
    unsigned tot = 0;
    for (int p = 0; p < g.n; ++p) {
      g.a[p] = args[i0 + p];
      g.gx[p] = gx[i0 + p];
      g.first[p] = tot;
      tot += gx[i0 + p] * gy[i0 + p];
    }
    for (int p = g.n; p <= WGRAD_GROUP_MAX; ++p) g.first[p] = tot;
    for (int p = g.n; p < WGRAD_GROUP_MAX; ++p) { g.gx[p] = 1; g.a[p] = args[i0]; }
    int rc = family == 1 ? launch_wgrad_glds_group(ctx, g)
                         : (args[i0].relu_in ? launch_wgrad3_group<true>(ctx, g) : launch_wgrad3_group<false>(ctx, g));
    if (rc) return rc;
  }
  return RCGAN_OK;
}

int mfma_prepare_launch(rcgan_ctx* ctx, const float* w, const float* sigma, bf16_t* wt, bf16_t* wd, int T, int Cin, int Cout) {
  long total = (long)T * Cin * Cout;
  hipLaunchKernelGGL(conv_prepare_mfma_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, w, sigma, wt, wd, T, Cin, Cout);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

int direct_prepare_launch(rcgan_ctx* ctx, const float* w, const float* sigma, float* out, long total) {
  hipLaunchKernelGGL(conv_prepare_direct_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, w, sigma, out, total);
  RC_LAUNCH_CHECK(ctx);
  return RCGAN_OK;
}

// ---------------------------------------------------------------------------------------------
// self test of the fragment layouts this file assumes
// ---------------------------------------------------------------------------------------------
__global__ void selftest_kernel(int* result /* [0]=mfma mismatches, [1]=tr mismatches */) {
  __shared__ __attribute__((aligned(16))) bf16_t A[16 * 32];     // A[row][k]
  __shared__ __attribute__((aligned(16))) bf16_t B[16 * 32];     // B^T[col][k]
  __shared__ __attribute__((aligned(16))) bf16_t Tm[32 * 16];    // [row][col] for the transpose read
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) {
    int r = i / 32, k = i % 32;
    A[i] = f32_to_bf16((float)((r * 7 + k * 3) % 11) - 5.f);
    B[i] = f32_to_bf16((float)((r * 5 + k * 2 + 1) % 13) - 6.f);
    Tm[i] = f32_to_bf16((float)(i % 256));
  }
  __syncthreads();
  bf16x8_t af = *(const bf16x8_t*)(A + (lane & 15) * 32 + (lane >> 4) * 8);
  bf16x8_t bf = *(const bf16x8_t*)(B + (lane & 15) * 32 + (lane >> 4) * 8);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  acc = mfma16(af, bf, acc);
  int bad = 0;
  for (int r = 0; r < 4; ++r) {
    int row = (lane >> 4) * 4 + r, col = lane & 15;
    float ref = 0.f;
    for (int k = 0; k < 32; ++k) ref += bf16_to_f32(A[row * 32 + k]) * bf16_to_f32(B[col * 32 + k]);
    if (fabsf(ref - acc[r]) > 1e-3f) bad++;
  }
  if (bad) atomicAdd(&result[0], bad);
  // transpose read: 16-lane group g, lane i supplies &Tm[g*4 + (i>>2)][(i&3)*4]; expects column i, rows g*4+e
  const int g = lane >> 4, i = lane & 15;
  const bf16_t* p = Tm + (g * 4 + (i >> 2)) * 16 + (i & 3) * 4;
  s16x4_t tr = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  int badt = 0;
  for (int e = 0; e < 4; ++e) {
    bf16_t expect = Tm[(g * 4 + e) * 16 + i];
    if ((bf16_t)tr[e] != expect) badt++;
  }
  if (badt) atomicAdd(&result[1], badt);
}

int mfma_selftest(rcgan_ctx* ctx, int* host_result /* [2] */) {
  int* d;
  RC_HIP(ctx, hipMalloc(&d, 2 * sizeof(int)));
  RC_HIP(ctx, hipMemsetAsync(d, 0, 2 * sizeof(int), ctx->stream));
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, ctx->stream, d);
  RC_LAUNCH_CHECK(ctx);
  RC_HIP(ctx, hipMemcpyAsync(host_result, d, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  RC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  RC_HIP(ctx, hipFree(d));
  return RCGAN_OK;
}
