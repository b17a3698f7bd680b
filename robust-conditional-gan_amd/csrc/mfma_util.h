// Device helpers shared by the MFMA kernels (conv_mfma.hip, conv_image.hip): vector types, the packed-max ReLU,
// asm-issued LDS-DMA with hand-placed waits, the gfx950 LDS transpose read.
#pragma once
#include "common.h"

// 16-lane (one DPP row) all-reduce in a fixed order: pairs, quads, half rows, rows
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}


#if RCGAN_HALF_FP16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8_t;      // (historical name: eight 16-bit operand elements)
#define H16_ONE 0x3C00u                                             /* 1.0 */
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
#define H16_ONE 0x3F80u
#endif
#define H16_ONE_X2 (H16_ONE | (H16_ONE << 16))
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

// D = A(16x32) * B(32x16) + C on the matrix cores in the build's 16-bit format, fp32 accumulate
__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
#if RCGAN_HALF_FP16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}


// packed signed-16 max on bf16 bit patterns (v_pk_max_i16): with bound 0 this is ReLU (every negative bf16,
// -0 included, has the int16 sign bit set); with bound 0x8000 per half it is the identity.
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t w, uint32_t bound) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, w), __builtin_bit_cast(s16x2_t, bound)));
}
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t w) { return pk_max_i16(w, 0u); }


// NS LDS stages: tile t+NS-1 is requested while tile t is multiplied; a counted s_waitcnt vmcnt leaves the
// NS-2 newest tiles in flight across the (raw) barrier, so small-grid layers are not serialised on one
// HBM/L2 round trip per K-tile.
// LDS-DMA issued from inline asm: hipcc then neither counts it in its own s_waitcnt bookkeeping nor drains it
// (vmcnt(0)) in front of every LDS read that might alias the destination -- the waits are placed by hand.
// M0 (LDS destination base) is written in the same statement that uses it and restored afterwards.
__device__ __forceinline__ void glds16_asm(const void* gptr, unsigned lds_byte_addr /* wave-uniform */) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(lds_byte_addr) : "memory");
}

// The same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: the address of a burst needs no vector ALU work in the
// loop -- a stream whose lanes keep their offsets advances with one scalar add per step (glds16_asm's callers add a 64-bit per-lane pointer
// per burst); every VALU instruction a "loading" wavefront issues comes out of the matrix pipe's time of the wavefront it shares the SIMD with.
__device__ __forceinline__ void glds16_sbase(const void* sbase /* wave-uniform */, unsigned voff, unsigned lds_byte_addr /* wave-uniform */) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N == 0, "unsupported vmcnt immediate");
}


template <int N> __device__ __forceinline__ void wait_vmcnt_any() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else static_assert(N == 0, "unsupported vmcnt immediate");
}

__device__ __forceinline__ bf16x8_t tr_pair(const unsigned char* p, int hi_off) {
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + hi_off));
  s16x8_t r = (s16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}



// ---------------------------------------------------------------------------------------------------------
// Shared epilogue of the MFMA convolution kernels: a wavefront's accumulators acc[NI][NJ] (NI channel fragments of 16,
// NJ pixel fragments of 16; lane holds pixel lane&15, channels 4*(lane>>4) .. +3 of every fragment) -> bias, optional
// ReLU-backward mask / accumulate / residual, 16-bit rounding, store to out[m][Cout].
//
// Why it looks like this (scripts/probes/epilogue_store.hip, MI355X, 256 x 256 tile per CU):
//   * one 8-byte store per fragment is bound by store ISSUE at ~7.5 B/clk/CU (8.1 us per tile); pairing two channel
//     fragments with v_permlane16_swap gives every lane 8 consecutive channels = one 16-byte store (5.3 us: 6.3 TB/s,
//     the chip's write bandwidth); non-temporal stores are slower.
//   * the per-fragment `if (bias) load; if (mask) load; ...` form this replaces made every fragment wait vmcnt(0) for its
//     own loads AND for the previous fragment's store (CDNA4 counts stores in vmcnt): 32 dependent memory round trips
//     per wavefront.  Here the bias is read once, and the extra operands of pixel row j+1 are requested BEFORE row j is
//     stored, so a wait for them never covers a store younger than one row.
// The swap: rows (16-lane groups) 1,3 of fragment i trade places with rows 0,2 of fragment i+1; afterwards row r holds
// channels (r>>1)*8 .. +7 of fragment i + (r&1) -- x[] the first four, y[] the last four.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void swap16(float& a, float& b) {
  auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  // (through named scalars: __builtin_bit_cast applied to the vector ELEMENT r[1] reads element 0 with this clang)
  const unsigned r0 = r[0], r1 = r[1];
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}

__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ float h16_lo(uint32_t w) { return bf16_to_f32((bf16_t)(w & 0xffffu)); }
__device__ __forceinline__ float h16_hi(uint32_t w) { return bf16_to_f32((bf16_t)(w >> 16)); }

// Row maps: the pixel index a tile row stands for -> the row of the output tensor.  RowIdent for every ordinary convolution;
// RowPhase for the sub-pixel form of a 3x3 convolution behind a nearest 2x upsample (conv_mfma8.hip): there the tile rows
// enumerate (phase, n, i, j) over the LOW-resolution grid and land at output pixel (n, 2i + ph, 2j + pw).
struct RowIdent { __device__ __forceinline__ long operator()(long m) const { return m; } };
struct RowPhase {
  int on, lws, lhs, ph, pw; long base;          // base = phase * (pixels per phase); on = 0: identity
  __device__ __forceinline__ long operator()(long m) const {
    if (!on) return m;
    const unsigned ms = (unsigned)(m - base);
    const unsigned j = ms & ((1u << lws) - 1u), i = (ms >> lws) & ((1u << lhs) - 1u), n = ms >> (lws + lhs);
    return (long)(((((n << lhs) + i) * 2u + (unsigned)ph) << (lws + 1)) + 2u * j + (unsigned)pw);
  }
};

// STATS: also accumulate, per lane, the column sums of the STORED values (after the 16-bit rounding) and of their squares over
// the wavefront's pixel rows -- st[p][e] / st[p][8 + e] for channel co_wave + (2p + (row & 1)) * 16 + (row >> 1) * 8 + e, row =
// lane >> 4 -- from which the producing kernel builds per-tile batch-norm statistics (conv_mfma8.hip).
template <int NI, int NJ, typename Map = RowIdent, bool STATS = false>
__device__ __forceinline__ void conv_epilogue(f32x4_t (&acc)[NI][NJ], const float* __restrict__ bias, const bf16_t* mask,
                                              const bf16_t* resid, bf16_t* out, int accumulate, long M, int Cout,
                                              long m_wave /* first pixel of the wavefront's rows */,
                                              int co_wave /* first channel of the wavefront's columns */, int lane, Map rowmap = Map(),
                                              int res_lw = -1 /* >= 1: resid lives on the half-resolution grid of a 2^res_lh x 2^res_lw image */,
                                              int res_lh = 0, float (*st)[16] = nullptr) {
  static_assert(NI % 2 == 0, "channel fragments are stored in pairs");
  constexpr int NP = NI / 2;
  const int row = lane >> 4;
  float4 b4[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
    b4[i] = bias ? *(const float4*)(bias + co_wave + i * 16 + row * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  // element offset of this lane's 8-channel run of pair p in pixel row j
  auto offs = [&](int j, int p) __attribute__((always_inline)) -> long {
    return rowmap(m_wave + j * 16 + (lane & 15)) * Cout + co_wave + (2 * p + (row & 1)) * 16 + (row >> 1) * 8;
  };
  auto valid = [&](int j) __attribute__((always_inline)) -> bool { return m_wave + j * 16 + (lane & 15) < M; };
  const bool extras = mask != nullptr || resid != nullptr || accumulate != 0;
  uint4 e_mask[NP], e_acc[NP], e_res[NP];
  auto fetch = [&](int j) __attribute__((always_inline)) {
    if (!valid(j)) return;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const long o = offs(j, p);
      if (mask) e_mask[p] = *(const uint4*)(mask + o);
      if (accumulate) e_acc[p] = *(const uint4*)(out + o);
      if (resid) {
        long ro = o;
        if (res_lw >= 1) {      // output pixel (n, oh, ow) reads the residual's pixel (n, oh/2, ow/2)
          const unsigned r = (unsigned)rowmap(m_wave + j * 16 + (lane & 15));
          const unsigned ow = r & ((1u << res_lw) - 1u), oh = (r >> res_lw) & ((1u << res_lh) - 1u), n = r >> (res_lw + res_lh);
          const unsigned rl = (((n << (res_lh - 1)) + (oh >> 1)) << (res_lw - 1)) + (ow >> 1);
          ro = o + ((long)rl - (long)r) * Cout;
        }
        e_res[p] = *(const uint4*)(resid + ro);
      }
    }
  };
  if (extras) fetch(0);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    uint4 c_mask[NP], c_acc[NP], c_res[NP];
    if (extras) {
#pragma unroll
      for (int p = 0; p < NP; ++p) { c_mask[p] = e_mask[p]; c_acc[p] = e_acc[p]; c_res[p] = e_res[p]; }
      if (j + 1 < NJ) fetch(j + 1);            // requested before row j is stored (see above)
    }
    const bool ok = valid(j);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      float x[4], y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[e] = acc[2 * p][j][e] + (&b4[2 * p].x)[e];
        y[e] = acc[2 * p + 1][j][e] + (&b4[2 * p + 1].x)[e];
        swap16(x[e], y[e]);
      }
      if (extras) {
        if (mask) {
          const uint32_t mw[4] = {c_mask[p].x, c_mask[p].y, c_mask[p].z, c_mask[p].w};
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (!(h16_lo(mw[q]) > 0.f)) x[2 * q] = 0.f;
            if (!(h16_hi(mw[q]) > 0.f)) x[2 * q + 1] = 0.f;
            if (!(h16_lo(mw[2 + q]) > 0.f)) y[2 * q] = 0.f;
            if (!(h16_hi(mw[2 + q]) > 0.f)) y[2 * q + 1] = 0.f;
          }
        }
        if (accumulate) {
          const uint32_t w[4] = {c_acc[p].x, c_acc[p].y, c_acc[p].z, c_acc[p].w};
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            x[2 * q] += h16_lo(w[q]); x[2 * q + 1] += h16_hi(w[q]);
            y[2 * q] += h16_lo(w[2 + q]); y[2 * q + 1] += h16_hi(w[2 + q]);
          }
        }
        if (resid) {
          const uint32_t w[4] = {c_res[p].x, c_res[p].y, c_res[p].z, c_res[p].w};
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            x[2 * q] += h16_lo(w[q]); x[2 * q + 1] += h16_hi(w[q]);
            y[2 * q] += h16_lo(w[2 + q]); y[2 * q + 1] += h16_hi(w[2 + q]);
          }
        }
      }
      const uint4 pk = make_uint4(pack_h16x2(x[0], x[1]), pack_h16x2(x[2], x[3]), pack_h16x2(y[0], y[1]), pack_h16x2(y[2], y[3]));
      if (ok) *(uint4*)(out + offs(j, p)) = pk;
      if (STATS && ok) {
        const uint32_t w[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = h16_lo(w[q]), hi = h16_hi(w[q]);
          st[p][2 * q] += lo; st[p][2 * q + 1] += hi;
          st[p][8 + 2 * q] += lo * lo; st[p][8 + 2 * q + 1] += hi * hi;
        }
      }
    }
  }
}

// ---- prepared-filter layouts of the image-end kernels (conv_image.hip; also filled by the batched prepare) ----
// element e of the extra region: wK[Cb][32] followed by wS[T][16][Cb]   (side 1: cin small, side 2: cout small)
//   side 1:  wK[n][t*Cs+c] = W[t][c][n]        (forward)        wS[t][j][co] = W[T-1-t][j][co]   (dX from dY)
//   side 2:  wK[ci][t*Cs+co] = W[T-1-t][ci][co] (dX from dY)    wS[t][j][ci] = W[t][ci][j]       (forward)
__device__ __forceinline__ bf16_t img_prepare_elem(long e, int side, int T, int Cin, int Cout, const float* w, float inv) {
  const int Cs = side == 1 ? Cin : Cout, Cb = side == 1 ? Cout : Cin;
  float v = 0.f;
  if (e < (long)Cb * 32) {
    const int n = (int)(e >> 5), k = (int)(e & 31);
    if (k < T * Cs) {
      const int t = k / Cs, c = k - t * Cs;
      v = side == 1 ? w[((long)t * Cin + c) * Cout + n] : w[((long)(T - 1 - t) * Cin + n) * Cout + c];
    }
  } else {
    long r = e - (long)Cb * 32;
    const int cbi = (int)(r % Cb); r /= Cb;
    const int j = (int)(r & 15), t = (int)(r >> 4);
    if (j < Cs) v = side == 1 ? w[((long)(T - 1 - t) * Cin + j) * Cout + cbi] : w[((long)t * Cin + cbi) * Cout + j];
  }
  return f32_to_bf16(v * inv);
}

