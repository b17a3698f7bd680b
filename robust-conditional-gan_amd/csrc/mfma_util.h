// Device helpers shared by the MFMA kernels (conv_mfma.hip, conv_image.hip): vector types, the packed-max ReLU,
// asm-issued LDS-DMA with hand-placed waits, the gfx950 LDS transpose read.
#pragma once
#include "common.h"

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

