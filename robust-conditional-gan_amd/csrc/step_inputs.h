// The critic step's input work as ONE body that rides in the batched filter-preparation launch (conv_mfma.hip): dequantisation
// noise, image preprocessing, the 2x2 mean pool of the images for D.Block.1's shortcut, the zero-fill of the gradient slab and the
// advance of the random stream -- five launches of 4-7 us each that depend on the step's inputs only, not on its parameters, and sat
// at the head of every critic step's dependency chain (gan_resnet.py:548-551, 239-240, 346).  Values are those of the separate
// kernels bit for bit: noise[src] = quad src/4 of the stream (rcgan_rng_fill, kind 0), x = 2 (img / 256 - .5) + noise in fp32
// (rcgan_preprocess_cifar), the pool sums the STORED values in the order of meanpool2_fwd_kernel.
#pragma once
#include "rng.h"

struct StepInputsArgs {
  int n;                       // critic batch B
  int is16;                    // x / pooled are 16-bit (the build's half type), else fp32
  const int32_t* img;          // [n][3][32][32] pixel values 0..255
  void* x;                     // [2n][32][32][3]: rows [0, n) are written here, rows [n, 2n) (the fakes) only read for the pool
  void* pooled;                // [2n][16][16][3] or null
  float lo, hi;                // noise ~ U[lo, hi)
  uint64_t seed; uint64_t* state;
  unsigned* counter;           // arrival counter (self-resetting): the last workgroup advances the stream by n*3072/4 quads
  float* fill; size_t fill4;   // fill4 float4 of zeros at fill (or null)
  // optional: the step's fake batch is slice slice[0] of fakes [nslices][n][32][32][3] (the generator forwards of the iteration's
  // critic steps, evaluated as one pass): copied into rows [n, 2n) of x here, pooled from the source; the last workgroup moves
  // slice[0] on (mod nslices).  Null: the fakes are in x already.
  const void* fakes; unsigned* slice; int nslices;
};

__device__ __forceinline__ float si_round(float v, int is16) { return is16 ? bf16_to_f32(f32_to_bf16(v)) : v; }
__device__ __forceinline__ void si_store(void* p, size_t i, float v, int is16) {
  if (is16) ((bf16_t*)p)[i] = f32_to_bf16(v); else ((float*)p)[i] = v;
}
__device__ __forceinline__ float si_load(const void* p, size_t i, int is16) {
  return is16 ? bf16_to_f32(((const bf16_t*)p)[i]) : ((const float*)p)[i];
}

// bid / nb: this workgroup's index / the number of workgroups of the rider (256 threads each)
__device__ __forceinline__ void step_inputs_body(const StepInputsArgs& a, int bid, int nb) {
  const uint64_t base = a.state[0];
  const unsigned k = a.fakes ? a.slice[0] : 0u;
  const size_t esz = a.is16 ? 2 : 4;
  const char* const fsrc = a.fakes ? (const char*)a.fakes + (size_t)k * a.n * 3072 * esz : (const char*)a.x + (size_t)a.n * 3072 * esz;
  const size_t u_real = (size_t)a.n * 3 * 16 * 8;                 // (image, channel, row pair, column quad)
  const size_t u_fake = a.pooled ? (size_t)a.n * 16 * 16 * 3 : 0;   // pooled outputs of the fake half
  const size_t u_copy = a.fakes ? (size_t)a.n * 3072 * esz / 16 : 0;  // 16-byte pieces of the fake batch
  const size_t total = u_real + u_fake + a.fill4 + u_copy;
  for (size_t u = (size_t)bid * 256 + threadIdx.x; u < total; u += (size_t)nb * 256) {
    if (u < u_real) {
      const int t = (int)(u & 7), i = (int)((u >> 3) & 15);
      const size_t bc = u >> 7;
      const int ch = (int)(bc % 3), b = (int)(bc / 3);
      const size_t src0 = (size_t)b * 3072 + (size_t)ch * 1024 + (size_t)(2 * i) * 32 + 4 * t;      // CHW index = noise index
      float v[2][4];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        uint32_t w[4];
        philox4(base + (src0 + 32 * r) / 4, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), w);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float y = 2.f * ((float)a.img[src0 + 32 * r + k] / 256.f - .5f);
          y += philox_uniform(w[k], a.lo, a.hi);
          const int px = (2 * i + r) * 32 + 4 * t + k;
          si_store(a.x, ((size_t)b * 1024 + px) * 3 + ch, y, a.is16);
          v[r][k] = si_round(y, a.is16);
        }
      }
      if (a.pooled) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {      // add_n order of meanpool2_fwd_kernel: (0,0) + (1,0) + (0,1) + (1,1)
          const float m = (v[0][2 * jj] + v[1][2 * jj] + v[0][2 * jj + 1] + v[1][2 * jj + 1]) * 0.25f;
          si_store(a.pooled, (((size_t)b * 16 + i) * 16 + 2 * t + jj) * 3 + ch, m, a.is16);
        }
      }
    } else if (u < u_real + u_fake) {
      const size_t o = u - u_real;
      const int ch = (int)(o % 3);
      const size_t p = o / 3;
      const int x2 = (int)(p & 15), y2 = (int)((p >> 4) & 15);
      const size_t b = p >> 8;                                       // image of the fake batch
      const size_t s0 = ((b * 32 + 2 * y2) * 32 + 2 * x2) * 3 + ch;
      const float m = (si_load(fsrc, s0, a.is16) + si_load(fsrc, s0 + 96, a.is16) + si_load(fsrc, s0 + 3, a.is16) + si_load(fsrc, s0 + 99, a.is16)) * 0.25f;
      si_store(a.pooled, (size_t)a.n * 768 + o, m, a.is16);
    } else if (u < u_real + u_fake + a.fill4) {
      ((float4*)a.fill)[u - u_real - u_fake] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const size_t c = u - u_real - u_fake - a.fill4;
      ((uint4*)((char*)a.x + (size_t)a.n * 3072 * esz))[c] = ((const uint4*)fsrc)[c];
    }
  }
  // every workgroup read the stream offset above; the last one to arrive moves it on
  __syncthreads();
  if (threadIdx.x == 0) {
    // acq_rel: this workgroup's reads of state[0] / slice[0] (above, before the barrier) are ordered before its arrival, and the
    // last arriver's stores below after every other workgroup's arrival
    const unsigned prev = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (unsigned)nb - 1u) {
      a.state[0] = base + ((uint64_t)a.n * 3072 + 3) / 4;
      if (a.fakes) a.slice[0] = (k + 1u) % (unsigned)a.nslices;
      __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
