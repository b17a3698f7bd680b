// Probe: how fast can one workgroup per CU (512 threads) store a 256 x 256 bf16 output tile that sits in MFMA accumulator
// layout?  (The 256x256 conv kernel pays 12-16 us per round outside its K loop; this isolates the store part.)
//   V0  one 8-byte store per (pixel fragment, channel fragment): 16 rows x 32 B per instruction   (the kernel's epilogue)
//   V1  v_permlane16_swap pairs two channel fragments: 16-byte stores, 16 rows x 64 B per instruction
//   V2  tile staged through LDS (XOR-swizzled), written back as whole 512-byte rows: 2 rows x 512 B per instruction
//   V3/V4/V5 = V0/V1/V2 with non-temporal stores
// usage: ./epilogue_store        (prints us per round of G workgroups and TB/s, for G = 256 and G = 32)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef unsigned short bf16_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16_t f2b(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

template <int V>
__global__ __launch_bounds__(512) void store_kernel(bf16_t* out, int tiles_per_wg, int Cout, float seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  constexpr bool NT = V >= 3;
  constexpr int M = V % 3;
  for (int t = 0; t < tiles_per_wg; ++t) {
    const long m0 = ((long)t * gridDim.x + blockIdx.x) * 256;
    float acc[4][8][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int p = wm * 128 + j * 16 + (lane & 15), co = wn * 64 + i * 16 + (lane >> 4) * 4 + e;
          acc[i][j][e] = seed * (float)((m0 + p) & 1023) + (float)co;     // value = f(pixel, channel): checkable
        }
    if (M == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long m = m0 + wm * 128 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int co = wn * 64 + i * 16 + (lane >> 4) * 4;
          uint2 pk;
          pk.x = (uint32_t)f2b(acc[i][j][0]) | ((uint32_t)f2b(acc[i][j][1]) << 16);
          pk.y = (uint32_t)f2b(acc[i][j][2]) | ((uint32_t)f2b(acc[i][j][3]) << 16);
          uint2* dst = (uint2*)(out + m * Cout + co);
          if (NT) __builtin_nontemporal_store((u32x2){pk.x, pk.y}, (u32x2*)dst); else *dst = pk;
        }
      }
    } else if (M == 1) {
      const int row = lane >> 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long m = m0 + wm * 128 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          uint32_t ax = (uint32_t)f2b(acc[i][j][0]) | ((uint32_t)f2b(acc[i][j][1]) << 16);
          uint32_t ay = (uint32_t)f2b(acc[i][j][2]) | ((uint32_t)f2b(acc[i][j][3]) << 16);
          uint32_t bx = (uint32_t)f2b(acc[i + 1][j][0]) | ((uint32_t)f2b(acc[i + 1][j][1]) << 16);
          uint32_t by = (uint32_t)f2b(acc[i + 1][j][2]) | ((uint32_t)f2b(acc[i + 1][j][3]) << 16);
          // odd 16-lane rows of a <-> even rows of b: afterwards (a, b) of a lane are 8 consecutive channels
          auto rx = __builtin_amdgcn_permlane16_swap(ax, bx, false, false);
          auto ry = __builtin_amdgcn_permlane16_swap(ay, by, false, false);
          uint4 pk = make_uint4(rx[0], ry[0], rx[1], ry[1]);
          const int co = wn * 64 + (i + (row & 1)) * 16 + (row >> 1) * 8;
          uint4* dst = (uint4*)(out + m * Cout + co);
          if (NT) __builtin_nontemporal_store((u32x4){pk.x, pk.y, pk.z, pk.w}, (u32x4*)dst); else *dst = pk;
        }
      }
    } else {
      // stage: pixel p, channel pair-of-pairs -> smem[p * 512 + ((slot ^ (p & 15)) within each 128-byte window) * 8]
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int p = wm * 128 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int co = wn * 64 + i * 16 + (lane >> 4) * 4;
          uint2 pk;
          pk.x = (uint32_t)f2b(acc[i][j][0]) | ((uint32_t)f2b(acc[i][j][1]) << 16);
          pk.y = (uint32_t)f2b(acc[i][j][2]) | ((uint32_t)f2b(acc[i][j][3]) << 16);
          const int slot = co >> 2;                           // 8-byte slot index in the row (64 per row)
          const int sw = (slot & ~15) | ((slot ^ p) & 15);
          *(uint2*)(smem + p * 512 + sw * 8) = pk;
        }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = wave * 32 + r * 2 + (lane >> 5);
        const int c = lane & 31;                              // 16-byte chunk of the row = slots 2c, 2c+1
        const int s0 = (2 * c & ~15) | ((2 * c ^ p) & 15);    // where slot 2c went; its partner is s0 ^ 1
        uint4 v = *(const uint4*)(smem + p * 512 + (s0 & ~1) * 8);
        if (p & 1) v = make_uint4(v.z, v.w, v.x, v.y);
        uint4* dst = (uint4*)(out + (m0 + p) * Cout + c * 8);
        if (NT) __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, (u32x4*)dst); else *dst = v;
      }
    }
  }
}

template <int V>
static void run(bf16_t* out, int G, int tiles, hipStream_t s, bool check) {
  const size_t lds = (V % 3 == 2) ? 131072 : 0;
  hipFuncSetAttribute((const void*)store_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<V>, dim3(G), dim3(512), lds, s, out, tiles, 256, 0.25f);
  const int reps = 20;
  hipEventRecord(e0, s);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_kernel<V>, dim3(G), dim3(512), lds, s, out, tiles, 256, 0.25f);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, bytes = (double)G * tiles * 131072;
  int bad = 0;
  if (check) {
    std::vector<bf16_t> h((size_t)G * tiles * 65536);
    hipMemcpy(h.data(), out, h.size() * 2, hipMemcpyDeviceToHost);
    for (size_t idx = 0; idx < h.size(); idx += 997) {
      const long m = idx / 256; const int co = idx % 256;
      float f = 0.25f * (float)(m & 1023) + (float)co;
      uint32_t u; std::memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u);
      if (h[idx] != (bf16_t)(u >> 16)) ++bad;
    }
  }
  printf("V%d G=%3d tiles/wg=%d: %8.1f us per launch, %6.2f us per round, %5.2f TB/s%s\n", V, G, tiles, us, us / tiles, bytes / us / 1e6,
         check ? (bad ? "  MISMATCH" : "  ok") : "");
}

int main() {
  bf16_t* out; hipMalloc(&out, (size_t)256 * 5 * 131072);
  hipStream_t s; hipStreamCreate(&s);
  for (int G : {256, 32})
    for (int tiles : {1, 5}) {
      run<0>(out, G, tiles, s, true); run<1>(out, G, tiles, s, true); run<2>(out, G, tiles, s, true);
      run<3>(out, G, tiles, s, false); run<4>(out, G, tiles, s, false); run<5>(out, G, tiles, s, false);
    }
  return 0;
}
