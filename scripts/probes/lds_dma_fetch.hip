// Probe: the same question as filter_fetch.hip for the LDS-DMA path (global_load_lds_dwordx4: every lane supplies its own global
// address, the wavefront's 1 KiB lands contiguously in LDS) -- how fast does a CU ingest L2-resident data as a function of what ONE
// instruction touches?
//   pattern 1:  8 rows x 128 B  -- a 64-deep K-tile row of the convolution kernels (filters [Cout][K] and NHWC pixels alike)
//   pattern 2:  4 rows x 256 B  -- a 128-deep K-tile row
//   pattern 3:  1 KiB contiguous -- a tile-major filter copy
//   pattern 4:  8 rows x 128 B with the kernels' XOR swizzle of the 16-byte slots inside a row (source side)
//   pattern 5:  1 KiB contiguous block, 16-byte slots permuted inside each 128-byte line (tile-major + swizzle)
// WPC workgroups of 256 threads per CU; every wavefront issues 72 loads per "layer", NS of them in flight (vmcnt), 8 layers.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void glds16(const void* g, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds) : "memory", "m0");
}

template <int PAT>
__global__ __launch_bounds__(256) void fetch(const char* w, int* sink, int layers) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long ROW = 2304;
  long off, step;
  const int sw = ((lane >> 3) & 7);
  if (PAT == 1) { off = (long)(wave * 32 + (lane >> 3)) * ROW + (lane & 7) * 16; step = 128; }
  else if (PAT == 2) { off = (long)(wave * 32 + (lane >> 4)) * ROW + (lane & 15) * 16; step = 256; }
  else if (PAT == 3) { off = (long)wave * 72 * 1024 + lane * 16; step = 1024; }
  else if (PAT == 4) { off = (long)(wave * 32 + (lane >> 3)) * ROW + ((lane & 7) ^ sw) * 16; step = 128; }
  else { off = (long)wave * 72 * 1024 + (lane >> 3) * 128 + ((lane & 7) ^ sw) * 16; step = 1024; }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * 8192;
  for (int L = 0; L < layers; ++L) {
    const char* base = w + (long)L * 128 * ROW + off;
#pragma unroll
    for (int s = 0; s < 72; ++s) {
      long o;
      if (PAT == 1 || PAT == 4) o = (long)(s % 18) * step + (s / 18) * 8 * ROW;
      else if (PAT == 2) o = (long)(s % 9) * step + (s / 9) * 4 * ROW;
      else o = (long)s * step;
      glds16(base + o, lds0 + (s & 7) * 1024);
      if ((s & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (((int*)smem)[threadIdx.x] == 0x12345678) sink[0] = 1;
}

int main() {
  const size_t bytes = (size_t)8 * 128 * 2304 + (1 << 20);
  char* w; int* sink;
  hipMalloc(&w, bytes); hipMalloc(&sink, 64);
  hipMemset(w, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc : {1, 2, 4}) {
    for (int pat = 1; pat <= 5; ++pat) {
      const int wgs = 256 * wpc;
      auto run = [&]() {
        if (pat == 1) hipLaunchKernelGGL(fetch<1>, dim3(wgs), dim3(256), 32768, 0, w, sink, 8);
        if (pat == 2) hipLaunchKernelGGL(fetch<2>, dim3(wgs), dim3(256), 32768, 0, w, sink, 8);
        if (pat == 3) hipLaunchKernelGGL(fetch<3>, dim3(wgs), dim3(256), 32768, 0, w, sink, 8);
        if (pat == 4) hipLaunchKernelGGL(fetch<4>, dim3(wgs), dim3(256), 32768, 0, w, sink, 8);
        if (pat == 5) hipLaunchKernelGGL(fetch<5>, dim3(wgs), dim3(256), 32768, 0, w, sink, 8);
      };
      run(); run();
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) run();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / 20;
      printf("workgroups per CU %d  pattern %d  %7.1f us per launch  (%6.1f GB/s per CU)\n", wpc, pat, us, wpc * 8 * 294912.0 / us / 1e3);
    }
  }
  return 0;
}
