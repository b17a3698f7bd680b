// Probe: how fast does ONE workgroup per CU stream a filter of 128 rows x 1152 16-bit elements (295 KB, L2-resident after the first
// pass) into registers, as a function of what ONE load instruction of a wavefront touches?
//   pattern 0: 16 rows x 64 B   (lane l: row l & 15, bytes (l >> 4) * 16 of the K-step)         -- the MFMA A-operand fetch from row-major filters
//   pattern 1:  8 rows x 128 B  (lane l: row l >> 3, bytes (l & 7) * 16)                        -- an LDS-DMA tile row of the convolution kernels
//   pattern 2:  4 rows x 256 B
//   pattern 3:  1 KiB contiguous                                                                -- fragment-major / tile-major layouts
// Every wavefront issues 72 x dwordx4 loads per "layer" (its quarter of the filter), all in flight together, 8 layers per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int PAT>
__global__ __launch_bounds__(256) void fetch(const char* w, int* sink, int layers) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long off;       // byte offset of this lane inside the wavefront's first KiB-equivalent
  long step;      // bytes from one load instruction to the next
  const long ROW = 2304;
  if (PAT == 0) { off = (long)(wave * 32 + (lane & 15)) * ROW + (lane >> 4) * 16; step = 64; }
  else if (PAT == 1) { off = (long)(wave * 32 + (lane >> 3)) * ROW + (lane & 7) * 16; step = 128; }
  else if (PAT == 2) { off = (long)(wave * 32 + (lane >> 4)) * ROW + (lane & 15) * 16; step = 256; }
  else { off = (long)wave * 72 * 1024 + lane * 16; step = 1024; }
  i32x4 acc = {0, 0, 0, 0};
  for (int L = 0; L < layers; ++L) {
    const char* base = w + (long)L * 128 * ROW + off;
    i32x4 v[72];
#pragma unroll
    for (int s = 0; s < 72; ++s) {
      long o;
      if (PAT == 0) o = (long)(s >> 1) * step + (s & 1) * 16 * ROW;                  // 36 K-steps x 2 channel tiles of 16 rows
      else if (PAT == 1) o = (long)(s % 18) * step + (s / 18) * 8 * ROW;             // 18 K-tiles x 4 groups of 8 rows
      else if (PAT == 2) o = (long)(s % 9) * step + (s / 9) * 4 * ROW;               // 9 x 8 groups of 4 rows
      else o = (long)s * step;
      v[s] = *(const i32x4*)(base + o);
    }
#pragma unroll
    for (int s = 0; s < 72; ++s) acc += v[s];
  }
  if (acc.x == 0x12345678) sink[0] = acc.y + acc.z + acc.w;
}

int main() {
  const size_t bytes = (size_t)8 * 128 * 2304 + (1 << 20);
  char* w; int* sink;
  hipMalloc(&w, bytes); hipMalloc(&sink, 64);
  hipMemset(w, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs : {8, 128, 256}) {
    for (int pat = 0; pat < 4; ++pat) {
      auto run = [&]() {
        if (pat == 0) hipLaunchKernelGGL(fetch<0>, dim3(wgs), dim3(256), 0, 0, w, sink, 8);
        if (pat == 1) hipLaunchKernelGGL(fetch<1>, dim3(wgs), dim3(256), 0, 0, w, sink, 8);
        if (pat == 2) hipLaunchKernelGGL(fetch<2>, dim3(wgs), dim3(256), 0, 0, w, sink, 8);
        if (pat == 3) hipLaunchKernelGGL(fetch<3>, dim3(wgs), dim3(256), 0, 0, w, sink, 8);
      };
      run(); run();
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) run();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / 20;
      printf("workgroups %3d  pattern %d  %7.1f us per launch  (%6.1f GB/s per CU, 8 x 295 KB each)\n", wgs, pat, us, 8 * 294912.0 / us / 1e3);
    }
  }
  return 0;
}
