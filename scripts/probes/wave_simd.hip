// Which wavefronts of a 512-thread workgroup share a SIMD?  Every wavefront records its HW_ID (gfx9: WAVE_ID[3:0], SIMD_ID[5:4],
// CU_ID[11:8]); the answer decides which wavefronts may "ping-pong" on a matrix pipe (conv_mfma8.hip, PP).
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/wave_simd.hip -o scripts/probes/_bin/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 64 * 16 * 4);
  for (int threads : {256, 512, 1024}) {
    hipMemset(d, 0, 64 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(4), dim3(threads), 0, 0, d);
    unsigned h[64 * 16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; ++b) {
      printf("threads %4d block %d: wave -> simd:", threads, b);
      for (int w = 0; w < threads / 64; ++w) printf(" %d->%u", w, (h[b * 16 + w] >> 4) & 3);
      printf("   (cu %u)\n", (h[b * 16] >> 8) & 15);
    }
  }
  return 0;
}
