// Stand-alone probe of the inline-asm LDS-DMA primitive (global_load_lds_dwordx4 with hand-set M0).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ void glds16_asm(const void* gptr, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void glds16_bi(const void* gptr, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int MODE>
__global__ void k(const unsigned* in, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // 4 waves x 2 deposits of 1 KiB; lane reads a permuted source so we can see the lane->slot mapping
  for (int i = 0; i < 2; ++i) {
    const unsigned* src = in + ((wave * 2 + i) * 64 + (lane ^ 1)) * 4;
    if (MODE == 0) glds16_asm(src, lds0 + (wave * 2 + i) * 1024);
    else glds16_bi(src, smem + (wave * 2 + i) * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < 2048; i += 256) out[i] = ((const unsigned*)smem)[i];
}
int main() {
  std::vector<unsigned> h(2048), o(2048);
  for (int i = 0; i < 2048; ++i) h[i] = i;
  unsigned *d, *e;
  hipMalloc(&d, 8192); hipMalloc(&e, 8192);
  hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(e, 0xff, 8192);
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 8192, 0, d, e);
    else hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 8192, 0, d, e);
    hipMemcpy(o.data(), e, 8192, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 2048; ++i) {
      int dep = i / 256, w = (i % 256) / 4, j = i % 4;       // deposit, slot(lane), word
      unsigned expect = (dep * 64 + (w ^ 1)) * 4 + j;
      if (o[i] != expect) { if (bad < 6) printf("mode %d: out[%d]=%u expect %u\n", mode, i, o[i], expect); bad++; }
    }
    printf("mode %d (%s): %d mismatches\n", mode, mode == 0 ? "asm" : "builtin", bad);
  }
  return 0;
}
