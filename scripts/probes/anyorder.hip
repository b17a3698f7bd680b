// Probe: does a kernel launched with hipExtAnyOrderLaunch (AQL packet without the barrier bit) overlap with the kernel in front
// of it on the SAME stream on gfx950, does the next ordinary launch wait for both, and does the flag survive stream capture?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>

__global__ void spin(unsigned long long ticks, int* out, int val) {
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0 && out) out[blockIdx.x] = val;
}
__global__ void check(const int* in, int n, int want, int* bad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && in[i] != want) atomicAdd(bad, 1);
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now().time_since_epoch()).count();
}

int main() {
  int *a, *b, *bad;
  hipMalloc(&a, 4096 * 4); hipMalloc(&b, 4096 * 4); hipMalloc(&bad, 4);
  hipMemset(bad, 0, 4);
  hipStream_t s; hipStreamCreate(&s);
  const unsigned long long T = 5000;   // wall_clock64 ticks at 100 MHz: 50 us
  const int N = 200;
  auto launch = [&](int* out, int val, int blocks, unsigned flags) {
    hipExtLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s, nullptr, nullptr, flags, T, out, val);
  };
  for (int blocks : {64, 256, 1024}) {
    for (int mode = 0; mode < 3; ++mode) {
      // mode 0: A, B both ordinary; mode 1: B any-order; mode 2: same as 1 inside a captured graph
      hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
      auto body = [&]() {
        for (int i = 0; i < N; ++i) {
          launch(a, i, blocks, 0);
          launch(b, i, blocks, mode ? hipExtAnyOrderLaunch : 0);
          hipLaunchKernelGGL(check, dim3((blocks + 255) / 256), dim3(256), 0, s, a, blocks, i, bad);
          hipLaunchKernelGGL(check, dim3((blocks + 255) / 256), dim3(256), 0, s, b, blocks, i, bad);
        }
      };
      if (mode == 2) {
        hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        body();
        hipError_t e2 = hipStreamEndCapture(s, &g);
        if (e != hipSuccess || e2 != hipSuccess || !g) { printf("blocks %d: capture failed (%d %d)\n", blocks, e, e2); (void)hipGetLastError(); continue; }
        if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); continue; }
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
      } else { body(); hipStreamSynchronize(s); }
      double t0 = now_us();
      if (mode == 2) hipGraphLaunch(ge, s); else body();
      hipStreamSynchronize(s);
      double us = (now_us() - t0) / N;
      int hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
      printf("blocks %4d mode %d (%s): %.1f us per (A, B, check, check)   [one spin = 50 us]   order violations %d\n", blocks, mode,
             mode == 0 ? "ordinary" : mode == 1 ? "B any-order, eager" : "B any-order, graph", us, hb);
    }
  }
  return 0;
}
