// Probe: per-kernel cost of back-to-back dependent launches (stream vs hipGraph replay) on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void tiny(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1.f;
}
int main() {
  float* p; hipMalloc(&p, 1 << 24); hipMemset(p, 0, 1 << 24);
  hipStream_t s; hipStreamCreate(&s);
  const int N = 2000;
  for (int blocks : {1, 256, 4096}) {
    int n = blocks * 256;
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, p, n);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, p, n);
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / N;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, p, n);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    t0 = std::chrono::high_resolution_clock::now();
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    double usg = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / N;
    printf("blocks %5d: stream %.2f us/kernel, graph %.2f us/kernel\n", blocks, us, usg);
  }
  return 0;
}
