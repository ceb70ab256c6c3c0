// hipGraphLaunch of a LINEAR chain of small kernels against launching them one by one (development aid).
// Build: hipcc --offload-arch=gfx950 -O2 -w -o tools/bin/graphbench tools/graphbench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_small(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
  float *d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int nk : {8, 13, 26}) {
    auto chain = [&]() { for (int k = 0; k < nk; k++) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, d, 16384); };
    for (int i = 0; i < 50; i++) chain();
    hipStreamSynchronize(s);
    const int reps = 2000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) chain();
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    auto t2 = std::chrono::steady_clock::now();
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); chain(); hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 50; i++) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    auto t3 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) hipGraphLaunch(ge, s);
    auto t4 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    auto t5 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    printf("%2d kernels: plain launches  enqueue %.1f us  total %.1f us per chain | graph  enqueue %.1f us  total %.1f us per replay\n",
           nk, us(t0, t1) / reps, us(t0, t2) / reps, us(t3, t4) / reps, us(t3, t5) / reps);
  }
  return 0;
}
