// Host cost of a kernel launch as a function of the by-value argument size (development aid).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N> struct Blob { char b[N]; };
template <int N> __global__ void k(Blob<N> a, int *out) { if (out && threadIdx.x == 1000) *out = a.b[0]; }
template <int N> double run(int reps) {
  Blob<N> a{};
  for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, a, (int *)nullptr);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, a, (int *)nullptr);
  auto t1 = std::chrono::steady_clock::now();
  hipDeviceSynchronize();
  auto t2 = std::chrono::steady_clock::now();
  printf("arg %5d B: host %.2f us per launch, %.2f us per launch until idle\n", N,
         std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
         std::chrono::duration<double, std::micro>(t2 - t0).count() / reps);
  return 0;
}
int main() { run<8>(2000); run<256>(2000); run<1480>(2000); run<1800>(2000); run<3900>(2000); return 0; }
