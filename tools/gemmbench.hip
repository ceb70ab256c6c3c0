// Development aid: k_gemm_p (ao_marl_amd/csrc/aomarl_gemm_p.h) alone on the loop's product shapes, every tile of
// the menu x a range of k splits, with the slab reduce a consumer would run, checked against a float64 host
// product on sampled entries.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/bin/gemmbench tools/gemmbench.hip
//   tools/bin/gemmbench [all|pick]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../ao_marl_amd/csrc/aomarl_gemm_p.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_reduce(int M, int N, int nz, float alpha, const float *__restrict__ P, float *__restrict__ C, int ldc) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * N) return;
  const int row = (int)(i / N), col = (int)(i - (long long)row * N);
  float s = 0.f;
  for (int z = 0; z < nz; z++) s += P[(long long)z * M * N + i];
  C[(long long)row * ldc + col] = alpha * s;
}

struct Shape { const char *name; int M, N, K; };

static float frand() { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; }

int main(int argc, char **argv) {
  const bool all = argc > 1 && !strcmp(argv[1], "all");
  const Shape shapes[] = {
      {"do_control (bench sys)", 256, 1286, 2400}, {"v2m", 256, 1283, 1286}, {"m2v", 256, 1286, 1283},
      {"extrusion 3 layers", 768, 648, 1957},      {"extrusion 2 layers", 512, 648, 1957},
      {"extrusion 1 layer", 256, 648, 1957},       {"reset half (128 envs x 3 layers)", 384, 648, 1957},       {"do_control (unfiltered)", 256, 1430, 2400},
      {"v2m (unfiltered)", 256, 1427, 1430},       {"ragged", 250, 1285, 1283}};
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (const Shape &sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    const int lda = (K + 3) & ~3, ldb = lda, ldc = (N + 3) & ~3;
    std::vector<float> hA((size_t)M * lda), hB((size_t)N * ldb);
    for (auto &v : hA) v = frand();
    for (auto &v : hB) v = frand();
    float *dA, *dB, *dC, *dP;
    const size_t wsf = (size_t)24 * M * N;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4));
    CK(hipMalloc(&dC, (size_t)M * ldc * 4)); CK(hipMalloc(&dP, wsf * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    const double gf = 2e-9 * M * N * K;
    printf("== %s: %d x %d x %d  (%.3f GFLOP; matrix floor %.1f us)\n", sh.name, M, N, K, gf, gf / 157.3 * 1e3);
    auto run = [&](const GemmPCfg &c, const char *tag) {
      const int G = c.tiles_m * c.tiles_n * c.nz;
      auto once = [&](bool red) {
        if (!gemm_p_launch(c, M, N, K, 0.5f, dA, lda, dB, ldb, 0.f, dC, ldc, dP, 1, s)) { printf("launch failed\n"); exit(1); }
        if (red && c.nz > 1)
          hipLaunchKernelGGL(k_reduce, dim3((unsigned)(((long long)M * N + 255) / 256)), dim3(256), 0, s, M, N, c.nz, 0.5f, dP, dC, ldc);
      };
      CK(hipMemsetAsync(dC, 0xff, (size_t)M * ldc * 4, s));
      once(true);
      CK(hipStreamSynchronize(s));
      std::vector<float> hC((size_t)M * ldc);
      CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0.0;
      for (int it = 0; it < 400; it++) {
        const int m = it < 4 ? (it & 1 ? M - 1 : 0) : rand() % M, n = it < 4 ? (it & 2 ? N - 1 : 0) : rand() % N;
        double ref = 0.0, mag = 0.0;
        for (int k = 0; k < K; k++) { const double p = (double)hA[(size_t)m * lda + k] * hB[(size_t)n * ldb + k]; ref += p; mag += fabs(p); }
        const double err = fabs(0.5 * ref - hC[(size_t)m * ldc + n]) / (0.5 * mag);
        if (!(err <= worst)) worst = err;     // NaN-safe
      }
      float ms[2];
      for (int red = 0; red < 2; red++) {
        for (int i = 0; i < 5; i++) once(red);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 50; i++) once(red);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[red], e0, e1));
      }
      printf("  %-5s tile %3dx%-3d nz %2d kchunk %4d  G %4d (%.2f/CU)  gemm %6.1f us %5.1f TF | +reduce %6.1f us %5.1f TF | err %.1e%s\n", tag,
             32 * c.wm, 32 * c.wn, c.nz, c.kchunk, G, G / 256.0, ms[0] * 20.0, gf / (ms[0] * 20.0) * 1e3, ms[1] * 20.0,
             gf / (ms[1] * 20.0) * 1e3, worst, worst < 1e-6 ? "" : "  <-- WRONG");
    };
    const GemmPCfg pick = gemm_p_pick(M, N, K, wsf, 16);
    run(pick, "pick");
    if (argc > 4 && !strcmp(argv[1], "cfg")) {       // cfg wm wn ns
      GemmPCfg c;
      gemm_p_cost(M, N, K, atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), &c);
      run(c, "cfg");
    }
    if (all) {
#define GB_TILE(a, b)                                                                                     \
      for (int ns = 1; ns <= 16; ns++) {                                                                  \
        GemmPCfg c;                                                                                       \
        gemm_p_cost(M, N, K, a, b, ns, &c);                                                               \
        if (c.nz < ns) continue;                                                                          \
        const int G = c.tiles_m * c.tiles_n * c.nz;                                                       \
        if (G < 160 || G > 1100) continue;                                                                \
        run(c, "");                                                                                       \
      }
      GP_FOR_EACH_TILE(GB_TILE)
#undef GB_TILE
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dP));
  }
  return 0;
}
