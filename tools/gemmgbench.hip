// Development aid: k_gemm_g (ao_marl_amd/csrc/aomarl_gemm_g.h) alone: every operand form x every wave tile on ragged
// shapes with every epilogue, checked against a float64 host product on sampled entries, then timed on the shapes of
// aomarl_sac_update at production size (14 agents, batch 256).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/bin/gemmgbench tools/gemmgbench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../ao_marl_amd/csrc/aomarl_gemm_g.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static float frand() { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; }
static int up4(int v) { return (v + 3) & ~3; }

struct Case { const char *name; int G, M, N, K; bool ak, bk; int bias, relu, mask, cs; };

static int bad = 0;

static void run(const Case &c, int fwm, int fwn, bool timeit, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  const int G = c.G, M = c.M, N = c.N, K = c.K;
  const int lda = c.ak ? up4(K) + 4 : up4(M) + 4, ldb = c.bk ? up4(K) : up4(N) + 8, ldc = up4(N) + 4, ldm = up4(N);
  const size_t nA = (size_t)(c.ak ? M : K) * lda, nB = (size_t)(c.bk ? N : K) * ldb;
  std::vector<float> hA(nA * G), hB(nB * G), hBias((size_t)G * up4(N)), hMask((size_t)G * M * ldm);
  for (auto &v : hA) v = frand();
  for (auto &v : hB) v = frand();
  for (auto &v : hBias) v = frand();
  for (auto &v : hMask) v = frand();
  float *dA, *dB, *dC, *dBias, *dMask, *dCs;
  CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)G * M * ldc * 4));
  CK(hipMalloc(&dBias, hBias.size() * 4)); CK(hipMalloc(&dMask, hMask.size() * 4)); CK(hipMalloc(&dCs, (size_t)G * up4(N) * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dBias, hBias.data(), hBias.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dMask, hMask.data(), hMask.size() * 4, hipMemcpyHostToDevice));
  GemmGArgs a;
  memset(&a, 0, sizeof(a));
  a.M = M; a.N = N; a.K = K; a.A = dA; a.lda = lda; a.sA = (long long)nA; a.B = dB; a.ldb = ldb; a.sB = (long long)nB;
  a.C = dC; a.ldc = ldc; a.sC = (long long)M * ldc;
  if (c.bias) { a.bias = dBias; a.sBias = up4(N); }
  a.relu = c.relu;
  if (c.mask) { a.mask = dMask; a.ldm = ldm; a.sM = (long long)M * ldm; }
  if (c.cs) { a.colsum = dCs; a.sCs = up4(N); }
  CK(hipMemsetAsync(dC, 0xff, (size_t)G * M * ldc * 4, s));
  CK(hipMemsetAsync(dCs, 0xff, (size_t)G * up4(N) * 4, s));
  if (gemm_g_launch(G, c.ak, c.bk, a, fwm, fwn, s)) { printf("launch failed\n"); exit(1); }
  CK(hipStreamSynchronize(s));
  std::vector<float> hC((size_t)G * M * ldc), hCs((size_t)G * up4(N));
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hCs.data(), dCs, hCs.size() * 4, hipMemcpyDeviceToHost));
  auto Aat = [&](int g, int m, int k) { return hA[(size_t)g * nA + (c.ak ? (size_t)m * lda + k : (size_t)k * lda + m)]; };
  auto Bat = [&](int g, int n, int k) { return hB[(size_t)g * nB + (c.bk ? (size_t)n * ldb + k : (size_t)k * ldb + n)]; };
  double worst = 0.0;
  const int nsamp = timeit ? 300 : 1500;
  for (int it = 0; it < nsamp; it++) {
    const int g = it < 8 ? (it & 4 ? G - 1 : 0) : rand() % G;
    const int m = it < 8 ? (it & 1 ? M - 1 : 0) : rand() % M, n = it < 8 ? (it & 2 ? N - 1 : 0) : rand() % N;
    double ref = 0.0, mag = 1e-30;
    for (int k = 0; k < K; k++) { const double p = (double)Aat(g, m, k) * Bat(g, n, k); ref += p; mag += fabs(p); }
    if (c.bias) { ref += hBias[(size_t)g * up4(N) + n]; mag += 1.0; }
    if (c.relu) ref = ref > 0 ? ref : 0;
    if (c.mask && !(hMask[(size_t)g * M * ldm + (size_t)m * ldm + n] > 0.f)) ref = 0;
    const double err = fabs(ref - hC[(size_t)g * M * ldc + (size_t)m * ldc + n]) / mag;
    if (!(err <= worst)) worst = err;
  }
  // nothing written outside [M][N]
  for (int g = 0; g < G && !timeit; g++)
    for (int m = 0; m < M; m++)
      for (int n = N; n < ldc; n++) {
        uint32_t u; memcpy(&u, &hC[(size_t)g * M * ldc + (size_t)m * ldc + n], 4);
        if (u != 0xffffffffu) worst = 1.0;
      }
  double wcs = 0.0;
  if (c.cs)
    for (int g = 0; g < G; g++)
      for (int n = 0; n < N; n++) {
        double ref = 0.0, mag = 1e-30;
        for (int k = 0; k < K; k++) { ref += Bat(g, n, k); mag += fabs(Bat(g, n, k)); }
        const double err = fabs(ref - hCs[(size_t)g * up4(N) + n]) / mag;
        if (!(err <= wcs)) wcs = err;
      }
  float ms = 0.f;
  if (timeit) {
    for (int i = 0; i < 5; i++) gemm_g_launch(G, c.ak, c.bk, a, fwm, fwn, s);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 50; i++) gemm_g_launch(G, c.ak, c.bk, a, fwm, fwn, s);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  int wm = fwm, wn = fwn;
  if (!wm || !wn) { int pm, pn; gemm_g_pick(G, M, N, K, &pm, &pn); if (!wm) wm = pm; if (!wn) wn = pn; }
  const double gf = 2e-9 * G * (double)M * N * K;
  const bool ok = worst < 2e-6 && wcs < 2e-6;
  if (!ok) bad++;
  printf("  %-26s %s%s G %2d %4d x %4d x %4d tile %3dx%-3d%s  err %.1e cs %.1e", c.name, c.ak ? "A[m][k]" : "A[k][m]", c.bk ? " B[n][k]" : " B[k][n]",
         G, M, N, K, 32 * wm, 32 * wn, (fwm || fwn) ? "" : "*", worst, wcs);
  if (timeit) printf("  %6.1f us %5.1f TF", ms * 20.0, gf / (ms * 20.0) * 1e3);
  printf("%s\n", ok ? "" : "   <-- WRONG");
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dBias)); CK(hipFree(dMask)); CK(hipFree(dCs));
}

int main(int argc, char **argv) {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("== correctness: forms x tiles, ragged shapes, epilogues\n");
  for (int form = 0; form < 4; form++)
    for (int wm = 2; wm <= 4; wm += 2)
      for (int wn = 2; wn <= 4; wn += 2) {
        const bool ak = form & 1, bk = form & 2;
        Case cs[] = {{"ragged plain", 3, 70, 50, 45, ak, bk, 0, 0, 0, !bk},
                     {"ragged bias relu", 2, 131, 197, 98, ak, bk, 1, 1, 0, 0},
                     {"ragged mask", 3, 257, 129, 33, ak, bk, 0, 0, 1, !bk},
                     {"tiny", 1, 1, 2, 3, ak, bk, 1, 0, 0, !bk},
                     {"k=1", 2, 65, 66, 1, ak, bk, 0, 0, 0, !bk}};
        for (const Case &c : cs) run(c, wm, wn, false, s, e0, e1);
      }
  printf("== SAC update shapes (14 agents, batch 256, in 552, act 98, hidden 256)\n");
  const Case sac[] = {
      {"policy L1 fwd (x', x)", 14, 512, 256, 552, true, false, 1, 1, 0, 0},
      {"policy L2 fwd (x', x)", 14, 512, 256, 256, true, false, 1, 1, 0, 0},
      {"policy head fwd (x', x)", 14, 512, 196, 256, true, false, 1, 0, 0, 0},
      {"critic hidden fwd", 14, 256, 512, 650, true, false, 1, 1, 0, 0},
      {"critic dWin", 14, 650, 512, 256, false, false, 0, 0, 0, 0},
      {"dpi = dh Win_a^T", 14, 256, 98, 512, true, true, 0, 0, 0, 0},
      {"head dW + db", 14, 256, 196, 256, false, false, 0, 0, 0, 1},
      {"dA = dhd Whead^T [mask]", 14, 256, 256, 196, true, true, 0, 0, 1, 0},
      {"L2 dW + db", 14, 256, 256, 256, false, false, 0, 0, 0, 1},
      {"dA0 = dA1 W2^T [mask]", 14, 256, 256, 256, true, true, 0, 0, 1, 0},
      {"L1 dW + db", 14, 552, 256, 256, false, false, 0, 0, 0, 1},
  };
  for (const Case &c : sac) {
    run(c, 0, 0, true, s, e0, e1);
    if (argc > 1 && !strcmp(argv[1], "all"))
      for (int wm = 2; wm <= 4; wm += 2)
        for (int wn = 2; wn <= 4; wn += 2) run(c, wm, wn, true, s, e0, e1);
  }
  printf(bad ? "FAILED: %d case(s)\n" : "all cases ok\n", bad);
  return bad ? 1 : 0;
}
