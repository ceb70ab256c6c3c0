// Development aid: what does the frame kernel's access pattern cost on its own?
// Reads 16x16 tiles (rows of 64 B at an arbitrary 4-byte alignment, row stride 2608 B) of NL planes
// per environment exactly like k_frame_wave does and sums them; variants change the lane -> pixel
// mapping, the prefetch depth and the number of resident waves.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/membench tools/membench.hip && /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct __attribute__((packed, aligned(4))) f4u { float v[4]; };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int DIM = 648, LD = 652, NT = 40, NL = 3;   // the 40x40 configuration's screens

// MAP 0: lane -> (row = lane >> 2, col = 4 (lane & 3));  MAP 1: (row = lane & 15, col = 4 (lane >> 4))
// block = 4 waves; MODE 0: wave = (env, stripe), walks 40 tiles;  MODE 1: waves of a block split a stripe
template <int MAP, int DEPTH, int MODE>
__global__ __launch_bounds__(256) void k_read(const float *__restrict__ scr, long long env_stride,
                                              const int *__restrict__ org, float *__restrict__ out,
                                              int nenv, int lds_pad) {
  extern __shared__ float pad[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = blockIdx.x;
  const int e = MODE == 0 ? 4 * blockIdx.y + wv : blockIdx.y;
  if (e >= nenv) return;
  const int row = MAP == 0 ? lane >> 2 : lane & 15, col = MAP == 0 ? 4 * (lane & 3) : 4 * (lane >> 4);
  const float *lay[NL];
  unsigned lpx[NL], lrow[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    lay[l] = scr + (long long)e * env_stride + (long long)l * DIM * LD;
    unsigned px = org[(e * NL + l) * 2] + 4, py = org[(e * NL + l) * 2 + 1] + 4 + 16 * r + row;
    px -= px >= DIM ? DIM : 0; py -= py >= DIM ? DIM : 0;
    lpx[l] = px + col; lrow[l] = py * LD;
  }
  float raw[DEPTH][NL][4];
  auto fetch = [&](int t, int slot) {
#pragma unroll
    for (int l = 0; l < NL; l++) {
      unsigned px = 16u * t + lpx[l]; px = min(px, px - DIM);
      const f4u v = *reinterpret_cast<const f4u *>(lay[l] + (lrow[l] + px));
#pragma unroll
      for (int j = 0; j < 4; j++) raw[slot][l][j] = v.v[j];
    }
  };
  const int t0 = MODE == 0 ? 0 : wv, dt = MODE == 0 ? 1 : 4;
  const int ntw = MODE == 0 ? NT : (NT - wv + 3) / 4;       // tiles of this wave
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < DEPTH; d++) if (d < ntw) fetch(t0 + d * dt, d);
  for (int k = 0; k < ntw; k += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      if (k + d < ntw) {
#pragma unroll
        for (int l = 0; l < NL; l++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc += raw[d][l][j];
        if (k + d + DEPTH < ntw) fetch(t0 + (k + d + DEPTH) * dt, d);
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc + pad[lds_pad ? 0 : 0];
}

// WIDE pattern: one instruction = (16 / W) rows x (W x 64) contiguous bytes (the rows of W adjacent tiles);
// W instructions per layer fetch a group of W tiles.  Same bytes as MAP 0 / 1, longer contiguous pieces.
template <int W, int DEPTH>
__global__ __launch_bounds__(256) void k_read_wide(const float *__restrict__ scr, long long env_stride,
                                                   const int *__restrict__ org, float *__restrict__ out, int nenv) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = blockIdx.x, e = 4 * blockIdx.y + wv;
  if (e >= nenv) return;
  constexpr int CH = 4 * W, RPI = 64 / CH;                 // 16-byte chunks per row, rows per instruction
  const int row = lane / CH, col = 4 * (lane % CH);
  const float *lay[NL];
  unsigned lpx[NL], lrow[NL][W];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    lay[l] = scr + (long long)e * env_stride + (long long)l * DIM * LD;
    unsigned px = org[(e * NL + l) * 2] + 4;
    px -= px >= DIM ? DIM : 0;
    lpx[l] = px + col;
#pragma unroll
    for (int i = 0; i < W; i++) {
      unsigned py = org[(e * NL + l) * 2 + 1] + 4 + 16 * r + RPI * i + row;
      py -= py >= DIM ? DIM : 0;
      lrow[l][i] = py * LD;
    }
  }
  float raw[DEPTH][NL][W][4];
  auto fetch = [&](int g, int slot) {
#pragma unroll
    for (int l = 0; l < NL; l++) {
      unsigned px = 16u * W * g + lpx[l]; px = min(px, px - DIM);
#pragma unroll
      for (int i = 0; i < W; i++) {
        const f4u v = *reinterpret_cast<const f4u *>(lay[l] + (lrow[l][i] + px));
#pragma unroll
        for (int j = 0; j < 4; j++) raw[slot][l][i][j] = v.v[j];
      }
    }
  };
  constexpr int NG = NT / W;
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < DEPTH; d++) fetch(d, d);
  for (int k = 0; k < NG; k += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      if (k + d < NG) {
#pragma unroll
        for (int l = 0; l < NL; l++)
#pragma unroll
          for (int i = 0; i < W; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc += raw[d][l][i][j];
        if (k + d + DEPTH < NG) fetch(k + d + DEPTH, d);
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

// PAIR pattern: the lane -> pixel mapping of MAP 1 (what the frame kernel computes on), but the loads of two
// adjacent tiles (the two 64-byte halves of the same rows) are issued back to back, DEPTH pairs in flight.
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read_pair(const float *__restrict__ scr, long long env_stride,
                                                   const int *__restrict__ org, float *__restrict__ out, int nenv) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = blockIdx.x, e = 4 * blockIdx.y + wv;
  if (e >= nenv) return;
  const int row = lane & 15, col = 4 * (lane >> 4);
  const float *lay[NL];
  unsigned lpx[NL], lrow[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    lay[l] = scr + (long long)e * env_stride + (long long)l * DIM * LD;
    unsigned px = org[(e * NL + l) * 2] + 4, py = org[(e * NL + l) * 2 + 1] + 4 + 16 * r + row;
    px -= px >= DIM ? DIM : 0; py -= py >= DIM ? DIM : 0;
    lpx[l] = px + col; lrow[l] = py * LD;
  }
  float raw[DEPTH][2][NL][4];
  auto fetch = [&](int g, int slot) {
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int l = 0; l < NL; l++) {
        unsigned px = 16u * (2 * g + h) + lpx[l]; px = min(px, px - DIM);
        const f4u v = *reinterpret_cast<const f4u *>(lay[l] + (lrow[l] + px));
#pragma unroll
        for (int j = 0; j < 4; j++) raw[slot][h][l][j] = v.v[j];
      }
  };
  constexpr int NG = NT / 2;
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < DEPTH; d++) fetch(d, d);
  for (int k = 0; k < NG; k += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      if (k + d < NG) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
          for (int l = 0; l < NL; l++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc += raw[d][h][l][j];
        if (k + d + DEPTH < NG) fetch(k + d + DEPTH, d);
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int DEPTH>
float run_pair(const float *scr, long long es, const int *org, float *out, int nenv) {
  dim3 grid(NT, (nenv + 3) / 4), blk(256);
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_read_pair<DEPTH>), grid, blk, 0, 0, scr, es, org, out, nenv);
  CK(hipEventRecord(a));
  const int reps = 10;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_read_pair<DEPTH>), grid, blk, 0, 0, scr, es, org, out, nenv);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

template <int W, int DEPTH>
float run_wide(const float *scr, long long es, const int *org, float *out, int nenv) {
  dim3 grid(NT, (nenv + 3) / 4), blk(256);
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_read_wide<W, DEPTH>), grid, blk, 0, 0, scr, es, org, out, nenv);
  CK(hipEventRecord(a));
  const int reps = 10;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_read_wide<W, DEPTH>), grid, blk, 0, 0, scr, es, org, out, nenv);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

template <int MAP, int DEPTH, int MODE>
float run(const float *scr, long long es, const int *org, float *out, int nenv, int lds) {
  dim3 grid(NT, MODE == 0 ? (nenv + 3) / 4 : nenv), blk(256);
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_read<MAP, DEPTH, MODE>), grid, blk, lds, 0, scr, es, org, out, nenv, lds);
  CK(hipEventRecord(a));
  const int reps = 10;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_read<MAP, DEPTH, MODE>), grid, blk, lds, 0, scr, es, org, out, nenv, lds);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const int nenv = 256;
  const long long es = (long long)NL * DIM * LD;
  float *scr, *out; int *org;
  CK(hipMalloc(&scr, sizeof(float) * es * nenv));
  CK(hipMemset(scr, 0, sizeof(float) * es * nenv));
  CK(hipMalloc(&out, 64));
  std::vector<int> h(nenv * NL * 2);
  srand(1);
  for (auto &v : h) v = rand() % DIM;
  CK(hipMalloc(&org, sizeof(int) * h.size()));
  CK(hipMemcpy(org, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
  const double bytes = (double)nenv * NT * NT * NL * 1024.0;
  printf("algorithmic bytes per launch: %.3f GB (all 1600 tiles)\n", bytes * 1e-9);
#define R(MAP, DEPTH, MODE, LDS)                                                              \
  do {                                                                                        \
    float ms = run<MAP, DEPTH, MODE>(scr, es, org, out, nenv, LDS);                            \
    printf("map %d depth %d mode %d lds %6d : %.3f ms  %.2f TB/s\n", MAP, DEPTH, MODE, LDS, ms, \
           bytes / ms * 1e-9);                                                                \
  } while (0)
  R(0, 1, 0, 0); R(0, 2, 0, 0); R(0, 4, 0, 0);
  R(1, 1, 0, 0); R(1, 2, 0, 0); R(1, 4, 0, 0);
  R(0, 1, 1, 0); R(0, 2, 1, 0); R(0, 4, 1, 0);
  R(1, 1, 1, 0); R(1, 2, 1, 0);
  R(0, 2, 0, 40000); R(0, 2, 0, 60000); R(1, 2, 0, 40000);
#define RW(W, DEPTH)                                                                          \
  do {                                                                                        \
    float ms = run_wide<W, DEPTH>(scr, es, org, out, nenv);                                    \
    printf("wide %d tiles per group, depth %d groups : %.3f ms  %.2f TB/s\n", W, DEPTH, ms, bytes / ms * 1e-9); \
  } while (0)
  for (int dp = 1; dp <= 2; dp++) {
    float ms = dp == 1 ? run_pair<1>(scr, es, org, out, nenv) : run_pair<2>(scr, es, org, out, nenv);
    printf("pair (map 1, two adjacent tiles back to back), depth %d pairs : %.3f ms  %.2f TB/s\n", dp, ms, bytes / ms * 1e-9);
  }
  RW(2, 1); RW(2, 2); RW(4, 1); RW(4, 2); RW(8, 1);
  // every environment at the SAME ring origin (what a simulation has: one wind for all) -- do the
  // environments' equal row offsets, one environment stride apart, fall on the same memory channels?
  for (auto &v : h) v = 123;
  CK(hipMemcpy(org, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
  printf("same origin in every environment, environment stride %lld B:\n", es * 4);
  R(1, 2, 0, 0); R(1, 4, 0, 0);
  for (long long padf : {64LL, 1024LL, 4160LL, 16448LL}) {
    float *scr2;
    const long long es2 = es + padf;
    CK(hipMalloc(&scr2, sizeof(float) * es2 * nenv));
    CK(hipMemset(scr2, 0, sizeof(float) * es2 * nenv));
    float ms = run<1, 2, 0>(scr2, es2, org, out, nenv, 0);
    printf("  stride + %lld B : map 1 depth 2 : %.3f ms  %.2f TB/s\n", padf * 4, ms, bytes / ms * 1e-9);
    CK(hipFree(scr2));
  }
  return 0;
}
