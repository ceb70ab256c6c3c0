// Development aid: the frame kernel's layer reads as FULL-LINE pieces through LDS.
// A wave fetches the rows of two adjacent 16 x 16 tiles with instructions that cover 8 rows x 128 contiguous
// bytes each (buffer_load_dwordx4 ... lds: the data lands in the wave's LDS image, no vector registers), and
// reads them back in the compute layout (lane (q, c) = row c, pixels 4q .. 4q + 3 of one tile) with
// conflict-free ds_read_b128.  Checks every value against the plain per-lane loads, then times both.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/dmabench tools/dmabench.hip && tools/bin/dmabench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct __attribute__((packed, aligned(4))) f4u { float v[4]; };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int DIM = 648, PAD = 32, LD = DIM + PAD, NT = 40, NL = 3, NG = NT / 2;
constexpr int IMG = 2 * 1024 + 128;          // LDS image of one layer of a tile pair: two 1 KB blocks, the second 128 B further
constexpr int WAVE_LDS = NL * IMG;

// reference: plain loads in the compute layout
__global__ __launch_bounds__(256) void k_plain(const float *__restrict__ scr, long long env_stride,
                                               const int *__restrict__ org, float *__restrict__ out, int nenv, int check) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = blockIdx.x, e = 4 * blockIdx.y + wv;
  if (e >= nenv) return;
  const int c = lane & 15, q = lane >> 4;
  float acc = 0.f;
  float raw[2][NL][4];
  const float *lay[NL];
  unsigned lpx[NL], lrow[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    lay[l] = scr + (long long)e * env_stride + (long long)l * DIM * LD;
    unsigned px = org[(e * NL + l) * 2], py = org[(e * NL + l) * 2 + 1] + 16 * r + c;
    py -= py >= DIM ? DIM : 0;
    lpx[l] = px; lrow[l] = py * LD;
  }
  auto fetch = [&](int t, int slot) {
#pragma unroll
    for (int l = 0; l < NL; l++) {
      unsigned px = 16u * t + lpx[l]; px -= px >= DIM ? DIM : 0;          // scalar
      const f4u v = *reinterpret_cast<const f4u *>(lay[l] + (lrow[l] + px + 4 * q));
#pragma unroll
      for (int j = 0; j < 4; j++) raw[slot][l][j] = v.v[j];
    }
  };
  fetch(0, 0); fetch(1, 1);
  for (int t = 0; t < NT; t += 2) {
#pragma unroll
    for (int d = 0; d < 2; d++) {
#pragma unroll
      for (int l = 0; l < NL; l++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          acc += raw[d][l][j];
          if (check) out[((((long long)e * NT + r) * NT + t + d) * NL + l) * 256 + c * 16 + 4 * q + j] = raw[d][l][j];
        }
      if (t + d + 2 < NT) fetch(t + d + 2, d);
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_dma(const float *__restrict__ scr, long long env_stride,
                                             const int *__restrict__ org, float *__restrict__ out, int nenv, int check) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = blockIdx.x, e = 4 * blockIdx.y + wv;
  if (e >= nenv) return;
  const int c = lane & 15, q = lane >> 4;
  const unsigned wave_lds = (unsigned)(unsigned long long)lds + (unsigned)wv * WAVE_LDS;    // LDS byte address
  // loader role of the lane: row rr of the 8-row block, chunk (j ^ rr) of the 8 16-byte chunks of a 128-byte row piece
  const int rr = lane >> 3, jj = lane & 7, kk = jj ^ rr;
  __amdgpu_buffer_rsrc_t rs[NL];
  unsigned lpx[NL], vo[NL][2];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    const float *lay = scr + (long long)e * env_stride + (long long)l * DIM * LD;
    rs[l] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(lay), 0, 4 * DIM * LD, 0x00020000);
    lpx[l] = org[(e * NL + l) * 2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
      unsigned py = org[(e * NL + l) * 2 + 1] + 16 * r + 8 * b + rr;
      py -= py >= DIM ? DIM : 0;
      vo[l][b] = 4u * (py * LD) + 16u * kk;
    }
  }
  // reader role: tile h, row c = 8 b + rc, chunk 4 h + q  ->  slot (rc * 8 + ((4 h + q) ^ rc)) of block b
  const int bc = c >> 3, rc = c & 7;
  unsigned rd[2];
#pragma unroll
  for (int h = 0; h < 2; h++) rd[h] = (unsigned)wv * WAVE_LDS + bc * 1152 + 16 * (rc * 8 + ((4 * h + q) ^ rc));
  auto dma = [&](int g) {
#pragma unroll
    for (int l = 0; l < NL; l++) {
      unsigned px = 32u * g + lpx[l]; px -= px >= DIM ? DIM : 0;            // scalar: a 32-pixel run never wraps (PAD = 32)
      const unsigned so = 4u * px;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const unsigned m0v = wave_lds + l * IMG + b * 1152;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                     :: "v"(vo[l][b]), "s"(rs[l]), "s"(m0v), "s"(so) : "memory");
      }
    }
  };
  float acc = 0.f;
  dma(0);
  for (int g = 0; g < NG; g++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float raw[2][NL][4];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const float4 v = *reinterpret_cast<const float4 *>(lds + rd[h] + l * IMG);
        raw[h][l][0] = v.x; raw[h][l][1] = v.y; raw[h][l][2] = v.z; raw[h][l][3] = v.w;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (g + 1 < NG) dma(g + 1);
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int l = 0; l < NL; l++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          acc += raw[h][l][j];
          if (check) out[((((long long)e * NT + r) * NT + 2 * g + h) * NL + l) * 256 + c * 16 + 4 * q + j] = raw[h][l][j];
        }
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const int nenv = 256;
  const long long es = (long long)NL * DIM * LD;
  std::vector<float> hs((size_t)es * nenv);
  for (int e = 0; e < nenv; e++)
    for (int l = 0; l < NL; l++)
      for (int y = 0; y < DIM; y++)
        for (int x = 0; x < LD; x++) {
          const int xs = x < DIM ? x : x - DIM;              // mirror columns
          hs[(size_t)e * es + ((size_t)l * DIM + y) * LD + x] = (float)((((e * 7 + l) * 131 + y) * 17 + xs) & 0xFFFFF);
        }
  float *scr, *out0, *out1; int *org;
  CK(hipMalloc(&scr, sizeof(float) * es * nenv));
  CK(hipMemcpy(scr, hs.data(), sizeof(float) * es * nenv, hipMemcpyHostToDevice));
  const int nchk = 8;                                          // environments compared value by value
  const size_t nout = (size_t)nchk * NT * NT * NL * 256;
  CK(hipMalloc(&out0, sizeof(float) * nout)); CK(hipMalloc(&out1, sizeof(float) * nout));
  std::vector<int> h(nenv * NL * 2);
  srand(1);
  for (auto &v : h) v = rand() % DIM;
  CK(hipMalloc(&org, sizeof(int) * h.size()));
  CK(hipMemcpy(org, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
  const size_t smem = 4 * WAVE_LDS;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  {
    dim3 grid(NT, nchk / 4), blk(256);
    hipLaunchKernelGGL(k_plain, grid, blk, 0, 0, scr, es, org, out0, nchk, 1);
    hipLaunchKernelGGL(k_dma, grid, blk, smem, 0, scr, es, org, out1, nchk, 1);
    CK(hipDeviceSynchronize());
    std::vector<float> a(nout), b(nout);
    CK(hipMemcpy(a.data(), out0, sizeof(float) * nout, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), out1, sizeof(float) * nout, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < nout; i++) bad += a[i] != b[i];
    printf("LDS-DMA image against plain loads: %zu of %zu values differ\n", bad, nout);
  }
  const double bytes = (double)nenv * NT * NT * NL * 1024.0;
  for (int which = 0; which < 2; which++) {
    dim3 grid(NT, nenv / 4), blk(256);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto launch = [&]() {
      if (which == 0) hipLaunchKernelGGL(k_plain, grid, blk, 0, 0, scr, es, org, out0, nenv, 0);
      else hipLaunchKernelGGL(k_dma, grid, blk, smem, 0, scr, es, org, out1, nenv, 0);
    };
    for (int i = 0; i < 2; i++) launch();
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    printf("%s : %.3f ms  %.2f TB/s (%.3f GB per launch)\n",
           which == 0 ? "plain loads, compute layout (16 rows x 64 B per instruction), 2 tiles in flight"
                      : "LDS-DMA, 8 rows x 128 B per instruction, 1 pair in flight                    ",
           ms, bytes / ms * 1e-9, bytes * 1e-9);
  }
  return 0;
}
