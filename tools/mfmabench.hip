// Development aid: fp32 MFMA (16x16x4) throughput, VALU throughput and how well the two overlap
// (a) inside one wave's instruction stream, (b) between waves of one SIMD.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/bin/mfmabench tools/mfmabench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: MFMA only; 1: VALU only; 2: both in every wave (interleaved by the compiler);
// 3: even waves MFMA, odd waves VALU (same total work as mode 2 per pair of waves)
// F16 = true: v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_16x16x4_f32 (same 64 per iteration)
template <int MODE, bool F16>
__global__ __launch_bounds__(512) void k(float *out, int iters, float a0, float b0) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = a0 + (float)(lane + i);
  const float a = a0 + lane, b = b0 + lane;
  half8 ha, hb;
#pragma unroll
  for (int i = 0; i < 8; i++) { ha[i] = (_Float16)(a0 * 0.01f + i); hb[i] = (_Float16)(b0 * 0.01f + lane * 0.001f); }
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wv < 4);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wv >= 4);
  const int rep = MODE == 3 ? 2 : 1;      // mode 3: each wave does twice its kind -> same totals as mode 2
  for (int it = 0; it < iters * rep; it++) {
    if (do_m) {
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < 8; i++)
          acc[i] = F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0)
                       : __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    if (do_v) {
#pragma unroll
      for (int r = 0; r < 32; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b0), "v"(a0));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i];
  if (s == 12345.f) out[0] = s;
}

template <int MODE, bool F16 = false>
void run(float *out, int blocks_per_cu, const char *what) {
  const int iters = 200;
  dim3 grid(256 * blocks_per_cu), blk(512);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, F16>), grid, blk, 0, 0, out, iters, 1.0f, 0.5f);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, F16>), grid, blk, 0, 0, out, iters, 1.0f, 0.5f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  // per wave per iteration: 64 MFMAs (2048 flop each), 512 VALU fma
  const double waves = 256.0 * blocks_per_cu * 8;
  const double mf = (MODE == 1) ? 0 : waves * iters * 64 * (F16 ? 16384.0 : 2048.0);
  const double vi = (MODE == 0) ? 0 : waves * iters * 512.0;
  printf("%-28s blocks/CU %d : %.3f ms   MFMA %.1f TFLOP/s   VALU %.2f Tinstr/s (wave64)\n", what, blocks_per_cu,
         ms, mf / ms * 1e-9, vi / ms * 1e-9);
}

int main() {
  float *out;
  CK(hipMalloc(&out, 64));
  for (int b : {1, 2}) {
    run<0>(out, b, "MFMA only");
    run<1>(out, b, "VALU only");
    run<2>(out, b, "both, same wave");
    run<3>(out, b, "both, specialised waves");
    run<0, true>(out, b, "f16 MFMA only");
    run<2, true>(out, b, "f16 MFMA + VALU, same wave");
    run<3, true>(out, b, "f16 MFMA + VALU, specialised");
  }
  return 0;
}
