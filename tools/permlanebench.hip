// Development aid (round 6): what v_permlane32_swap_b32 / v_permlane16_swap_b32 (new on gfx950) do to a wave, lane by
// lane, and what they cost beside the DPP adds of the frame kernel's reductions -- the three moments of a sub-aperture
// summed over the wave in 10 instructions instead of 3 x 7 (spot_cog_qf, ao_marl_amd/csrc/aomarl_kernels.hip).
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/permlanebench tools/permlanebench.hip && tools/bin/permlanebench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void k_map(int *o) {
  const unsigned x = threadIdx.x, y = 100 + threadIdx.x;
  const u32x2 a = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  const u32x2 b = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  o[threadIdx.x] = a[0]; o[64 + threadIdx.x] = a[1]; o[128 + threadIdx.x] = b[0]; o[192 + threadIdx.x] = b[1];
}

template <int B>
__device__ __forceinline__ float dppadd(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), B, 0xF, 0xF, true));
}
// the reduction as the frame kernel does it now: three sums of 64 lanes, each valid in lane 63
__device__ __forceinline__ float sum_last(float v) {
  v = dppadd<0xB1>(v); v = dppadd<0x4E>(v); v = dppadd<0x141>(v); v = dppadd<0x140>(v);
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
  return v;
}
// ... and through the swaps: row 0 <- sum a, row 1 <- sum c, row 2 <- sum b (every lane of the row)
__device__ __forceinline__ float sum3_rows(float a, float b, float c) {
  u32x2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  const float u = __uint_as_float(t[0]) + __uint_as_float(t[1]);        // rows 0, 1: a (r0 + r2, r1 + r3); rows 2, 3: b
  t = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(c), false, false);
  const float w = __uint_as_float(t[0]) + __uint_as_float(t[1]);        // every row pair: c
  t = __builtin_amdgcn_permlane16_swap(__float_as_uint(u), __float_as_uint(w), false, false);
  float z = __uint_as_float(t[0]) + __uint_as_float(t[1]);              // rows: a, c, b, c
  z = dppadd<0xB1>(z); z = dppadd<0x4E>(z); z = dppadd<0x141>(z); z = dppadd<0x140>(z);
  return z;
}

template <int KIND>
__global__ __launch_bounds__(768) void k_time(float *out, int iters, float seed) {
  float a = seed + threadIdx.x * 1e-3f, b = a * 0.5f, c = a * 0.25f, acc = 0.f;
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) { acc += sum_last(a) + sum_last(b) + sum_last(c); }
    else { acc += sum3_rows(a, b, c); }
    a += 1e-3f; b += 2e-3f; c -= 1e-3f;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void k_check(float *o) {
  const float a = 1.f + threadIdx.x, b = 1000.f + 3.f * threadIdx.x, c = -7.f * threadIdx.x;
  const float z = sum3_rows(a, b, c);
  o[threadIdx.x] = z;
  o[64 + threadIdx.x] = sum_last(a); o[128 + threadIdx.x] = sum_last(b); o[192 + threadIdx.x] = sum_last(c);
}

int main() {
  int *d; int h[256];
  CK(hipMalloc(&d, sizeof(h)));
  hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, d);
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  const char *nm[4] = {"permlane32_swap(x, y)[0]", "permlane32_swap(x, y)[1]", "permlane16_swap(x, y)[0]", "permlane16_swap(x, y)[1]"};
  for (int k = 0; k < 4; k++) {
    printf("%s (x = lane, y = 100 + lane), first lane of each row:", nm[k]);
    for (int r = 0; r < 4; r++) printf("  row %d: %d", r, h[64 * k + 16 * r]);
    printf("\n");
  }
  float *f; float hf[256];
  CK(hipMalloc(&f, sizeof(float) * 768 * 1024));
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, f);
  CK(hipMemcpy(hf, f, sizeof(hf), hipMemcpyDeviceToHost));
  printf("sum3_rows: row 0 %.1f (sum a = %.1f)  row 1 %.1f (sum c = %.1f)  row 2 %.1f (sum b = %.1f)  row 3 %.1f\n",
         hf[0], hf[64 + 63], hf[16], hf[192 + 63], hf[32], hf[128 + 63], hf[48]);
  int dev = 0, clk = 0;
  CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int waves = 1; waves <= 3; waves++) {
    for (int kind = 0; kind < 2; kind++) {
      const int iters = 20000, blocks = 256;
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        if (kind == 0) hipLaunchKernelGGL(k_time<0>, dim3(blocks), dim3(256 * waves), 0, 0, f, iters, 1.f);
        else hipLaunchKernelGGL(k_time<1>, dim3(blocks), dim3(256 * waves), 0, 0, f, iters, 1.f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      // one block per CU (256 blocks on 256 CUs), `waves` waves per SIMD: SIMD cycles per reduction of three sums
      printf("%d wave(s) per SIMD, %s: %.1f SIMD cycles per three sums per wave\n", waves,
             kind ? "2 x permlane32_swap + permlane16_swap + 4 DPP adds" : "3 x (4 DPP adds + 2 row_bcast adds)   ",
             best * 1e-3 * (double)clk * 1e3 / iters / waves);
    }
  }
  return 0;
}
