// Development aid: issue cost of the vector instructions the frame kernel is made of (one wave per SIMD, independent
// operations, 8 accumulators): cycles per wave-instruction from the wall time at the measured clock.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/valubench tools/valubench.hip && tools/bin/valubench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(768) void k(float *out, int iters, float seed) {
  float a[8];
  f32x2 p[8];
  f32x4 m[4];
  for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f32x2{a[i], a[i] + 1.f}; }
  for (int i = 0; i < 4; i++) m[i] = f32x4{a[i], a[i], a[i], a[i]};
  const float b = seed * 0.999f, c = seed * 1e-3f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
      if (KIND == 2) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 3) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 4) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 5) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 6) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 7) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[i & 3]) : "v"(b), "v"(c));
      if (KIND == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (KIND == 10) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
      if (KIND == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
      // the integer side of Philox (k_frame_wave's noisy instantiations): 32 x 32 products
      if (KIND == 12) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (KIND == 13) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (KIND == 14) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(p[i]) : "v"(b), "v"(c) : "vcc");
      if (KIND == 15) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (KIND == 16) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 17) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 18) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
  for (int i = 0; i < 4; i++) s += m[i][0];
  if (s == 12345.678f) out[0] = s;
}

template <int KIND>
int run(const char *name, float *out, double ghz, int wps = 1) {
  const int iters = 20000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, out, iters, 1.0f);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, out, iters, 1.0f);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("%-34s %d wave(s) per SIMD: %.3f ms  -> %.2f cycles of the SIMD per instruction at %.2f GHz\n", name, wps, ms, ms * 1e-3 * ghz * 1e9 / (iters * 8.0 * wps), ghz);
  return 0;
}

int main() {
  float *out; CK(hipMalloc(&out, 64));
  int khz = 0; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
  const double ghz = khz * 1e-6;
  for (int w = 2; w <= 3; w++) {
    run<0>("v_fma_f32", out, ghz, w); run<1>("v_pk_fma_f32", out, ghz, w); run<2>("v_sin_f32", out, ghz, w);
    run<4>("v_add_f32_dpp quad_perm", out, ghz, w); run<6>("v_add_f32_dpp row_bcast:15", out, ghz, w);
    run<3>("v_rndne_f32", out, ghz, w); run<7>("v_mfma_f32_16x16x4_f32", out, ghz, w);
  }
  run<0>("v_fma_f32", out, ghz); run<9>("v_add_f32", out, ghz); run<1>("v_pk_fma_f32", out, ghz);
  run<10>("v_pk_add_f32", out, ghz); run<11>("v_pk_mul_f32", out, ghz);
  run<2>("v_sin_f32", out, ghz); run<8>("v_rcp_f32", out, ghz); run<3>("v_rndne_f32", out, ghz);
  run<4>("v_add_f32_dpp quad_perm", out, ghz); run<5>("v_add_f32_dpp row_mirror", out, ghz);
  run<6>("v_add_f32_dpp row_bcast:15", out, ghz); run<7>("v_mfma_f32_16x16x4_f32", out, ghz);
  for (int w = 1; w <= 3; w += 2) {
    run<12>("v_mul_lo_u32", out, ghz, w); run<13>("v_mul_hi_u32", out, ghz, w); run<14>("v_mad_u64_u32", out, ghz, w);
    run<15>("v_xor_b32", out, ghz, w); run<16>("v_log_f32", out, ghz, w); run<17>("v_sqrt_f32", out, ghz, w);
    run<18>("v_cvt_f32_u32", out, ghz, w); run<8>("v_rcp_f32", out, ghz, w);
  }
  return 0;
}
