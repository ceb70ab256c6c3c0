// aomarl_host.h -- host-side pieces shared by the translation units of libaomarl_hip.so (aomarl_capi.hip: context,
// frame / atmosphere / control kernels and every entry point but the two below; aomarl_denoise.hip: the denoiser;
// aomarl_sac.hip: the learner's update and the grouped GEMM).  Not part of the C ABI: hidden visibility.
#pragma once
#include "aomarl_dev.h"

#define AOMARL_LOCAL __attribute__((visibility("hidden")))

// sets this thread's last-error text (aomarl_last_error), returns 1
AOMARL_LOCAL int fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define HIPCHK(x)                                                                            \
  do {                                                                                       \
    hipError_t _e = (x);                                                                     \
    if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e),    \
                                      __FILE__, __LINE__);                                   \
  } while (0)
#define LAUNCHCHK()                                                                          \
  do {                                                                                       \
    hipError_t _e = hipGetLastError();                                                       \
    if (_e != hipSuccess) return fail("kernel launch failed: %s (%s:%d)",                    \
                                      hipGetErrorString(_e), __FILE__, __LINE__);            \
  } while (0)

// launches per arithmetic family since aomarl_arith_reset (bench.py builds its `dtype` from them)
enum { AR_FRAME_F32 = 0, AR_FRAME_SPLIT, AR_GEMM_F32, AR_GEMM_SPLIT, AR_DENOISE_F32, AR_DENOISE_SPLIT, AR_ACTOR_F32, AR_N };
extern AOMARL_LOCAL unsigned long long g_arith[AR_N];
extern AOMARL_LOCAL int g_precision;           // process-wide default of every family (aomarl_set_precision)

// round 1's general batched product (aomarl_capi_composites.hip): rows that are not 16-byte aligned, accumulation into C
AOMARL_LOCAL int gemm_batched_launch(int batch, int transA, int transB, int M, int N, int K, const float *A, int lda,
                                     long long strideA, const float *B, int ldb, long long strideB,
                                     const float *bias, long long strideBias, float *C, int ldc, long long strideC,
                                     int relu, int accumulate, const float *mask, int ldm, long long strideM,
                                     hipStream_t s);
// one launch of the grouped kernel (aomarl_gemm_g.h, instantiated in aomarl_sac.hip): C[g] = act(opA(A[g]) opB(B[g]) + bias[g]);
// ak / bk: the operand is contiguous along k
AOMARL_LOCAL int gemm_g_batched(int batch, bool ak, bool bk, int M, int N, int K, const float *A, int lda, long long sA,
                                const float *B, int ldb, long long sB, const float *bias, long long sBias, float *C, int ldc,
                                long long sC, int relu, hipStream_t s);
