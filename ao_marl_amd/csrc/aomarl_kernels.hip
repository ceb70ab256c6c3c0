// aomarl_kernels.hip -- CDNA4 (gfx950) kernels of the AO environment hot path.
// wave = 64 lanes; fp32 MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2_f32) for the DFT and GEMM work.
#include "aomarl_host.h"
#include "aomarl_gemm_p.h"
#include <type_traits>

#define WAVE 64

// ---- split-fp16 helpers (used by the GEMM below and by the frame kernel; see the frame kernel's notes)
// (hx2 / hx8 and mfma_h: aomarl_dev.h -- the denoiser's translation unit uses them too)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ hx2 cvt_h2(float a, float b) {
  return __builtin_bit_cast(hx2, __builtin_amdgcn_cvt_pkrtz(a, b));
}
// a - f32(h.lo) and a - f32(h.hi) as single mixed-precision FMAs (v_fma_mix_f32 reads the f16 half
// directly; the compiler will not form it from a - (float)h).  Operands always come out of ordinary
// VALU instructions (v_cvt_pkrtz of the same value sits in between any producer and this read).
__device__ __forceinline__ float sub_lo(hx2 h, float a) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a));
  return r;
}
__device__ __forceinline__ float sub_hi(hx2 h, float a) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a));
  return r;
}

// 4 consecutive floats at a dword-aligned (not necessarily 16-byte aligned) address: one
// global_load_dwordx4 (gfx950 supports unaligned vector access; the compiler emits it for this type)
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };
__device__ __forceinline__ void add4(float acc[4], const float *p) {
  const f4u t = *reinterpret_cast<const f4u *>(p);
  acc[0] += t.v[0]; acc[1] += t.v[1]; acc[2] += t.v[2]; acc[3] += t.v[3];
}
// 4 consecutive logical pixels of a ring-buffered screen row starting at physical column px
__device__ __forceinline__ void add4_ring(float acc[4], const float *row, int px, int dim) {
  (void)dim;                     // rows carry RING_PAD mirror columns: never wraps
  add4(acc, row + px);
}

// =============================================================================================
// fp32 GEMM  C[M][N] = alpha * A[M][K] . B[N][K]^T + beta * C     (both operands K-contiguous), ANY alignment:
// the fallback behind k_gemm_p (aomarl_gemm_p.h), which wants 16-byte aligned rows.  Element-wise loads,
// 256 threads = 4 waves in 2x2, block tile 64x64, one v_mfma_f32_32x32x2_f32 accumulator/wave.
// =============================================================================================
__global__ __launch_bounds__(256) void k_gemm_nt(int M, int N, int K, float alpha,
                                                 const float *__restrict__ A, int lda,
                                                 const float *__restrict__ B, int ldb, float beta,
                                                 float *__restrict__ C, int ldc, int kchunk,
                                                 float *__restrict__ P) {
  // blockIdx.z = K split: the block reduces k in [z*kchunk, min(K, (z+1)*kchunk)); with more than
  // one split the raw partial tile goes to P[z][M][N] and k_gemm_reduce finishes (deterministic)
  __shared__ float As[64][17];
  __shared__ float Bs[64][17];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int kb = blockIdx.z * kchunk, ke = min(K, kb + kchunk);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  const int lr = tid >> 2, lc = (tid & 3) * 4;
  const int gm = m0 + lr, gn = n0 + lr;
  const float *pa = A + (long long)gm * lda;
  const float *pb = B + (long long)gn * ldb;
  for (int k0 = kb; k0 < ke; k0 += 16) {
    float va[4] = {0.f, 0.f, 0.f, 0.f}, vb[4] = {0.f, 0.f, 0.f, 0.f};
    const int gk = k0 + lc;
    if (gm < M) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (gk + j < ke) va[j] = pa[gk + j];
    }
    if (gn < N) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (gk + j < ke) vb[j] = pb[gk + j];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      As[lr][lc + j] = va[j];
      Bs[lr][lc + j] = vb[j];
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      float a = As[wm * 32 + (lane & 31)][2 * ks + (lane >> 5)];
      float b = Bs[wn * 32 + (lane & 31)][2 * ks + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int col = n0 + wn * 32 + (lane & 31);
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < M && col < N) {
      if (split) {
        P[((long long)blockIdx.z * M + row) * N + col] = acc[r];
      } else {
        float *c = C + (long long)row * ldc + col;
        float v = alpha * acc[r];
        if (beta != 0.f) v += beta * (*c);
        *c = v;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Pipelined variant (operands 16-byte aligned, lda / ldb multiples of 4): block tile 64 x 64 x 32,
// double-buffered LDS (one barrier per k-tile), the global loads of tile k+1 are issued before the
// MFMAs of tile k.  K order inside a tile is permuted so that every lane reads its 16 k-values of
// a row as four 128-bit LDS reads: MFMA number i of the tile uses k = 16 (lane >> 5) + i for both
// operands (any order is fine as long as A and B agree).  Row stride 36 floats: the 16 lanes of
// a 128-bit read pass hit 64 distinct banks.
// ---------------------------------------------------------------------------------------------
#define G2_LD 36


// ---------------------------------------------------------------------------------------------
// The same GEMM on the f16 matrix pipe with SPLIT operands: every fp32 value v is carried as
// hi = f16(v), lo = f16(v - hi), both rounded to nearest (23 significant bits, unbiased) and a product
// as hi.hi + lo.hi + hi.lo with fp32 accumulation (lo.lo, <= 2^-24 of a product, dropped): three v_mfma_f32_32x32x16_f16 of 32 cycles
// each per 16 k against sixteen 64-cycle v_mfma_f32_32x32x2_f32 (fp32 matrix instructions run at the
// packed-fp32 vector rate on this chip) -- 10x less matrix-pipe time; what is left is the splitting
// (2 vector instructions per element as it is staged into LDS) and the LDS traffic.
// Operands stay fp32 in memory, same interface as k_gemm_nt plus a power-of-two scale per operand
// (sa, sb; applied as the values are staged, undone through alpha): the scaled values must stay
// below 65504 (they saturate above) and lose low bits of `lo` below 6e-5 (absolute error <= 3e-8
// of the scaled value).  gemm_scale() picks the scale of a static matrix from its largest entry.
// LDS: hi and lo planes of the A and B tiles as f16, [row][32 k] with a row stride of 40 halfs (a
// 16-lane pass of a 128-bit read -- 8 k of one row per lane -- covers the 64 banks exactly once).
// ---------------------------------------------------------------------------------------------
#define GH_LD 40
typedef _Float16 hx4 __attribute__((ext_vector_type(4)));
// round-to-nearest-even pair (v_cvt_pk_f16_f32): unbiased, unlike the truncating v_cvt_pkrtz -- a GEMM
// adds thousands of products, a truncation bias would add up linearly
typedef float fx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hx2 cvt_rn2(float a, float b) {
  const fx2 v = {a, b};
  return __builtin_convertvector(v, hx2);
}
__device__ __forceinline__ void gh_split(float4 v, const float scale, hx4 &hi, hx4 &lo, float &amax) {
  // power of two: exact; clamped to the f16 range (a value beyond it -- a centroid whose total flux
  // came out ~0 -- saturates instead of turning into inf - inf = NaN).  amax: largest scaled magnitude this
  // thread staged; the kernel counts the threads that saw one above the range (aomarl_gemm_saturated)
  v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
  amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  v.x = __builtin_amdgcn_fmed3f(v.x, -65504.f, 65504.f); v.y = __builtin_amdgcn_fmed3f(v.y, -65504.f, 65504.f);
  v.z = __builtin_amdgcn_fmed3f(v.z, -65504.f, 65504.f); v.w = __builtin_amdgcn_fmed3f(v.w, -65504.f, 65504.f);
  const hx2 h01 = cvt_rn2(v.x, v.y), h23 = cvt_rn2(v.z, v.w);
  const hx2 l01 = cvt_rn2(sub_lo(h01, v.x), sub_hi(h01, v.y));   // |v - hi| <= 2^-12 |v|, lo keeps 11 bits of it
  const hx2 l23 = cvt_rn2(sub_lo(h23, v.z), sub_hi(h23, v.w));
  hi = hx4{h01[0], h01[1], h23[0], h23[1]};
  lo = hx4{l01[0], l01[1], l23[0], l23[1]};
}

// Three k-tiles of global loads are in flight per thread (register stages, loop unrolled by three):
// with the 10-20 k-tiles a split-K block walks, one tile ahead left the loop waiting for L2 / HBM on
// every iteration.  The loop body has NO branch around a load and no select on a load's result: the
// loads are unconditional (addresses clamped into the row; a tile past the end of the chunk is masked
// to zero as it is staged into LDS, so the loop simply runs whole groups of three tiles) -- with
// either, the compiler waits for the data where it is loaded and the three stages collapse into one
// (3 000 lines of branchy ISA and 20 us per call; measured).  Scheduling barriers keep the loads where
// they are written.  Callers make the chunk a multiple of 96 so that only the last chunk has padding.
__device__ __forceinline__ void gh_mainloop(const float *__restrict__ A, int lda,
                                            const float *__restrict__ B, int ldb, int M, int N,
                                            int m0, int n0, int kb, int ke, _Float16 *S, f32x16 &acc,
                                            const float sa, const float sb, float &amax) {
  // S: [2 buffers][4 planes: A hi, A lo, B hi, B lo][64 rows][GH_LD]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  const int lr = tid >> 3, lc = (tid & 7) * 4;
  const float *pa0 = A + (long long)min(m0 + lr, M - 1) * lda;
  const float *pa1 = A + (long long)min(m0 + lr + 32, M - 1) * lda;
  const float *pb0 = B + (long long)min(n0 + lr, N - 1) * ldb;
  const float *pb1 = B + (long long)min(n0 + lr + 32, N - 1) * ldb;
  const int klast = (ke - 1) & ~3;               // last 16-byte group that holds a valid element (lda, ldb >= its end)
  constexpr int ST = 3;                          // register stages
  float4 ra0[ST], ra1[ST], rb0[ST], rb1[ST];
  auto gload = [&](int k0, int st) {             // st: compile-time after unrolling
    const int k = min(k0 + lc, klast);
    ra0[st] = *reinterpret_cast<const float4 *>(pa0 + k); ra1[st] = *reinterpret_cast<const float4 *>(pa1 + k);
    rb0[st] = *reinterpret_cast<const float4 *>(pb0 + k); rb1[st] = *reinterpret_cast<const float4 *>(pb1 + k);
  };
  constexpr int PL = 64 * GH_LD;                 // halfs per plane
  auto lstore = [&](int buf, int st, int k0) {
    _Float16 *s = S + buf * 4 * PL;
    float4 a0 = ra0[st], a1 = ra1[st], b0 = rb0[st], b1 = rb1[st];
    if (k0 + 32 > ke) {                          // wave-uniform: the tile crosses the end of the chunk
      const int k = k0 + lc;
      const bool m0_ = k < ke, m1_ = k + 1 < ke, m2_ = k + 2 < ke, m3_ = k + 3 < ke;
      auto msk = [&](float4 &v) { v.x = m0_ ? v.x : 0.f; v.y = m1_ ? v.y : 0.f; v.z = m2_ ? v.z : 0.f; v.w = m3_ ? v.w : 0.f; };
      msk(a0); msk(a1); msk(b0); msk(b1);
    }
    hx4 h, l;
    gh_split(a0, sa, h, l, amax);
    *reinterpret_cast<hx4 *>(s + lr * GH_LD + lc) = h; *reinterpret_cast<hx4 *>(s + PL + lr * GH_LD + lc) = l;
    gh_split(a1, sa, h, l, amax);
    *reinterpret_cast<hx4 *>(s + (lr + 32) * GH_LD + lc) = h; *reinterpret_cast<hx4 *>(s + PL + (lr + 32) * GH_LD + lc) = l;
    gh_split(b0, sb, h, l, amax);
    *reinterpret_cast<hx4 *>(s + 2 * PL + lr * GH_LD + lc) = h; *reinterpret_cast<hx4 *>(s + 3 * PL + lr * GH_LD + lc) = l;
    gh_split(b1, sb, h, l, amax);
    *reinterpret_cast<hx4 *>(s + 2 * PL + (lr + 32) * GH_LD + lc) = h; *reinterpret_cast<hx4 *>(s + 3 * PL + (lr + 32) * GH_LD + lc) = l;
  };
  // operand of lane l for k-chunk c (16 k): row (l & 31) of the wave's 32, k = 16 c + 8 (l >> 5) .. + 7
  const int ro = (lane & 31) * GH_LD + 8 * (lane >> 5);
#pragma unroll
  for (int st = 0; st < ST; st++) gload(kb + 32 * st, st);
  __builtin_amdgcn_sched_barrier(0);
  lstore(0, 0, kb);
  gload(kb + 32 * ST, 0);
  __syncthreads();
  int buf = 0;
  for (int k0 = kb; k0 < ke; k0 += 32 * ST) {
#pragma unroll
    for (int u = 0; u < ST; u++) {
      const int kc = k0 + 32 * u;                // the tile in LDS buffer `buf` (all zeros past the end)
      // tile kc + 32 sits in register stage (u + 1) % ST: split it into the other LDS buffer, then
      // reuse that stage for tile kc + 32 (ST + 1); only then the matrix instructions on this tile
      lstore(buf ^ 1, (u + 1) % ST, kc + 32);
      gload(kc + 32 * (ST + 1), (u + 1) % ST);
      __builtin_amdgcn_sched_barrier(0);
      const _Float16 *s = S + buf * 4 * PL;
      const _Float16 *ah = s + wm * 32 * GH_LD + ro, *al = ah + PL;
      const _Float16 *bh = s + 2 * PL + wn * 32 * GH_LD + ro, *bl = bh + PL;
#pragma unroll
      for (int cch = 0; cch < 2; cch++) {
        const hx8 Ah = *reinterpret_cast<const hx8 *>(ah + 16 * cch), Al = *reinterpret_cast<const hx8 *>(al + 16 * cch);
        const hx8 Bh = *reinterpret_cast<const hx8 *>(bh + 16 * cch), Bl = *reinterpret_cast<const hx8 *>(bl + 16 * cch);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      buf ^= 1;
    }
  }
}

__global__ __launch_bounds__(256) void k_gemm_nt_h(int M, int N, int K, float alpha,
                                                   const float *__restrict__ A, int lda,
                                                   const float *__restrict__ B, int ldb, float beta,
                                                   float *__restrict__ C, int ldc, int kchunk,
                                                   float *__restrict__ P, float sa, float sb, int xcd,
                                                   unsigned *__restrict__ sat) {
  __shared__ __attribute__((aligned(16))) _Float16 S[2 * 4 * 64 * GH_LD];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wm = wv >> 1, wn = wv & 1;
  // Workgroups go to the 8 XCDs round-robin in launch order, each XCD with its own 4 MB L2.  With the
  // plain (x, y, z) order every XCD sees tiles of every k-chunk, i.e. streams BOTH operands whole
  // (5 + 6 MB for an extrusion round) through its L2; remapped, XCD q owns a contiguous range of the
  // z-major order -- about one k-chunk, 1.6 MB of operands.  Worth 2.5 % of a reset (45.8 -> 44.6 ms),
  // no more: the kernel is not bound by where its operands come from (the GEMM notes).
  int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
  if (xcd) {
    const int T = gridDim.x * gridDim.y * gridDim.z;
    const int L = bxi + gridDim.x * (byi + gridDim.y * bzi);
    const int q = L & 7, i = L >> 3;
    const int lg = q * (T >> 3) + min(q, T & 7) + i;
    const int xy = gridDim.x * gridDim.y;
    bzi = lg / xy;
    const int r = lg - bzi * xy;
    byi = r / gridDim.x; bxi = r - byi * gridDim.x;
  }
  const int m0 = byi * 64, n0 = bxi * 64;
  const int kb = bzi * kchunk, ke = min(K, kb + kchunk);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  float amax = 0.f;
  gh_mainloop(A, lda, B, ldb, M, N, m0, n0, kb, ke, S, acc, sa, sb, amax);
  if (amax > 65504.f) atomicAdd(sat, 1u);        // a scaled operand left the fp16 range and was clipped (rare: one atomic per such thread)
  const int col = n0 + wn * 32 + (lane & 31);
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < M && col < N) {
      if (split) {
        P[((long long)bzi * M + row) * N + col] = acc[r];
      } else {
        float *c = C + (long long)row * ldc + col;
        float v = alpha * acc[r];
        if (beta != 0.f) v += beta * (*c);
        *c = v;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// General batched GEMM for the SAC networks' forward AND backward passes:
//     C[b] = act( opA(A[b]) . opB(B[b]) + bias[b] ) (+ C[b] when accumulate)
// opA(A) is M x K, opB(B) is K x N.  TA = false: A stored [M][K] (K contiguous); TA = true: A stored
// [K][M].  TB = false: B stored [N][K] (the "NT" form above); TB = true: B stored [K][N].
// The three products of a linear layer y = x W (W stored [in][out]) are
//     forward  y  = x . W        TA = 0, TB = 1        backward dx = dy . W^T     TA = 0, TB = 0
//     weights  dW = x^T . dy     TA = 1, TB = 1
// 64 x 64 tile, LDS rows of 36 floats, one 32x32x2 accumulator per wave; a k-strided operand is read with 128-bit loads
// along its contiguous (row) direction and transposed on the way into LDS.
// ---------------------------------------------------------------------------------------------
template <bool T>
__device__ __forceinline__ void gg_load(const float *__restrict__ P, int ld, int rows, int r0, int k0,
                                        int ke, int tid, float (&v)[8], bool vec) {
  // this thread's 8 elements of the 64 (rows) x 32 (k) tile starting at (r0, k0)
  if (!T) {
    const int lr = tid >> 3, lc = (tid & 7) * 4;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int r = min(r0 + lr + 32 * h, rows - 1);
      const float *p = P + (long long)r * ld;
      const int k = k0 + lc;
      if (k + 3 < ke && vec) {
        const float4 t = *reinterpret_cast<const float4 *>(p + k);
        v[4 * h] = t.x; v[4 * h + 1] = t.y; v[4 * h + 2] = t.z; v[4 * h + 3] = t.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) v[4 * h + j] = (k + j < ke) ? p[k + j] : 0.f;
      }
    }
  } else {
    const int kk = tid >> 4, r4 = (tid & 15) * 4;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int k = k0 + kk + 16 * h;
      const float *p = P + (long long)k * ld;
      if (k < ke) {
        if (r0 + r4 + 3 < rows && vec) {
          const float4 t = *reinterpret_cast<const float4 *>(p + r0 + r4);
          v[4 * h] = t.x; v[4 * h + 1] = t.y; v[4 * h + 2] = t.z; v[4 * h + 3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) v[4 * h + j] = p[min(r0 + r4 + j, rows - 1)];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) v[4 * h + j] = 0.f;
      }
    }
  }
}

template <bool T>
__device__ __forceinline__ void gg_store(float *S, int tid, const float (&v)[8]) {
  if (!T) {
    const int lr = tid >> 3, lc = (tid & 7) * 4;
#pragma unroll
    for (int h = 0; h < 2; h++)
      *reinterpret_cast<float4 *>(S + (lr + 32 * h) * G2_LD + lc) = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
  } else {
    const int kk = tid >> 4, r4 = (tid & 15) * 4;
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int j = 0; j < 4; j++) S[(r4 + j) * G2_LD + kk + 16 * h] = v[4 * h + j];
  }
}

// G k-groups of 4 waves share one 64 x 64 output tile: group g runs the K slabs g, g + G, ... through
// its own double-buffered LDS stage, so G slabs are in flight per block (these products are small --
// 224 tiles for the SAC layers -- and with one wave per SIMD every slab paid the full L2 / MALL
// latency); the partial tiles are summed through LDS in a fixed order.
template <bool TA, bool TB, int G>
__global__ __launch_bounds__(256 * G) void k_gemm_batched_gen(int M, int N, int K,
                                                              const float *__restrict__ A, int lda, long long sA,
                                                              const float *__restrict__ B, int ldb, long long sB,
                                                              const float *__restrict__ bias, long long sBias,
                                                              float *__restrict__ C, int ldc, long long sC,
                                                              int relu, int accumulate, int vecA, int vecB,
                                                              const float *__restrict__ mask, int ldm, long long sM,
                                                              int tn, int tm, int ntile) {
  // mask (the layer's forward output, for the ReLU backward): C = acc where mask > 0, else 0
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  // XCD-aware tile order: workgroups go round-robin over the 8 XCDs (each with its own L2), so
  // workgroup L runs tile (L % 8) * per + L / 8: the tiles of one matrix -- which share A rows and
  // B columns -- land on one XCD and fetch them into its L2 once instead of once per XCD.
  const int per = gridDim.x >> 3;
  const int w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (w >= ntile) return;
  const int bz = w / (tn * tm), wt = w - bz * (tn * tm);
  const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
  float *As = gsm + grp * (4 * 64 * G2_LD), *Bs = As + 2 * 64 * G2_LD;
  const int lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  const int m0 = (wt / tn) * 64, n0 = (wt % tn) * 64;
  A += (long long)bz * sA; B += (long long)bz * sB; C += (long long)bz * sC;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  const int nslab = (K + 31) / 32, nloop = (nslab + G - 1) / G;      // block-uniform trip count
  float va[8], vb[8];
  if (grp < nslab) {
    gg_load<TA>(A, lda, M, m0, grp * 32, K, tid, va, vecA);
    gg_load<TB>(B, ldb, N, n0, grp * 32, K, tid, vb, vecB);
    gg_store<TA>(As, tid, va);
    gg_store<TB>(Bs, tid, vb);
  }
  __syncthreads();
  const int ro = (lane & 31) * G2_LD + 16 * (lane >> 5);
  int buf = 0;
  for (int it = 0; it < nloop; it++, buf ^= 1) {
    const int slab = it * G + grp;
    const bool live = slab < nslab, more = slab + G < nslab;
    if (more) {
      gg_load<TA>(A, lda, M, m0, (slab + G) * 32, K, tid, va, vecA);
      gg_load<TB>(B, ldb, N, n0, (slab + G) * 32, K, tid, vb, vecB);
    }
    if (live) {
      const float *as = As + buf * 64 * G2_LD + wm * 32 * G2_LD + ro;
      const float *bs = Bs + buf * 64 * G2_LD + wn * 32 * G2_LD + ro;
      float4 a4[4], b4[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        a4[j] = *reinterpret_cast<const float4 *>(as + 4 * j);
        b4[j] = *reinterpret_cast<const float4 *>(bs + 4 * j);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].x, b4[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].y, b4[j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].z, b4[j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].w, b4[j].w, acc, 0, 0, 0);
      }
    }
    if (more) {
      gg_store<TA>(As + (buf ^ 1) * 64 * G2_LD, tid, va);
      gg_store<TB>(Bs + (buf ^ 1) * 64 * G2_LD, tid, vb);
    }
    __syncthreads();
  }
  if (G > 1) {
    // partial tiles of groups 1 .. G-1 -> LDS [g-1][r][256 threads]; group 0 adds them in order
    if (grp > 0) {
      float *red = gsm + (grp - 1) * (16 * 256);
#pragma unroll
      for (int r = 0; r < 16; r++) red[r * 256 + tid] = acc[r];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < G; g++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] += gsm[(g - 1) * (16 * 256) + r * 256 + tid];
  }
  const int col = n0 + wn * 32 + (lane & 31);
  const float bv = (bias && col < N) ? bias[(long long)bz * sBias + col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < M && col < N) {
      float *c = C + (long long)row * ldc + col;
      float v = acc[r] + bv;
      if (accumulate) v += *c;
      if (relu) v = fmaxf(v, 0.f);
      if (mask && !(mask[(long long)bz * sM + (long long)row * ldm + col] > 0.f)) v = 0.f;
      *c = v;
    }
  }
}

__global__ void k_gemm_reduce(int M, int N, int nsplit, float alpha, const float *__restrict__ P,
                              float beta, float *__restrict__ C, int ldc) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * N) return;
  const int row = (int)(i / N), col = (int)(i - (long long)row * N);
  const float s = slab_sum<4>(nsplit, [&](int z) { return P[(long long)z * M * N + i]; });
  float *c = C + (long long)row * ldc + col;
  float v = alpha * s;
  if (beta != 0.f) v += beta * (*c);
  *c = v;
}

// split-K reduce with the consumer's element-wise step folded in (saves that launch):
//   mode 1: C = err, com += gain * err                                (Rtc.do_control)
//   mode 2: C = modes, modes[m] += action[j] * freedom[m] for the action modes  (rl_control)
struct GemmEpi {
  int mode;
  float *com; int ldcom; float gain;
  const float *gain_row;               // mode 1: per-row (per-environment) integrator gains, or null
  const float *action; int nact; const int32_t *amode_inv; const float *freedom;
};

__global__ void k_gemm_reduce_epi(int M, int N, int nsplit, float alpha, const float *__restrict__ P,
                                  float beta, float *__restrict__ C, int ldc, GemmEpi ep) {
  CHAIN_SETPRIO();
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * N) return;
  const int row = (int)(i / N), col = (int)(i - (long long)row * N);
  const float s = slab_sum<4>(nsplit, [&](int z) { return P[(long long)z * M * N + i]; });
  float *c = C + (long long)row * ldc + col;
  float v = alpha * s;
  if (beta != 0.f) v += beta * (*c);
  if (ep.mode == 1) {
    ep.com[(long long)row * ep.ldcom + col] += (ep.gain_row ? ep.gain_row[row] : ep.gain) * v;
  } else if (ep.mode == 2) {
    const int j = ep.amode_inv[col];
    if (j >= 0) v += ep.action[(long long)row * ep.nact + j] * ep.freedom[col];
  }
  *c = v;
}

static int g_gemm_target_blocks = 0;     // 0: split-K by the blocks-per-CU cost model (launch_gemm_nt); > 0: about that many blocks
// power-of-two scale that brings the largest magnitude of a matrix to ~4096 (f16: 11 bits, max 65504)
static float gemm_scale(const float *h, size_t n) {
  float m = 0.f;
  for (size_t i = 0; i < n; i++) m = std::max(m, fabsf(h[i]));
  if (!(m > 0.f) || !std::isfinite(m)) return 1.f;
  int e = (int)floorf(log2f(4096.f / m));
  e = std::max(-10, std::min(24, e));
  return ldexpf(1.f, e);
}
// Arithmetic of the library.  The reference computes in fp32 throughout (Rtc_FFF, shesha/sutra_wrap.py:49; every
// array cast to np.float32, shesha/init/wfs_init.py:76-101), and so does the DEFAULT here: fp32 operands on fp32
// matrix instructions (v_mfma_f32_*_f32), fp32 vector arithmetic.  "precision" = 1 (aomarl_set_precision) is the
// opt-in fast mode: split-fp16 operand pairs (hi + lo, 22-bit mantissa, fp32 accumulation) in the three kernel
// families that have such a form -- the frame kernel's DFTs, the internal GEMMs, the denoiser.
static bool g_gemm_split_f16 = false; // "gemm_split_f16": the internal GEMMs (extrusion, command matrix, Btt projections) on k_gemm_nt_h
int g_precision = 0;                  // process-wide default of every family (aomarl_set_precision)
// launches per arithmetic family since aomarl_arith_reset (bench.py builds its `dtype` from them)
// (the AR_* enumeration: aomarl_host.h)
unsigned long long g_arith[AR_N] = {0, 0, 0, 0, 0, 0, 0};
static const char *const g_arith_name[AR_N] = {
    "frame_kernel_dft:f32_mfma", "frame_kernel_dft:split_f16_mfma", "gemm:f32_mfma", "gemm:split_f16_mfma",
    "denoiser:f32_mfma", "denoiser:split_f16_mfma", "actor:f32_mfma"};
static int g_gemm_xcd = 1;            // "gemm_xcd_map": k_gemm_nt_h's / k_gemm_p's blocks grouped by k-chunk per XCD
static int g_gemm_kgroups = 0;       // batched general GEMM: 0 = by heuristic; 1 / 2 / 4 forced
// Retired after their A/B runs (profiles/r01g_*): the un-pipelined aligned kernel (30 us vs 22 us per
// call) and an in-kernel split-K reduction through ticket counters (4x slower: every block pays an
// L2 write-back for its __threadfence).  k_gemm_nt / k_gemm_nt_batched stay as the fallback
// for operands that are not 16-byte aligned.

// Threads of k_gemm_nt_h launches that staged an operand beyond the fp16 range (clipped to +-65504): one
// counter per device, read and cleared by aomarl_gemm_saturated.  The internal call sites scale their
// operands with margins of 10^2 .. 10^4 over what a closed loop produces (stencil differences x 2^8 up to
// 255 um, modes x 2^4 up to 4094, slopes x 1); a diverging policy or a runaway loop can leave them.
static unsigned *g_gemm_sat[64] = {nullptr};
static unsigned *gemm_sat_counter() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!g_gemm_sat[dev]) {
    void *p = nullptr;
    if (hipMalloc(&p, sizeof(unsigned)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned)) != hipSuccess) return nullptr;
    g_gemm_sat[dev] = (unsigned *)p;
  }
  return g_gemm_sat[dev];
}

// ws / ws_floats: optional split-K workspace (NULL: never split)
// epi: applied by the split-K reduce when there is one (returns true), else left to the caller
//      (returns false).  nsplit_out: when non-null and the GEMM was split, NO reduce is launched and
//      the caller's next kernel sums the partial tiles ws[z][M][N] itself (*nsplit_out = count,
//      0 = C is final).
// fast: the split-f16 kernel may be used (internal call sites whose operands are inside its range)
bool launch_gemm_nt(int M, int N, int K, float alpha, const float *A, int lda, const float *B,
                    int ldb, float beta, float *C, int ldc, hipStream_t s, float *ws = nullptr,
                    size_t ws_floats = 0, const GemmEpi *epi = nullptr, int *nsplit_out = nullptr,
                    bool fast = false, float sa = 1.f, float sb = 1.f, float *alpha_out = nullptr,
                    int min_chunk = 128, int pick_M = 0) {
  // alpha_out: the factor the caller must apply to the partial tiles when it sums them itself
  // pick_M > 0: tile and split-K as a product of pick_M rows would get them (a sum's order depends on the k split
  //             alone: M rows at once then give, bit for bit, what M / pick_M products of pick_M rows give)
  if (nsplit_out) *nsplit_out = 0;
  if (alpha_out) *alpha_out = alpha;
  if (M <= 0 || N <= 0) return false;
  const int Mp = pick_M > 0 ? pick_M : M;
  const size_t wsp = pick_M > 0 ? (size_t)((double)ws_floats * Mp / M) : ws_floats;     // the part's share of the workspace
  const int bx = (N + 63) / 64, by = (M + 63) / 64, byp = (Mp + 63) / 64;
  int nsplit = 1;
  bool al = (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A & 15) == 0) &&
            (((uintptr_t)B & 15) == 0);
  if (ws && bx * byp < 384) {
    // Split K so that the launch is as short as its slowest CU: blocks go round-robin over the 256 CUs, a CU
    // that gets one block more than the others sets the duration (528 blocks = 2.06 per CU took as long as 768
    // would: 132 tiles x 4 chunks lost to 132 x 3 = 396).  Cost model per candidate: blocks per CU (rounded
    // up) x k-tiles per block (whole groups of three for the pipelined kernels, + 2 tiles of fill / drain).
    // min_chunk: the control chain's products ask for at least three groups of three k-tiles per block
    // (288): below that the fill / drain of the load pipeline and the wider reduce cost more than the
    // extra blocks bring (round-2 script gemm_split_time.py, since removed).
    const int ncu = 256, tiles = bx * byp;
    if (g_gemm_target_blocks > 0) {              // "gemm_target_blocks" > 0: the plain rule (about that many blocks)
      nsplit = (g_gemm_target_blocks + tiles - 1) / tiles;
      if (nsplit > 8) nsplit = 8;
      while (nsplit > 1 && (K / nsplit < min_chunk || (size_t)nsplit * Mp * N > wsp)) nsplit--;
    } else {
      long long best = -1;
      for (int ns = 1; ns <= 8; ns++) {
        if (ns > 1 && (K / ns < min_chunk || (size_t)ns * Mp * N > wsp)) break;
        const int chunk = al ? ((K + ns - 1) / ns + 95) / 96 * 96 : (((K + ns - 1) / ns + 31) & ~31);
        const int nz = (K + chunk - 1) / chunk;
        const long long per_cu = ((long long)tiles * nz + ncu - 1) / ncu;
        const long long cost = per_cu * (chunk / 32 + 3);
        if (best < 0 || cost < best) { best = cost; nsplit = ns; }
      }
    }
  }
  const bool split_f16 = al && fast && g_gemm_split_f16;
  if (al && !split_f16) {
    // round 4: the balanced kernel; tile and k split from its own cost model (memoised per shape)
    struct Memo { int M, N, K; size_t ws; GemmPCfg c; };
    static thread_local Memo memo[16];
    static thread_local int memo_n = 0;
    const size_t wsf = ws ? wsp : 0;
    const GemmPCfg *cfg = nullptr;
    for (int i = 0; i < memo_n; i++)
      if (memo[i].M == Mp && memo[i].N == N && memo[i].K == K && memo[i].ws == wsf) { cfg = &memo[i].c; break; }
    if (!cfg) {
      Memo &m = memo[memo_n < 16 ? memo_n++ : (memo_n = 1, 0)];
      m.M = Mp; m.N = N; m.K = K; m.ws = wsf;
      m.c = gemm_p_pick(Mp, N, K, wsf, ws ? 16 : 1);
      cfg = &m.c;
    }
    GemmPCfg mine = *cfg;                        // (pick_M: the part's tile and k split over this product's rows)
    if (mine.wm > 0) mine.tiles_m = (M + 32 * mine.wm - 1) / (32 * mine.wm);
    cfg = &mine;
    if (cfg->wm > 0 && gemm_p_launch(*cfg, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, ws, g_gemm_xcd, s)) {
      g_arith[AR_GEMM_F32]++;
      nsplit = cfg->nz;
      if (nsplit > 1) {
        const long long tot = (long long)M * N;
        if (nsplit_out) { *nsplit_out = nsplit; return false; }
        if (epi) {
          hipLaunchKernelGGL(k_gemm_reduce_epi, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, M, N,
                             nsplit, alpha, ws, beta, C, ldc, *epi);
          return true;
        }
        hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, M, N,
                           nsplit, alpha, ws, beta, C, ldc);
      }
      return false;
    }
  }
  int kchunk = ((K + nsplit - 1) / nsplit + 31) & ~31;
  if (al)                                        // whole groups of three k-tiles (g3_mainloop / gh_mainloop)
    kchunk = ((K + nsplit - 1) / nsplit + 95) / 96 * 96;
  nsplit = (K + kchunk - 1) / kchunk;
  dim3 grid(bx, by, nsplit);
  unsigned *sat = (al && fast && g_gemm_split_f16) ? gemm_sat_counter() : nullptr;
  if (al && fast && g_gemm_split_f16 && sat) {
    alpha /= (sa * sb);                            // also what the split-K reduce below applies
    if (alpha_out) *alpha_out = alpha;
    hipLaunchKernelGGL(k_gemm_nt_h, grid, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C,
                       ldc, kchunk, ws, sa, sb, g_gemm_xcd, sat);
    g_arith[AR_GEMM_SPLIT]++;
  }
  else {
    hipLaunchKernelGGL(k_gemm_nt, grid, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb,
                       beta, C, ldc, kchunk, ws);
    g_arith[AR_GEMM_F32]++;
  }
  if (nsplit > 1) {
    const long long tot = (long long)M * N;
    if (nsplit_out) { *nsplit_out = nsplit; return false; }
    if (epi) {
      hipLaunchKernelGGL(k_gemm_reduce_epi, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, M, N,
                         nsplit, alpha, ws, beta, C, ldc, *epi);
      return true;
    }
    hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, M, N,
                       nsplit, alpha, ws, beta, C, ldc);
  }
  return false;
}

// =============================================================================================
// atmosphere: Fried-Clark extrusion on ring-buffered screens
// =============================================================================================
struct RoundOps {
  int nops;
  int layer[AOMARL_MAX_LAYERS];
  int dir[AOMARL_MAX_LAYERS];
  int tflag[AOMARL_MAX_LAYERS];     // 1: the screen holds the transpose (reset): dir is +-2, stencil istT
};
// the round behind a round, as the fused scatter + gather sees it: idx[i] = position of operation i's layer in the next
// round (-1: that layer is done), dir / tflag of the next round's operations
struct RoundNext {
  int nops;
  int idx[AOMARL_MAX_LAYERS];
  int dir[AOMARL_MAX_LAYERS];
  int tflag[AOMARL_MAX_LAYERS];
};

// Z[col][0..ns) = screen[stencil] - zref ; Z[col][ns..ns+n) = amplitude * N(0,1)
// work item j of column col, for a ring origin (ox, oy) and an extrusion counter cnt given by the caller
__device__ __forceinline__ void extrude_gather_item(const DevSys &sys, const DevState &st, int env_begin,
                                                    const RoundOps &ops, float *__restrict__ Z, int ldz,
                                                    float *__restrict__ ZREF, int col, int j, int ox, int oy,
                                                    uint32_t cnt) {
  const int e = env_begin + col / ops.nops, op = col % ops.nops;
  const int li = ops.layer[op], dir = ops.dir[op];
  const DevLayer &L = sys.layers[li];
  const int n = L.dim, ns = L.ns;
  const float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
  const bool top_right = (dir == 1 || dir == -2);
  const float zref = base[ring_idx(top_right ? n - 1 : 0, top_right ? 0 : n - 1, ox, oy, n)];
  const uint32_t *ist = ops.tflag[op] ? L.istT : ((dir == 1 || dir == -1) ? L.istx : L.isty);
  const uint32_t seed = st.seeds[e] + (uint32_t)li;
  // items [0, ns): stencil values; items [ns, ns + ceil(n / 4)): 4 normals each (one Philox block)
  if (j < ns) {
    uint32_t xy = ist[j];
    Z[(long long)col * ldz + j] = base[ring_idx(xy & 0xFFFF, xy >> 16, ox, oy, n)] - zref;
  } else if (j < ns + (n + 3) / 4) {
    const int g = j - ns;
    float z4[4];
    philox_normal4(seed, 0u, cnt, 0u, (uint32_t)g, z4);
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (4 * g + u < n) Z[(long long)col * ldz + ns + 4 * g + u] = L.amp * z4[u];
  }
  if (j == 0) ZREF[col] = zref;
}

__global__ __launch_bounds__(256) void k_extrude_gather(DevSys sys, DevState st, int env_begin,
                                                        RoundOps ops, float *__restrict__ Z,
                                                        int ldz, float *__restrict__ ZREF) {
  ATM_SETPRIO();
  const int col = blockIdx.x;
  const int e = env_begin + col / ops.nops, li = ops.layer[col % ops.nops];
  const int ox = st.origin[(e * sys.nlayers + li) * 2], oy = st.origin[(e * sys.nlayers + li) * 2 + 1];
  extrude_gather_item(sys, st, env_begin, ops, Z, ldz, ZREF, col, blockIdx.y * blockDim.x + threadIdx.x, ox, oy,
                      st.ext_count[e * sys.nlayers + li]);
}

// new line -> ring (+ mirror columns) for a ring origin (ox, oy); returns the advanced origin
__device__ __forceinline__ void extrude_scatter_col(const DevSys &sys, const DevState &st, int env_begin,
                                                    const RoundOps &ops, const float *__restrict__ NEWL, int ldn,
                                                    const float *__restrict__ ZREF, const float *__restrict__ P,
                                                    int nsplit, int ncol, int pn, float pscale, int col, int ox,
                                                    int oy, int &nox, int &noy, float *__restrict__ snew = nullptr) {
  // nsplit > 0: the new lines are still split-K partial tiles P[z][ncol][pn] of the extrusion GEMM
  // (times 1 / pscale when the split-f16 kernel produced them from scaled operands)
  const int e = env_begin + col / ops.nops, op = col % ops.nops;
  const int li = ops.layer[op], dir = ops.dir[op];
  const DevLayer &L = sys.layers[li];
  const int n = L.dim;
  float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
  const float zref = ZREF[col];
  const int stride = n + RING_PAD;
  for (int r = threadIdx.x; r < n; r += blockDim.x) {
    float v = zref;
    if (nsplit > 0) {
      const float acc = slab_sum<8>(nsplit, [&](int z) { return P[((long long)z * ncol + col) * pn + r]; });   // (tools/reset_trace.sh)
      v += acc * pscale;
    } else {
      v += NEWL[(long long)col * ldn + r];
    }
    int px, py;
    if (dir == 1) {
      px = ox;
      py = r + oy; py -= (py >= n) ? n : 0;
    } else if (dir == -1) {
      px = ox - 1; px += (px < 0) ? n : 0;
      py = n - 1 - r + oy; py -= (py >= n) ? n : 0;
    } else if (dir == 2) {
      py = oy;
      px = r + ox; px -= (px >= n) ? n : 0;
    } else {
      py = oy - 1; py += (py < 0) ? n : 0;
      px = n - 1 - r + ox; px -= (px >= n) ? n : 0;
    }
    base[py * stride + px] = v;
    if (px < RING_PAD) base[py * stride + n + px] = v;     // mirror columns
    if (snew) snew[r] = v;                                 // (k_extrude_sg: the next stencil's points on this line)
  }
  nox = ox; noy = oy;
  if (dir == 1) nox = (ox + 1 >= n) ? 0 : ox + 1;
  else if (dir == -1) nox = (ox - 1 < 0) ? n - 1 : ox - 1;
  else if (dir == 2) noy = (oy + 1 >= n) ? 0 : oy + 1;
  else noy = (oy - 1 < 0) ? n - 1 : oy - 1;
}

// new line -> ring, then the ring origin / extrusion counter of this (environment, layer) advance:
// one block per column, so nobody else reads that origin
__global__ __launch_bounds__(256) void k_extrude_scatter(DevSys sys, DevState st, int env_begin,
                                                         RoundOps ops,
                                                         const float *__restrict__ NEWL, int ldn,
                                                         const float *__restrict__ ZREF,
                                                         const float *__restrict__ P, int nsplit,
                                                         int ncol, int pn, float pscale) {
  ATM_SETPRIO();
  const int col = blockIdx.x;
  const int e = env_begin + col / ops.nops, li = ops.layer[col % ops.nops];
  int *o = st.origin + (e * sys.nlayers + li) * 2;
  const int ox = o[0], oy = o[1];
  int nox, noy;
  extrude_scatter_col(sys, st, env_begin, ops, NEWL, ldn, ZREF, P, nsplit, ncol, pn, pscale, col, ox, oy, nox, noy);
  __syncthreads();
  if (threadIdx.x == 0) {
    o[0] = nox; o[1] = noy;
    if (st.origin_snap) { st.origin_snap[(e * sys.nlayers + li) * 2] = nox; st.origin_snap[(e * sys.nlayers + li) * 2 + 1] = noy; }
    st.ext_count[e * sys.nlayers + li] += 1u;
  }
}

// scatter of round r and gather of round r + 1 in one launch: the dependency is column-local -- a
// column's next stencil reads only that column's screen, including the line just written -- so one
// block does both, with the advanced origin and counter carried in registers (never re-read: a
// uniform re-load could come from the scalar cache, which the block's own stores do not update).
// The next round may hold fewer layers (a frame's rounds thin out as the slower layers finish) and other
// directions: `nx` says where this column's layer sits in it; its Z / ZREF are a second pair of buffers
// (ZN / ZREFN: the next round numbers its columns anew, another block may still need this round's ZREF entry).
#define SG_MAX_N 1024                            // longest line k_extrude_sg keeps in LDS (checked on the host)
#ifndef SG_EARLY
#define SG_EARLY 1                               // 0 (A/B builds): every stencil value read back from the ring behind the barrier, as before round 6
#endif
#ifndef SG_THREADS
#define SG_THREADS 512                           // threads per column; SG_U stencil items per thread: ns <= SG_U * SG_THREADS
#define SG_U 4
#endif
__global__ __launch_bounds__(SG_THREADS) void k_extrude_sg(DevSys sys, DevState st, int env_begin, RoundOps ops,
                                                    const float *__restrict__ NEWL, int ldn,
                                                    const float *__restrict__ ZREF, const float *__restrict__ P,
                                                    int nsplit, int ncol, int pn, float pscale,
                                                    float *__restrict__ Z, int ldz, float *__restrict__ ZREFN,
                                                    RoundNext nx) {
  __shared__ float snew[SG_MAX_N];               // the line this block writes, for the next stencil's points on it
  ATM_SETPRIO();
  const int col = blockIdx.x;
  const int el = col / ops.nops, oi = col % ops.nops;
  const int e = env_begin + el, li = ops.layer[oi];
  int *o = st.origin + (e * sys.nlayers + li) * 2;
  const int ox = o[0], oy = o[1];
  const uint32_t cnt = st.ext_count[e * sys.nlayers + li];
  const DevLayer &L = sys.layers[li];
  const int n = L.dim, ns = L.ns;
  const int ni = nx.idx[oi];                     // this column's layer in the next round (-1: it has no operation there)
  const int col2 = el * nx.nops + (ni < 0 ? 0 : ni);
  const int dir = nx.dir[ni < 0 ? 0 : ni];
  // the origin as this round's scatter leaves it (the same arithmetic as extrude_scatter_col's)
  const int cdir = ops.dir[oi];
  int eox = ox, eoy = oy;
  if (cdir == 1) eox = (ox + 1 >= n) ? 0 : ox + 1;
  else if (cdir == -1) eox = (ox - 1 < 0) ? n - 1 : ox - 1;
  else if (cdir == 2) eoy = (oy + 1 >= n) ? 0 : oy + 1;
  else eoy = (oy - 1 < 0) ? n - 1 : oy - 1;
  // Is the logical point (x, y) of the ADVANCED screen on the line this block is about to write, and which of its
  // elements is it?  (dir +1: the new column is x = n - 1, element y; -1: x = 0, element n - 1 - y; +2: the row y = n - 1,
  // element x; -2: y = 0, element n - 1 - x -- the placement rules of extrude_scatter_col)
  auto on_new = [&](int x, int y, int &r) {
    if (cdir == 1) { r = y; return x == n - 1; }
    if (cdir == -1) { r = n - 1 - y; return x == 0; }
    if (cdir == 2) { r = x; return y == n - 1; }
    r = n - 1 - x; return y == 0;
  };
  // What the next round's gather needs and the scatter does not touch goes FIRST, so that it runs while the scatter's
  // loads are in flight instead of behind the barrier: the stencil's index list (the gather was a dependent pair of
  // loads per item), the noise half of Z (Philox + two Box-Muller pairs per item: nothing but arithmetic) and -- round
  // 6 -- the stencil's VALUES off the new line: half of the stencil (its second full line and the far points) is older
  // data that this block does not write, and the other half IS the line it writes, which it keeps in LDS: no value is
  // read back from the ring behind the barrier, one dependent round trip to memory less per launch.  Z / ZREFN are the
  // next round's buffers: the product that last read them is a whole round back.
  constexpr int U = SG_U;                        // ns <= U * blockDim.x (checked on the host: AOMARL_SG_MAX_NS)
  const float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
  float v[U] = {};
  int rr[U] = {};
  bool nw[U] = {};
  float zref = 0.f;
  int zr = 0;
  bool znew = false;
  if (ni >= 0) {
    const uint32_t *ist = nx.tflag[ni] ? L.istT : ((dir == 1 || dir == -1) ? L.istx : L.isty);
    uint32_t xy[U];
#pragma unroll
    for (int u = 0; u < U; u++) xy[u] = ist[min((int)threadIdx.x + u * (int)blockDim.x, ns - 1)];
    const bool top_right = (dir == 1 || dir == -2);
    const int zx = top_right ? n - 1 : 0, zy = top_right ? 0 : n - 1;
    znew = on_new(zx, zy, zr);
    if (SG_EARLY && !znew) zref = base[ring_idx(zx, zy, eox, eoy, n)];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int x = xy[u] & 0xFFFF, y = xy[u] >> 16;
      nw[u] = on_new(x, y, rr[u]);
      if (SG_EARLY && !nw[u]) v[u] = base[ring_idx(x, y, eox, eoy, n)];
      if (!SG_EARLY) rr[u] = ring_idx(x, y, eox, eoy, n);
    }
    if (!SG_EARLY) zr = ring_idx(zx, zy, eox, eoy, n);
    const uint32_t seed = st.seeds[e] + (uint32_t)li;
    for (int g = threadIdx.x; g < (n + 3) / 4; g += blockDim.x) {
      float z4[4];
      philox_normal4(seed, 0u, cnt + 1u, 0u, (uint32_t)g, z4);
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (4 * g + u < n) Z[(long long)col2 * ldz + ns + 4 * g + u] = L.amp * z4[u];
    }
  }
  int nox, noy;
  extrude_scatter_col(sys, st, env_begin, ops, NEWL, ldn, ZREF, P, nsplit, ncol, pn, pscale, col, ox, oy, nox, noy, snew);
  __syncthreads();                               // the new line is in the ring and in LDS; ZREF[col] was read by everyone
  if (threadIdx.x == 0) {
    o[0] = nox; o[1] = noy;
    if (st.origin_snap) { st.origin_snap[(e * sys.nlayers + li) * 2] = nox; st.origin_snap[(e * sys.nlayers + li) * 2 + 1] = noy; }
    st.ext_count[e * sys.nlayers + li] = cnt + 1u;
  }
  if (ni < 0) return;
  if (SG_EARLY) {
    if (znew) zref = snew[zr];
  } else {
    zref = base[zr];
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int j = (int)threadIdx.x + u * (int)blockDim.x;
    if (SG_EARLY) { if (nw[u]) v[u] = snew[rr[u]]; }
    else v[u] = base[rr[u]];
    if (j < ns) Z[(long long)col2 * ldz + j] = v[u] - zref;
  }
  if (threadIdx.x == 0) ZREFN[col2] = zref;
}

// Small screens (dim <= 256, ns + dim <= MOVE_SMALL_K): ALL the extrusions of one frame's move in ONE launch.
// One block per (environment, layer): per extrusion it gathers Z = [stencil - zref | amp N(0,1)] into LDS,
// forms the new line zref + [A|B] Z from the transposed matrix (thread = (row, K slice), coalesced over
// rows, the K slices summed in a fixed order through LDS), writes it into the ring and goes on with the
// origin and counter it carries in registers: |kx| x-extrusions, then |ky| y-extrusions, as run_plan
// orders them.  The matrix (a few hundred KB) is re-read from L2 by every block, which is what limits
// this to small screens: there the step is bound by the host's launches (configs[1]: 6 - 12 launches of
// the generic rounds become 1), not by the arithmetic.  fp32 vector FMAs in both precision modes.
constexpr int MOVE_SMALL_K = 4096, MOVE_SMALL_DIM = 256, MOVE_SMALL_T = 512;
struct MovePlan { int kx[AOMARL_MAX_LAYERS], ky[AOMARL_MAX_LAYERS]; };

__global__ __launch_bounds__(MOVE_SMALL_T) void k_move_small(DevSys sys, DevState st, int env_begin, MovePlan plan) {
  __shared__ float Zs[MOVE_SMALL_K];
  __shared__ __attribute__((aligned(16))) float Ps[4 * MOVE_SMALL_T];     // parts x ldt partial sums
  const int e = env_begin + blockIdx.x, li = blockIdx.y, tid = threadIdx.x;
  const int kx = plan.kx[li], ky = plan.ky[li];
  const int nx = abs(kx), nit = nx + abs(ky);
  if (nit == 0) return;
  const DevLayer &L = sys.layers[li];
  const int n = L.dim, ns = L.ns, K = n + ns, stride = n + RING_PAD, ldt = L.ldt;
  // thread = (group of 4 rows, K slice): one 16-byte load of the transposed matrix per k, ldt / 4 row groups,
  // 4 * MOVE_SMALL_T / ldt slices (8 .. 32): many independent loads in flight per thread instead of one per row
  const int rgs = ldt >> 2, parts = MOVE_SMALL_T / rgs, part = tid / rgs, rg = tid - part * rgs;
  float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
  int *o = st.origin + (e * sys.nlayers + li) * 2;
  int ox = o[0], oy = o[1];
  uint32_t cnt = st.ext_count[e * sys.nlayers + li];
  const uint32_t seed = st.seeds[e] + (uint32_t)li;
  const float amp = L.amp;
  const float *__restrict__ ABt = L.ABt;
  for (int it = 0; it < nit; it++) {
    const int dir = it < nx ? (kx > 0 ? 1 : -1) : (ky > 0 ? 2 : -2);
    const bool top_right = (dir == 1 || dir == -2);
    // the ring is rewritten by this block between iterations: a plain vector load, never the scalar cache
    const float zref = __hip_atomic_load(base + ring_idx(top_right ? n - 1 : 0, top_right ? 0 : n - 1, ox, oy, n),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t *ist = (dir == 1 || dir == -1) ? L.istx : L.isty;
    for (int j = tid; j < ns; j += MOVE_SMALL_T) {
      const uint32_t xy = ist[j];
      Zs[j] = __hip_atomic_load(base + ring_idx(xy & 0xFFFF, xy >> 16, ox, oy, n), __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_WORKGROUP) - zref;
    }
    for (int g = tid; g < (n + 3) / 4; g += MOVE_SMALL_T) {
      float z4[4];
      philox_normal4(seed, 0u, cnt, 0u, (uint32_t)g, z4);
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (4 * g + u < n) Zs[ns + 4 * g + u] = amp * z4[u];
    }
    __syncthreads();
    if (part < parts) {
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
      const float4 *p = reinterpret_cast<const float4 *>(ABt) + rg;
      auto acc = [](float4 &a, const float4 m, float z) {
        a.x = fmaf(m.x, z, a.x); a.y = fmaf(m.y, z, a.y); a.z = fmaf(m.z, z, a.z); a.w = fmaf(m.w, z, a.w);
      };
      int j = part;
      for (; j + 3 * parts < K; j += 4 * parts) {
        const float4 m0 = p[(long long)j * rgs], m1 = p[(long long)(j + parts) * rgs];
        const float4 m2 = p[(long long)(j + 2 * parts) * rgs], m3 = p[(long long)(j + 3 * parts) * rgs];
        acc(a0, m0, Zs[j]); acc(a1, m1, Zs[j + parts]); acc(a2, m2, Zs[j + 2 * parts]); acc(a3, m3, Zs[j + 3 * parts]);
      }
      for (; j < K; j += parts) acc(a0, p[(long long)j * rgs], Zs[j]);
      float4 t;
      t.x = (a0.x + a1.x) + (a2.x + a3.x); t.y = (a0.y + a1.y) + (a2.y + a3.y);
      t.z = (a0.z + a1.z) + (a2.z + a3.z); t.w = (a0.w + a1.w) + (a2.w + a3.w);
      reinterpret_cast<float4 *>(Ps)[part * rgs + rg] = t;
    }
    __syncthreads();
    if (tid < n) {
      float v = 0.f;
      for (int q = 0; q < parts; q++) v += Ps[q * ldt + tid];
      v += zref;
      const int r = tid;
      int px, py;
      if (dir == 1) {
        px = ox;
        py = r + oy; py -= (py >= n) ? n : 0;
      } else if (dir == -1) {
        px = ox - 1; px += (px < 0) ? n : 0;
        py = n - 1 - r + oy; py -= (py >= n) ? n : 0;
      } else if (dir == 2) {
        py = oy;
        px = r + ox; px -= (px >= n) ? n : 0;
      } else {
        py = oy - 1; py += (py < 0) ? n : 0;
        px = n - 1 - r + ox; px -= (px >= n) ? n : 0;
      }
      base[py * stride + px] = v;
      if (px < RING_PAD) base[py * stride + n + px] = v;     // mirror columns
    }
    if (dir == 1) ox = (ox + 1 >= n) ? 0 : ox + 1;
    else if (dir == -1) ox = (ox - 1 < 0) ? n - 1 : ox - 1;
    else if (dir == 2) oy = (oy + 1 >= n) ? 0 : oy + 1;
    else oy = (oy - 1 < 0) ? n - 1 : oy - 1;
    cnt += 1u;
    __syncthreads();                             // the new line is in the ring (block-visible); Zs / Ps are free again
  }
  if (tid == 0) {
    o[0] = ox; o[1] = oy;
    if (st.origin_snap) { st.origin_snap[(e * sys.nlayers + li) * 2] = ox; st.origin_snap[(e * sys.nlayers + li) * 2 + 1] = oy; }
    st.ext_count[e * sys.nlayers + li] = cnt;
  }
}

// in-place transposition of the n x n ring of (environment, layer li): 32 x 32 tiles, block = the tile
// pair (i, j) / (j, i), i <= j; the ring origin is exchanged with it.  The mirror columns are rebuilt by
// k_refresh_mirror afterwards.
__global__ __launch_bounds__(256) void k_transpose_ring(DevSys sys, DevState st, int env_begin, int li, int T) {
  __shared__ float A[32][33], B[32][33];
  const DevLayer &L = sys.layers[li];
  const int n = L.dim, stride = n + RING_PAD;
  const int e = env_begin + blockIdx.y;
  float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
  // pair index -> (i, j), i <= j (row-major over the upper triangle)
  int k = blockIdx.x, i = 0;
  while (k >= T - i) { k -= T - i; i++; }
  const int j = i + k;
  const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;
  for (int ty = ty0; ty < 32; ty += 8) {
    const int ya = 32 * i + ty, xa = 32 * j + tx;          // tile (i, j): rows of tile row i, columns of tile column j
    const int yb = 32 * j + ty, xb = 32 * i + tx;
    A[ty][tx] = (ya < n && xa < n) ? base[ya * stride + xa] : 0.f;
    B[ty][tx] = (yb < n && xb < n) ? base[yb * stride + xb] : 0.f;
  }
  __syncthreads();
  for (int ty = ty0; ty < 32; ty += 8) {
    const int ya = 32 * i + ty, xa = 32 * j + tx;
    const int yb = 32 * j + ty, xb = 32 * i + tx;
    if (ya < n && xa < n) base[ya * stride + xa] = B[tx][ty];
    if (i != j && yb < n && xb < n) base[yb * stride + xb] = A[tx][ty];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int *o = st.origin + (e * sys.nlayers + li) * 2;
    const int ox = o[0], oy = o[1];
    o[0] = oy; o[1] = ox;
  }
}

__global__ void k_refresh_mirror(DevSys sys, DevState st, int env_begin, int li) {
  const DevLayer &L = sys.layers[li];
  const int n = L.dim, stride = n + RING_PAD;
  const int e = env_begin + blockIdx.y;
  float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n * RING_PAD) return;
  const int y = p / RING_PAD, x = p - y * RING_PAD;
  base[y * stride + n + x] = base[y * stride + x];
}

__global__ void k_fill_f32(float *p, long long n, float v) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

__global__ void k_reset_env(DevSys sys, DevState st, int env_begin, int env_count,
                            const uint32_t *__restrict__ seeds_dev, int ld_actu, int origin_in_y) {
  // small per-env state: origin, counters, integrator vectors
  const int e = env_begin + blockIdx.x;
  if (threadIdx.x == 0) {
    st.seeds[e] = seeds_dev[blockIdx.x];
    st.frame[e] = 0u;
  }
  for (int i = threadIdx.x; i < sys.nlayers; i += blockDim.x) {
#if FW_ALIGN_ORIGIN
    // Where the ring starts is free (the logical screen does not depend on it).  Start it so that the first pupil
    // pixel of a row sits on a 128-byte line once the reset is through: the frame kernel's 32-pixel pieces of a layer
    // that does not move along x (ground layers under a wind along y: this ring origin never changes) then ARE whole
    // lines, for the whole episode.  The reset's 2 dim extrusions bring the origin back to where it started, and its
    // final transposition exchanges x and y (origin_in_y: start the offset in y).
    const int a = sys.fused_ok ? (32 - (sys.layers[i].tox & 31)) & 31 : 0;
#else
    const int a = 0;
#endif
    st.origin[(e * sys.nlayers + i) * 2] = origin_in_y ? 0 : a;
    st.origin[(e * sys.nlayers + i) * 2 + 1] = origin_in_y ? a : 0;
    st.ext_count[e * sys.nlayers + i] = 0u;
  }
  for (int i = threadIdx.x; i < ld_actu; i += blockDim.x) {
    long long o = (long long)e * ld_actu + i;
    st.com[o] = 0.f; st.com1[o] = 0.f; st.com2[o] = 0.f; st.err[o] = 0.f; st.voltage[o] = 0.f;
  }
}

// upload logical screens src [env_count][n*n] (origin reset to 0, mirror columns filled)
__global__ void k_set_screen(DevSys sys, DevState st, int env_begin, int li, const float *src) {
  const DevLayer &L = sys.layers[li];
  const int n = L.dim, stride = n + RING_PAD;
  const int e = env_begin + blockIdx.y;
  float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n * stride; p += gridDim.x * blockDim.x) {
    int y = p / stride, x = p - y * stride;
    base[p] = src[(long long)blockIdx.y * n * n + y * n + (x >= n ? x - n : x)];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st.origin[(e * sys.nlayers + li) * 2] = 0;
    st.origin[(e * sys.nlayers + li) * 2 + 1] = 0;
  }
}

__global__ void k_get_screen(DevSys sys, DevState st, int env_begin, int li, float *dst) {
  const DevLayer &L = sys.layers[li];
  const int n = L.dim;
  const int e = env_begin + blockIdx.y;
  const float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
  const int ox = st.origin[(e * sys.nlayers + li) * 2], oy = st.origin[(e * sys.nlayers + li) * 2 + 1];
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n * n; p += gridDim.x * blockDim.x) {
    int y = p / n, x = p - y * n;
    dst[(long long)blockIdx.y * n * n + p] = base[ring_idx(x, y, ox, oy, n)];
  }
}

// =============================================================================================
// deformable mirrors
// Stack-array shape: materialised per env (dim x dim).  Tip-tilt: NOT materialised -- the two
// commands are kept in the DM's 4-float slot of st->dm_shape and every consumer evaluates
// c0*f0[p] + c1*f1[p] from the static planes (L2/MALL-resident, shared by all environments).
// =============================================================================================
__device__ __forceinline__ float dm_value(const DevSys &sys, const DevState &st, int e, int k, int x,
                                          int y) {
  const DevDm &D = sys.dms[k];
  const float *slot = st.dm_shape + (long long)e * sys.shape_stride + D.shape_off;
  if (D.type == AOMARL_DM_PZT) return slot[y * D.dim + x];
  const float2 f = reinterpret_cast<const float2 *>(D.influ)[y * D.dim + x];
  return slot[0] * f.x + slot[1] * f.y;
}

// tip-tilt mirror k of environment env_begin + row: its two coefficients, and (p == 2) the stack-array
// DM's value at the centre of the pupil grid
__device__ __forceinline__ void dm_shape_tt_body(const DevSys &sys, const DevState &st, int env_begin, int row, int k,
                                                 const float *__restrict__ volts, int ldv, int p) {
  const DevDm &D = sys.dms[k];
  const float *com = volts + (long long)row * ldv + D.com_off;
  float *shape = st.dm_shape + (long long)(env_begin + row) * sys.shape_stride + D.shape_off;
  if (p < 2) shape[p] = com[p];
  if (p == 2 && sys.fused_ok) {
    // stack-array DM value at the centre of the pupil grid (pivot of the one-pass frame kernel's
    // variance sums when the stack-array shape is evaluated from the commands on the fly)
    const DevDm &Z = sys.dms[0];
    const float *zc = volts + (long long)row * ldv + Z.com_off;
    const int half = sys.pupdiam / 2, zp = (half + Z.toy) * Z.dim + half + Z.tox;
    const int ss2 = Z.ss * Z.ss, s0 = Z.influstart[zp], cn = Z.ninflu[zp];
    float acc = 0.f;
    for (int t = 0; t < cn; t++) {
      const int pos = Z.influpos[s0 + t];
      acc += Z.influ[pos] * zc[pos / ss2];
    }
    shape[2] = acc;
  }
}

// generic per-pixel gather over the reference's influpos / ninflu / influstart tables
__global__ __launch_bounds__(256) void k_dm_shape(DevSys sys, DevState st, int env_begin, int k,
                                                  const float *__restrict__ volts, int ldv) {
  const DevDm &D = sys.dms[k];
  const int e = env_begin + blockIdx.y;
  const float *com = volts + (long long)blockIdx.y * ldv + D.com_off;
  float *shape = st.dm_shape + (long long)e * sys.shape_stride + D.shape_off;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (D.type == AOMARL_DM_TT) {
    dm_shape_tt_body(sys, st, env_begin, blockIdx.y, k, volts, ldv, p);
    return;
  }
  if (p >= D.dim * D.dim) return;
  const int ss2 = D.ss * D.ss;
  const int s0 = D.influstart[p], c = D.ninflu[p];
  float acc = 0.f;
  for (int t = 0; t < c; t++) {
    int pos = D.influpos[s0 + t];
    acc += D.influ[pos] * com[pos / ss2];
  }
  shape[p] = acc;
}

// separable lattice fast path: tile of 64 x 32 pixels per block; the <= 9 x 8 lattice cells that
// can reach the tile are staged in LDS; shape = sum_gy u(y - Yg) sum_gx u(x - Xg) C[gy][gx]
#define DMS_TX 64
#define DMS_TY 32
#define DMS_GX 9
#define DMS_GY 8
__global__ __launch_bounds__(256) void k_dm_shape_sep(DevSys sys, DevState st, int env_begin, int k,
                                                      const float *__restrict__ volts, int ldv) {
  __shared__ float sC[DMS_GY][DMS_GX + 1];
  __shared__ float sU[128];
  const DevDm &D = sys.dms[k];
  const int e = env_begin + blockIdx.z;
  const float *com = volts + (long long)blockIdx.z * ldv + D.com_off;
  float *shape = st.dm_shape + (long long)e * sys.shape_stride + D.shape_off;
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * DMS_TX, y0 = blockIdx.y * DMS_TY;
  const int ss = D.ss, pitch = D.pitch;
  // first lattice column / row that can touch the tile: X_g + ss - 1 >= x0
  auto first = [&](int p0, int pmin) {
    int num = p0 - pmin - (ss - 1);
    int g = num >= 0 ? (num + pitch - 1) / pitch : -((-num) / pitch);
    return g;
  };
  const int gxb = first(x0, D.i1min), gyb = first(y0, D.j1min);
  if (tid < ss) sU[tid] = D.prof[tid];
  if (tid < DMS_GY * DMS_GX) {
    const int r = tid / DMS_GX, cc = tid - r * DMS_GX;
    const int gx = gxb + cc, gy = gyb + r;
    float v = 0.f;
    if (gx >= 0 && gx < D.gw && gy >= 0 && gy < D.gh) {
      int a = D.grid[gy * D.gw + gx];
      if (a >= 0) v = com[a];
    }
    sC[r][cc] = v;
  }
  __syncthreads();
  const int x = x0 + (tid & 63);
  float wx[DMS_GX];
#pragma unroll
  for (int cc = 0; cc < DMS_GX; cc++) {
    int a = x - (D.i1min + pitch * (gxb + cc));
    wx[cc] = (a >= 0 && a < ss) ? sU[a] : 0.f;
  }
  float rs[DMS_GY];
#pragma unroll
  for (int r = 0; r < DMS_GY; r++) {
    float acc = 0.f;
#pragma unroll
    for (int cc = 0; cc < DMS_GX; cc++) acc += wx[cc] * sC[r][cc];
    rs[r] = acc;
  }
  if (x >= D.dim) return;
#pragma unroll
  for (int j = 0; j < DMS_TY / 4; j++) {
    const int y = y0 + (tid >> 6) + 4 * j;
    if (y >= D.dim) continue;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < DMS_GY; r++) {
      int a = y - (D.j1min + pitch * (gyb + r));
      float wy = (a >= 0 && a < ss) ? sU[a] : 0.f;
      acc += wy * rs[r];
    }
    shape[y * D.dim + x] = acc;
  }
}

// materialise any DM's shape into dst [env_count][dim*dim] (getter for tests / host API)
__global__ void k_get_dm_shape(DevSys sys, DevState st, int env_begin, int k, float *dst) {
  const DevDm &D = sys.dms[k];
  const int e = env_begin + blockIdx.y;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < D.dim * D.dim; p += gridDim.x * blockDim.x) {
    int y = p / D.dim, x = p - y * D.dim;
    dst[(long long)blockIdx.y * D.dim * D.dim + p] = dm_value(sys, st, e, k, x, y);
  }
}

// =============================================================================================
// raytrace into a phase buffer (generic, bilinear) -- unfused API path
// =============================================================================================
__device__ __forceinline__ float bilinear_plain(const float *in, int N, float fx, float fy) {
  int ix = (int)floorf(fx), iy = (int)floorf(fy);
  float wx = fx - (float)ix, wy = fy - (float)iy;
  if (ix < 0 || iy < 0 || ix >= N || iy >= N) return 0.f;
  int ix1 = ix + 1 < N ? ix + 1 : ix, iy1 = iy + 1 < N ? iy + 1 : iy;
  float v00 = in[iy * N + ix], v01 = in[iy * N + ix1], v10 = in[iy1 * N + ix], v11 = in[iy1 * N + ix1];
  return (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
}

__device__ __forceinline__ float bilinear_ring(const float *in, int N, int ox, int oy, float fx,
                                               float fy) {
  int ix = (int)floorf(fx), iy = (int)floorf(fy);
  float wx = fx - (float)ix, wy = fy - (float)iy;
  if (ix < 0 || iy < 0 || ix >= N || iy >= N) return 0.f;
  int ix1 = ix + 1 < N ? ix + 1 : ix, iy1 = iy + 1 < N ? iy + 1 : iy;
  float v00 = in[ring_idx(ix, iy, ox, oy, N)], v01 = in[ring_idx(ix1, iy, ox, oy, N)];
  float v10 = in[ring_idx(ix, iy1, ox, oy, N)], v11 = in[ring_idx(ix1, iy1, ox, oy, N)];
  return (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
}

template <bool TARGET>
__global__ __launch_bounds__(256) void k_raytrace(DevSys sys, DevState st, int env_begin,
                                                  int flags) {
  const int nn = TARGET ? sys.pupdiam : sys.n;
  const int e = env_begin + blockIdx.y;
  float *out = (TARGET ? st.tar_phase : st.wfs_phase) + (long long)e * nn * nn;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nn * nn) return;
  const int y = p / nn, x = p - y * nn;
  float v = (flags & AOMARL_TRACE_RESET) ? 0.f : out[p];
  if (flags & AOMARL_TRACE_ATMOS) {
    for (int l = 0; l < sys.nlayers; l++) {
      const DevLayer &L = sys.layers[l];
      const float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
      const int ox = st.origin[(e * sys.nlayers + l) * 2], oy = st.origin[(e * sys.nlayers + l) * 2 + 1];
      v += bilinear_ring(base, L.dim, ox, oy, (float)x + (TARGET ? L.txo : L.wxo),
                         (float)y + (TARGET ? L.tyo : L.wyo));
    }
  }
  if (flags & AOMARL_TRACE_DMS) {
    for (int k = 0; k < sys.ndm; k++) {
      const DevDm &D = sys.dms[k];
      const float fx = (float)x + (TARGET ? D.txo : D.wxo), fy = (float)y + (TARGET ? D.tyo : D.wyo);
      const int ix = (int)floorf(fx), iy = (int)floorf(fy);
      if (ix >= 0 && iy >= 0 && ix < D.dim && iy < D.dim) {
        const float wx = fx - (float)ix, wy = fy - (float)iy;
        const int ix1 = ix + 1 < D.dim ? ix + 1 : ix, iy1 = iy + 1 < D.dim ? iy + 1 : iy;
        float v00 = dm_value(sys, st, e, k, ix, iy);
        if (wx == 0.f && wy == 0.f) {
          v += v00;
        } else {
          float v01 = dm_value(sys, st, e, k, ix1, iy), v10 = dm_value(sys, st, e, k, ix, iy1);
          float v11 = dm_value(sys, st, e, k, ix1, iy1);
          v += (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
        }
      }
    }
  }
  if (flags & AOMARL_TRACE_MASK) v *= (TARGET ? sys.spupil : sys.mpupil)[p];
  out[p] = v;
}

// =============================================================================================
// Shack-Hartmann spot image + centre of gravity.
// One wavefront per (environment, sub-aperture); specialised for the production sampling
// pdiam = 16 phase pixels, Nfft = 64, nrebin = 2, npix = 16 (SURVEY Appendix A / golden fixtures).
//
// The 16x16 complex pupil tile is staged in LDS; the zero-padded 64x64 FFT is never formed:
// only the 32x32 central frequencies feed the 16x16 binned image, and the input has 16 non-zero
// rows/columns, so the transform is the pruned DFT  X = E^T . a . E  with E[16][32] =
// exp(-2 pi i x k / 64), k in [-16, 16): two small complex GEMMs (96 v_mfma_f32_16x16x4_f32).
// |X|^2, the 2x2 binning (binmap), the flux normalisation, optional noise and the COG are done on
// the accumulator registers.
// =============================================================================================
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// An all-zero accumulator the compiler cannot rematerialise: MFMAs that start a chain read it as
// their C operand (dst != srcC form) instead of zero-filling a fresh accumulator with v_mov each
// time (fp32 MFMAs and VALU instructions share the issue slots, so every v_mov counts).
__device__ __forceinline__ f32x4 opaque_zero4() {
  float z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  f32x4 r = {z, z, z, z};
  return r;
}

__device__ __forceinline__ float poisson_draw(float lam, float u, float zn) {
  if (!(lam > 0.f)) return 0.f;
  if (lam < 30.f) {
    // hardware exp2 / reciprocal (1 ulp): the cumulative sums move in their last bit against the
    // oracle's libm, which flips a count where u sits within ~1e-7 of a threshold
#ifdef AOMARL_LIBM_NOISE
    float p = expf(-lam), c = p;
    int k = 0;
    while (u > c && k < 200) {
      k++;
      p *= lam / (float)k;
      c += p;
    }
#else
    float p = __expf(-lam), c = p;
    int k = 0;
    while (u > c && k < 200) {
      k++;
      p *= lam * __builtin_amdgcn_rcpf((float)k);
      c += p;
    }
#endif
    return (float)k;
  }
  float v = floorf(lam + sqrtf(lam) * zn + 0.5f);
  return v < 0.f ? 0.f : v;
}

// photon + read noise of one pixel: ONE Philox block per (pixel, frame) -- word 0: uniform of the
// Poisson inversion, words 1, 2: a Box-Muller pair (cosine branch: rounded-normal Poisson for
// lam >= 30, sine branch: read noise).  Same rule as oracle/aoref.c:aoref_sh_noise.
__device__ __forceinline__ float sh_noise(float lam, float sigma, uint32_t seed, uint32_t frame, uint32_t idx) {
  uint32_t x[4];
  philox4x32_10(idx, frame, 0u, 4u, seed, 0x414F4D52u, x);
  // Box-Muller on the transcendental unit: log2, and sin / cos of an angle given in revolutions
  // (v_sin_f32 / v_cos_f32 take exactly that) -- ~6 instructions against ~150 for libm's logf, sinf,
  // cosf; the normals agree with the oracle's to ~1e-6
#ifdef AOMARL_LIBM_NOISE
  const float r = sqrtf(-2.0f * logf(u01(x[1])));
  const float a = 6.28318530717958647692f * u01(x[2]);
  float v = poisson_draw(lam, u01(x[0]), r * cosf(a));
  if (sigma > 0.f) v += sigma * (r * sinf(a));
  return v;
#else
  const float r = sqrtf(-2.0f * __logf(u01(x[1])));
  const float t = u01(x[2]);
  float v = poisson_draw(lam, u01(x[0]), r * __builtin_amdgcn_cosf(t));
  if (sigma > 0.f) v += sigma * (r * __builtin_amdgcn_sinf(t));
  return v;
#endif
}

// MFMA stages + binning + normalisation (+noise) + COG of one sub-aperture whose complex amplitude
// tile b = mask * exp(i phi 2pi/lambda) is in sAr/sAi[wv]; shared by the generic and fast kernels.
//
// The 32 kept frequencies [-16, 15] of the 64-point transform of  a = b * exp(-i pi (x+y)/64)
// (the reference's half-pixel ramp `halfxy`, geom_init.py:689-701) are the HALF-INTEGER
// frequencies kappa = +-(k + 1/2), k = 0..15, of b itself:  kx = k  <->  +(k+1/2),
// kx = -(k+1)  <->  -(k+1/2).  The +- pairs share cos and differ in the sign of sin, so with
//   C[x][k] = cos(2 pi (2k+1) x / 128),  S[x][k] = sin(2 pi (2k+1) x / 128)      (16 x 16, real)
// stage 1 needs the 4 real products  b_r C, b_i C, b_r S, b_i S  (16 MFMAs) and gives both
// T+ = (PCr + PSi, PCi - PSr) and T- = (PCr - PSi, PCi + PSr); stage 2 does the same along y for
// T+ and T- (32 MFMAs): 48 v_mfma_f32_16x16x4_f32 instead of 96 for the plain complex products.
// Accumulator layout of stage 1 (row y = 4q + reg, col k = c) is again exactly the B-operand
// K-order stage 2 wants, and the twiddle registers serve as B operand (stage 1) and A operand
// (stage 2) alike, so nothing moves between the stages.
// ---------------------------------------------------------------------------------------------
// Split-fp16 operands for the real matrix cores.  v_mfma_f32_16x16x32_f16 costs 16 cycles against
// the 4 x 32 of the four fp32 MFMAs it replaces (fp32 MFMAs run at the vector rate).
// A value v is carried as hi = f16(v), lo = f16(v - hi): 22 mantissa bits.  The K = 32 slots of
// one instruction hold the 16 real terms twice, so
//     mfma(pack(a), dup_hi(b)) + mfma(pack(a), dup_lo(b))     with pack(a) = [a_hi | a_lo]
// is the full (a_hi + a_lo)(b_hi + b_lo) product, accumulated in fp32.  Lane (q, c) owns the 4
// values (row c, k = 4q .. 4q+3) of an operand -- for the amplitude tile those are exactly the 4
// pixels the lane computed, and for stage 2 exactly its 4 accumulator registers of stage 1: no LDS.
// ---------------------------------------------------------------------------------------------
// [hi(a0..a3) | lo(a0..a3)]
__device__ __forceinline__ hx8 pack_hl(float a0, float a1, float a2, float a3) {
  const hx2 h01 = cvt_h2(a0, a1), h23 = cvt_h2(a2, a3);
  const hx2 l01 = cvt_h2(sub_lo(h01, a0), sub_hi(h01, a1));
  const hx2 l23 = cvt_h2(sub_lo(h23, a2), sub_hi(h23, a3));
  return hx8{h01[0], h01[1], h23[0], h23[1], l01[0], l01[1], l23[0], l23[1]};
}
// [hi | hi] and [lo | lo]
__device__ __forceinline__ void dup_hl(float a0, float a1, float a2, float a3, hx8 &H, hx8 &L) {
  const hx2 h01 = cvt_h2(a0, a1), h23 = cvt_h2(a2, a3);
  const hx2 l01 = cvt_h2(sub_lo(h01, a0), sub_hi(h01, a1));
  const hx2 l23 = cvt_h2(sub_lo(h23, a2), sub_hi(h23, a3));
  H = hx8{h01[0], h01[1], h23[0], h23[1], h01[0], h01[1], h23[0], h23[1]};
  L = hx8{l01[0], l01[1], l23[0], l23[1], l01[0], l01[1], l23[0], l23[1]};
}
// twiddle operands of the spot DFT in split-fp16 form (constant per lane)
struct SpotTwH { hx8 CH, CL, SH, SL; };
__device__ __forceinline__ SpotTwH spot_tw_h(const float (&Cc)[4], const float (&Ss)[4]) {
  SpotTwH t;
  dup_hl(Cc[0], Cc[1], Cc[2], Cc[3], t.CH, t.CL);
  dup_hl(Ss[0], Ss[1], Ss[2], Ss[3], t.SH, t.SL);
  return t;
}

// the two DFT stages on fp32 MFMAs: br[s] / bi[s] = complex amplitude of pixel (y = c, x = 4q + s)
__device__ __forceinline__ void spot_dft_f32(const float (&Cc)[4], const float (&Ss)[4],
                                             const float (&br)[4], const float (&bi)[4],
                                             const f32x4 z4, f32x4 (&Xr)[2][2], f32x4 (&Xi)[2][2]) {
  // ---- stage 1 (y = c on M, x = 4q + s on K, k = c on N)
  f32x4 PCr = z4, PCi = z4, PSr = z4, PSi = z4;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    PCr = mfma16(br[s], Cc[s], PCr);
    PCi = mfma16(bi[s], Cc[s], PCi);
    PSr = mfma16(br[s], Ss[s], PSr);
    PSi = mfma16(bi[s], Ss[s], PSi);
  }
  f32x4 Tr[2], Ti[2];                 // [0]: kx = +(k+1/2)   [1]: kx = -(k+1/2)
  Tr[0] = PCr + PSi; Ti[0] = PCi - PSr;
  Tr[1] = PCr - PSi; Ti[1] = PCi + PSr;
  // ---- stage 2 (ky' = c on M, y = 4q + s on K = accumulator reg s, kx' on N)
#pragma unroll
  for (int m = 0; m < 2; m++) {
    f32x4 QCr = z4, QCi = z4, QSr = z4, QSi = z4;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      QCr = mfma16(Cc[s], Tr[m][s], QCr);
      QCi = mfma16(Cc[s], Ti[m][s], QCi);
      QSr = mfma16(Ss[s], Tr[m][s], QSr);
      QSi = mfma16(Ss[s], Ti[m][s], QSi);
    }
    Xr[0][m] = QCr + QSi; Xi[0][m] = QCi - QSr;
    Xr[1][m] = QCr - QSi; Xi[1][m] = QCi + QSr;
  }
}

// |X|^2 and 2x2 binning of one quadrant tile.  Reg r, lane (q, c): ky' = 4q + r, kx' = c;
// + tiles: k = k' -> LR index 8 + (k' >> 1);  - tiles: k = -(k'+1) -> LR index 7 - (k' >> 1)
__device__ __forceinline__ void spot_bin(const f32x4 Xr, const f32x4 Xi, float (&v)[2]) {
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const float a0 = Xr[2 * h], b0 = Xi[2 * h], a1 = Xr[2 * h + 1], b1 = Xi[2 * h + 1];
    v[h] = add_xor1((a0 * a0 + b0 * b0) + (a1 * a1 + b1 * b1));
  }
}

// the split-fp16 DFT with the binning folded in per kx-sign half: only 2 of the 4 quadrant tiles
// are ever live (32 accumulator registers instead of 64)
__device__ __forceinline__ void spot_dft_h_v(const SpotTwH &tw, const float (&br)[4],
                                             const float (&bi)[4], const f32x4 z4,
                                             float (&v)[2][2][2]) {
  const hx8 ar = pack_hl(br[0], br[1], br[2], br[3]), ai = pack_hl(bi[0], bi[1], bi[2], bi[3]);
  f32x4 PCr = mfma_h(ar, tw.CL, mfma_h(ar, tw.CH, z4));
  f32x4 PCi = mfma_h(ai, tw.CL, mfma_h(ai, tw.CH, z4));
  f32x4 PSr = mfma_h(ar, tw.SL, mfma_h(ar, tw.SH, z4));
  f32x4 PSi = mfma_h(ai, tw.SL, mfma_h(ai, tw.SH, z4));
#pragma unroll
  for (int m = 0; m < 2; m++) {                // [0]: kx = +(k+1/2)   [1]: kx = -(k+1/2)
    const f32x4 Tr = m == 0 ? PCr + PSi : PCr - PSi;
    const f32x4 Ti = m == 0 ? PCi - PSr : PCi + PSr;
    const hx8 tr = pack_hl(Tr[0], Tr[1], Tr[2], Tr[3]);
    const hx8 ti = pack_hl(Ti[0], Ti[1], Ti[2], Ti[3]);
    const f32x4 QCr = mfma_h(tw.CL, tr, mfma_h(tw.CH, tr, z4));
    const f32x4 QCi = mfma_h(tw.CL, ti, mfma_h(tw.CH, ti, z4));
    const f32x4 QSr = mfma_h(tw.SL, tr, mfma_h(tw.SH, tr, z4));
    const f32x4 QSi = mfma_h(tw.SL, ti, mfma_h(tw.SH, ti, z4));
    // X(+ky) = (QCr + QSi, QCi - QSr), X(-ky) = (QCr - QSi, QCi + QSr):
    //   |X(+-ky)|^2 = P +- 2 D,  P = QCr^2 + QSi^2 + QCi^2 + QSr^2,  D = QCr QSi - QCi QSr
    // and the 2 x 2 binning adds register pairs (2h, 2h + 1) and lane pairs (c, c ^ 1): P and D are
    // summed over the register pair first (14 instructions per pair for both signs instead of 20)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int a = 2 * h, b = 2 * h + 1;
      float P = QCr[a] * QCr[a];
      P = fmaf(QSi[a], QSi[a], P); P = fmaf(QCi[a], QCi[a], P); P = fmaf(QSr[a], QSr[a], P);
      P = fmaf(QCr[b], QCr[b], P); P = fmaf(QSi[b], QSi[b], P); P = fmaf(QCi[b], QCi[b], P); P = fmaf(QSr[b], QSr[b], P);
      float D = QCr[a] * QSi[a];
      D = fmaf(-QCi[a], QSr[a], D); D = fmaf(QCr[b], QSi[b], D); D = fmaf(-QCi[b], QSr[b], D);
      v[0][m][h] = add_xor1(fmaf(2.f, D, P));
      v[1][m][h] = add_xor1(fmaf(-2.f, D, P));
    }
  }
}

// Slopes only (no noise, no image wanted), fp32: the centre of gravity straight from the stage-2 products.
// With X(+-ky) = (QCr +- QSi, QCi -+ QSr):  |X(+-ky)|^2 = P +- 2 D,  P = QCr^2 + QSi^2 + QCi^2 + QSr^2,
// D = QCr QSi - QCi QSr.  The image is never formed: the 2 x 2 binning only gives pairs of frequencies the same
// pixel coordinate, so the three moments are weighted sums of P and D with per-lane constants --
//   pixel row of +ky: 8 + 2q + (r >> 1), of -ky: 7 - 2q - (r >> 1): their sum is 15, their difference
//   1 + 4q + 2 (r >> 1); pixel column of +kx: Xp = 8 + (c >> 1), of -kx: Xm = 7 - (c >> 1) --
//   sum I   = 2 (P0 + P1)                    (Pm: P summed over the lane's 4 registers of half m)
//   sum x I = 2 (Xp P0 + Xm P1)
//   sum y I = 15 (P0 + P1) + 2 [(1 + 4q) (D over r = 0, 1) + (3 + 4q) (D over r = 2, 3)]
// (the common factor 2 drops out of the ratios): 48 multiply-adds + 8 for the moments, against 125 vector
// instructions for combine / square / bin / weight -- on fp32 matrix instructions vector and matrix work
// share the issue slots, so every one of them is kernel time.
__device__ __forceinline__ void spot_cog_f32(const DevSys &sys, const DevState &st, int e, int i, int lane,
                                             const float (&Cc)[4], const float (&Ss)[4], const float (&br)[4],
                                             const float (&bi)[4], int do_cog, const f32x4 z4) {
  const int q = lane >> 4, c = lane & 15;
  // ---- stage 1 (y = c on M, x = 4q + s on K, k = c on N)
  f32x4 PCr = z4, PCi = z4, PSr = z4, PSi = z4;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    PCr = mfma16(br[s], Cc[s], PCr);
    PCi = mfma16(bi[s], Cc[s], PCi);
    PSr = mfma16(br[s], Ss[s], PSr);
    PSi = mfma16(bi[s], Ss[s], PSi);
  }
  float Pm[2], Da = 0.f, Db = 0.f;
#pragma unroll
  for (int m = 0; m < 2; m++) {                  // [0]: kx = +(k+1/2)   [1]: kx = -(k+1/2)
    const f32x4 Tr = m == 0 ? PCr + PSi : PCr - PSi;
    const f32x4 Ti = m == 0 ? PCi - PSr : PCi + PSr;
    f32x4 QCr = z4, QCi = z4, QSr = z4, QSi = z4;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      QCr = mfma16(Cc[s], Tr[s], QCr);
      QCi = mfma16(Cc[s], Ti[s], QCi);
      QSr = mfma16(Ss[s], Tr[s], QSr);
      QSi = mfma16(Ss[s], Ti[s], QSi);
    }
    float P = QCr[0] * QCr[0];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (r) P = fmaf(QCr[r], QCr[r], P);
      P = fmaf(QSi[r], QSi[r], P); P = fmaf(QCi[r], QCi[r], P); P = fmaf(QSr[r], QSr[r], P);
    }
    Pm[m] = P;
    Da = fmaf(QCr[0], QSi[0], Da); Da = fmaf(-QCi[0], QSr[0], Da);
    Da = fmaf(QCr[1], QSi[1], Da); Da = fmaf(-QCi[1], QSr[1], Da);
    Db = fmaf(QCr[2], QSi[2], Db); Db = fmaf(-QCi[2], QSr[2], Db);
    Db = fmaf(QCr[3], QSi[3], Db); Db = fmaf(-QCi[3], QSr[3], Db);
  }
  const float Xp = (float)(8 + (c >> 1)), Xm = (float)(7 - (c >> 1));
  float s0 = Pm[0] + Pm[1];
  float sx = fmaf(Xm, Pm[1], Xp * Pm[0]);
  float sy = fmaf((float)(3 + 4 * q), Db, fmaf((float)(1 + 4 * q), Da, 7.5f * s0));
  s0 = wave_sum_last(s0);
  sx = wave_sum_last(sx);
  sy = wave_sum_last(sy);
  if (do_cog && lane == 63) {
    float *sl = st.slopes + (long long)e * sys.nslope;
    if (s0 > 0.f) {
      const float inv = __builtin_amdgcn_rcpf(s0);       // 1 ulp; slopes are compared at 1e-4"
      sl[i] = (sx * inv - sys.cog_offset) * sys.cog_scale;
      sl[sys.nvalid + i] = (sy * inv - sys.cog_offset) * sys.cog_scale;
    } else {
      sl[i] = 0.f;
      sl[sys.nvalid + i] = 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Packed fp32 vector arithmetic, hand-emitted.  On this chip fp32 matrix instructions run at the PACKED-fp32
// vector rate and share the issue slots with the vector unit, so a v_pk_fma_f32 does twice
// the work of a v_fma_f32 for the same 4 issue cycles -- but the compiler's pre-emit peephole splits packed fp32
// operations that follow matrix instructions back into scalar ones (it assumes a separate matrix pipe), and the
// vector work of the frame kernel sits exactly there.  Inline asm keeps them packed.  The price: the compiler's
// hazard recogniser does not look inside inline asm, so the software-visible wait states of gfx950 are kept BY
// HAND -- a result of v_mfma_f32_16x16x4_f32 (8 passes) must not be read by a vector instruction for 10 wait
// states, a transcendental's result not by a non-transcendental for 1: every group of packed instructions whose
// inputs come from matrix or transcendental instructions opens with PK_GUARD_*: a scheduling barrier (everything
// written before it in the source is issued before it) and an s_nop that covers the longest such distance.
// All statements are `asm volatile`: they stay in source order among themselves.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifdef FW_DBG_NONOP                              // (timing experiment: hazards unprotected, garbage results)
#define PK_GUARD_MFMA() do { __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PK_GUARD_MFMA() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 10"); } while (0)
#endif
#define PK_GUARD_TRANS() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 1"); } while (0)
// (a group's results feed matrix instructions: two wait states of margin, although gfx950 documents none)
#define PK_END_TO_MFMA() do { asm volatile("s_nop 1"); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ f32x2 pk_lo(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 pk_hi(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// a * b - c
__device__ __forceinline__ f32x2 pk_fma_nc(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// c - a * b
__device__ __forceinline__ f32x2 pk_fma_na(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
  f32x2 d;
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x4 pk_join(f32x2 lo, f32x2 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3); }
// in-place forms for accumulators that live across branches / loop iterations (an "=v" result is a new register: the
// compiler then copies it back where control flow joins) and forms whose constant operand stays in a scalar register
// pair (a loop-invariant pair in vector registers is re-materialised from its scalars every iteration: one
// v_mov_b64 each; a packed instruction may read one scalar pair)
__device__ __forceinline__ void pk_acc_add(f32x2 &acc, f32x2 x) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x)); }
__device__ __forceinline__ void pk_acc_fma(f32x2 &acc, f32x2 a, f32x2 b) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }
// acc -= a * b
__device__ __forceinline__ void pk_acc_fnma(f32x2 &acc, f32x2 a, f32x2 b) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ f32x2 pk_fma_s(f32x2 a, f32x2 bs, f32x2 c) {
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(bs), "v"(c));
  return d;
}
__device__ __forceinline__ f32x2 pk_fma_nc_s(f32x2 a, f32x2 bs, f32x2 c) {
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(a), "s"(bs), "v"(c));
  return d;
}
__device__ __forceinline__ f32x2 pk_mul_s(f32x2 a, f32x2 bs) {
  f32x2 d;
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "s"(bs));
  return d;
}

// spot_cog_f32 with its vector arithmetic packed (same formulas, the sums in pairs): 8 + 24 packed instructions for
// the stage-1 combinations and the P / D sums instead of 16 + 48 scalar ones.
__device__ __forceinline__ void spot_cog_f32_pk(const DevSys &sys, const DevState &st, int e, int i, int lane,
                                                const float (&Cc)[4], const float (&Ss)[4], const float (&br)[4],
                                                const float (&bi)[4], int do_cog, const f32x4 z4) {
  const int q = lane >> 4, c = lane & 15;
  // ---- stage 1 (y = c on M, x = 4q + s on K, k = c on N)
  f32x4 PCr = z4, PCi = z4, PSr = z4, PSi = z4;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    PCr = mfma16(br[s], Cc[s], PCr);
    PCi = mfma16(bi[s], Cc[s], PCi);
    PSr = mfma16(br[s], Ss[s], PSr);
    PSi = mfma16(bi[s], Ss[s], PSi);
  }
  // [0]: kx = +(k+1/2): (PCr + PSi, PCi - PSr)   [1]: kx = -(k+1/2): (PCr - PSi, PCi + PSr)
  PK_GUARD_MFMA();
  const f32x2 tr0l = pk_add(pk_lo(PCr), pk_lo(PSi)), ti0l = pk_sub(pk_lo(PCi), pk_lo(PSr));
  const f32x2 tr0h = pk_add(pk_hi(PCr), pk_hi(PSi)), ti0h = pk_sub(pk_hi(PCi), pk_hi(PSr));
  const f32x2 tr1l = pk_sub(pk_lo(PCr), pk_lo(PSi)), ti1l = pk_add(pk_lo(PCi), pk_lo(PSr));
  const f32x2 tr1h = pk_sub(pk_hi(PCr), pk_hi(PSi)), ti1h = pk_add(pk_hi(PCi), pk_hi(PSr));
  PK_END_TO_MFMA();
  const f32x4 TrA[2] = {pk_join(tr0l, tr0h), pk_join(tr1l, tr1h)};
  const f32x4 TiA[2] = {pk_join(ti0l, ti0h), pk_join(ti1l, ti1h)};
  f32x2 Pa[2], Pb[2];                            // P of half m in two partial pairs
  f32x2 Da, Db;                                  // D over registers (0, 1) and (2, 3), both halves
#pragma unroll
  for (int m = 0; m < 2; m++) {
    const f32x4 Tr = TrA[m], Ti = TiA[m];
    f32x4 QCr = z4, QCi = z4, QSr = z4, QSi = z4;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      QCr = mfma16(Cc[s], Tr[s], QCr);
      QCi = mfma16(Cc[s], Ti[s], QCi);
      QSr = mfma16(Ss[s], Tr[s], QSr);
      QSi = mfma16(Ss[s], Ti[s], QSi);
    }
    PK_GUARD_MFMA();
    const f32x2 crl = pk_lo(QCr), crh = pk_hi(QCr), cil = pk_lo(QCi), cih = pk_hi(QCi);
    const f32x2 srl = pk_lo(QSr), srh = pk_hi(QSr), sil = pk_lo(QSi), sih = pk_hi(QSi);
    f32x2 pa = pk_mul(crl, crl), pb = pk_mul(crh, crh);
    if (m == 0) { Da = pk_mul(crl, sil); Db = pk_mul(crh, sih); }
    else { pk_acc_fma(Da, crl, sil); pk_acc_fma(Db, crh, sih); }
    pk_acc_fma(pa, sil, sil); pk_acc_fma(pb, sih, sih);
    pk_acc_fnma(Da, cil, srl); pk_acc_fnma(Db, cih, srh);
    pk_acc_fma(pa, cil, cil); pk_acc_fma(pb, cih, cih);
    pk_acc_fma(pa, srl, srl); pk_acc_fma(pb, srh, srh);
    Pa[m] = pa; Pb[m] = pb;
    __builtin_amdgcn_sched_barrier(0);
  }
  const f32x2 P0 = pk_add(Pa[0], Pb[0]), P1 = pk_add(Pa[1], Pb[1]);
  const float Xp = (float)(8 + (c >> 1)), Xm = (float)(7 - (c >> 1));
  const float p0 = P0.x + P0.y, p1 = P1.x + P1.y, da = Da.x + Da.y, db = Db.x + Db.y;
  float s0 = p0 + p1;
  float sx = fmaf(Xm, p1, Xp * p0);
  float sy = fmaf((float)(3 + 4 * q), db, fmaf((float)(1 + 4 * q), da, 7.5f * s0));
  s0 = wave_sum_last(s0);
  sx = wave_sum_last(sx);
  sy = wave_sum_last(sy);
  if (do_cog && lane == 63) {
    float *sl = st.slopes + (long long)e * sys.nslope;
    if (s0 > 0.f) {
      const float inv = __builtin_amdgcn_rcpf(s0);       // 1 ulp; slopes are compared at 1e-4"
      sl[i] = (sx * inv - sys.cog_offset) * sys.cog_scale;
      sl[sys.nvalid + i] = (sy * inv - sys.cog_offset) * sys.cog_scale;
    } else {
      sl[i] = 0.f;
      sl[sys.nvalid + i] = 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Slopes only, fp32, WITHOUT the spot: the three moments of the binned image as quadratic forms of the
// pupil field (28 matrix instructions per sub-aperture instead of 48, 8 packed vector instructions instead of 32).
//
// The binned image covers the central 32 x 32 frequencies f = +-(j + 1/2), j = 0 .. 15 of the half-pixel-shifted
// 64-point transform, every one of them exactly once (npix * nrebin = 32), and the pixel coordinate of a frequency
// depends on one axis only: X(+-(j + 1/2)) = 7.5 +- u_j, u_j = 1/2 + (j >> 1).  A sum  sum_f w_x(fx) w_y(fy) |F(fx, fy)|^2
// with F = sum_{x, y} E[y][x] exp(-2 pi i (fx x + fy y) / 64) is then
//     sum_{x, x', y, y'} E[y][x] conj(E[y'][x']) K_wx[x'][x] K_wy[y'][y],    K_w[x'][x] = sum_f w(f) exp(2 pi i f (x' - x) / 64),
// and for w = 1 the kernel is the real symmetric Toeplitz matrix  M[d] = 2 sum_j cos(2 pi (j + 1/2) d / 64), for the
// odd weight w = +-u_j it is i S with  S[d] = 2 sum_j u_j sin(2 pi (j + 1/2) d / 64)  real and antisymmetric.  With
// E = Er + i Ei (rows y, columns x) the imaginary parts cancel and
//     sum I           = < M, Er M Er^T + Ei M Ei^T >
//     sum (X - 7.5) I =  2 < M, Ei S Er^T >
//     sum (Y - 7.5) I =  2 < S, Ei M Er^T >            (< A, B > = sum_{y', y} A[y'][y] B[y'][y])
// -- the same numbers as transform, |.|^2, 2 x 2 binning and centre of gravity, to fp32 round-off (2e-7 pixels
// against the 64 x 64 FFT in float64, tools/qf_cog_check.py).  Not usable with noise or when the image is wanted.
//
// On the matrix cores: lane (q, c) holds E[y = c][x = 4q + s], which is the A operand of a product E . (..) AND
// the B operand of a product (..) . E^T:  W = M Er^T, M Ei^T, S Er^T  (A = the constant [x' = c][x = 4q + s], 12
// instructions) come out as [x' = 4q + r][y = c], the B operand of  Er W, Ei W  (16 instructions), whose results
// [y' = 4q + r][y = c] meet the same constants again (M symmetric, S antisymmetric) in 8 packed multiply-adds.
struct SpotQf { f32x2 Ml, Mh, Sl, Sh; };          // M[c][4q + s], S[c - (4q + s)], s = 0 .. 3

__device__ __forceinline__ SpotQf spot_qf_consts(int lane, const float2 *sTw /* [128]: cos, sin(2 pi k / 128) */) {
  const int q = lane >> 4, c = lane & 15;
  float m[4], sv[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int d = c - (4 * q + s);
    float a = 0.f, b = 0.f;
    for (int j = 0; j < 16; j++) {
      const float2 w = sTw[((2 * j + 1) * d) & 127];
      a += w.x;
      b = fmaf(0.5f + (float)(j >> 1), w.y, b);
    }
    // exact zeros of the Dirichlet kernel (even d != 0) come out as round-off: clear them
    m[s] = (d != 0 && (d & 1) == 0) ? 0.f : 2.f * a;
    sv[s] = 2.f * b;
  }
  SpotQf k;
  k.Ml = f32x2{m[0], m[1]}; k.Mh = f32x2{m[2], m[3]};
  k.Sl = f32x2{sv[0], sv[1]}; k.Sh = f32x2{sv[2], sv[3]};
  return k;
}

// sys.qf_tab: the constants of every lane, once per context (the frame kernel computed them at the head of every
// workgroup: a 128-entry twiddle table by two of its waves, a barrier, 64 LDS reads and ~250 vector instructions per
// wave).  Same table expression, same function: the same bits.
__global__ __launch_bounds__(128) void k_fill_qf_tab(float *__restrict__ tab) {
  __shared__ float2 sTw[128];
  const int tid = threadIdx.x;
  float sn, cs;
  sincospif((float)tid * (1.0f / 64.0f), &sn, &cs);
  sTw[tid] = make_float2(cs, sn);
  __syncthreads();
  if (tid < 64) {
    const SpotQf k = spot_qf_consts(tid, sTw);
    float *o = tab + 8 * tid;
    o[0] = k.Ml.x; o[1] = k.Ml.y; o[2] = k.Mh.x; o[3] = k.Mh.y;
    o[4] = k.Sl.x; o[5] = k.Sl.y; o[6] = k.Sh.x; o[7] = k.Sh.y;
  }
}

// The three moments of one sub-aperture, summed over the wave: returns z with  row 0 (lanes 0 .. 15): sum I,
// row 1: sum (Y - 7.5) I / 2 (sign: see qf_slopes),  row 2: sum (X - 7.5) I / 2  -- every lane of a row holds the total.
// Round 6: 10 cross-lane instructions instead of 3 x 7 (tools/permlanebench.hip): gfx950's row swaps fold the four
// 16-lane rows of TWO registers into one (v_permlane32_swap: the upper half of the first operand against the lower
// half of the second -- [a_lo | b_lo] + [a_hi | b_hi]; v_permlane16_swap: the odd rows of the first against the even
// rows of the second), so that one register carries all three sums through the four in-row DPP steps; and the two
// products that make sum I share an accumulator (two packed instructions less).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float swap32_add(float a, float b) {
  const u32x2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);      // rows 0, 1: a (rows 0 + 2, 1 + 3); rows 2, 3: b
}
__device__ __forceinline__ float swap16_add(float a, float b) {
  const u32x2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);      // rows: a.0 + a.1, b.0 + b.1, a.2 + a.3, b.2 + b.3
}
__device__ __forceinline__ float spot_qf_moments(const SpotQf &K, const float (&er)[4], const float (&ei)[4], const f32x4 z4) {
  const float Mc[4] = {K.Ml.x, K.Ml.y, K.Mh.x, K.Mh.y}, Sc[4] = {K.Sl.x, K.Sl.y, K.Sh.x, K.Sh.y};
  f32x4 Wr = z4, Wi = z4, V = z4;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    Wr = mfma16(Mc[s], er[s], Wr);               // (M Er^T)[x'][y]
    Wi = mfma16(Mc[s], ei[s], Wi);               // (M Ei^T)[x'][y]
    V = mfma16(Sc[s], er[s], V);                 // (S Er^T)[x'][y]
  }
  f32x4 G1 = z4, G2 = z4, G3 = z4;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    G1 = mfma16(er[s], Wr[s], G1);               // (Er M Er^T + Ei M Ei^T)[y'][y]: both products into one accumulator
    G2 = mfma16(ei[s], Wr[s], G2);               // (Ei M Er^T)[y'][y]
    G3 = mfma16(ei[s], V[s], G3);                // (Ei S Er^T)[y'][y]
    G1 = mfma16(ei[s], Wi[s], G1);
  }
  // constants in the result layout: M[4q + r][c] = M[c][4q + r] = Mc[r];  S[(4q + r) - c] = -Sc[r]
  PK_GUARD_MFMA();
  f32x2 p0 = pk_mul(pk_lo(G1), K.Ml);
  f32x2 px = pk_mul(pk_lo(G3), K.Ml);
  f32x2 py = pk_mul(pk_lo(G2), K.Sl);
  pk_acc_fma(p0, pk_hi(G1), K.Mh);
  pk_acc_fma(px, pk_hi(G3), K.Mh);
  pk_acc_fma(py, pk_hi(G2), K.Sh);
  const float s0 = p0.x + p0.y, tx = px.x + px.y, ty = py.x + py.y;
  // (the second operand of the second fold is a register that is dead by now: rows 2, 3 of that fold -- row 3 of z --
  // are never read, and a swap of ty with itself would need a copy first)
  float z = swap16_add(swap32_add(s0, tx), swap32_add(ty, p0.y));   // rows: s0, ty, tx, (unused)
  z += dpp_f<0xB1>(z);      // quad_perm [1,0,3,2]
  z += dpp_f<0x4E>(z);      // quad_perm [2,3,0,1]
  z += dpp_f<0x141>(z);     // row_half_mirror
  z += dpp_f<0x140>(z);     // row_mirror -> every lane holds its 16-lane row sum
  return z;
}
// the two slopes of a sub-aperture from its moments (sum I, sum (X - 7.5) I / 2, -sum (Y - 7.5) I / 2 as
// spot_qf_moments leaves them)
__device__ __forceinline__ void qf_slopes(const DevSys &sys, float s0, float tx, float ty, float &sx, float &sy) {
  if (s0 > 0.f) {
    const float inv = 2.f * __builtin_amdgcn_rcpf(s0);  // 1 ulp; slopes are compared at 1e-4"
    sx = (fmaf(tx, inv, 7.5f) - sys.cog_offset) * sys.cog_scale;
    sy = (fmaf(-ty, inv, 7.5f) - sys.cog_offset) * sys.cog_scale;
  } else {
    sx = 0.f; sy = 0.f;
  }
}

// flux normalisation (+noise), COG on the binned quadrant values v[sy][sx][h]
template <bool NOISE, bool WRITE_CUBE>
__device__ __forceinline__ void spot_finish_v(const DevSys &sys, const DevState &st, int e, int i,
                                              int lane, const float (&v)[2][2][2], int do_cog,
                                              float flux_i) {
  const int q = lane >> 4, c = lane & 15;
  const bool owner = (c & 1) == 0;
  float tot = 0.f;
#pragma unroll
  for (int sy = 0; sy < 2; sy++)
#pragma unroll
    for (int sx = 0; sx < 2; sx++)
#pragma unroll
      for (int h = 0; h < 2; h++) tot += v[sy][sx][h];
  const int Xp = 8 + (c >> 1), Xm = 7 - (c >> 1);
  if (!NOISE && !WRITE_CUBE) {
    // only the slopes are wanted and nothing depends on the normalised pixel values: the COG is
    // invariant to the flux scale, so reduce (sum, sum x, sum y) of the raw image in ONE pass.  Every
    // binned pixel sits in both lanes of a pair (c, c ^ 1) with the same bits, so the sums over all
    // 64 lanes are exactly twice the image's: the factor drops out of the ratios, no lane is masked.
    const float A0 = (v[0][0][0] + v[0][0][1]) + (v[1][0][0] + v[1][0][1]);        // columns Xp
    const float A1 = (v[0][1][0] + v[0][1][1]) + (v[1][1][0] + v[1][1][1]);        // columns Xm
    float sx_ = (float)Xp * A0;
    sx_ = fmaf((float)Xm, A1, sx_);
    float sy_ = (float)(8 + 2 * q) * (v[0][0][0] + v[0][1][0]);
    sy_ = fmaf((float)(9 + 2 * q), v[0][0][1] + v[0][1][1], sy_);
    sy_ = fmaf((float)(7 - 2 * q), v[1][0][0] + v[1][1][0], sy_);
    sy_ = fmaf((float)(6 - 2 * q), v[1][0][1] + v[1][1][1], sy_);
    tot = wave_sum_last(A0 + A1);
    sx_ = wave_sum_last(sx_);
    sy_ = wave_sum_last(sy_);
    if (do_cog && lane == 63) {
      float *sl = st.slopes + (long long)e * sys.nslope;
      if (tot > 0.f) {
        const float inv = __builtin_amdgcn_rcpf(tot);       // 1 ulp; slopes are compared at 1e-4"
        sl[i] = (sx_ * inv - sys.cog_offset) * sys.cog_scale;
        sl[sys.nvalid + i] = (sy_ * inv - sys.cog_offset) * sys.cog_scale;
      } else {
        sl[i] = 0.f;
        sl[sys.nvalid + i] = 0.f;
      }
    }
    (void)flux_i;
    return;
  }
  // ---- total flux (each LR pixel is held by two lanes: count even lanes only)
  tot = wave_sum(owner ? tot : 0.f);
  const float g = tot > 0.f ? sys.nphot * flux_i / tot : 0.f;
  float s0 = 0.f, sx = 0.f, sy = 0.f;
  if (NOISE) {
    // the two lanes of a pair hold the same 8 pixels: lane parity h takes pixel h of each (ty, tx),
    // so every pixel gets its noise once and all 64 lanes work
    const int hh = c & 1;
    const uint32_t sd = st.seeds[e], fr = st.frame[e];
#pragma unroll
    for (int ty_ = 0; ty_ < 2; ty_++)
#pragma unroll
      for (int tx_ = 0; tx_ < 2; tx_++) {
        const int Y = ty_ ? 7 - (2 * q + hh) : 8 + 2 * q + hh, X = tx_ ? Xm : Xp;
        const uint32_t idx = (uint32_t)i * 256u + (uint32_t)(Y * 16 + X);
        const float val = sh_noise((hh ? v[ty_][tx_][1] : v[ty_][tx_][0]) * g, sys.noise, sd, fr, idx);
        if (WRITE_CUBE) st.bincube[((long long)e * sys.nvalid + i) * 256 + Y * 16 + X] = val;
        s0 += val;
        sx += val * (float)X;
        sy += val * (float)Y;
      }
  } else {
#pragma unroll
    for (int ty_ = 0; ty_ < 2; ty_++)
#pragma unroll
      for (int tx_ = 0; tx_ < 2; tx_++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int Y = ty_ ? 7 - (2 * q + h) : 8 + 2 * q + h, X = tx_ ? Xm : Xp;
          const float val = v[ty_][tx_][h] * g;
          if (WRITE_CUBE && owner)
            st.bincube[((long long)e * sys.nvalid + i) * 256 + Y * 16 + X] = val;
          s0 += val;
          sx += val * (float)X;
          sy += val * (float)Y;
        }
  }
  if (do_cog) {
    s0 = wave_sum((NOISE || owner) ? s0 : 0.f);
    sx = wave_sum((NOISE || owner) ? sx : 0.f);
    sy = wave_sum((NOISE || owner) ? sy : 0.f);
    if (lane == 0) {
      float *sl = st.slopes + (long long)e * sys.nslope;
      if (s0 != 0.f) {
        sl[i] = (sx / s0 - sys.cog_offset) * sys.cog_scale;
        sl[sys.nvalid + i] = (sy / s0 - sys.cog_offset) * sys.cog_scale;
      } else {
        sl[i] = 0.f;
        sl[sys.nvalid + i] = 0.f;
      }
    }
  }
}

template <bool NOISE, bool WRITE_CUBE>
__device__ __forceinline__ void spot_finish(const DevSys &sys, const DevState &st, int e, int i,
                                            int lane, const f32x4 (&Xr)[2][2],
                                            const f32x4 (&Xi)[2][2], int do_cog, float flux_i) {
  float v[2][2][2];
#pragma unroll
  for (int sy = 0; sy < 2; sy++)
#pragma unroll
    for (int sx = 0; sx < 2; sx++) spot_bin(Xr[sy][sx], Xi[sy][sx], v[sy][sx]);
  spot_finish_v<NOISE, WRITE_CUBE>(sys, st, e, i, lane, v, do_cog, flux_i);
}

// operands: br[s] / bi[s] = complex amplitude of pixel (y = c, x = 4q + s) of the tile
template <bool NOISE, bool WRITE_CUBE>
__device__ __forceinline__ void spot_core(const DevSys &sys, const DevState &st, int e, int i,
                                          int lane, const float (&Cc)[4], const float (&Ss)[4],
                                          const float (&br)[4], const float (&bi)[4], int do_cog,
                                          float flux_i, const f32x4 z4) {
  f32x4 Xr[2][2], Xi[2][2];           // [sy][sx], 0: +, 1: -
  spot_dft_f32(Cc, Ss, br, bi, z4, Xr, Xi);
  spot_finish<NOISE, WRITE_CUBE>(sys, st, e, i, lane, Xr, Xi, do_cog, flux_i);
}

template <bool NOISE, bool WRITE_CUBE>
__device__ __forceinline__ void spot_compute(const DevSys &sys, const DevState &st, int e, int i,
                                             int lane, int wv, const float (&Cc)[4],
                                             const float (&Ss)[4],
                                             float (*sAr)[16][17], float (*sAi)[16][17], int do_cog,
                                             float flux_i) {
  const int q = lane >> 4, c = lane & 15;
  float br[4], bi[4];
#pragma unroll
  for (int s = 0; s < 4; s++) { br[s] = sAr[wv][c][4 * q + s]; bi[s] = sAi[wv][c][4 * q + s]; }
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  spot_core<NOISE, WRITE_CUBE>(sys, st, e, i, lane, Cc, Ss, br, bi, do_cog, flux_i, z4);
}

// phase (sum of all sources) and pupil mask of the 4 pixels this lane owns in sub-aperture i
template <bool FROM_BUF>
__device__ __forceinline__ void spot_load(const DevSys &sys, const DevState &st, int e, int i,
                                          int lane, int no_atmos, int no_dms, float ph[4],
                                          float mk[4]) {
  const int xy0 = sys.sub_xy[i];
  const int gx0 = xy0 & 0xFFFF, gy0 = xy0 >> 16;
  const int ty = lane >> 2, tx0 = (lane & 3) * 4;
  const int gy = gy0 + ty;
#pragma unroll
  for (int j = 0; j < 4; j++) { ph[j] = 0.f; mk[j] = 0.f; }
  if (FROM_BUF) {
    add4(ph, st.wfs_phase + (long long)e * sys.n * sys.n + gy * sys.n + gx0 + tx0);
  } else {
    if (!no_atmos) {
      for (int l = 0; l < sys.nlayers; l++) {
        const DevLayer &L = sys.layers[l];
        const float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
        const int ox = st.origin[(e * sys.nlayers + l) * 2], oy = st.origin[(e * sys.nlayers + l) * 2 + 1];
        int py = gy + L.woy + oy; py -= (py >= L.dim) ? L.dim : 0;
        int px = gx0 + tx0 + L.wox + ox; px -= (px >= L.dim) ? L.dim : 0;
        add4_ring(ph, base + py * (L.dim + RING_PAD), px, L.dim);
      }
    }
    if (!no_dms) {
      for (int k = 0; k < sys.ndm; k++) {
        const DevDm &D = sys.dms[k];
        const float *slot = st.dm_shape + (long long)e * sys.shape_stride + D.shape_off;
        const int o = (gy + D.woy) * D.dim + gx0 + tx0 + D.wox;
        if (D.type == AOMARL_DM_PZT) {
          add4(ph, slot + o);
        } else {
          const float c0 = slot[0], c1 = slot[1];
          float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          add4(f, D.influ + 2 * o);
          add4(f + 4, D.influ + 2 * o + 4);
#pragma unroll
          for (int j = 0; j < 4; j++) ph[j] += c0 * f[2 * j] + c1 * f[2 * j + 1];
        }
      }
    }
  }
  add4(mk, sys.mpupil + gy * sys.n + gx0 + tx0);
}

// Persistent waves: block = 4 waves, wave w of block bx walks sub-apertures
// (bx*4 + w) + k * gridDim.x*4 of environment blockIdx.y; the phase of the NEXT sub-aperture is
// fetched into registers while the MFMAs of the current one run (hides the load latency chain).
template <bool FROM_BUF, bool NOISE, bool WRITE_CUBE>
__global__ __launch_bounds__(256) void k_wfs_spot(DevSys sys, DevState st, int env_begin,
                                                  int no_atmos, int no_dms, int do_cog) {
  __shared__ float sAr[4][16][17];
  __shared__ float sAi[4][16][17];
  __shared__ float2 sTw[128];                    // (cos, sin)(2 pi m / 128)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int e = env_begin + blockIdx.y;
  if (threadIdx.x < 128) {
    float sn, cs;
    sincospif((float)threadIdx.x * (1.0f / 64.0f), &sn, &cs);
    sTw[threadIdx.x] = make_float2(cs, sn);
  }
  __syncthreads();                               // the only block-wide barrier
  const int q = lane >> 4, c = lane & 15;
  const int stride = gridDim.x * 4;
  int i = blockIdx.x * 4 + wv;
  if (i >= sys.nvalid) return;

  // ---- twiddles: idx = 4q + s (pixel), half-integer frequency k + 1/2 with k = c
  float Cc[4], Ss[4];                            // cos / sin(2 pi (2c+1)(4q+s) / 128)
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const float2 w = sTw[((4 * q + s) * (2 * c + 1)) & 127];
    Cc[s] = w.x; Ss[s] = w.y;
  }
  const int ty = lane >> 2, tx0 = (lane & 3) * 4;

  float ph[4], mk[4];
  spot_load<FROM_BUF>(sys, st, e, i, lane, no_atmos, no_dms, ph, mk);
  for (; i < sys.nvalid; i += stride) {
    // ---- stage 0: complex amplitude tile -> LDS (4 pixels per lane)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float t = ph[j] * sys.wfs_inv_lambda;          // revolutions (the half-pixel ramp is folded
                                                     // into the half-integer frequencies)
      t -= rintf(t);
      const float sn = __builtin_amdgcn_sinf(t), cs = __builtin_amdgcn_cosf(t);   // v_sin/v_cos
      sAr[wv][ty][tx0 + j] = mk[j] * cs;
      sAi[wv][ty][tx0 + j] = mk[j] * sn;
    }
    // ---- prefetch the next sub-aperture of this wave
    const int inext = i + stride;
    if (inext < sys.nvalid) spot_load<FROM_BUF>(sys, st, e, inext, lane, no_atmos, no_dms, ph, mk);
    __builtin_amdgcn_wave_barrier();

    spot_compute<NOISE, WRITE_CUBE>(sys, st, e, i, lane, wv, Cc, Ss, sAr, sAi, do_cog, sys.flux[i]);
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// Fast variant for the production layout: NL atmosphere layers, DMs = [stack array, tip-tilt],
// every offset an integer, atmosphere and DMs both seen.  All loads of a tile are independent
// (raw values land in separate registers, the sum is formed at first use), so one wait covers the
// whole prefetch and it overlaps the MFMA block of the previous sub-aperture.  Per-environment
// constants (ring origins, TT commands) are read once per wave.
// ---------------------------------------------------------------------------------------------
template <int NL>
struct SpotEnv {
  const float *lay[NL];
  int px0[NL], py0[NL], dim[NL];
  const float *pzt;
  int pzt_dim, pzt_ox, pzt_oy;
  const float *tt;
  int tt_dim, tt_ox, tt_oy;
  float c0, c1;
};

template <int NL>
struct SpotRaw {
  float L[NL][4];
  float P[4], T[8], M[4];
  float F;   // fluxPerSub of the sub-aperture
};

// All loads of one tile.  Addresses are  uniform base pointer (SGPR) + 32-bit lane offset, so
// the compiler emits global_load ... v_off, s[base] with no 64-bit VALU arithmetic; ring wraps
// cost one v_sub + v_min_u32 per axis (px, px - dim as unsigned: the smaller one is in range).
template <int NL>
__device__ __forceinline__ void spot_fetch(const DevSys &sys, const SpotEnv<NL> &E, int xy0, int ty,
                                           int tx0, int i, SpotRaw<NL> &r) {
  const unsigned gx = (unsigned)((xy0 & 0xFFFF) + tx0), gy = (unsigned)((xy0 >> 16) + ty);
  r.F = sys.flux[i];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    const unsigned dim = (unsigned)E.dim[l];
    unsigned py = gy + (unsigned)E.py0[l]; py = min(py, py - dim);
    unsigned px = gx + (unsigned)E.px0[l]; px = min(px, px - dim);
    const unsigned off = py * (dim + RING_PAD) + px;
    const f4u t = *reinterpret_cast<const f4u *>(E.lay[l] + off);
#pragma unroll
    for (int j = 0; j < 4; j++) r.L[l][j] = t.v[j];
  }
  {
    const unsigned off = (gy + (unsigned)E.pzt_oy) * (unsigned)E.pzt_dim + gx + (unsigned)E.pzt_ox;
    const f4u t = *reinterpret_cast<const f4u *>(E.pzt + off);
#pragma unroll
    for (int j = 0; j < 4; j++) r.P[j] = t.v[j];
  }
  {
    const unsigned off = 2u * ((gy + (unsigned)E.tt_oy) * (unsigned)E.tt_dim + gx + (unsigned)E.tt_ox);
    const f4u t0 = *reinterpret_cast<const f4u *>(E.tt + off);
    const f4u t1 = *reinterpret_cast<const f4u *>(E.tt + off + 4u);
#pragma unroll
    for (int j = 0; j < 4; j++) { r.T[j] = t0.v[j]; r.T[4 + j] = t1.v[j]; }
  }
  {
    const unsigned off = gy * (unsigned)sys.n + gx;
    const f4u t = *reinterpret_cast<const f4u *>(sys.mpupil + off);
#pragma unroll
    for (int j = 0; j < 4; j++) r.M[j] = t.v[j];
  }
}

// stage-1 MFMAs + T combinations of the tile in sAr/sAi (see spot_compute for the algebra)
__device__ __forceinline__ void spot_stage1(int lane, const float (&Cc)[4], const float (&Ss)[4],
                                            const float (*bAr)[17], const float (*bAi)[17],
                                            f32x4 (&Tr)[2], f32x4 (&Ti)[2]) {
  const int q = lane >> 4, c = lane & 15;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 PCr = z4, PCi = z4, PSr = z4, PSi = z4;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const float br = bAr[c][4 * q + s], bi = bAi[c][4 * q + s];
    PCr = mfma16(br, Cc[s], PCr);
    PCi = mfma16(bi, Cc[s], PCi);
    PSr = mfma16(br, Ss[s], PSr);
    PSi = mfma16(bi, Ss[s], PSi);
  }
  Tr[0] = PCr + PSi; Ti[0] = PCi - PSr;
  Tr[1] = PCr - PSi; Ti[1] = PCi + PSr;
}

// Software-pipelined fast kernel.  Per wave and iteration k (tile k of this wave):
//   R1  stage-1 MFMAs of tile k (LDS buffer k&1), T combinations
//   R2  stage-2 MFMAs of tile k  INTERLEAVED (sched_group_barrier) with independent work:
//         amplitude of tile k+1 (sum of the prefetched sources, sin/cos) -> LDS buffer (k+1)&1,
//         address arithmetic + issue of the loads of tile k+2
//   R3  X combinations, |X|^2, binning, COG, store of tile k
// so the VALU / VMEM / LDS-write work of the neighbouring tiles hides under the matrix pipe of
// the current one instead of serialising with it.  The loop body is one basic block (indices are
// clamped instead of branched on).
template <int NL, bool NOISE, bool WRITE_CUBE>
__global__ __launch_bounds__(256) void k_wfs_spot_fast(DevSys sys, DevState st, int env_begin,
                                                       int do_cog) {
  __shared__ float sAr[4][2][16][17];
  __shared__ float sAi[4][2][16][17];
  __shared__ float2 sTw[128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int e = env_begin + blockIdx.y;
  if (threadIdx.x < 128) {
    float sn, cs;
    sincospif((float)threadIdx.x * (1.0f / 64.0f), &sn, &cs);
    sTw[threadIdx.x] = make_float2(cs, sn);
  }
  __syncthreads();
  const int q = lane >> 4, c = lane & 15;
  const int stride = gridDim.x * 4;
  const int i0 = blockIdx.x * 4 + wv;
  if (i0 >= sys.nvalid) return;
  const int nit = (sys.nvalid - i0 + stride - 1) / stride;      // tiles of this wave
  const int ilast = i0 + (nit - 1) * stride;
  float Cc[4], Ss[4];                            // cos / sin(2 pi (2c+1)(4q+s) / 128)
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const float2 w = sTw[((4 * q + s) * (2 * c + 1)) & 127];
    Cc[s] = w.x; Ss[s] = w.y;
  }
  const int ty = lane >> 2, tx0 = (lane & 3) * 4;
  const bool owner = (c & 1) == 0;
  // ---- per-environment constants
  SpotEnv<NL> E;
#pragma unroll
  for (int l = 0; l < NL; l++) {
    const DevLayer &L = sys.layers[l];
    E.lay[l] = st.screens + (long long)e * sys.screen_stride + L.screen_off;
    E.dim[l] = L.dim;
    int px = L.wox + st.origin[(e * sys.nlayers + l) * 2]; px -= (px >= L.dim) ? L.dim : 0;
    int py = L.woy + st.origin[(e * sys.nlayers + l) * 2 + 1]; py -= (py >= L.dim) ? L.dim : 0;
    E.px0[l] = px; E.py0[l] = py;
  }
  {
    const DevDm &D0 = sys.dms[0], &D1 = sys.dms[1];
    E.pzt = st.dm_shape + (long long)e * sys.shape_stride + D0.shape_off;
    E.pzt_dim = D0.dim; E.pzt_ox = D0.wox; E.pzt_oy = D0.woy;
    const float *slot = st.dm_shape + (long long)e * sys.shape_stride + D1.shape_off;
    E.c0 = slot[0]; E.c1 = slot[1];
    E.tt = D1.influ; E.tt_dim = D1.dim; E.tt_ox = D1.wox; E.tt_oy = D1.woy;
  }
  const float inv_lambda = sys.wfs_inv_lambda;
  SpotRaw<NL> raw;
  auto amplitude_to_lds = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float ph = raw.P[j] + (E.c0 * raw.T[2 * j] + E.c1 * raw.T[2 * j + 1]);
#pragma unroll
      for (int l = 0; l < NL; l++) ph += raw.L[l][j];
      float t = ph * inv_lambda;
      t -= rintf(t);
      const float sn = __builtin_amdgcn_sinf(t), cs = __builtin_amdgcn_cosf(t);
      sAr[wv][buf][ty][tx0 + j] = raw.M[j] * cs;
      sAi[wv][buf][ty][tx0 + j] = raw.M[j] * sn;
    }
  };
  // ---- prologue: tile 0 -> LDS buffer 0, loads of tile 1 in flight
  spot_fetch<NL>(sys, E, sys.sub_xy[i0], ty, tx0, i0, raw);
  float flux_cur = raw.F;
  amplitude_to_lds(0);
  {
    const int i1 = min(i0 + stride, ilast);
    spot_fetch<NL>(sys, E, sys.sub_xy[i1], ty, tx0, i1, raw);
  }
  int xy2 = sys.sub_xy[min(i0 + 2 * stride, ilast)];
  __builtin_amdgcn_wave_barrier();

  for (int k = 0; k < nit; k++) {
    const int i = i0 + k * stride;
    const int buf = k & 1;
    // ---- R1
    f32x4 Tr[2], Ti[2];
    spot_stage1(lane, Cc, Ss, sAr[wv][buf], sAi[wv][buf], Tr, Ti);
    // ---- R2: stage-2 MFMAs ...
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 Xr[2][2], Xi[2][2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
      f32x4 QCr = z4, QCi = z4, QSr = z4, QSi = z4;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        QCr = mfma16(Cc[s], Tr[m][s], QCr);
        QCi = mfma16(Cc[s], Ti[m][s], QCi);
        QSr = mfma16(Ss[s], Tr[m][s], QSr);
        QSi = mfma16(Ss[s], Ti[m][s], QSi);
      }
      Xr[0][m] = QCr + QSi; Xi[0][m] = QCi - QSr;
      Xr[1][m] = QCr - QSi; Xi[1][m] = QCi + QSr;
    }
    // ... with the neighbours' work: amplitude of tile k+1, loads of tile k+2
    const float flux_next = raw.F;
    amplitude_to_lds(buf ^ 1);
    {
      const int i2 = min(i + 2 * stride, ilast);
      spot_fetch<NL>(sys, E, xy2, ty, tx0, i2, raw);
      xy2 = sys.sub_xy[min(i + 3 * stride, ilast)];
    }
#pragma unroll
    for (int r = 0; r < 32; r++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);     // 5 VALU
      __builtin_amdgcn_sched_group_barrier(0x220, 1, 0);     // 1 VMEM read or DS write
    }
    // ---- R3: |X|^2, 2x2 binning, flux / COG, store (same mapping as spot_compute)
    float v[2][2][2];
    float tot = 0.f;
#pragma unroll
    for (int sy = 0; sy < 2; sy++)
#pragma unroll
      for (int sx = 0; sx < 2; sx++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
          float a0 = Xr[sy][sx][2 * h], b0 = Xi[sy][sx][2 * h];
          float a1 = Xr[sy][sx][2 * h + 1], b1 = Xi[sy][sx][2 * h + 1];
          float t = (a0 * a0 + b0 * b0) + (a1 * a1 + b1 * b1);
          t = add_xor1(t);
          v[sy][sx][h] = t;
          tot += t;
        }
    const int Xp = 8 + (c >> 1), Xm = 7 - (c >> 1);
    float *sl = st.slopes + (long long)e * sys.nslope;
    if (!NOISE && !WRITE_CUBE) {
      float sx_ = 0.f, sy_ = 0.f;
#pragma unroll
      for (int sy = 0; sy < 2; sy++)
#pragma unroll
        for (int sx = 0; sx < 2; sx++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int Y = sy ? 7 - (2 * q + h) : 8 + 2 * q + h;
            sx_ += v[sy][sx][h] * (float)(sx ? Xm : Xp);
            sy_ += v[sy][sx][h] * (float)Y;
          }
      tot = wave_sum(owner ? tot : 0.f);
      sx_ = wave_sum(owner ? sx_ : 0.f);
      sy_ = wave_sum(owner ? sy_ : 0.f);
      if (do_cog && lane == 0) {
        const float inv = tot > 0.f ? 1.0f / tot : 0.f;
        sl[i] = tot > 0.f ? (sx_ * inv - sys.cog_offset) * sys.cog_scale : 0.f;
        sl[sys.nvalid + i] = tot > 0.f ? (sy_ * inv - sys.cog_offset) * sys.cog_scale : 0.f;
      }
    } else {
      tot = wave_sum(owner ? tot : 0.f);
      const float g = tot > 0.f ? sys.nphot * flux_cur / tot : 0.f;
      float s0 = 0.f, sx = 0.f, sy = 0.f;
      if (NOISE) {
        // lane parity h takes pixel h of each (ty, tx) of the pair's 8 pixels (see spot_finish)
        const int hh = c & 1;
        const uint32_t sd = st.seeds[e], fr = st.frame[e];
#pragma unroll
        for (int ty_ = 0; ty_ < 2; ty_++)
#pragma unroll
          for (int tx_ = 0; tx_ < 2; tx_++) {
            const int Y = ty_ ? 7 - (2 * q + hh) : 8 + 2 * q + hh, X = tx_ ? Xm : Xp;
            const uint32_t idx = (uint32_t)i * 256u + (uint32_t)(Y * 16 + X);
            const float val = sh_noise((hh ? v[ty_][tx_][1] : v[ty_][tx_][0]) * g, sys.noise, sd, fr, idx);
            if (WRITE_CUBE) st.bincube[((long long)e * sys.nvalid + i) * 256 + Y * 16 + X] = val;
            s0 += val;
            sx += val * (float)X;
            sy += val * (float)Y;
          }
      } else {
#pragma unroll
        for (int ty_ = 0; ty_ < 2; ty_++)
#pragma unroll
          for (int tx_ = 0; tx_ < 2; tx_++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const int Y = ty_ ? 7 - (2 * q + h) : 8 + 2 * q + h, X = tx_ ? Xm : Xp;
              const float val = v[ty_][tx_][h] * g;
              if (WRITE_CUBE && owner)
                st.bincube[((long long)e * sys.nvalid + i) * 256 + Y * 16 + X] = val;
              s0 += val;
              sx += val * (float)X;
              sy += val * (float)Y;
            }
      }
      if (do_cog) {
        s0 = wave_sum((NOISE || owner) ? s0 : 0.f);
        sx = wave_sum((NOISE || owner) ? sx : 0.f);
        sy = wave_sum((NOISE || owner) ? sy : 0.f);
        if (lane == 0) {
          sl[i] = s0 != 0.f ? (sx / s0 - sys.cog_offset) * sys.cog_scale : 0.f;
          sl[sys.nvalid + i] = s0 != 0.f ? (sy / s0 - sys.cog_offset) * sys.cog_scale : 0.f;
        }
      }
    }
    flux_cur = flux_next;
    __builtin_amdgcn_wave_barrier();
  }
}

// =============================================================================================
// ONE-PASS FRAME KERNEL: WFS spot images + COG  and  science-path PSF rows from ONE pass over the
// phase.  The sub-aperture tiles (16 x 16 phase pixels starting at pad + 16 k of the WFS grid) are
// exactly the 16 x 16 tiles of the pupil grid the target sees, at the same screen / DM pixels, so
// every phase pixel is read once and feeds both paths: per tile 48 MFMAs (pruned spot DFT) if it
// is a valid sub-aperture + 16 MFMAs (PSF row DFT).
//
// Work split: ONE WAVE = ONE (environment, stripe of 16 pupil rows); it walks the tiles of the
// stripe left to right, so its 16 PSF rows stay in its accumulators (no cross-wave reduction) and
// its per-environment constants are scalars.  The 4 waves of a block work on the same stripe of
// 4 consecutive environments: the data shared by all environments (tip-tilt planes, pupil mask
// rows, PSF twiddles) is fetched at about the same time by the 4 waves and hits in the L1.
// The pupil mask comes as 16-bit rows; tiles outside the pupil are skipped.
//
// Lane (q, c) owns the pixels (y = c, x = 4q .. 4q+3) of the tile -- the accumulator layout of a
// 16x16x4 MFMA whose M index is x and N index is y, so the stack-array DM phase can be produced IN
// PLACE on the matrix cores (OTF = true): with the separable influence profile u and the command
// lattice C of the environment,
//     S[x][y] = sum_j u(x - X_j) sum_i C[j][i] u(y - Y_i)
// is two tiny GEMMs (lattice nodes that reach a tile: 4 .. 8 per axis): stage A over i (A operand
// = lattice patch from LDS, B operand = profile column, a per-lane constant), stage B over j
// (A operand = profile, B operand = the accumulator registers of stage A, whose row order
// j(m) = (m >> 2) + 4 (m & 3) makes register s the K-slice of MFMA s).  2 MFMAs for <= 4 nodes
// (NB = 1), 4 for <= 8 (NB = 2): the DM shape of the pupil never exists in memory.
// LDS tiles are stored transposed ([x][y], row stride 20 floats): writes by (q, c) and the
// operand reads (x = 4q + s, y = c) both touch 32 distinct banks per half wave.
// =============================================================================================
#define FW_LD 20
// tile_info bits: [15:0] sub-aperture index, 16 lit, 17 every pixel lit, 18 has a valid sub-aperture
#define FW_LIT 0x10000
#define FW_FULL 0x20000
#define FW_SUB 0x40000
// science-path phase in revolutions for v_sin / v_cos.  The instructions reduce their argument
// themselves for |x| <= 256 revolutions (ISA: valid input domain [-256, 256], 0 outside); the
// science wavelength sees |phase| / lambda of a few tens at most (an 8 m pupil at r0 = 0.16 m has
// 2 um rms of optical path, 1.65 um wavelength: 422 um would be needed to leave the domain), so
// the explicit round-and-subtract the WFS path keeps (its fused multiply-subtract also saves one
// rounding, which the 2e-5 image tolerance needs) is dropped here.
__device__ __forceinline__ float sci_rev(float ph, float inv_lambda) { return ph * inv_lambda; }

// Uniform tables that nothing writes during a launch, read through the CONSTANT address space: the compiler
// then uses scalar loads (s_load, counted by lgkmcnt) instead of a vector load + v_readfirstlane.  It matters
// for the per-tile walk of the lit-tile list: vector loads return in order, so waiting for the list entry --
// the newest load in flight -- made every wave wait for ALL its prefetched tile loads at the top of each
// iteration (s_waitcnt vmcnt(0)): the software pipeline of the frame kernel collapsed to one stage.
typedef const int __attribute__((address_space(4))) *const_int_p;
typedef const float __attribute__((address_space(4))) *const_float_p;

template <int NL, bool OTF>
struct FrameRaw {
  float L[NL][4];
  float P[OTF ? 1 : 4], T[8];
  unsigned mrow;      // 16-bit mask row of the tile (this lane's row)
  float F;
  f4u SH;             // this wave's quarter of the tile's environment-independent data
  float4 CS;          // the PSF operand of the lane (pair walk: read out of the block's slot by the caller, like T)
};

// Both arithmetics share one skeleton.  The data of a tile that does not depend on the environment
// (tip-tilt planes: 2 x 16 bytes per lane; PSF twiddle operand: 16 bytes per lane) is fetched once per
// BLOCK -- each of the 4 waves loads one quarter a tile ahead, parks it in a double-buffered LDS slot, one
// barrier per lit tile -- instead of once per wave from L2, and two tiles of loads are in flight per wave.
// HP = false (default, the reference's arithmetic): fp32 operands on v_mfma_f32_16x16x4_f32.  Lane (q, c)
// owns the pixels (y = c, x = 4q + s): with K step s of an MFMA taking x = 4q + s from K lane q, the lane's
// own four amplitudes ARE the A operands of the first DFT stage and of the PSF rows, and the accumulator
// registers of stage 1 the B operands of stage 2: no amplitude tile ever goes through LDS.
// HP = true (fast mode): the same products on split-fp16 MFMAs (hi + lo operand pairs).
//
// Science path: R[y][kx] = sum_x a(y, x) exp(-2 pi i kx X / Npsf), X = 16 t + x, kx in [-8, 8).  cos is
// even and sin odd in kx, so the 16 columns of ONE real operand  [cos k X (k = 1..8) | sin k X (k = 1..8)]
// serve every kx != 0:  PA = a_r . [C|S],  PB = a_i . [C|S]  (8 fp32 MFMAs per tile instead of 16 for the
// plain complex product; 4 instead of 8 split-fp16 ones), accumulated over the whole stripe and combined
// once at its end:  kx = +k: (PA_C + PB_S, PB_C - PA_S);  kx = -k: (PA_C - PB_S, PB_C + PA_S).  The
// kx = 0 column is the plain row sum of the amplitudes: 8 vector adds per tile.
template <int NL, int NB, bool OTF, bool NOISE, bool WRITE_CUBE, bool HP>
// 3 waves per SIMD (<= 168 VGPRs).  Forcing the fp32 slopes-only instantiation (140 VGPRs) into four (<= 128, 8
// registers spilled) measured 0.466 against 0.459 ms: the kernel is bound by issue slots, not by latency
#ifndef FW_WAVES
#define FW_WAVES 3
#endif
#ifndef FW_QF
#define FW_QF 1            // 0: the slopes-only fp32 instantiation goes through the pruned transform (spot_cog_f32_pk)
#endif
#ifndef FW_DMA
#define FW_DMA 1           // 0: layer rows fetched per tile into vector registers (16 rows x 64 B per instruction)
#endif
#ifndef FW_DEPTH
#define FW_DEPTH 2         // lit tiles of loads in flight per wave in the per-tile walk (register sets)
#endif
#ifndef FW_DMA_F32
#define FW_DMA_F32 1       // 0: the pair walk for the split-fp16 instantiations only
#endif
// Round 6 experiment (FW_SLOTS1 = 1; built, measured, NOT the default): the pair walk's shared-data slots single-buffered,
// three quarters per tile (2 x 3 KB instead of 2 x 2 x 4 KB; with no twiddle table in the slopes-only instantiation:
// 37.5 KB of LDS per workgroup instead of 49).  Three frame workgroups then leave 46 KB of a CU's 160, and one workgroup
// of the chains' products (k_gemm_p<2, 3>: 46 080 B, 156 registers beside 3 x 112) is resident beside them on EVERY CU
// instead of starting when a frame workgroup retires.  Price: the slots are read into registers between two barriers.
// Measured (alternating runs on one box, gpurun_out/r06f_*, r06g_*): the frame kernel in the loop gets FASTER (0.365
// against 0.386 ms: it always has its three workgroups) and the step SLOWER (0.494 against 0.482 ms, 512 against 524 k):
// a product's wave beside three frame waves gets a quarter of the SIMD's issue slots, one that took a retired frame
// workgroup's place a third -- and the step waits for the products' chain, not for the frame kernel.  Wave priorities on
// top (GP_PRIO / ATM_PRIO / CHAIN_PRIO) change nothing, in either layout.
#ifndef FW_SLOTS1
#define FW_SLOTS1 0
#endif
#define FW_SLOT_BYTES(dma) ((dma) ? (FW_SLOTS1 ? 2 * 3 * 1024 : 16384) : 16384)
// Layer rows of a PAIR of adjacent tiles as whole 128-byte pieces, straight into LDS (FW_DMA, the stack-array-from-
// voltages instantiations).  A load instruction that covers 16 rows x 64 B (the compute layout: lane (q, c) = row c,
// pixels 4q .. 4q + 3) makes the memory pipeline handle every 128-byte line twice, half a line at a time; as
// 8 rows x 128 contiguous bytes the same bytes arrive 25 % faster (tools/dmabench.hip: 3.9 -> 4.9 TB/s on this access
// pattern alone).  (HBM-side bytes do not go down -- a 128-byte run at a 4-byte-aligned start straddles two lines 7
// times out of 8: 1.6 GB per launch against 1.47 -- the number of requests does.)  The pieces are written by
// buffer_load_dwordx4 ... lds -- no vector registers in flight -- into the wave's own LDS image: per layer two blocks of
// 8 rows x 8 chunks of 16 bytes in LANE order (what that instruction can write), the second block 128 bytes further,
// the chunk a lane fetches XOR-ed with its row (lane = 8 row + (chunk ^ row)): the compute layout's ds_read_b128
// (row c, chunk 4 h + q of tile h) then finds its 16 lanes in 16 different bank groups.
#define FWD_BLK 1152                       // bytes from block 0 to block 1 of a layer image
#define FWD_IMG (2 * 1024 + 128)           // bytes of a layer image
#define FWD_WAVE(nl) ((nl) * FWD_IMG)      // bytes per wave
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FW_WAVES, FW_WAVES)))
void k_frame_wave(DevSys sys, DevState st, int env_begin,
                                                    int env_count, int do_cog,
                                                    float *__restrict__ TR,
                                                    float *__restrict__ TPART, int nblk) {
  extern __shared__ float smem[];
  const int pd = sys.pupdiam, ntl = sys.ntiles;
  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, c = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // slopes only, fp32: the moments as quadratic forms of the field (spot_qf_moments): no transform, no Cc / Ss, no
  // twiddle table -- the lane's constants come from sys.qf_tab
  constexpr bool QF = FW_QF && OTF && !HP && !NOISE && !WRITE_CUBE;
  constexpr bool DMA = FW_DMA && OTF && (HP || FW_DMA_F32);
  float2 *sTw = reinterpret_cast<float2 *>(smem);            // [128] WFS twiddles (none in the slopes-only instantiation)
  float *lat_all = reinterpret_cast<float *>(sTw + (QF ? 0 : 128));     // [4 waves][4 NB][latw]
  // per tile walk: [2][4][64]; pair walk: [2 tiles][3 quarters][64] (FW_SLOTS1) or [2 parities][2 tiles][4][64]
  float4 *shb = reinterpret_cast<float4 *>(lat_all + 4 * 4 * NB * (OTF ? sys.otf_latw : 0));
  char *dimg = reinterpret_cast<char *>(shb) + FW_SLOT_BYTES(DMA) + (DMA ? wv * FWD_WAVE(NL) : 0);   // this wave's layer images
  // slopes-only fp32 instantiation: the moments of the stripe's sub-apertures, [tile][s0, ty, tx, -] per wave,
  // turned into slopes once per stripe by lane = tile (the per-tile finish was a dozen instructions on ONE lane)
  float4 *qmom = reinterpret_cast<float4 *>(reinterpret_cast<char *>(shb) + FW_SLOT_BYTES(DMA) + (DMA ? 4 * FWD_WAVE(NL) : 0)) + wv * ntl;
  const int dbg = do_cog >> 8;                               // development switches (kbench)
  do_cog &= 1;
  // blocks are dispatched x-fastest: x = group of 4 environments, y = rank of the stripe by
  // decreasing number of lit tiles -- the longest stripes start first and the launch ends on the
  // shortest ones (longest-processing-time order: smaller tail)
  const int r = sys.stripe_order[blockIdx.y];                // stripe: pupil rows 16 r .. 16 r + 15
  const int el = 4 * blockIdx.x + wv;                        // environment of this wave
  if constexpr (!QF) {
    if (tid < 128) {
      float sn, cs;
      sincospif((float)tid * (1.0f / 64.0f), &sn, &cs);
      sTw[tid] = make_float2(cs, sn);
    }
    __syncthreads();
  }
  const bool active = el < env_count;          // a wave past the last environment repeats it (see the tile)
  const int e = env_begin + (active ? el : env_count - 1);
  float Cc[4] = {0.f, 0.f, 0.f, 0.f}, Ss[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (!QF) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const float2 w = sTw[((4 * q + s) * (2 * c + 1)) & 127];
      Cc[s] = w.x; Ss[s] = w.y;
    }
  }
  SpotTwH twh;
  if (HP) twh = spot_tw_h(Cc, Ss);
  SpotQf qfk;
  if constexpr (QF) {
    const float4 ka = reinterpret_cast<const float4 *>(sys.qf_tab)[2 * lane], kb = reinterpret_cast<const float4 *>(sys.qf_tab)[2 * lane + 1];
    qfk.Ml = f32x2{ka.x, ka.y}; qfk.Mh = f32x2{ka.z, ka.w};
    qfk.Sl = f32x2{kb.x, kb.y}; qfk.Sh = f32x2{kb.z, kb.w};
  }
  // shared loads: wave 0 / 1: the two 16-byte halves of the lane's tip-tilt pairs, wave 2 (and 3, a
  // duplicate that hits in the L1): the PSF operand of the lane, [t][64] x 16 B
  unsigned shstep;
  const int y = 16 * r + c;                                  // pupil row of this lane
  // ---- per-environment constants (target-path offsets; the WFS sees the same pixels).
  // Every load address is  scalar base (advances with the tile)  +  per-lane byte offset (fixed
  // for the stripe): rows carry RING_PAD = 16 mirror columns, so the 16-byte load of a lane may
  // start up to 12 floats past the scalar wrap point without wrapping itself.
  // Loads go through buffer descriptors (scalar base + 32-bit per-lane offset + scalar offset of
  // the tile): no 64-bit address arithmetic on the vector unit, no address registers.
  const char *layb[NL];
  __amdgpu_buffer_rsrc_t lrs[NL];
  unsigned lpxs[NL], ldim[NL], lvo[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    const DevLayer &L = sys.layers[l];
    layb[l] = reinterpret_cast<const char *>(st.screens + (long long)e * sys.screen_stride + L.screen_off);
    lrs[l] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(layb[l]), 0,
                                               4 * L.dim * (L.dim + RING_PAD), 0x00020000);
    ldim[l] = (unsigned)L.dim;
    int px = L.tox + st.origin[(e * sys.nlayers + l) * 2]; px -= (px >= L.dim) ? L.dim : 0;
    int py = L.toy + st.origin[(e * sys.nlayers + l) * 2 + 1]; py -= (py >= L.dim) ? L.dim : 0;
    lpxs[l] = (unsigned)px;
    unsigned pr = (unsigned)y + (unsigned)py; pr = min(pr, pr - ldim[l]);
    lvo[l] = 4u * (pr * (ldim[l] + RING_PAD) + 4u * (unsigned)q);
  }
  const DevDm &D0 = sys.dms[0], &D1 = sys.dms[1];
  const float *pzt = st.dm_shape + (long long)e * sys.shape_stride + D0.shape_off;
  const float *ttslot = st.dm_shape + (long long)e * sys.shape_stride + D1.shape_off;
  const float c0 = ttslot[0], c1 = ttslot[1];
  const int half = pd / 2;
  const unsigned pvo = 4u * ((unsigned)(y + D0.toy) * (unsigned)D0.dim + (unsigned)D0.tox + 4u * (unsigned)q);
  // tip-tilt planes re-arranged per 4 pixels of a pupil row: [x0 x1 x2 x3 | y0 y1 y2 y3] (sys.tt_pk), so that a
  // lane's two 16-byte halves are the x-plane and the y-plane values of its 4 pixels (pairs for packed FMAs)
  const unsigned tvo = 32u * ((unsigned)y * (unsigned)(pd >> 2) + (unsigned)q);
  const unsigned mvo = 2u * (unsigned)y * (unsigned)ntl;
  const char *pztb = reinterpret_cast<const char *>(pzt);
  const char *ttb = reinterpret_cast<const char *>(sys.tt_pk);
  const char *mkb = reinterpret_cast<const char *>(sys.tile_mask);
  // (selected with scalar selects, not in two branches: the descriptor must stay provably wave-uniform, or
  // every load through it is wrapped in a waterfall loop)
  const bool sh_tt = wv < 2;
  const void *shbase = sh_tt ? static_cast<const void *>(ttb) : (HP ? sys.psf_tw_h : sys.psf_tw_f);
  const __amdgpu_buffer_rsrc_t shrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(shbase), 0, sh_tt ? 8 * pd * pd : ntl * 1024, 0x00020000);
  const unsigned shvo = sh_tt ? tvo + 16u * (unsigned)wv : 16u * (unsigned)lane;
  shstep = sh_tt ? 7u : 10u;          // log2 of the bytes per tile (a shift stays on the scalar unit; the
                                      // compiler turns a multiplication into a VECTOR mul24 + waterfall loop)
  // pivot of the variance sums: phase at the grid centre (ttslot[2]: stack-array value there)
  float pivot;
  {
    float v = OTF ? ttslot[2] : pzt[(half + D0.toy) * D0.dim + half + D0.tox];
    const float2 f = reinterpret_cast<const float2 *>(D1.influ)[(half + D1.toy) * D1.dim + half + D1.tox];
    v += c0 * f.x + c1 * f.y;
#pragma unroll
    for (int l = 0; l < NL; l++) {
      const DevLayer &L = sys.layers[l];
      int py = L.toy + st.origin[(e * sys.nlayers + l) * 2 + 1]; py -= (py >= L.dim) ? L.dim : 0;
      unsigned pyc = (unsigned)(half + py); pyc = min(pyc, pyc - ldim[l]);
      unsigned pxc = (unsigned)half + lpxs[l]; pxc = min(pxc, pxc - ldim[l]);
      v += reinterpret_cast<const float *>(layb[l])[pyc * (ldim[l] + RING_PAD) + pxc];
    }
    pivot = v;
  }
  // ---- on-the-fly stack-array DM: profile operands + the lattice rows of this stripe in LDS
  float Pxr[NB], Pyr[NB];
  const int jm = (c >> 2) + 4 * (c & 3);
  const bool jm_ok = jm < 4 * NB;
  const int latw = sys.otf_latw, tpn = sys.otf_tpn;
  float *lat = lat_all + wv * (4 * NB * latw);
  if (OTF) {
#pragma unroll
    for (int s = 0; s < NB; s++) {
      const int ax = sys.otf_xoff + c - D0.pitch * (q + 4 * s);
      const int ay = sys.otf_yoff + c - D0.pitch * (q + 4 * s);
      Pxr[s] = (ax >= 0 && ax < D0.ss) ? D0.prof[ax] : 0.f;
      Pyr[s] = (ay >= 0 && ay < D0.ss) ? D0.prof[ay] : 0.f;
    }
    const float *volt = st.voltage + (long long)e * st.ld_actu + D0.com_off;
    // (row by row, a lane per column: the flat index over rows x columns cost an integer division per element, a
    // hundred vector instructions at the head of every wave)
    for (int i = 0; i < 4 * NB; i++) {
      const int gy = sys.otf_gy0 + r * tpn + i;
      for (int jj = lane; jj < latw; jj += 64) {
        const int gx = sys.otf_gx0 + jj;
        float v = 0.f;
        if (gx >= 0 && gx < D0.gw && gy >= 0 && gy < D0.gh) {
          const int a = D0.grid[gy * D0.gw + gx];
          if (a >= 0) v = volt[a];
        }
        lat[i * latw + jj] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  const float *latq = lat + (jm_ok ? jm : 0);
  const float wfs_il = sys.wfs_inv_lambda, tar_il = sys.tar_inv_lambda;
  const f32x4 Z4 = opaque_zero4();
  f32x4 PA = {0.f, 0.f, 0.f, 0.f}, PB = {0.f, 0.f, 0.f, 0.f};     // a_r . [C|S], a_i . [C|S] of the stripe
  float R0r = 0.f, R0i = 0.f;                                     // kx = 0: row sums of this lane's 4 columns
  float sd = 0.f, sd2 = 0.f, sm = 0.f;
  int nlit = 0;
  // packed-fp32 form of the tile's vector arithmetic (fp32 arithmetic with the stack-array DM evaluated in here:
  // the instantiations the loop launches).  The pivot of the variance sums goes into the C operand of the
  // lattice product, so the phase of the tile comes out as phase - pivot: a piston, invisible to |.|^2 of either
  // path, and the variance sums take the phase as it stands.
  constexpr bool PK = OTF && !HP;
  const f32x4 NP4 = {-pivot, -pivot, -pivot, -pivot};
  const f32x2 c0c0 = {c0, c0}, c1c1 = {c1, c1};
  const f32x2 wil2 = {sys.wfs_inv_lambda, sys.wfs_inv_lambda}, til2 = {sys.tar_inv_lambda, sys.tar_inv_lambda};
  f32x2 sdp = {0.f, 0.f}, sd2p = {0.f, 0.f}, R0rp = {0.f, 0.f}, R0ip = {0.f, 0.f};
  int nfull = 0;

  const const_float_p cflux = (const_float_p)(unsigned long long)sys.flux;
  // Loads of one lit tile.  UNCONDITIONAL: the loops below walk the compact list of this stripe's
  // lit tiles (sys.lit_info), so no load sits under a branch and the prefetch registers need no
  // copies where control flow joins (those copies were 9 % of the vector instructions).
  auto fetch = [&](int info, FrameRaw<NL, OTF> &raw) {
    const int t = (info >> 24) & 0x7F;
    {   // consumed first (slot write at the top of the tile): issued first, so that waiting for it never
        // waits for the layer loads behind it (vector loads return in order)
      const f32x4 v4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(shrs, shvo, (unsigned)t << shstep, 0));
#pragma unroll
      for (int j = 0; j < 4; j++) raw.SH.v[j] = v4[j];
    }
    raw.mrow = *reinterpret_cast<const uint16_t *>(mkb + 2u * (unsigned)t + mvo);
#pragma unroll
    for (int l = 0; l < NL; l++) {
      unsigned sx = 16u * (unsigned)t + lpxs[l]; sx -= (sx >= ldim[l]) ? ldim[l] : 0u;   // scalar
      // (whole-vector bit cast: a per-element __builtin_bit_cast of the result is narrowed to ONE dword
      // load by this compiler, ROCm 7.2)
      const f32x4 v4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs[l], lvo[l], 4u * sx, 0));
#pragma unroll
      for (int j = 0; j < 4; j++) raw.L[l][j] = v4[j];
    }
    if (!OTF) {
      const f4u p4 = *reinterpret_cast<const f4u *>(pztb + 64u * (unsigned)t + pvo);
#pragma unroll
      for (int j = 0; j < 4; j++) raw.P[OTF ? 0 : j] = p4.v[j];
    }
    raw.F = cflux[info & 0xFFFF];                            // scalar load; no sub-aperture: index 0, unused
  };

  // one lit tile: consume `cur`, then issue the loads of the tile `infon` describes into `nxt`
  // (dma_slot != nullptr: the pair walk below -- cur.L / mrow / F are filled in by the caller, the block's shared data
  // of this tile is already in `dma_slot`, nothing is fetched in here)
  // mode 0: the tile of the per-tile walk; 2: the same without the fetch of a further tile (the last tiles of a stripe)
  auto tile = [&](int info, int infon, FrameRaw<NL, OTF> &cur, FrameRaw<NL, OTF> &nxt, auto mode_tag, float4 *dma_slot) {
    constexpr bool DM = decltype(mode_tag)::value == 1;
    constexpr bool NOFETCH = decltype(mode_tag)::value != 0;
    const int t = (info >> 24) & 0x7F;
    // ---- stack-array DM phase of the tile on the matrix cores (independent of the loads)
    f32x4 S = PK ? NP4 : Z4;
    if (OTF) {
      f32x4 U = Z4;
#pragma unroll
      for (int kb = 0; kb < NB; kb++) {
        const float a = latq[(q + 4 * kb) * latw + t * tpn];
        U = mfma16(jm_ok ? a : 0.f, Pyr[kb], U);
      }
#pragma unroll
      for (int s = 0; s < NB; s++) S = mfma16(Pxr[s], U[s], S);
    }
    const float flux_i = cur.F;
    // environment-independent data of the tile through the block's LDS slot
    float4 *slot = DM ? dma_slot : shb + (nlit & 1) * 256;
    if constexpr (!DM) {
      nlit++;
      slot[wv * 64 + lane] = make_float4(cur.SH.v[0], cur.SH.v[1], cur.SH.v[2], cur.SH.v[3]);
#ifndef FW_DBG_NOBAR                             // (timing experiment: garbage results)
      __syncthreads();
#endif
    }
    if (!(DM && FW_SLOTS1)) {                      // (pair walk with single-buffered slots: the caller has read them)
      const float4 t0 = slot[lane], t1 = slot[64 + lane];
      cur.T[0] = t0.x; cur.T[1] = t0.y; cur.T[2] = t0.z; cur.T[3] = t0.w;
      cur.T[4] = t1.x; cur.T[5] = t1.y; cur.T[6] = t1.z; cur.T[7] = t1.w;
      cur.CS = slot[128 + lane];
    }
    const float4 csP = cur.CS;                     // HP: [hi | lo] halfs; fp32: the 4 K steps
    // (a wave without an environment of its own -- env_count not a multiple of 4 -- repeats the block's last
    // one: same loads, same arithmetic, same values stored twice.  No branch on `active` in here: a join behind
    // a branch that issues loads makes the compiler wait for ALL loads in flight, vmcnt(0), on both sides)
    // ---- phase of the 4 pixels, both complex amplitudes (registers)
    float wr[4], wi[4], ar[4], ai[4];
    if constexpr (PK) {
      // phase - pivot of the 4 pixels as two pairs: S + c0 X + c1 Y + layers (10 packed instructions)
      const f32x2 X01 = {cur.T[0], cur.T[1]}, X23 = {cur.T[2], cur.T[3]};
      const f32x2 Y01 = {cur.T[4], cur.T[5]}, Y23 = {cur.T[6], cur.T[7]};
      PK_GUARD_MFMA();
      f32x2 p01 = pk_fma_s(X01, c0c0, pk_lo(S)), p23 = pk_fma_s(X23, c0c0, pk_hi(S));
      p01 = pk_fma_s(Y01, c1c1, p01); p23 = pk_fma_s(Y23, c1c1, p23);
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const f32x2 l01 = {cur.L[l][0], cur.L[l][1]}, l23 = {cur.L[l][2], cur.L[l][3]};
        p01 = pk_add(p01, l01); p23 = pk_add(p23, l23);
      }
      if (info & FW_FULL) {
        const f32x2 t01 = pk_mul_s(p01, wil2), t23 = pk_mul_s(p23, wil2);
        const f32x2 b01 = pk_mul_s(p01, til2), b23 = pk_mul_s(p23, til2);
        if constexpr (QF) {
          // Slopes only: the sensor's phase goes to v_sin / v_cos as it stands, in revolutions, like the science
          // path's (the instructions reduce |x| <= 256 themselves; an 8 m pupil sees a few tens of revolutions at
          // 0.5 um).  The explicit round-and-subtract below keeps one rounding less in the REDUCED argument -- what
          // the 2e-5 image tolerance of the image-producing instantiations needs; the centre of gravity does not see
          // it (6e-8 x |revolutions| of phase per pixel, unbiased: 1e-7 pixel of centroid), and the reference itself
          // takes cos / sin of the unreduced float phase.  Six vector instructions less per tile.
          sdp += p01; sd2p = __builtin_elementwise_fma(p01, p01, sd2p);
          sdp += p23; sd2p = __builtin_elementwise_fma(p23, p23, sd2p);
          wr[0] = __builtin_amdgcn_cosf(t01.x); wi[0] = __builtin_amdgcn_sinf(t01.x);
          wr[1] = __builtin_amdgcn_cosf(t01.y); wi[1] = __builtin_amdgcn_sinf(t01.y);
          wr[2] = __builtin_amdgcn_cosf(t23.x); wi[2] = __builtin_amdgcn_sinf(t23.x);
          wr[3] = __builtin_amdgcn_cosf(t23.y); wi[3] = __builtin_amdgcn_sinf(t23.y);
        } else {
        // (plain vector arithmetic, not the in-place asm forms: behind an asm the allocator copied both accumulators
        // into fresh registers in front of the full tile and back at the join with the partial tile's path -- four
        // v_mov_b64 per full tile; the compiler packs these itself)
        sdp += p01; sd2p = __builtin_elementwise_fma(p01, p01, sd2p);
        const f32x2 r01 = {rintf(t01.x), rintf(t01.y)}, r23 = {rintf(t23.x), rintf(t23.y)};
        sdp += p23; sd2p = __builtin_elementwise_fma(p23, p23, sd2p);
        const f32x2 a01 = pk_fma_nc_s(p01, wil2, r01), a23 = pk_fma_nc_s(p23, wil2, r23);
        wr[0] = __builtin_amdgcn_cosf(a01.x); wi[0] = __builtin_amdgcn_sinf(a01.x);
        wr[1] = __builtin_amdgcn_cosf(a01.y); wi[1] = __builtin_amdgcn_sinf(a01.y);
        wr[2] = __builtin_amdgcn_cosf(a23.x); wi[2] = __builtin_amdgcn_sinf(a23.x);
        wr[3] = __builtin_amdgcn_cosf(a23.y); wi[3] = __builtin_amdgcn_sinf(a23.y);
        }
        ar[0] = __builtin_amdgcn_cosf(b01.x); ai[0] = __builtin_amdgcn_sinf(b01.x);
        ar[1] = __builtin_amdgcn_cosf(b01.y); ai[1] = __builtin_amdgcn_sinf(b01.y);
        ar[2] = __builtin_amdgcn_cosf(b23.x); ai[2] = __builtin_amdgcn_sinf(b23.x);
        ar[3] = __builtin_amdgcn_cosf(b23.y); ai[3] = __builtin_amdgcn_sinf(b23.y);
        nfull++;
      } else {
        const float ph4[4] = {p01.x, p01.y, p23.x, p23.y};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float ph = ph4[j];
          const bool m = (cur.mrow >> (4 * q + j)) & 1u;
          float a_ = ph * wfs_il; a_ -= rintf(a_);
          const float b_ = sci_rev(ph, tar_il);
          wr[j] = m ? __builtin_amdgcn_cosf(a_) : 0.f; wi[j] = m ? __builtin_amdgcn_sinf(a_) : 0.f;
          ar[j] = m ? __builtin_amdgcn_cosf(b_) : 0.f; ai[j] = m ? __builtin_amdgcn_sinf(b_) : 0.f;
          const float d = m ? ph : 0.f;
          sd += d; sd2 += d * d; sm += m ? 1.f : 0.f;
        }
      }
    } else if (info & FW_FULL) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float ph = (OTF ? S[j] : cur.P[OTF ? 0 : j]) + (c0 * cur.T[j] + c1 * cur.T[4 + j]);
#pragma unroll
        for (int l = 0; l < NL; l++) ph += cur.L[l][j];
        float a_ = ph * wfs_il; a_ -= rintf(a_);
        const float b_ = sci_rev(ph, tar_il);
        wr[j] = __builtin_amdgcn_cosf(a_); wi[j] = __builtin_amdgcn_sinf(a_);
        ar[j] = __builtin_amdgcn_cosf(b_); ai[j] = __builtin_amdgcn_sinf(b_);
        const float d = ph - pivot;
        sd += d; sd2 += d * d;
      }
      sm += 4.f;
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float ph = (OTF ? S[j] : cur.P[OTF ? 0 : j]) + (c0 * cur.T[j] + c1 * cur.T[4 + j]);
#pragma unroll
        for (int l = 0; l < NL; l++) ph += cur.L[l][j];
        const bool m = (cur.mrow >> (4 * q + j)) & 1u;
        float a_ = ph * wfs_il; a_ -= rintf(a_);
        const float b_ = sci_rev(ph, tar_il);
        wr[j] = m ? __builtin_amdgcn_cosf(a_) : 0.f; wi[j] = m ? __builtin_amdgcn_sinf(a_) : 0.f;
        ar[j] = m ? __builtin_amdgcn_cosf(b_) : 0.f; ai[j] = m ? __builtin_amdgcn_sinf(b_) : 0.f;
        const float d = m ? ph - pivot : 0.f;
        sd += d; sd2 += d * d; sm += m ? 1.f : 0.f;
      }
    }
    if constexpr (!NOFETCH) fetch(infon, nxt);               // these loads fly during the MFMAs
    // ---- science path (see the kernel's header)
    if (!(dbg & 2)) {
      if constexpr (PK) {
        const f32x2 ar01 = {ar[0], ar[1]}, ar23 = {ar[2], ar[3]}, ai01 = {ai[0], ai[1]}, ai23 = {ai[2], ai[3]};
        PK_GUARD_TRANS();
        const f32x2 sr_ = pk_add(ar01, ar23), si_ = pk_add(ai01, ai23);
        pk_acc_add(R0rp, sr_); pk_acc_add(R0ip, si_);
        PK_END_TO_MFMA();
      } else {
        R0r += (ar[0] + ar[1]) + (ar[2] + ar[3]);
        R0i += (ai[0] + ai[1]) + (ai[2] + ai[3]);
      }
      if (HP) {
        const hx8 csH = __builtin_bit_cast(hx8, csP);
        hx8 arH, arL, aiH, aiL;
        dup_hl(ar[0], ar[1], ar[2], ar[3], arH, arL);
        dup_hl(ai[0], ai[1], ai[2], ai[3], aiH, aiL);
        PA = mfma_h(arH, csH, PA); PA = mfma_h(arL, csH, PA);
        PB = mfma_h(aiH, csH, PB); PB = mfma_h(aiL, csH, PB);
      } else {
        PA = mfma16(ar[0], csP.x, PA); PB = mfma16(ai[0], csP.x, PB);
        PA = mfma16(ar[1], csP.y, PA); PB = mfma16(ai[1], csP.y, PB);
        PA = mfma16(ar[2], csP.z, PA); PB = mfma16(ai[2], csP.z, PB);
        PA = mfma16(ar[3], csP.w, PA); PB = mfma16(ai[3], csP.w, PB);
      }
    }
    // ---- WFS path (valid sub-apertures only; wave-uniform branch)
    if ((info & FW_SUB) && !(dbg & 1)) {
      if (HP) {
        float v[2][2][2];
        spot_dft_h_v(twh, wr, wi, Z4, v);
        spot_finish_v<NOISE, WRITE_CUBE>(sys, st, e, info & 0xFFFF, lane, v, do_cog, flux_i);
      } else if (!NOISE && !WRITE_CUBE) {
        if constexpr (QF) {
          const float z = spot_qf_moments(qfk, wr, wi, Z4);
          if (c == 0 && q < 3) reinterpret_cast<float *>(qmom + t)[q] = z;     // lanes 0, 16, 32: the three row totals
        } else if constexpr (PK) spot_cog_f32_pk(sys, st, e, info & 0xFFFF, lane, Cc, Ss, wr, wi, do_cog, Z4);
        else spot_cog_f32(sys, st, e, info & 0xFFFF, lane, Cc, Ss, wr, wi, do_cog, Z4);
      } else {
        spot_core<NOISE, WRITE_CUBE>(sys, st, e, info & 0xFFFF, lane, Cc, Ss, wr, wi, do_cog, flux_i, Z4);
      }
    }
  };

  // The stripe's lit tiles, compact (sys.lit_info[r][k], tile index in bits 24..30; the entries past
  // the last one repeat it, so the prefetch at the end of the list needs no test: the last tile is
  // loaded again, from L2, and dropped).  TWO tiles of loads are in flight (two register sets, list
  // walked in pairs; an odd first tile goes on its own).  The list is read with scalar loads.
  if constexpr (DMA) {
    // ---- the stripe's lit tiles in PAIRS (sys.pair_info): per pair ONE wait for everything fetched a pair ago, the
    // layer rows out of the wave's LDS image into registers, the block's shared data of both tiles into its slots,
    // ONE barrier, then the fetches of the next pair (in flight during the two tiles of this one) and the tiles.
    const std::integral_constant<int, 1> dm;
    const int np = sys.pair_count[r];
    const const_int_p pinfo = (const_int_p)(unsigned long long)(sys.pair_info + r * (2 * ((ntl + 1) / 2 + 2)));
    // loader role of the lane: row rr of an 8-row block, chunk (lane & 7) ^ rr of the 128-byte row piece
    const int rr = lane >> 3, kk = (lane & 7) ^ rr;
    unsigned dvo[NL][2];
#pragma unroll
    for (int l = 0; l < NL; l++) {
      const DevLayer &L = sys.layers[l];
      int py = L.toy + st.origin[(e * sys.nlayers + l) * 2 + 1]; py -= (py >= L.dim) ? L.dim : 0;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        unsigned pr = (unsigned)(16 * r + 8 * b + rr) + (unsigned)py; pr = min(pr, pr - ldim[l]);
        dvo[l][b] = 4u * (pr * (ldim[l] + RING_PAD)) + 16u * (unsigned)kk;
      }
    }
    // reader role: row c = 8 bc + rc, chunk 4 h + q of tile h
    const int bc = c >> 3, rc = c & 7;
    const char *rdp[2];
#pragma unroll
    for (int h = 0; h < 2; h++) rdp[h] = dimg + bc * FWD_BLK + 16 * (rc * 8 + ((4 * h + q) ^ rc));
    const unsigned dimg_lds = (unsigned)(unsigned long long)dimg;       // LDS byte address of the wave's images
    FrameRaw<NL, OTF> A, B;
    unsigned mrA, mrB;
    float fA, fB;
    const unsigned shb_lds = (unsigned)(unsigned long long)shb + 1024u * (unsigned)wv;   // this wave's quarter of a slot
    auto dma16 = [&](unsigned vo, __amdgpu_buffer_rsrc_t rs, unsigned lds_addr, unsigned so) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(vo), "s"(rs), "s"(lds_addr), "s"(so) : "memory");
    };
    // the block's shared data of a pair, one quarter per wave, filled in place (par: parity of the pair with
    // double-buffered slots; single-buffered: three quarters per tile, wave 3 -- whose quarter was a duplicate -- idles)
    auto issue_shared = [&](int ia, int ib, int par) {
      const int ta = (ia >> 24) & 0x7F, tb = (ib >> 24) & 0x7F;
      if (FW_SLOTS1) {
        if (wv < 3) {
          dma16(shvo, shrs, shb_lds, (unsigned)ta << shstep);
          dma16(shvo, shrs, shb_lds + 3072u, (unsigned)tb << shstep);
        }
      } else {
        dma16(shvo, shrs, shb_lds + (unsigned)par * 8192u, (unsigned)ta << shstep);
        dma16(shvo, shrs, shb_lds + (unsigned)par * 8192u + 4096u, (unsigned)tb << shstep);
      }
    };
    auto issue = [&](int ia, int ib) {
      const int ta = (ia >> 24) & 0x7F, tb = (ib >> 24) & 0x7F;
      mrA = *reinterpret_cast<const uint16_t *>(mkb + 2u * (unsigned)ta + mvo);
      mrB = *reinterpret_cast<const uint16_t *>(mkb + 2u * (unsigned)tb + mvo);
      fA = cflux[ia & 0xFFFF];
      fB = cflux[ib & 0xFFFF];
#pragma unroll
      for (int l = 0; l < NL; l++) {
        unsigned sx = 16u * (unsigned)(ta & ~1) + lpxs[l]; sx -= (sx >= ldim[l]) ? ldim[l] : 0u;   // scalar; 32 pixels never wrap
        const unsigned so = 4u * sx;
#pragma unroll
        for (int b = 0; b < 2; b++) dma16(dvo[l][b], lrs[l], dimg_lds + l * FWD_IMG + b * FWD_BLK, so);
      }
    };
    int ia = pinfo[0], ib = pinfo[1];
    if (np > 0) { issue_shared(ia, ib, 0); issue(ia, ib); }
    for (int k = 0; k < np; k++) {
      const int ja = pinfo[2 * k + 2], jb = pinfo[2 * k + 3];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this pair's layer images, shared quarters and masks have landed
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const float4 va = *reinterpret_cast<const float4 *>(rdp[0] + l * FWD_IMG);
        const float4 vb = *reinterpret_cast<const float4 *>(rdp[1] + l * FWD_IMG);
        A.L[l][0] = va.x; A.L[l][1] = va.y; A.L[l][2] = va.z; A.L[l][3] = va.w;
        B.L[l][0] = vb.x; B.L[l][1] = vb.y; B.L[l][2] = vb.z; B.L[l][3] = vb.w;
      }
      A.mrow = mrA; A.F = fA; B.mrow = mrB; B.F = fB;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the image has been read: the next pair may overwrite it
      if (FW_SLOTS1) {
        if (k + 1 < np) issue(ja, jb);                         // (the wave's own images, masks, flux: nobody else reads them)
        __syncthreads();                                       // every wave's quarters of THIS pair have landed
        float4 *slots = shb;
        {
          const float4 a0 = slots[lane], a1 = slots[64 + lane];
          A.CS = slots[128 + lane];
          A.T[0] = a0.x; A.T[1] = a0.y; A.T[2] = a0.z; A.T[3] = a0.w; A.T[4] = a1.x; A.T[5] = a1.y; A.T[6] = a1.z; A.T[7] = a1.w;
        }
        if (ia & FW_LIT) tile(ia, 0, A, A, dm, slots);
        {   // (the second tile's slot only now: twelve registers less across the first tile -- the kernel has to stay at
            //  112, the product's workgroup beside three of its waves needs the other 160 of the SIMD's 512)
          const float4 b0 = slots[192 + lane], b1 = slots[256 + lane];
          B.CS = slots[320 + lane];
          B.T[0] = b0.x; B.T[1] = b0.y; B.T[2] = b0.z; B.T[3] = b0.w; B.T[4] = b1.x; B.T[5] = b1.y; B.T[6] = b1.z; B.T[7] = b1.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                       // ... and everybody has read them: the next pair's may land
        if (k + 1 < np) issue_shared(ja, jb, 0);
        if (ib & FW_LIT) tile(ib, 0, B, B, dm, slots);
      } else {
        float4 *slots = shb + (k & 1) * 512;
        __syncthreads();
        if (k + 1 < np) { issue_shared(ja, jb, (k + 1) & 1); issue(ja, jb); }
        if (ia & FW_LIT) tile(ia, 0, A, A, dm, slots);
        if (ib & FW_LIT) tile(ib, 0, B, B, dm, slots + 256);
      }
      ia = ja; ib = jb;
    }
  } else {
  // The stripe's lit tiles, compact (sys.lit_info[r][k], tile index in bits 24..30; the entries past the last one
  // repeat it, so the prefetches at the end of the list need no test: the last tile is loaded again, from L2, and
  // dropped).  FW_DEPTH tiles of loads are in flight (as many register sets, the list walked in groups of that
  // many; the tiles left over come last, on data the last group prefetched).  The list is read with scalar loads.
  const std::integral_constant<int, 0> nd;
  const std::integral_constant<int, 2> nf;
  const int nl = sys.lit_count[r];
  const const_int_p linfo = (const_int_p)(unsigned long long)(sys.lit_info + r * (ntl + 8));
  {
    constexpr int D = FW_DEPTH;
    FrameRaw<NL, OTF> raw[D];
    int inf[D];
#pragma unroll
    for (int d = 0; d < D; d++) { inf[d] = linfo[d]; fetch(inf[d], raw[d]); }
    const int rem = nl % D, nfull = nl - rem;
    for (int k = 0; k < nfull; k += D) {
      int nx[D];
#pragma unroll
      for (int d = 0; d < D; d++) nx[d] = linfo[k + D + d];
#pragma unroll
      for (int d = 0; d < D; d++) tile(inf[d], nx[d], raw[d], raw[d], nd, nullptr);
#pragma unroll
      for (int d = 0; d < D; d++) inf[d] = nx[d];
    }
#pragma unroll
    for (int d = 0; d + 1 < D; d++)
      if (rem > d) tile(inf[d], inf[d], raw[d], raw[d], nf, nullptr);
  }
  }
  if (!active) return;
  if constexpr (QF) {
    // ---- slopes of the stripe's sub-apertures: lane t = tile t (the moments were written by this wave: its LDS
    // accesses complete in order)
    if (do_cog && !(dbg & 1)) {
      for (int tt = lane; tt < ntl; tt += 64) {
        const int ti = sys.tile_info[r * ntl + tt];
        if (ti & FW_SUB) {
          const float4 m = qmom[tt];
          float sx, sy;
          qf_slopes(sys, m.x, m.z, m.y, sx, sy);
          float *sl = st.slopes + (long long)e * sys.nslope;
          sl[ti & 0xFFFF] = sx;
          sl[sys.nvalid + (ti & 0xFFFF)] = sy;
        }
      }
    }
  }
  // ---- PSF rows of this stripe.  Register j of lane (q, c): row y = 4q + j, column c of the merged operand
  // (c < 8: cos of k = c + 1; c >= 8: sin of k = c - 7); the other half of a +-k pair sits in lane c ^ 8 of
  // the same 16-lane row (row_ror:8).  Lanes c < 8 write kx = -(c + 1), lanes 8 .. 14 kx = +(c - 7), lane 15
  // (k = 8 has no +8 in the window [-8, 8)) writes kx = 0 from the row sums.
  if constexpr (PK) {
    R0r += R0rp.x + R0rp.y; R0i += R0ip.x + R0ip.y;
    sd += sdp.x + sdp.y; sd2 += sd2p.x + sd2p.y; sm += 4.f * (float)nfull;
  }
  R0r += __shfl_xor(R0r, 16); R0r += __shfl_xor(R0r, 32);          // -> every lane: full sum of row y = c
  R0i += __shfl_xor(R0i, 16); R0i += __shfl_xor(R0i, 32);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float oa = dpp_f<0x128>(PA[j]), ob = dpp_f<0x128>(PB[j]);
    const float z0r = __shfl(R0r, 4 * q + j), z0i = __shfl(R0i, 4 * q + j);
    float rr, ri;
    int kxi;
    if (c < 8) { rr = PA[j] - ob; ri = PB[j] + oa; kxi = 7 - c; }
    else if (c < 15) { rr = oa + PB[j]; ri = ob - PA[j]; kxi = c + 1; }
    else { rr = z0r; ri = z0i; kxi = 8; }
    float2 *o = reinterpret_cast<float2 *>(TR) + ((long long)el * pd + (16 * r + 4 * q + j)) * 16 + kxi;
    *o = make_float2(rr, ri);
  }
  sd = wave_sum(sd); sd2 = wave_sum(sd2); sm = wave_sum(sm);
  if (lane == 0) {
    float *pp = TPART + ((long long)el * nblk + r) * 4;
    pp[0] = sd; pp[1] = sd2; pp[2] = sm; pp[3] = 0.f;
  }
}

// COG from a stored bincube (Rtc.do_centroids on its own): one wave per sub-aperture
__global__ __launch_bounds__(256) void k_cog(DevSys sys, DevState st, int env_begin) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wv;
  const int e = env_begin + blockIdx.y;
  if (i >= sys.nvalid) return;
  const int np2 = sys.npix * sys.npix;
  const float *im = st.bincube + ((long long)e * sys.nvalid + i) * np2;
  float s0 = 0.f, sx = 0.f, sy = 0.f;
  for (int p = lane; p < np2; p += 64) {
    float val = im[p];
    int y = p / sys.npix, x = p - y * sys.npix;
    s0 += val; sx += val * (float)x; sy += val * (float)y;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    s0 += __shfl_xor(s0, o); sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o);
  }
  if (lane == 0) {
    float *sl = st.slopes + (long long)e * sys.nslope;
    if (s0 != 0.f) {
      sl[i] = (sx / s0 - sys.cog_offset) * sys.cog_scale;
      sl[sys.nvalid + i] = (sy / s0 - sys.cog_offset) * sys.cog_scale;
    } else {
      sl[i] = 0.f; sl[sys.nvalid + i] = 0.f;
    }
  }
}

// geometric slopes from st->wfs_phase (calibration only): one wave per sub-aperture
__global__ __launch_bounds__(256) void k_slopes_geom(DevSys sys, DevState st, int env_begin) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wv;
  const int e = env_begin + blockIdx.y;
  if (i >= sys.nvalid) return;
  const int pd = sys.pdiam, n = sys.n;
  const int xy0 = sys.sub_xy[i];
  const int gx0 = xy0 & 0xFFFF, gy0 = xy0 >> 16;
  const float *ph = st.wfs_phase + (long long)e * n * n;
  float gx = 0.f, gyv = 0.f;
  for (int k = lane; k < pd * pd; k += 64) {
    int y = k / pd, x = k - y * pd;
    int xm = x > 0 ? x - 1 : x, xp = x < pd - 1 ? x + 1 : x;
    int ym = y > 0 ? y - 1 : y, yp = y < pd - 1 ? y + 1 : y;
    float m = sys.mpupil[(gy0 + y) * n + gx0 + x];
    float dx = (ph[(gy0 + y) * n + gx0 + xp] - ph[(gy0 + y) * n + gx0 + xm]) / (float)(xp - xm);
    float dy = (ph[(gy0 + yp) * n + gx0 + x] - ph[(gy0 + ym) * n + gx0 + x]) / (float)(yp - ym);
    gx += m * dx; gyv += m * dy;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { gx += __shfl_xor(gx, o); gyv += __shfl_xor(gyv, o); }
  if (lane == 0) {
    const float alpha = 0.206265f / sys.subapd;
    const float den = (float)pd * sys.flux[i];
    float *sl = st.slopes + (long long)e * sys.nslope;
    sl[i] = alpha * gx / den;
    sl[sys.nvalid + i] = alpha * gyv / den;
  }
}

// =============================================================================================
// controller glue
// =============================================================================================
// com += gain * err     (err = -cmat.s was produced by the GEMM with alpha = -1)
__global__ void k_integrate(float *__restrict__ com, const float *__restrict__ err, int nactu,
                            int ld, float gain, int env_begin, const float *__restrict__ env_gain) {
  const int e = env_begin + blockIdx.y;
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a < nactu) com[(long long)e * ld + a] += (env_gain ? env_gain[e] : gain) * err[(long long)e * ld + a];
}

// voltage = a com + b com1 + c com2 ; com2 <- com1 ; com1 <- com
__global__ void k_delay(DevState st, int nactu, int ld, float a, float b, float c, int env_begin,
                        int comp_voltage) {
  const int e = env_begin + blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nactu) return;
  const long long o = (long long)e * ld + i;
  const float c0 = st.com[o];
  if (comp_voltage) {
    const float c1 = st.com1[o], c2 = st.com2[o];
    st.voltage[o] = a * c0 + b * c1 + c * c2;
    st.com2[o] = c1;
    st.com1[o] = c0;
  } else {
    st.voltage[o] = c0;
  }
}

// k_delay whose newest command is still the split-K partial tiles P[z][n][nactu] of the m2v product:
// com = alpha * sum_z P[z] (k_gemm_reduce's expression, same order), then the delay line
__global__ void k_delay_sum(DevState st, int nactu, int ld, float a, float b, float c, int env_begin,
                            int comp_voltage, const float *__restrict__ P, int nsplit, float alpha, int nrows) {
  const int e = env_begin + blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nactu) return;
  const long long o = (long long)e * ld + i;
  const float s = slab_sum<4>(nsplit, [&](int z) { return P[((long long)z * nrows + blockIdx.y) * nactu + i]; });
  const float c0 = alpha * s;
  st.com[o] = c0;
  if (comp_voltage) {
    const float c1 = st.com1[o], c2 = st.com2[o];
    st.voltage[o] = a * c0 + b * c1 + c * c2;
    st.com2[o] = c1;
    st.com1[o] = c0;
  } else {
    st.voltage[o] = c0;
  }
}

// Frame pipeline, loop delay of exactly one frame: the voltages of the NEXT frame are the new commands
// (k_delay's expression with weights (0, 1, 0), evaluated one shift earlier), the delay line shifts, and the
// tip-tilt slot of that frame's dm_shape is filled by the same launch (rows >= n: dm_shape_tt_body with the
// voltages it needs recomputed by the rows' own expression), so that the frame kernel waits for ONE kernel
// behind the m2v product.  P: split-K partial tiles of that product (nsplit > 0) or null (st.com holds it).
__global__ void k_delay_ahead(DevSys sys, DevState st, int nactu, int ld, int n, const float *__restrict__ P,
                              int nsplit, float alpha, int ktt) {
  CHAIN_SETPRIO();
  auto newest = [&](int row, int a) -> float {
    if (nsplit > 0) {
      return alpha * slab_sum<4>(nsplit, [&](int z) { return P[((long long)z * n + row) * nactu + a]; });
    }
    return st.com[(long long)row * ld + a];
  };
  if ((int)blockIdx.y < n) {
    const int e = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nactu) return;
    const long long o = (long long)e * ld + i;
    const float c0 = newest(e, i);
    if (nsplit > 0) st.com[o] = c0;
    const float c1 = st.com1[o];
    st.voltage[o] = c0;
    st.com2[o] = c1;
    st.com1[o] = c0;
  } else if (blockIdx.x == 0 && ktt >= 0) {
    const int row = blockIdx.y - n, p = threadIdx.x;
    const DevDm &D = sys.dms[ktt];
    float *shape = st.dm_shape + (long long)row * sys.shape_stride + D.shape_off;
    if (p < 2) shape[p] = newest(row, D.com_off + p);
    if (p == 2 && sys.fused_ok) {
      const DevDm &Z = sys.dms[0];
      const int half = sys.pupdiam / 2, zp = (half + Z.toy) * Z.dim + half + Z.tox;
      const int ss2 = Z.ss * Z.ss, s0 = Z.influstart[zp], cn = Z.ninflu[zp];
      float acc = 0.f;
      for (int t = 0; t < cn; t++) {
        const int pos = Z.influpos[s0 + t];
        acc += Z.influ[pos] * newest(row, Z.com_off + pos / ss2);
      }
      shape[2] = acc;
    }
  }
}

__global__ void k_copy_rows(float *__restrict__ dst, int ldd, const float *__restrict__ src,
                            int lds, int ncols) {
  const int r = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ncols) dst[(long long)r * ldd + i] = src[(long long)r * lds + i];
}

// modes[e][action_modes[j]] += action[e][j] * freedom[action_modes[j]]
__global__ void k_modal_add(float *__restrict__ modes, int ldm, const float *__restrict__ action,
                            int nact, const int32_t *__restrict__ amodes,
                            const float *__restrict__ freedom) {
  const int r = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nact) {
    int m = amodes[j];
    modes[(long long)r * ldm + m] += action[(long long)r * nact + j] * freedom[m];
  }
}

// =============================================================================================
// science target: fused raytrace + windowed PSF (separable DFT) + phase variance
// stage 1: R[y][kx] = sum_x a(y, x) exp(-2 pi i kx x / Npsf), kx in [-hw, hw)
// =============================================================================================
#define TGT_XC 64
template <bool FROM_BUF>
__global__ __launch_bounds__(256) void k_target_rows(DevSys sys, DevState st, int env_begin,
                                                     float *__restrict__ TR,
                                                     float *__restrict__ TPART, int nblk) {
  extern __shared__ float smem[];
  const int W = 2 * sys.hw, RB = 256 / W, pd = sys.pupdiam, np = sys.npsf;
  float *sar = smem;                      // [RB][TGT_XC]
  float *sai = sar + RB * TGT_XC;
  float *red = sai + RB * TGT_XC;         // [3][256]
  float2 *stw = reinterpret_cast<float2 *>(red + 3 * 256);   // [np] (only if it fits)
  const bool tw_lds = np <= 4096;
  const int tid = threadIdx.x;
  const int e = env_begin + blockIdx.y;
  const int y0 = blockIdx.x * RB;
  const float2 *gtw = reinterpret_cast<const float2 *>(sys.psf_tw);
  if (tw_lds)
    for (int j = tid; j < np; j += 256) stw[j] = gtw[j];
  const float2 *tw = tw_lds ? stw : gtw;
  // pivot for the variance sums: phase at the pupil-grid centre
  float pivot;
  {
    const int cx = pd / 2, cy = pd / 2;
    float v = 0.f;
    if (FROM_BUF) {
      v = st.tar_phase[(long long)e * pd * pd + cy * pd + cx];
    } else {
      for (int l = 0; l < sys.nlayers; l++) {
        const DevLayer &L = sys.layers[l];
        const float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
        const int ox = st.origin[(e * sys.nlayers + l) * 2], oy = st.origin[(e * sys.nlayers + l) * 2 + 1];
        v += base[ring_idx(cx + L.tox, cy + L.toy, ox, oy, L.dim)];
      }
      for (int k = 0; k < sys.ndm; k++)
        v += dm_value(sys, st, e, k, cx + sys.dms[k].tox, cy + sys.dms[k].toy);
    }
    pivot = v;
  }
  const int kxi = tid % W, ys = tid / W;
  const int kxf = kxi - sys.hw;
  float accr = 0.f, acci = 0.f;
  float sd = 0.f, sd2 = 0.f, sm = 0.f;
  for (int x0 = 0; x0 < pd; x0 += TGT_XC) {
    __syncthreads();
    for (int p = tid; p < RB * TGT_XC; p += 256) {
      const int yy = p / TGT_XC, xx = p - yy * TGT_XC;
      const int y = y0 + yy, x = x0 + xx;
      float ar = 0.f, ai = 0.f;
      if (y < pd && x < pd) {
        const float m = sys.spupil[y * pd + x];
        if (m != 0.f) {
          float v = 0.f;
          if (FROM_BUF) {
            v = st.tar_phase[(long long)e * pd * pd + y * pd + x];
          } else {
            for (int l = 0; l < sys.nlayers; l++) {
              const DevLayer &L = sys.layers[l];
              const float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
              const int ox = st.origin[(e * sys.nlayers + l) * 2];
              const int oy = st.origin[(e * sys.nlayers + l) * 2 + 1];
              v += base[ring_idx(x + L.tox, y + L.toy, ox, oy, L.dim)];
            }
            for (int k = 0; k < sys.ndm; k++)
              v += dm_value(sys, st, e, k, x + sys.dms[k].tox, y + sys.dms[k].toy);
          }
          float t = v * sys.tar_inv_lambda;
          t -= rintf(t);
          float sn, cs;
          sincospif(2.0f * t, &sn, &cs);
          ar = m * cs; ai = m * sn;
          const float d = v - pivot;
          sd += d; sd2 += d * d; sm += 1.f;
        }
      }
      sar[p] = ar; sai[p] = ai;
    }
    __syncthreads();
    if (ys < RB) {
      const int xmax = (pd - x0) < TGT_XC ? (pd - x0) : TGT_XC;
      for (int xx = 0; xx < xmax; xx++) {
        const float ar = sar[ys * TGT_XC + xx], ai = sai[ys * TGT_XC + xx];
        const float2 w = tw[(kxf * (x0 + xx)) & (np - 1)];   // (cos, sin); e^{-i t} = cos - i sin
        accr += ar * w.x + ai * w.y;
        acci += ai * w.x - ar * w.y;
      }
    }
  }
  if (ys < RB && y0 + ys < pd) {
    float *o = TR + (((long long)blockIdx.y * pd + (y0 + ys)) * W + kxi) * 2;
    o[0] = accr; o[1] = acci;
  }
  // block partial sums for the variance
  red[tid] = sd; red[256 + tid] = sd2; red[512 + tid] = sm;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) {
      red[tid] += red[tid + o]; red[256 + tid] += red[256 + tid + o]; red[512 + tid] += red[512 + tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    float *pp = TPART + ((long long)blockIdx.y * nblk + blockIdx.x) * 4;
    pp[0] = red[0]; pp[1] = red[256]; pp[2] = red[512]; pp[3] = 0.f;
  }
}

// Target.comp_strehl(do_fit = True), the default of every get_strehl call in the reference
// (shesha/supervisor/components/targetCompass.py:139-159): the PSF peak fitted by two 1-D sincs, along x and along y
// through the maximum and its two neighbours: y(x) = A sinc(w (x - x0)); gain of one axis = A / y(0) = 1 / sinc(w x0).
// COMPASS's kernel is not in the reference tree: restated from its name and docstring (UNPINNED); the same
// algorithm as oracle/aoref.c:aoref_sinc_gain (Newton from the parabola through the three points, fall-backs).
__device__ __forceinline__ float sincf_(float t) { return fabsf(t) < 1e-2f ? 1.f - t * t * (1.f / 6.f) * (1.f - t * t * 0.05f) : __sinf(t) / t; }
__device__ __forceinline__ float dsincf_(float t) {
  return fabsf(t) < 1e-2f ? -t * (1.f / 3.f) * (1.f - t * t * 0.1f) : (__cosf(t) - __sinf(t) / t) / t;
}
__device__ float sinc_gain(float ym, float y0, float yp) {
  if (!(y0 > 0.f)) return 1.f;
  const float rm = ym / y0, rp = yp / y0;
  const float a = 0.5f * (rm + rp) - 1.f, b = 0.5f * (rp - rm);
  if (!(a < -1e-6f)) return 1.f;
  float x0 = fminf(0.5f, fmaxf(-0.5f, -b / (2.f * a)));
  const float gpar = 1.f - b * b / (4.f * a);
  float w = fminf(3.f, fmaxf(1e-3f, sqrtf(-6.f * a / gpar)));
  for (int it = 0; it < 8; it++) {
    const float f0 = sincf_(w * x0), d0 = dsincf_(w * x0);
    const float fm = sincf_(w * (1.f + x0)), dm = dsincf_(w * (1.f + x0));
    const float fp = sincf_(w * (1.f - x0)), dp = dsincf_(w * (1.f - x0));
    const float F1 = fm - rm * f0, F2 = fp - rp * f0;
    const float J11 = (1.f + x0) * dm - rm * x0 * d0, J12 = w * dm - rm * w * d0;
    const float J21 = (1.f - x0) * dp - rp * x0 * d0, J22 = -w * dp - rp * w * d0;
    const float det = J11 * J22 - J12 * J21;
    if (!(fabsf(det) > 1e-12f)) break;
    const float dw = (F1 * J22 - F2 * J12) / det, dx = (J11 * F2 - J21 * F1) / det;
    w = fminf(3.f, fmaxf(1e-3f, w - dw));
    x0 = fminf(0.6f, fmaxf(-0.6f, x0 - dx));
    if (fabsf(dw) + fabsf(dx) < 1e-6f) break;          // (quadratic convergence: 2 - 4 iterations)
  }
  const float g = 1.f / sincf_(w * x0);
  if (g >= 1.f && g < 1.5f) return g;
  return (gpar >= 1.f && gpar < 1.5f) ? gpar : 1.f;
}
// one axis (0: x, 1: y) of the fit of a W x W window whose maximum sits at `arg` (1 on the border: no fit)
__device__ float fit_axis_gain(const float *img, int W, int arg, int axis) {
  const int ay = arg / W, ax = arg - ay * W;
  if (ax == 0 || ay == 0 || ax == W - 1 || ay == W - 1) return 1.f;
  const int d = axis ? W : 1;
  return sinc_gain(img[arg - d], img[arg], img[arg + d]);
}

// The short-exposure window's fit, done where the window is formed (the PSF finish kernels run on the library's side
// stream, off the control chain): gain of both axes -> pend[W * W + 1].  One wave: argmax (lowest index on ties,
// like the commit's reduction and the oracle's scan), then lanes 0 / 1 solve the two axes.  Call behind a barrier that
// makes pend[0 .. W*W) visible.
__device__ __forceinline__ void psf_window_fit(float *pend, int W) {
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  float m = -1.f;
  int arg = 0;
  for (int o = lane; o < W * W; o += 64) { const float v = pend[o]; if (v > m) { m = v; arg = o; } }
  for (int d = 32; d >= 1; d >>= 1) {
    const float m2 = __shfl_xor(m, d);
    const int a2 = __shfl_xor(arg, d);
    if (m2 > m || (m2 == m && a2 < arg)) { m = m2; arg = a2; }
  }
  const float g = lane < 2 ? fit_axis_gain(pend, W, arg, lane) : 1.f;
  const float g1 = __shfl(g, 1);
  if (lane == 0) pend[W * W + 1] = g * g1;
}

// stage 2: G[ky][kx] = sum_y R[y][kx] exp(-2 pi i ky y / Npsf); |G|^2 -> pending window;
// variance from the block partials -> pending[W*W]
__global__ __launch_bounds__(256) void k_target_finish(DevSys sys, const float *__restrict__ TR,
                                                       const float *__restrict__ TPART, int nblk,
                                                       float *__restrict__ PEND) {
  const int W = 2 * sys.hw, pd = sys.pupdiam, np = sys.npsf;
  const int b = blockIdx.x;     // env within the batch
  const float2 *tw = reinterpret_cast<const float2 *>(sys.psf_tw);
  const float2 *R = reinterpret_cast<const float2 *>(TR) + (long long)b * pd * W;
  float *pend = PEND + (long long)b * (W * W + 4);
  for (int o = threadIdx.x; o < W * W; o += blockDim.x) {
    const int ky = o / W, kx = o - ky * W;
    const int kyf = ky - sys.hw;
    float gr = 0.f, gi = 0.f;
    for (int y = 0; y < pd; y++) {
      const float2 r = R[y * W + kx];
      const float2 w = tw[(kyf * y) & (np - 1)];
      gr += r.x * w.x + r.y * w.y;
      gi += r.y * w.x - r.x * w.y;
    }
    pend[o] = gr * gr + gi * gi;
  }
  __syncthreads();
  psf_window_fit(pend, W);
  if (threadIdx.x == 0) {
    double sd = 0., sd2 = 0., sm = 0.;
    for (int k = 0; k < nblk; k++) {
      const float *pp = TPART + ((long long)b * nblk + k) * 4;
      sd += pp[0]; sd2 += pp[1]; sm += pp[2];
    }
    double var = 0.;
    if (sm > 0.) { double mean = sd / sm; var = sd2 / sm - mean * mean; }
    pend[W * W] = (float)var;
  }
}


// ---------------------------------------------------------------------------------------------
// MFMA version (hw == 8 -> 16 kept frequencies per axis = one 16-wide MFMA tile).
// Block = 16 pupil rows; the complex amplitude of a 16 x 64 chunk is staged in LDS; the 4 waves
// split the chunk's 64 columns (K) and accumulate R[16 y][16 kx] with v_mfma_f32_16x16x4_f32:
//   Rr += ar*C + ai*S ; Ri += ai*C - ar*S ,  C/S[x][kx] = cos/sin(2 pi kx x / Npsf)
// ---------------------------------------------------------------------------------------------
template <bool FROM_BUF>
__global__ __launch_bounds__(256) void k_target_rows_mfma(DevSys sys, DevState st, int env_begin,
                                                          float *__restrict__ TR,
                                                          float *__restrict__ TPART, int nblk) {
  extern __shared__ float smem[];
  const int pd = sys.pupdiam, np = sys.npsf;
  float *sar = smem;                       // [16][65]
  float *sai = sar + 16 * 65;
  float *red = sai + 16 * 65;              // [4 waves][2][256] partial tiles; reused for sums
  float2 *stw = reinterpret_cast<float2 *>(red + 4 * 2 * 256);   // [np] if it fits
  const bool tw_lds = np <= 4096;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane >> 4, c = lane & 15;
  const int e = env_begin + blockIdx.y;
  const int y0 = blockIdx.x * 16;
  const float2 *gtw = reinterpret_cast<const float2 *>(sys.psf_tw);
  if (tw_lds)
    for (int j = tid; j < np; j += 256) stw[j] = gtw[j];
  const float2 *tw = tw_lds ? stw : gtw;
  float pivot;
  {
    const int cx = pd / 2, cy = pd / 2;
    float v = 0.f;
    if (FROM_BUF) {
      v = st.tar_phase[(long long)e * pd * pd + cy * pd + cx];
    } else {
      for (int l = 0; l < sys.nlayers; l++) {
        const DevLayer &L = sys.layers[l];
        const float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
        const int ox = st.origin[(e * sys.nlayers + l) * 2], oy = st.origin[(e * sys.nlayers + l) * 2 + 1];
        v += base[ring_idx(cx + L.tox, cy + L.toy, ox, oy, L.dim)];
      }
      for (int k = 0; k < sys.ndm; k++)
        v += dm_value(sys, st, e, k, cx + sys.dms[k].tox, cy + sys.dms[k].toy);
    }
    pivot = v;
  }
  f32x4 Rr = {0.f, 0.f, 0.f, 0.f}, Ri = {0.f, 0.f, 0.f, 0.f};
  float sd = 0.f, sd2 = 0.f, sm = 0.f;
  const int fy = tid >> 4, fx0 = (tid & 15) * 4;     // fill: row fy, 4 consecutive columns
  const int y = y0 + fy;
  const int kxf = c - 8;
  for (int x0 = 0; x0 < pd; x0 += 64) {
    __syncthreads();
    {
      float ph[4] = {0.f, 0.f, 0.f, 0.f};
      float mk[4] = {0.f, 0.f, 0.f, 0.f};
      const int xb = x0 + fx0;
      if (y < pd) {
        // pd % 4 == 0 (checked at create): a group of 4 columns is entirely inside or outside
        if (xb < pd) add4(mk, sys.spupil + y * pd + xb);
        if (FROM_BUF) {
          if (xb < pd) add4(ph, st.tar_phase + (long long)e * pd * pd + y * pd + xb);
        } else if (xb < pd) {
          for (int l = 0; l < sys.nlayers; l++) {
            const DevLayer &L = sys.layers[l];
            const float *base = st.screens + (long long)e * sys.screen_stride + L.screen_off;
            const int ox = st.origin[(e * sys.nlayers + l) * 2], oy = st.origin[(e * sys.nlayers + l) * 2 + 1];
            int py = y + L.toy + oy; py -= (py >= L.dim) ? L.dim : 0;
            int px = xb + L.tox + ox; px -= (px >= L.dim) ? L.dim : 0;
            add4_ring(ph, base + py * (L.dim + RING_PAD), px, L.dim);
          }
          for (int k = 0; k < sys.ndm; k++) {
            const DevDm &D = sys.dms[k];
            const float *slot = st.dm_shape + (long long)e * sys.shape_stride + D.shape_off;
            const int o = (y + D.toy) * D.dim + xb + D.tox;
            if (D.type == AOMARL_DM_PZT) {
              add4(ph, slot + o);
            } else {
              const float c0 = slot[0], c1 = slot[1];
              float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              add4(f, D.influ + 2 * o);
              add4(f + 4, D.influ + 2 * o + 4);
#pragma unroll
              for (int j = 0; j < 4; j++) ph[j] += c0 * f[2 * j] + c1 * f[2 * j + 1];
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float ar = 0.f, ai = 0.f;
        if (mk[j] != 0.f) {
          float t = ph[j] * sys.tar_inv_lambda;
          t -= rintf(t);
          ar = mk[j] * __builtin_amdgcn_cosf(t);
          ai = mk[j] * __builtin_amdgcn_sinf(t);
          const float d = ph[j] - pivot;
          sd += d; sd2 += d * d; sm += 1.f;
        }
        sar[fy * 65 + fx0 + j] = ar;
        sai[fy * 65 + fx0 + j] = ai;
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int xl = 16 * wv + 4 * s + q;
      const float ar = sar[c * 65 + xl], ai = sai[c * 65 + xl];
      const float2 w = tw[(kxf * (x0 + xl)) & (np - 1)];
      Rr = mfma16(ar, w.x, Rr);
      Ri = mfma16(ai, w.x, Ri);
      Rr = mfma16(ai, w.y, Rr);
      Ri = mfma16(ar, -w.y, Ri);
    }
  }
  __syncthreads();
  // cross-wave reduction of the 4 partial tiles; acc reg r of lane (q, c): y = 4q + r, kx = c
#pragma unroll
  for (int r = 0; r < 4; r++) {
    red[(wv * 2 + 0) * 256 + (4 * q + r) * 16 + c] = Rr[r];
    red[(wv * 2 + 1) * 256 + (4 * q + r) * 16 + c] = Ri[r];
  }
  __syncthreads();
  {
    const int yy = tid >> 4, kx = tid & 15;
    if (y0 + yy < pd) {
      float vr = 0.f, vi = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; w4++) { vr += red[(w4 * 2) * 256 + tid]; vi += red[(w4 * 2 + 1) * 256 + tid]; }
      float *o = TR + (((long long)blockIdx.y * pd + (y0 + yy)) * 16 + kx) * 2;
      o[0] = vr; o[1] = vi;
    }
  }
  __syncthreads();
  red[tid] = sd; red[256 + tid] = sd2; red[512 + tid] = sm;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) {
      red[tid] += red[tid + o]; red[256 + tid] += red[256 + tid + o]; red[512 + tid] += red[512 + tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    float *pp = TPART + ((long long)blockIdx.y * nblk + blockIdx.x) * 4;
    pp[0] = red[0]; pp[1] = red[256]; pp[2] = red[512]; pp[3] = 0.f;
  }
}

// Fast variant (production layout: NL layers, DMs = [stack array, tip-tilt], integer offsets):
// per-environment constants hoisted, all loads of a chunk independent and issued one chunk ahead
// of their use (software pipeline), double-buffered LDS tile -> one barrier per chunk.
template <int NL>
__global__ __launch_bounds__(256) void k_target_rows_fast(DevSys sys, DevState st, int env_begin,
                                                          float *__restrict__ TR,
                                                          float *__restrict__ TPART, int nblk) {
  extern __shared__ float smem[];
  const int pd = sys.pupdiam, np = sys.npsf;
  float *sar = smem;                       // [2][16][65]
  float *sai = sar + 2 * 16 * 65;
  float *red = sai + 2 * 16 * 65;          // [4 waves][2][256]
  float2 *stw = reinterpret_cast<float2 *>(red + 4 * 2 * 256);
  const bool tw_lds = np <= 4096;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane >> 4, c = lane & 15;
  const int e = env_begin + blockIdx.y;
  const int y0 = blockIdx.x * 16;
  const float2 *gtw = reinterpret_cast<const float2 *>(sys.psf_tw);
  if (tw_lds)
    for (int j = tid; j < np; j += 256) stw[j] = gtw[j];
  const float2 *tw = tw_lds ? stw : gtw;
  // ---- per-environment constants
  const float *lay[NL];
  int lpx[NL], lpy[NL], ldim[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    const DevLayer &L = sys.layers[l];
    lay[l] = st.screens + (long long)e * sys.screen_stride + L.screen_off;
    ldim[l] = L.dim;
    int px = L.tox + st.origin[(e * sys.nlayers + l) * 2]; px -= (px >= L.dim) ? L.dim : 0;
    int py = L.toy + st.origin[(e * sys.nlayers + l) * 2 + 1]; py -= (py >= L.dim) ? L.dim : 0;
    lpx[l] = px; lpy[l] = py;
  }
  const DevDm &D0 = sys.dms[0], &D1 = sys.dms[1];
  const float *pzt = st.dm_shape + (long long)e * sys.shape_stride + D0.shape_off;
  const float *ttslot = st.dm_shape + (long long)e * sys.shape_stride + D1.shape_off;
  const float c0 = ttslot[0], c1 = ttslot[1];
  const int fy = tid >> 4, fx0 = (tid & 15) * 4;
  const int y = y0 + fy;
  const bool row_ok = y < pd;
  // pivot of the variance sums: phase at the grid centre
  float pivot;
  {
    const int cx = pd / 2, cy = pd / 2;
    float v = pzt[(cy + D0.toy) * D0.dim + cx + D0.tox];
    const float2 f = reinterpret_cast<const float2 *>(D1.influ)[(cy + D1.toy) * D1.dim + cx + D1.tox];
    v += c0 * f.x + c1 * f.y;
#pragma unroll
    for (int l = 0; l < NL; l++) {
      int py = cy + lpy[l]; py -= (py >= ldim[l]) ? ldim[l] : 0;
      int px = cx + lpx[l]; px -= (px >= ldim[l]) ? ldim[l] : 0;
      v += lay[l][py * (ldim[l] + RING_PAD) + px];
    }
    pivot = v;
  }
  // row pointers of this thread
  const float *lrow[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    int py = (row_ok ? y : 0) + lpy[l]; py -= (py >= ldim[l]) ? ldim[l] : 0;
    lrow[l] = lay[l] + py * (ldim[l] + RING_PAD);
  }
  const float *prow = pzt + ((row_ok ? y : 0) + D0.toy) * D0.dim + D0.tox;
  const float *trow = D1.influ + 2 * (((row_ok ? y : 0) + D1.toy) * D1.dim + D1.tox);
  const float *mrow = sys.spupil + (row_ok ? y : 0) * pd;

  float rL[NL][4], rP[4], rT[8], rM[4];
  auto fetch = [&](int x0) {
    const int xb = x0 + fx0;
    const bool ok = row_ok && xb < pd;        // pd % 4 == 0: the 4 columns are in or out together
    const int xs = ok ? xb : 0;
#pragma unroll
    for (int l = 0; l < NL; l++) {
      int px = xs + lpx[l]; px -= (px >= ldim[l]) ? ldim[l] : 0;
      const f4u t = *reinterpret_cast<const f4u *>(lrow[l] + px);
#pragma unroll
      for (int j = 0; j < 4; j++) rL[l][j] = t.v[j];
    }
    const f4u tp = *reinterpret_cast<const f4u *>(prow + xs);
    const f4u t0 = *reinterpret_cast<const f4u *>(trow + 2 * xs);
    const f4u t1 = *reinterpret_cast<const f4u *>(trow + 2 * xs + 4);
    const f4u tm = *reinterpret_cast<const f4u *>(mrow + xs);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      rP[j] = tp.v[j]; rT[j] = t0.v[j]; rT[4 + j] = t1.v[j];
      rM[j] = ok ? tm.v[j] : 0.f;
    }
  };

  f32x4 Rr = {0.f, 0.f, 0.f, 0.f}, Ri = {0.f, 0.f, 0.f, 0.f};
  float sd = 0.f, sd2 = 0.f, sm = 0.f;
  const int kxf = c - 8;
  fetch(0);
  int buf = 0;
  for (int x0 = 0; x0 < pd; x0 += 64, buf ^= 1) {
    float *bar = sar + buf * 16 * 65, *bai = sai + buf * 16 * 65;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float ph = rP[j] + (c0 * rT[2 * j] + c1 * rT[2 * j + 1]);
#pragma unroll
      for (int l = 0; l < NL; l++) ph += rL[l][j];
      float ar = 0.f, ai = 0.f;
      if (rM[j] != 0.f) {
        float t = ph * sys.tar_inv_lambda;
        t -= rintf(t);
        ar = rM[j] * __builtin_amdgcn_cosf(t);
        ai = rM[j] * __builtin_amdgcn_sinf(t);
        const float d = ph - pivot;
        sd += d; sd2 += d * d; sm += 1.f;
      }
      bar[fy * 65 + fx0 + j] = ar;
      bai[fy * 65 + fx0 + j] = ai;
    }
    if (x0 + 64 < pd) fetch(x0 + 64);       // next chunk's loads fly during the barrier + MFMAs
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int xl = 16 * wv + 4 * s + q;
      const float ar = bar[c * 65 + xl], ai = bai[c * 65 + xl];
      const float2 w = tw[(kxf * (x0 + xl)) & (np - 1)];
      Rr = mfma16(ar, w.x, Rr);
      Ri = mfma16(ai, w.x, Ri);
      Rr = mfma16(ai, w.y, Rr);
      Ri = mfma16(ar, -w.y, Ri);
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    red[(wv * 2 + 0) * 256 + (4 * q + r) * 16 + c] = Rr[r];
    red[(wv * 2 + 1) * 256 + (4 * q + r) * 16 + c] = Ri[r];
  }
  __syncthreads();
  {
    const int yy = tid >> 4, kx = tid & 15;
    if (y0 + yy < pd) {
      float vr = 0.f, vi = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; w4++) { vr += red[(w4 * 2) * 256 + tid]; vi += red[(w4 * 2 + 1) * 256 + tid]; }
      float *o = TR + (((long long)blockIdx.y * pd + (y0 + yy)) * 16 + kx) * 2;
      o[0] = vr; o[1] = vi;
    }
  }
  __syncthreads();
  red[tid] = sd; red[256 + tid] = sd2; red[512 + tid] = sm;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) {
      red[tid] += red[tid + o]; red[256 + tid] += red[256 + tid + o]; red[512 + tid] += red[512 + tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    float *pp = TPART + ((long long)blockIdx.y * nblk + blockIdx.x) * 4;
    pp[0] = red[0]; pp[1] = red[256]; pp[2] = red[512]; pp[3] = 0.f;
  }
}

// stage 2 on MFMA: G[ky][kx] = sum_y E[ky][y] R[y][kx]; 4 waves split y, one block per env
__global__ __launch_bounds__(256) void k_target_finish_mfma(DevSys sys, const float *__restrict__ TR,
                                                            const float *__restrict__ TPART, int nblk,
                                                            float *__restrict__ PEND,
                                                            uint32_t *__restrict__ frame) {
  __shared__ float red[4 * 2 * 256];
  __shared__ double dred[3][64];
  const int pd = sys.pupdiam, np = sys.npsf;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane >> 4, c = lane & 15;
  const float2 *tw = reinterpret_cast<const float2 *>(sys.psf_tw);
  const float2 *R = reinterpret_cast<const float2 *>(TR) + (long long)b * pd * 16;
  float *pend = PEND + (long long)b * (256 + 4);
  const int kyf = c - 8;
  f32x4 Gr = {0.f, 0.f, 0.f, 0.f}, Gi = {0.f, 0.f, 0.f, 0.f};
  const int per = ((pd + 15) / 16) * 4;              // rows per wave, multiple of 4
  const int yb = wv * per, ye = min(pd, yb + per);
  // 8 steps (32 rows) of operands are fetched before their MFMAs: the loop is latency-bound
  for (int y0 = yb; y0 < ye; y0 += 32) {
    float2 r[8], w[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int y = y0 + 4 * u + q;
      r[u] = make_float2(0.f, 0.f); w[u] = make_float2(0.f, 0.f);
      if (y < ye) {
        r[u] = R[y * 16 + c];                          // B operand: R[y][kx = c]
        w[u] = tw[(kyf * y) & (np - 1)];               // A operand: E[ky = c - 8][y]
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      Gr = mfma16(w[u].x, r[u].x, Gr);
      Gi = mfma16(w[u].x, r[u].y, Gi);
      Gr = mfma16(w[u].y, r[u].y, Gr);
      Gi = mfma16(-w[u].y, r[u].x, Gi);
    }
  }
#pragma unroll
  for (int r4 = 0; r4 < 4; r4++) {
    red[(wv * 2 + 0) * 256 + (4 * q + r4) * 16 + c] = Gr[r4];
    red[(wv * 2 + 1) * 256 + (4 * q + r4) * 16 + c] = Gi[r4];
  }
  if (tid < 64) {
    double sd = 0., sd2 = 0., sm = 0.;
    for (int k = tid; k < nblk; k += 64) {
      const float *pp = TPART + ((long long)b * nblk + k) * 4;
      sd += pp[0]; sd2 += pp[1]; sm += pp[2];
    }
    dred[0][tid] = sd; dred[1][tid] = sd2; dred[2][tid] = sm;
  }
  __syncthreads();
  {
    float gr = 0.f, gi = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < 4; w4++) { gr += red[(w4 * 2) * 256 + tid]; gi += red[(w4 * 2 + 1) * 256 + tid]; }
    pend[tid] = gr * gr + gi * gi;                     // tid = ky * 16 + kx
  }
  __syncthreads();
  psf_window_fit(pend, 16);
  if (tid == 0) {
    double sd = 0., sd2 = 0., sm = 0.;
    for (int k = 0; k < 64; k++) { sd += dred[0][k]; sd2 += dred[1][k]; sm += dred[2][k]; }
    double var = 0.;
    if (sm > 0.) { double mean = sd / sm; var = sd2 / sm - mean * mean; }
    pend[256] = (float)var;
    if (frame) frame[b] += 1u;                         // WFS noise frame counter (one-pass path)
  }
}

// publish the pending PSF: LE accumulation, Strehl SE / LE, variance bookkeeping
__device__ __forceinline__ void strehl_commit_body(const DevSys &sys, const DevState &st, int env_begin, int blk,
                                                   const float *__restrict__ PEND) {
  __shared__ float r0[256], r1[256];
  __shared__ int ri[256], rl[256];
  const int W = 2 * sys.hw;
  const int e = env_begin + blk;
  const float *pend = PEND + (long long)blk * (W * W + 4);
  float *le = st.le_img + (long long)e * W * W;
  float mse = 0.f, mle = 0.f;
  int arg = 0, argl = 0;
  // (the first 256 threads of the block do the work: blocks of 256 -- every caller but k_small_actor_head -- all of them)
  const int nthr = min((int)blockDim.x, 256);
  if ((int)threadIdx.x < nthr) {
    for (int o = threadIdx.x; o < W * W; o += nthr) {
      const float p = pend[o];
      const float l = le[o] + p;
      le[o] = l;
      if (p > mse) { mse = p; arg = o; }
      if (l > mle) { mle = l; argl = o; }
    }
    r0[threadIdx.x] = mse; r1[threadIdx.x] = mle; ri[threadIdx.x] = arg; rl[threadIdx.x] = argl;
  }
  __syncthreads();
  // (ties go to the lower index, like the scans of the oracle: the sinc fit reads the neighbours of THAT pixel)
  for (int o = 128; o >= 1; o >>= 1) {
    if (threadIdx.x < o) {
      const int t = threadIdx.x, u = t + o;
      if (r0[u] > r0[t] || (r0[u] == r0[t] && ri[u] < ri[t])) { r0[t] = r0[u]; ri[t] = ri[u]; }
      if (r1[u] > r1[t] || (r1[u] == r1[t] && rl[u] < rl[t])) { r1[t] = r1[u]; rl[t] = rl[u]; }
    }
    __syncthreads();
  }
  // comp_strehl(do_fit = True): the short exposure's fitted gain comes with the window (psf_window_fit, computed
  // where the window was formed: this kernel sits on the control chain's critical path); the long exposure's is
  // solved when somebody reads it (k_strehl_fit_le): slot 7 holds the un-fitted value until then.
  if (threadIdx.x == 0) {
    float *s = st.strehl + (long long)e * 8;
    const float cnt = s[4] + 1.f;
    const float var = pend[W * W];
    const int ay = ri[0] / W, ax = ri[0] - ay * W;
    s[0] = r0[0] / sys.ref_peak;
    s[1] = r1[0] / cnt / sys.ref_peak;
    s[2] = var;
    s[3] = s[3] + var;
    s[4] = cnt;
    s[5] = (ay == 0 || ax == 0 || ay == W - 1 || ax == W - 1) ? 1.f : 0.f;
    s[6] = r0[0] * pend[W * W + 1] / sys.ref_peak;
    s[7] = r1[0] / cnt / sys.ref_peak;
  }
}

__global__ __launch_bounds__(256) void k_strehl_commit(DevSys sys, DevState st, int env_begin,
                                                       const float *__restrict__ PEND) {
  strehl_commit_body(sys, st, env_begin, blockIdx.x, PEND);
}

// What follows the delay line in one launch: blocks [0, n) commit the pending PSF window (k_strehl_commit),
// blocks [n, 2n) refresh the tip-tilt mirror's coefficients and the frame kernel's pivot from the
// new voltages (k_dm_shape's tip-tilt branch).  The two halves touch different data.
__global__ __launch_bounds__(256) void k_post_delay(DevSys sys, DevState st, int env_begin, int n,
                                                    const float *__restrict__ PEND, int do_strehl, int ktt,
                                                    const float *__restrict__ volts, int ldv) {
  CHAIN_SETPRIO();
  if ((int)blockIdx.x < n) {
    if (do_strehl) strehl_commit_body(sys, st, env_begin, blockIdx.x, PEND);
  } else {
    dm_shape_tt_body(sys, st, env_begin, blockIdx.x - n, ktt, volts, ldv, threadIdx.x);
  }
}

// comp_strehl(do_fit = True), long exposure: the fitted peak of the accumulated window -> slot 7 (on demand:
// aomarl_strehl_fit; one wave per environment)
__global__ __launch_bounds__(64) void k_strehl_fit_le(DevSys sys, DevState st, int env_begin) {
  __shared__ float g2[4];
  const int W = 2 * sys.hw, e = env_begin + blockIdx.x, lane = threadIdx.x;
  float *le = st.le_img + (long long)e * W * W;
  float m = -1.f;
  int arg = 0;
  for (int o = lane; o < W * W; o += 64) { const float v = le[o]; if (v > m) { m = v; arg = o; } }
  for (int d = 32; d >= 1; d >>= 1) {
    const float m2 = __shfl_xor(m, d);
    const int a2 = __shfl_xor(arg, d);
    if (m2 > m || (m2 == m && a2 < arg)) { m = m2; arg = a2; }
  }
  if (lane < 2) g2[lane] = fit_axis_gain(le, W, arg, lane);
  __syncthreads();
  if (lane == 0) {
    float *s = st.strehl + (long long)e * 8;
    s[7] = s[4] > 0.f ? m * g2[0] * g2[1] / s[4] / sys.ref_peak : 0.f;
  }
}

__global__ void k_strehl_reset(DevSys sys, DevState st, int env_begin) {
  const int W = 2 * sys.hw;
  const int e = env_begin + blockIdx.x;
  for (int o = threadIdx.x; o < W * W; o += blockDim.x) st.le_img[(long long)e * W * W + o] = 0.f;
  if (threadIdx.x < 8) st.strehl[(long long)e * 8 + threadIdx.x] = 0.f;
}
