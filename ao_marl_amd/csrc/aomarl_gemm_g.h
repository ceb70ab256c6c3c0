// aomarl_gemm_g.h -- grouped fp32 GEMM of the learner (round 5), gfx950 only: k_gemm_p's inner loop (independent
// v_mfma_f32_16x16x4_f32 accumulators per wave, global -> registers -> LDS with two k-tiles in flight, ONE barrier per
// k-tile in the middle of its matrix instructions, exchanged operands for 16-byte stores) for the products of
// aomarl_sac_update, where the GROUP is the agent and an operand may lie either way in memory:
//
//   C[g][M][N] = opA(A[g]) . opB(B[g])        AK: A is [M][K] (k contiguous)   !AK: A is [K][M] (m contiguous)
//                                             BK: B is [N][K] (k contiguous)   !BK: B is [K][N] (n contiguous)
//
//   forward            Y  = X W + b, ReLU       AK, !BK   (model_rpc.py:72-84,121-135: x [B][in], W kept [in][out])
//   input gradient     dX = dY W^T [ReLU mask]  AK,  BK   (autograd of the same lines, train_rpc.py:1030-1064)
//   weight gradient    dW = X^T dY, db = 1^T dY !AK, !BK
//
// An operand that is contiguous along m / n is staged as it lies ([k][m] rows in LDS) and read by the matrix
// instruction's lanes as W consecutive floats of one k row: lane (q, i) gets rows m = W i + g of the wave's tile for its
// W accumulators -- a permutation of the tile's rows that the epilogue undoes for free (a lane then holds 4 W
// consecutive columns of C).  No transposition through LDS, one ds_read per W matrix instructions, as for a
// k-contiguous operand (one ds_read_b128 = 4 k steps of one accumulator row).
// Epilogues: + bias[n], ReLU, zero where mask <= 0; the k-sums of the B operand's columns (the bias gradient) are
// accumulated from the staged pieces by the workgroups of the first tile row and written in a fixed order.
// Operands: 16-byte aligned, leading dimensions and group strides multiples of 4 floats; an m- / n-contiguous
// operand's rows must hold roundup4(M or N) readable floats (they do: the leading dimension is a multiple of 4).
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <type_traits>

#ifndef GP_TYPES
#define GP_TYPES
typedef float gp_f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) gp_f4u { float v[4]; };
#endif

#define GG_KT 32          // k-tile

struct GemmGArgs {
  int M, N, K;
  const float *A; int lda; long long sA;
  const float *B; int ldb; long long sB;
  float *C; int ldc; long long sC;
  const float *bias; long long sBias;            // + bias[g][n]
  int relu;
  const float *mask; int ldm; long long sM;      // C = mask[g][m][n] > 0 ? v : 0
  float *colsum; long long sCs;                  // [g][N] = sum_k B[k][n]  (!BK only)
  int tiles_m, tiles_n, ntile;                   // tiles per group (tiles_m x tiles_n) and in all
};

template <int W> struct gg_vec;
template <> struct gg_vec<2> { typedef float2 t; };
template <> struct gg_vec<4> { typedef float4 t; };

// one workgroup = tile `w` of the problem's (group-major) tile list
template <int WM, int WN, bool AK, bool BK, bool CS = false>
__device__ __forceinline__ void gemm_g_body(const GemmGArgs &a, const int w) {
  static_assert((WM == 2 || WM == 4) && (WN == 2 || WN == 4), "wave tiles of 32 or 64 rows / columns");
  constexpr int BM = 32 * WM, BN = 32 * WN;
  constexpr int PA = AK ? GG_KT : BM + 4, PB = BK ? GG_KT : BN + 4;       // LDS row pitch (floats)
  constexpr int SA = AK ? BM * GG_KT : GG_KT * (BM + 4), SB = BK ? BN * GG_KT : GG_KT * (BN + 4);
  extern __shared__ __attribute__((aligned(16))) float gg_lds[];
  float *As = gg_lds, *Bs = gg_lds + 2 * SA;
  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, li = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wv >> 1, wn = wv & 1;

  const int tiles = a.tiles_m * a.tiles_n;
  const int grp = w / tiles, t = w - grp * tiles;
  const int tm = t / a.tiles_n, tn = t - tm * a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = a.M, N = a.N, K = a.K;
  const float *A = a.A + (long long)grp * a.sA, *B = a.B + (long long)grp * a.sB;
  const int nkt = (K + GG_KT - 1) / GG_KT;

  // ---- staging.  k-contiguous operand: thread -> (row lr + 32 p, 4 floats at k = lc), 16-byte pieces XOR-swizzled
  // by row in LDS (aomarl_gemm_p.h).  m-contiguous operand: thread -> (k row kr + (32 / W) p, 4 floats at column
  // 4 pc), LDS rows of BM + 4 floats (rows 8 apart shifted by half the banks: the 64-bit fragment reads of two q).
  const int lr = tid >> 3, lc = (tid & 7) * 4;
  const int lcs = (((tid & 7) ^ ((lr >> 1) & 7)) << 2);
  constexpr int PRA = 8 * WM, PRB = 8 * WN;      // 16-byte pieces per LDS row of an m- / n-contiguous operand
  const int kra = tid / PRA, pca = tid % PRA, krb = tid / PRB, pcb = tid % PRB;
  const float *pa[WM], *pb[WN];
  if (AK) {
#pragma unroll
    for (int p = 0; p < WM; p++) pa[p] = A + (long long)min(m0 + lr + 32 * p, M - 1) * a.lda;
  } else {
    const int col = min(m0 + 4 * pca, ((M + 3) & ~3) - 4);
#pragma unroll
    for (int p = 0; p < WM; p++) pa[p] = A + col;
  }
  if (BK) {
#pragma unroll
    for (int p = 0; p < WN; p++) pb[p] = B + (long long)min(n0 + lr + 32 * p, N - 1) * a.ldb;
  } else {
    const int col = min(n0 + 4 * pcb, ((N + 3) & ~3) - 4);
#pragma unroll
    for (int p = 0; p < WN; p++) pb[p] = B + col;
  }
  const int klast = (K - 1) & ~3;
  float4 ra[2][WM], rb[2][WN];
  auto gload = [&](int kt, int st) {
    if (AK) {
      const int k = min(kt * GG_KT + lc, klast);
#pragma unroll
      for (int p = 0; p < WM; p++) ra[st][p] = *reinterpret_cast<const float4 *>(pa[p] + k);
    } else {
#pragma unroll
      for (int p = 0; p < WM; p++)
        ra[st][p] = *reinterpret_cast<const float4 *>(pa[p] + (long long)min(kt * GG_KT + kra + (32 / WM) * p, K - 1) * a.lda);
    }
    if (BK) {
      const int k = min(kt * GG_KT + lc, klast);
#pragma unroll
      for (int p = 0; p < WN; p++) rb[st][p] = *reinterpret_cast<const float4 *>(pb[p] + k);
    } else {
#pragma unroll
      for (int p = 0; p < WN; p++)
        rb[st][p] = *reinterpret_cast<const float4 *>(pb[p] + (long long)min(kt * GG_KT + krb + (32 / WN) * p, K - 1) * a.ldb);
    }
  };
  const bool want_cs = CS && !BK && a.colsum != nullptr && tm == 0;   // block-uniform (CS: instantiations without the sums carry no code for them)
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  // TAILc: the tile crosses the end of K (only ever the last one): its own instantiation of the step, so that the
  // k-tiles in front of it carry no masks (as selects they cost 27 vector instructions per k-tile of 32 matrix ones)
  auto lstore = [&](int buf, int st, int kt, auto TAILc) {
    constexpr bool tail = decltype(TAILc)::value;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if (AK) {
      float *as = As + buf * SA + lr * PA + lcs;
      const int k = kt * GG_KT + lc;
#pragma unroll
      for (int p = 0; p < WM; p++) {
        float4 v = ra[st][p];
        if (tail) { v.x = k < K ? v.x : 0.f; v.y = k + 1 < K ? v.y : 0.f; v.z = k + 2 < K ? v.z : 0.f; v.w = k + 3 < K ? v.w : 0.f; }
        *reinterpret_cast<float4 *>(as + 32 * p * PA) = v;
      }
    } else {
      float *as = As + buf * SA + kra * PA + 4 * pca;
#pragma unroll
      for (int p = 0; p < WM; p++) {
        float4 v = ra[st][p];
        if (tail && kt * GG_KT + kra + (32 / WM) * p >= K) v = zero;
        *reinterpret_cast<float4 *>(as + (32 / WM) * p * PA) = v;
      }
    }
    if (BK) {
      float *bs = Bs + buf * SB + lr * PB + lcs;
      const int k = kt * GG_KT + lc;
#pragma unroll
      for (int p = 0; p < WN; p++) {
        float4 v = rb[st][p];
        if (tail) { v.x = k < K ? v.x : 0.f; v.y = k + 1 < K ? v.y : 0.f; v.z = k + 2 < K ? v.z : 0.f; v.w = k + 3 < K ? v.w : 0.f; }
        *reinterpret_cast<float4 *>(bs + 32 * p * PB) = v;
      }
    } else {
      float *bs = Bs + buf * SB + krb * PB + 4 * pcb;
#pragma unroll
      for (int p = 0; p < WN; p++) {
        float4 v = rb[st][p];
        if (tail && kt * GG_KT + krb + (32 / WN) * p >= K) v = zero;
        *reinterpret_cast<float4 *>(bs + (32 / WN) * p * PB) = v;
        if (want_cs) { cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w; }
      }
    }
  };

  gp_f32x4 acc[WM][WN];
#pragma unroll
  for (int g = 0; g < WM; g++)
#pragma unroll
    for (int h = 0; h < WN; h++) acc[g][h] = (gp_f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- operand fragments: oa[j][c][g] is what matrix instruction (j, c) takes for accumulator row g; its k lane q
  // stands for k = 8 q + 4 j + c of the tile on both operands.
  float oa[2][4][WM], ob[2][4][WN];
  const int fo[2] = {li * GG_KT + (((2 * q) ^ ((li >> 1) & 7)) << 2), li * GG_KT + (((2 * q + 1) ^ ((li >> 1) & 7)) << 2)};
  auto fread = [&](int buf, int j) {
    if (AK) {
      const float *as = As + buf * SA + wm * (16 * WM) * PA + fo[j];
#pragma unroll
      for (int g = 0; g < WM; g++) {
        const float4 v = *reinterpret_cast<const float4 *>(as + g * 16 * PA);
        oa[j][0][g] = v.x; oa[j][1][g] = v.y; oa[j][2][g] = v.z; oa[j][3][g] = v.w;
      }
    } else {
      const float *as = As + buf * SA + (8 * q + 4 * j) * PA + wm * (16 * WM) + WM * li;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const typename gg_vec<WM>::t v = *reinterpret_cast<const typename gg_vec<WM>::t *>(as + c * PA);
        const float *vf = reinterpret_cast<const float *>(&v);
#pragma unroll
        for (int g = 0; g < WM; g++) oa[j][c][g] = vf[g];
      }
    }
    if (BK) {
      const float *bs = Bs + buf * SB + wn * (16 * WN) * PB + fo[j];
#pragma unroll
      for (int h = 0; h < WN; h++) {
        const float4 v = *reinterpret_cast<const float4 *>(bs + h * 16 * PB);
        ob[j][0][h] = v.x; ob[j][1][h] = v.y; ob[j][2][h] = v.z; ob[j][3][h] = v.w;
      }
    } else {
      const float *bs = Bs + buf * SB + (8 * q + 4 * j) * PB + wn * (16 * WN) + WN * li;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const typename gg_vec<WN>::t v = *reinterpret_cast<const typename gg_vec<WN>::t *>(bs + c * PB);
        const float *vf = reinterpret_cast<const float *>(&v);
#pragma unroll
        for (int h = 0; h < WN; h++) ob[j][c][h] = vf[h];
      }
    }
  };
#define GG_MMA(j, c)                                                                                              \
  _Pragma("unroll") for (int g = 0; g < WM; g++)                                                                  \
  _Pragma("unroll") for (int h = 0; h < WN; h++)                                                                  \
      acc[g][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(ob[j][c][h], oa[j][c][g], acc[g][h], 0, 0, 0);
  // software pipeline of aomarl_gemm_p.h: tile kt in LDS buffer kt & 1, tile kt + 1 in register stage (kt + 1) & 1,
  // tile kt + 2 in flight into stage kt & 1; one barrier per k-tile between its two halves.
  const std::integral_constant<bool, false> full_c;
  const std::integral_constant<bool, true> tail_c;
  auto step = [&](int kt, int S, auto TAILc) {
    fread(S, 1);
    __builtin_amdgcn_sched_barrier(0);
    GG_MMA(0, 0) GG_MMA(0, 1)
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nkt) lstore(S ^ 1, S ^ 1, kt + 1, TAILc);   // wave-uniform
    gload(kt + 3, S ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    GG_MMA(0, 2) GG_MMA(0, 3)
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    fread(S ^ 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    GG_MMA(1, 0) GG_MMA(1, 1) GG_MMA(1, 2) GG_MMA(1, 3)
    __builtin_amdgcn_sched_barrier(0);
  };
  const bool ktail = (K & (GG_KT - 1)) != 0;        // the last k-tile is a partial one
  gload(0, 0);
  gload(1, 1);
  __builtin_amdgcn_sched_barrier(0);
  if (ktail && nkt == 1) lstore(0, 0, 0, tail_c); else lstore(0, 0, 0, full_c);
  gload(2, 0);
  __syncthreads();
  fread(0, 0);
  // (the step that STAGES tile nkt - 1 is step nkt - 2)
  for (int kt = 0; kt < nkt; kt += 2) {
    if (ktail && kt + 2 == nkt) step(kt, 0, tail_c); else step(kt, 0, full_c);
    if (kt + 1 >= nkt) break;
    if (ktail && kt + 3 == nkt) step(kt + 1, 1, tail_c); else step(kt + 1, 1, full_c);
  }
#undef GG_MMA

  // ---- bias gradient: the staged B pieces' k-sums, reduced over the threads that share a column piece
  if (want_cs) {                                   // block-uniform; every wave is past the loop's last barrier
    float *red = gg_lds;                           // [256 / PRB][BN]
    *reinterpret_cast<float4 *>(red + krb * BN + 4 * pcb) = cs;
    __syncthreads();
    if (tid < BN && n0 + tid < N) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 256 / PRB; r++) s += red[r * BN + tid];
      a.colsum[(long long)grp * a.sCs + n0 + tid] = s;
    }
  }

  // ---- epilogue.  Lane (q, i) of accumulator (g, h) holds component r: the B-side index t = 4 q + r and the A-side
  // index i of the instruction, i.e. column n = 16 h + t (BK) or WN t + h (!BK), row m = 16 g + i (AK) or WM i + g.
  float *C = a.C + (long long)grp * a.sC;
  const float *bias = a.bias ? a.bias + (long long)grp * a.sBias : nullptr;
  const float *mask = a.mask ? a.mask + (long long)grp * a.sM : nullptr;
  const int mw = m0 + wm * (16 * WM), nw = n0 + wn * (16 * WN);
  auto emit4 = [&](int m, int n, float v0, float v1, float v2, float v3) {
    if (m >= M || n >= N) return;
    float v[4] = {v0, v1, v2, v3};
    float *c = C + (long long)m * a.ldc + n;
    if (n + 3 < N) {
      if (bias) { const gp_f4u b = *reinterpret_cast<const gp_f4u *>(bias + n); v[0] += b.v[0]; v[1] += b.v[1]; v[2] += b.v[2]; v[3] += b.v[3]; }
      if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      if (mask) {
        const gp_f4u k = *reinterpret_cast<const gp_f4u *>(mask + (long long)m * a.ldm + n);
        v[0] = k.v[0] > 0.f ? v[0] : 0.f; v[1] = k.v[1] > 0.f ? v[1] : 0.f; v[2] = k.v[2] > 0.f ? v[2] : 0.f; v[3] = k.v[3] > 0.f ? v[3] : 0.f;
      }
      gp_f4u o; o.v[0] = v[0]; o.v[1] = v[1]; o.v[2] = v[2]; o.v[3] = v[3];
      *reinterpret_cast<gp_f4u *>(c) = o;
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) if (n + r < N) {
        float o = v[r];
        if (bias) o += bias[n + r];
        if (a.relu) o = fmaxf(o, 0.f);
        if (mask && !(mask[(long long)m * a.ldm + n + r] > 0.f)) o = 0.f;
        c[r] = o;
      }
    }
  };
#pragma unroll
  for (int g = 0; g < WM; g++) {
    const int m = AK ? mw + 16 * g + li : mw + WM * li + g;
    if (BK) {
#pragma unroll
      for (int h = 0; h < WN; h++) emit4(m, nw + 16 * h + 4 * q, acc[g][h][0], acc[g][h][1], acc[g][h][2], acc[g][h][3]);
    } else if (WN == 4) {
#pragma unroll
      for (int r = 0; r < 4; r++) emit4(m, nw + 16 * q + 4 * r, acc[g][0][r], acc[g][1][r], acc[g][2][r], acc[g][3][r]);
    } else {
      emit4(m, nw + 8 * q, acc[g][0][0], acc[g][1][0], acc[g][0][1], acc[g][1][1]);
      emit4(m, nw + 8 * q + 4, acc[g][0][2], acc[g][1][2], acc[g][0][3], acc[g][1][3]);
    }
  }
}


// hardware block b lands on XCD b % 8: XCD x takes a contiguous range of the (group-major) tile list, so the tiles of
// one group -- which share A rows and B columns -- meet in one L2
__device__ __forceinline__ int gemm_g_tile_of_block() {
  const int per = gridDim.x >> 3;
  return (blockIdx.x & 7) * per + (blockIdx.x >> 3);
}

template <int WM, int WN, bool AK, bool BK>
__global__ __launch_bounds__(256, 2) void k_gemm_g(const GemmGArgs a) {
  const int w = gemm_g_tile_of_block();
  if (w >= a.ntile) return;
  if (!BK && a.colsum) gemm_g_body<WM, WN, AK, BK, true>(a, w);      // block-uniform
  else gemm_g_body<WM, WN, AK, BK>(a, w);
}

// Up to GG_MAXP independent products in ONE launch (64 x 64 tiles), each with its own shapes, operand forms and
// epilogue: the update's products that do not depend on each other (a layer's input gradient and its weight
// gradient; the policy's first layer and the critics' hidden layer on the gathered batch) share a grid instead of
// meeting through events on two streams (an event record or wait on the critical stream cost 6 - 7.5 us each,
// profiles/r05a_sac_update_timeline.txt).  A workgroup belongs to one problem: the form switch is block-uniform.
#define GG_MAXP 3
struct GemmGMulti {
  int np;
  int form[GG_MAXP];                             // 2 AK + BK
  int tile_end[GG_MAXP];                         // prefix sums of the problems' tile counts
  GemmGArgs p[GG_MAXP];
};

__global__ __launch_bounds__(256, 2) void k_gemm_g_multi(const GemmGMulti mp) {
  const int w = gemm_g_tile_of_block();
  if (w >= mp.tile_end[mp.np - 1]) return;
  int i = 0;
  while (i < mp.np - 1 && w >= mp.tile_end[i]) i++;
  const int wl = w - (i ? mp.tile_end[i - 1] : 0);
  switch (mp.form[i]) {
    case 0:
      if (mp.p[i].colsum) gemm_g_body<2, 2, false, false, true>(mp.p[i], wl);
      else gemm_g_body<2, 2, false, false>(mp.p[i], wl);
      break;
    case 1: gemm_g_body<2, 2, false, true>(mp.p[i], wl); break;
    case 2: gemm_g_body<2, 2, true, false>(mp.p[i], wl); break;
    default: gemm_g_body<2, 2, true, true>(mp.p[i], wl); break;
  }
}

// ---- host side ---------------------------------------------------------------------------------------------
static inline size_t gemm_g_lds_bytes(int wm, int wn, bool ak, bool bk) {
  const size_t sa = ak ? (size_t)32 * wm * GG_KT : (size_t)GG_KT * (32 * wm + 4);
  const size_t sb = bk ? (size_t)32 * wn * GG_KT : (size_t)GG_KT * (32 * wn + 4);
  return 2 * (sa + sb) * sizeof(float);
}

typedef void (*gemm_g_kernel_t)(const GemmGArgs);
template <bool AK, bool BK>
static inline gemm_g_kernel_t gemm_g_kernel_f(int wm, int wn) {
  if (wm == 2 && wn == 2) return k_gemm_g<2, 2, AK, BK>;
  if (wm == 4 && wn == 2) return k_gemm_g<4, 2, AK, BK>;
  if (wm == 2 && wn == 4) return k_gemm_g<2, 4, AK, BK>;
  if (wm == 4 && wn == 4) return k_gemm_g<4, 4, AK, BK>;
  return nullptr;
}
static inline gemm_g_kernel_t gemm_g_kernel(int wm, int wn, bool ak, bool bk) {
  return ak ? (bk ? gemm_g_kernel_f<true, true>(wm, wn) : gemm_g_kernel_f<true, false>(wm, wn))
            : (bk ? gemm_g_kernel_f<false, true>(wm, wn) : gemm_g_kernel_f<false, false>(wm, wn));
}

// Block tile.  Measured on the update's shapes (tools/gemmgbench.hip, profiles/r05_gemmgbench.txt): 64 x 64 wins on
// every one of them (31 us against 33 - 36 for 128 x 64 / 64 x 128 on the largest, 10 against 15 on the smallest) --
// these products have 8 - 20 k-tiles and 224 - 1232 tiles of 64 x 64, and two co-resident workgroups per CU fill
// each other's barrier and prologue stalls where one bigger tile per CU has nothing beside it.  The larger tiles
// stay instantiated for shapes that would otherwise put more than 8 workgroups of 64 x 64 on a CU.
static inline void gemm_g_pick(int groups, int M, int N, int K, int *wm, int *wn) {
  (void)K;
  *wm = 2; *wn = 2;
  const long long G = (long long)groups * ((M + 63) / 64) * ((N + 63) / 64);
  if (G > 8 * 256) { *wm = M >= N ? 4 : 2; *wn = M >= N ? 2 : 4; }
  if (G > 16 * 256) { *wm = 4; *wn = 4; }
}

// 0 = launched.  `force_wm / force_wn` (0 = pick) for the bench.
static inline int gemm_g_launch(int groups, bool ak, bool bk, GemmGArgs a, int force_wm, int force_wn, hipStream_t s) {
  if (groups <= 0 || a.M <= 0 || a.N <= 0 || a.K <= 0) return 0;
  int wm = 2, wn = 2;
  gemm_g_pick(groups, a.M, a.N, a.K, &wm, &wn);
  if (force_wm) wm = force_wm;
  if (force_wn) wn = force_wn;
  gemm_g_kernel_t f = gemm_g_kernel(wm, wn, ak, bk);
  if (!f) return 1;
  const size_t lds = gemm_g_lds_bytes(wm, wn, ak, bk);
  static bool done[2][2][8][8] = {};
  if (!done[ak][bk][wm][wn]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return 1;
    done[ak][bk][wm][wn] = true;
  }
  a.tiles_m = (a.M + 32 * wm - 1) / (32 * wm);
  a.tiles_n = (a.N + 32 * wn - 1) / (32 * wn);
  a.ntile = groups * a.tiles_m * a.tiles_n;
  hipLaunchKernelGGL(f, dim3((unsigned)((a.ntile + 7) / 8 * 8)), dim3(256), lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Several products in one launch.  `groups` tiles lists are concatenated problem after problem.
static inline int gemm_g_launch_multi(int groups, int np, const GemmGArgs *probs, const bool *ak, const bool *bk,
                                      hipStream_t s) {
  if (np < 1 || np > GG_MAXP) return 1;
  GemmGMulti mp;
  memset(&mp, 0, sizeof(mp));
  int total = 0, n = 0;
  for (int i = 0; i < np; i++) {
    GemmGArgs a = probs[i];
    if (groups <= 0 || a.M <= 0 || a.N <= 0 || a.K <= 0) continue;
    a.tiles_m = (a.M + 63) / 64;
    a.tiles_n = (a.N + 63) / 64;
    a.ntile = groups * a.tiles_m * a.tiles_n;
    total += a.ntile;
    mp.p[n] = a; mp.form[n] = (ak[i] ? 2 : 0) + (bk[i] ? 1 : 0); mp.tile_end[n] = total;
    n++;
  }
  if (!n) return 0;
  mp.np = n;
  const size_t lds = gemm_g_lds_bytes(2, 2, false, false);       // the largest of the four forms
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_g_multi), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return 1;
    done = true;
  }
  hipLaunchKernelGGL(k_gemm_g_multi, dim3((unsigned)((total + 7) / 8 * 8)), dim3(256), lds, s, mp);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
