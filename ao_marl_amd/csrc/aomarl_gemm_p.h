// aomarl_gemm_p.h -- the fp32 GEMM of the loop's products (round 4), gfx950 only.
//
//   C[M][N] = alpha * A[M][K] . B[N][K]^T (+ beta C)        both operands K-contiguous, fp32 in / fp32 accumulate
//
// The products the AO loop makes every frame have M = environments (256 .. 768 rows) against a static matrix:
// the Fried-Clark extrusion Z.[A|B]^T (shesha/util/iterkolmo.py:255-288), err = -cmat.s
// (shesha/supervisor/components/rtcCompass.py:527-547), and the two Btt projections v2m / m2v
// (src/reinforcement_learning/.../rlSupervisor.py:784-818).  Each is 0.8 - 2 GFLOP: a handful of output tiles for
// 256 CUs, so K is split and the consumers sum the partial slabs P[z][M][N] (k_gemm_reduce_epi, the extrusion
// scatter, the delay line, the state assembly).  What this kernel changes against k_gemm_nt2 (64x64 tiles, one
// 32x32x2 accumulator per wave, a barrier every 16 matrix instructions, 84 - 396 workgroups):
//   * the grid is CHOSEN so that every CU gets the same number of workgroups (gemm_p_pick: wave tile, block tile
//     and k-chunk from a menu, cost = the busiest CU's matrix time + the partial slabs' traffic);
//   * a wave owns (16 WM) x (16 WN) outputs = WM x WN independent v_mfma_f32_16x16x4_f32 accumulators (12 - 16
//     chains: the instruction's 40-cycle dependent latency never shows), fed by WM + WN ds_read_b128 per 4 WM WN
//     matrix instructions; one block barrier per 8 WM WN of them (96 - 128);
//   * operands go global -> registers (two k-tiles in flight) -> LDS (two buffers; rows unpadded, their 16-byte
//     pieces XOR-swizzled by row: the 16 lanes of a 128-bit pass hit 16 different bank groups, writing and reading);
//   * the matrix instruction is issued with the operands exchanged (a = B fragment, b = A fragment), so a lane
//     holds four CONSECUTIVE columns of one output row: 16-byte stores;
//   * workgroups are numbered so that the ones sharing a k-chunk sit on one XCD (its L2 sees that chunk of both
//     operands once).
// Tails: rows / columns past M / N are clamped on load and not stored; a k-tile that crosses the end of the
// block's chunk is zeroed as it is staged (wave-uniform branch).  Operands must be 16-byte aligned with leading
// dimensions that are multiples of 4 (the caller falls back to k_gemm_nt otherwise).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#ifndef GP_TYPES
#define GP_TYPES
typedef float gp_f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) gp_f4u { float v[4]; };
#endif

#ifndef GP_SWZ
#define GP_SWZ 1
#endif
#if GP_SWZ
// LDS rows unpadded (32 floats = 8 pieces of 16 bytes), piece p of row r stored at piece p ^ ((r >> 1) & 7): the 16 lanes
// of a 128-bit access pass -- 2 rows x 8 pieces when staging, 16 rows x 1 piece when reading operands -- then always
// touch 16 different bank groups (a padded pitch of 36 made the staging writes of rows r, r + 1 collide: 34 % of the
// LDS cycles were conflicts, profiles/r04_pmc_gemm_p.txt)
#define GP_LD 32
#else
#define GP_LD 36          // LDS row pitch (floats): 32 k + 4; pitch / 4 odd -> conflict-free 128-bit operand reads
#endif
#define GP_KT 32          // k-tile

#ifndef GP_DBG
#define GP_DBG 0
#endif
#ifndef GP_GLDS
#define GP_GLDS 0         // 1: k-tiles go global -> LDS directly (buffer_load_dwordx4 ... lds), three LDS buffers
#endif
#define GP_NB (GP_GLDS ? 3 : 2)                   // LDS buffers per operand
template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void k_gemm_p(int M, int N, int K, float alpha,
                                                   const float *__restrict__ A, int lda,
                                                   const float *__restrict__ B, int ldb, float beta,
                                                   float *__restrict__ C, int ldc, int kchunk, int nz,
                                                   float *__restrict__ P, int tiles_n, int tiles, int xcd) {
  constexpr int BM = 32 * WM, BN = 32 * WN;
#ifdef GP_PRIO
  if (GP_PRIO) __builtin_amdgcn_s_setprio(GP_PRIO);
#endif
  extern __shared__ __attribute__((aligned(16))) float gp_lds[];
  float *As = gp_lds;                              // [GP_NB][BM][GP_LD]
  float *Bs = gp_lds + GP_NB * BM * GP_LD;         // [GP_NB][BN][GP_LD]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wv >> 1, wn = wv & 1;

  // workgroup -> (k-chunk z, tile t): hardware block b lands on XCD b % 8; XCD x takes the contiguous range
  // [off_x, off_x + cnt_x) of the (z-major) work list
  int w = blockIdx.x;
  if (xcd) {
    const int G = gridDim.x, x = w & 7, q = G >> 3, r = G & 7;
    w = x * q + min(x, r) + (w >> 3);
  }
  const int z = w / tiles, t = w - z * tiles;
  const int tm = t / tiles_n, tn = t - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kb = z * kchunk, ke = min(K, kb + kchunk);
  const int nkt = (ke - kb + GP_KT - 1) / GP_KT;

  // staging: thread -> (row lr + 32 p, 4 floats at lc) of each operand's k-tile
  const int lr = tid >> 3, lc = (tid & 7) * 4;
  const float *pa[WM], *pb[WN];
#pragma unroll
  for (int p = 0; p < WM; p++) pa[p] = A + (long long)min(m0 + lr + 32 * p, M - 1) * lda;
#pragma unroll
  for (int p = 0; p < WN; p++) pb[p] = B + (long long)min(n0 + lr + 32 * p, N - 1) * ldb;
  const int klast = (ke - 1) & ~3;                 // last 16-byte group with a valid element
  float4 ra[2][WM], rb[2][WN];
  auto gload = [&](int kt, int st) {
    const int k = min(kb + kt * GP_KT + lc, klast);
#pragma unroll
    for (int p = 0; p < WM; p++) ra[st][p] = *reinterpret_cast<const float4 *>(pa[p] + k);
#pragma unroll
    for (int p = 0; p < WN; p++) rb[st][p] = *reinterpret_cast<const float4 *>(pb[p] + k);
  };
  auto lstore = [&](int buf, int st, int kt) {
#if GP_SWZ
    const int lcs = (((tid & 7) ^ ((lr >> 1) & 7)) << 2);
#else
    const int lcs = lc;
#endif
    float *as = As + buf * BM * GP_LD + lr * GP_LD + lcs;
    float *bs = Bs + buf * BN * GP_LD + lr * GP_LD + lcs;
    const int k = kb + kt * GP_KT + lc;
    if (kb + (kt + 1) * GP_KT > ke) {              // wave-uniform: the tile crosses the end of the chunk
      const bool v0 = k < ke, v1 = k + 1 < ke, v2 = k + 2 < ke, v3 = k + 3 < ke;
      auto msk = [&](float4 v) { v.x = v0 ? v.x : 0.f; v.y = v1 ? v.y : 0.f; v.z = v2 ? v.z : 0.f; v.w = v3 ? v.w : 0.f; return v; };
#pragma unroll
      for (int p = 0; p < WM; p++) *reinterpret_cast<float4 *>(as + 32 * p * GP_LD) = msk(ra[st][p]);
#pragma unroll
      for (int p = 0; p < WN; p++) *reinterpret_cast<float4 *>(bs + 32 * p * GP_LD) = msk(rb[st][p]);
    } else {
#pragma unroll
      for (int p = 0; p < WM; p++) *reinterpret_cast<float4 *>(as + 32 * p * GP_LD) = ra[st][p];
#pragma unroll
      for (int p = 0; p < WN; p++) *reinterpret_cast<float4 *>(bs + 32 * p * GP_LD) = rb[st][p];
    }
  };

  gp_f32x4 acc[WM][WN];
#pragma unroll
  for (int g = 0; g < WM; g++)
#pragma unroll
    for (int h = 0; h < WN; h++) acc[g][h] = (gp_f32x4){0.f, 0.f, 0.f, 0.f};

  // operand fragments: lane (q = lane >> 4, i = lane & 15) reads 4 floats of row i at k = 8 q + 4 j; component c of
  // them is the K lane q of matrix instruction (j, c) -- the same k permutation on both operands.  Two fragment
  // sets (j = 0, 1) so that every LDS read is issued a half-tile of matrix instructions before its first use.
#if GP_SWZ
  const int fo0 = (lane & 15) * GP_LD + (((2 * (lane >> 4)) ^ (((lane & 15) >> 1) & 7)) << 2);
  const int fo1 = (lane & 15) * GP_LD + (((2 * (lane >> 4) + 1) ^ (((lane & 15) >> 1) & 7)) << 2);
#else
  const int fo0 = (lane & 15) * GP_LD + 8 * (lane >> 4), fo1 = fo0 + 4;
#endif
  float4 fa[2][WM], fb[2][WN];
  auto fread = [&](int buf, int j) {
    const float *as = As + buf * BM * GP_LD + wm * (16 * WM) * GP_LD + (j ? fo1 : fo0);
    const float *bs = Bs + buf * BN * GP_LD + wn * (16 * WN) * GP_LD + (j ? fo1 : fo0);
#pragma unroll
    for (int g = 0; g < WM; g++) fa[j][g] = GP_DBG >= 4 ? ra[0][g] : *reinterpret_cast<const float4 *>(as + g * 16 * GP_LD);
#pragma unroll
    for (int h = 0; h < WN; h++) fb[j][h] = GP_DBG >= 4 ? rb[0][h] : *reinterpret_cast<const float4 *>(bs + h * 16 * GP_LD);
  };
#if GP_GLDS
  // ---- operands straight into LDS.  One wave-instruction writes 64 x 16 bytes in LANE order = 8 rows of a k-tile
  // (128 bytes each); the XOR swizzle of the 16-byte pieces goes on the SOURCE address (lane = 8 row + slot loads
  // piece slot ^ ((row >> 1) & 7)).  Wave w stages row groups w, w + 4, ... of both operands: WM + WN instructions
  // per k-tile.  Three LDS buffers, tiles kt + 1 and kt + 2 in flight; a wave waits for ITS pieces of tile kt + 1
  // (counted vmcnt) in front of the k-tile's barrier.  A k-tile that crosses the end of the chunk (only ever the
  // last one) is staged through registers with the mask, as before.
  {
    const int rl = lane >> 3, slot = lane & 7;
    __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, 0x7fffffff, 0x00020000);
    unsigned voa[WM], vob[WN];
#pragma unroll
    for (int p = 0; p < WM; p++) {
      const int r = 8 * (4 * p + wv) + rl;
      voa[p] = 4u * ((unsigned)min(m0 + r, M - 1) * (unsigned)lda) + 16u * (unsigned)(slot ^ ((r >> 1) & 7));
    }
#pragma unroll
    for (int p = 0; p < WN; p++) {
      const int r = 8 * (4 * p + wv) + rl;
      vob[p] = 4u * ((unsigned)min(n0 + r, N - 1) * (unsigned)ldb) + 16u * (unsigned)(slot ^ ((r >> 1) & 7));
    }
    const unsigned as_lds = (unsigned)(unsigned long long)As, bs_lds = (unsigned)(unsigned long long)Bs;
    auto dma16 = [&](unsigned vo, __amdgpu_buffer_rsrc_t rs, unsigned lds_addr, unsigned so) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(vo), "s"(rs), "s"(lds_addr), "s"(so) : "memory");
    };
    // stage k-tile kt into buffer `buf`: by DMA when it is whole, through registers (synchronously) when it is not
    auto stage = [&](int kt, int buf) {
      if (kb + (kt + 1) * GP_KT <= ke) {           // wave-uniform
        const unsigned so = 4u * (unsigned)(kb + kt * GP_KT);
#pragma unroll
        for (int p = 0; p < WM; p++) dma16(voa[p], rsa, as_lds + 4u * (unsigned)(buf * BM * GP_LD + 8 * (4 * p + wv) * GP_LD), so);
#pragma unroll
        for (int p = 0; p < WN; p++) dma16(vob[p], rsb, bs_lds + 4u * (unsigned)(buf * BN * GP_LD + 8 * (4 * p + wv) * GP_LD), so);
      } else {
        gload(kt, 0);
        lstore(buf, 0, kt);
      }
    };
    constexpr int PW = WM + WN;                    // DMA instructions of one wave per k-tile
    auto wait_tiles_in_flight = [&](bool one) {    // all of this wave's stagings but (at most) the newest tile's have landed
      if (one) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
#define GP_MMA(j, c)                                                                                              \
  _Pragma("unroll") for (int g = 0; g < WM; g++)                                                                  \
  _Pragma("unroll") for (int h = 0; h < WN; h++)                                                                  \
      acc[g][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][h].c, fa[j][g].c, acc[g][h], 0, 0, 0);
    auto step = [&](int kt, auto Sc) {
      constexpr int S = decltype(Sc)::value, S1 = (S + 1) % 3, S2 = (S + 2) % 3;
      fread(S, 1);
      __builtin_amdgcn_sched_barrier(0);
      GP_MMA(0, x) GP_MMA(0, y)
      __builtin_amdgcn_sched_barrier(0);
      const bool more = kt + 2 < nkt;              // wave-uniform
      if (more) stage(kt + 2, S2);                 // buffer S2 held tile kt - 1: every wave is past the barrier behind its reads
      __builtin_amdgcn_sched_barrier(0);
      GP_MMA(0, z) GP_MMA(0, w)
      __builtin_amdgcn_sched_barrier(0);
      wait_tiles_in_flight(more);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      fread(S1, 0);
      __builtin_amdgcn_sched_barrier(0);
      GP_MMA(1, x) GP_MMA(1, y) GP_MMA(1, z) GP_MMA(1, w)
      __builtin_amdgcn_sched_barrier(0);
    };
    stage(0, 0);
    if (nkt > 1) stage(1, 1);
    wait_tiles_in_flight(nkt > 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fread(0, 0);
    const std::integral_constant<int, 0> c0;
    const std::integral_constant<int, 1> c1;
    const std::integral_constant<int, 2> c2;
    for (int kt = 0; kt < nkt; kt += 3) {
      step(kt, c0);
      if (kt + 1 >= nkt) break;
      step(kt + 1, c1);
      if (kt + 2 >= nkt) break;
      step(kt + 2, c2);
    }
#undef GP_MMA
  }
#else
#define GP_MMA(j, c)                                                                                              \
  _Pragma("unroll") for (int g = 0; g < WM; g++)                                                                  \
  _Pragma("unroll") for (int h = 0; h < WN; h++)                                                                  \
      acc[g][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][h].c, fa[j][g].c, acc[g][h], 0, 0, 0);
  // software pipeline: tile kt in LDS buffer kt & 1, tile kt + 1 in register stage (kt + 1) & 1, tile kt + 2 in
  // flight into stage kt & 1.  Loads past the end of the chunk are clamped (never staged).  ONE barrier per
  // k-tile, in the middle of its matrix instructions: before it a wave stages tile kt + 1 and reads the second
  // half of tile kt, behind it it reads the first half of tile kt + 1 under the second half's instructions.
  auto step = [&](int kt, int S) {
    fread(S, 1);
    __builtin_amdgcn_sched_barrier(0);
    GP_MMA(0, x) GP_MMA(0, y)
    __builtin_amdgcn_sched_barrier(0);
    if (GP_DBG < 2) if (kt + 1 < nkt) lstore(S ^ 1, S ^ 1, kt + 1);   // wave-uniform
    if (GP_DBG < 1) gload(kt + 3, S ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    GP_MMA(0, z) GP_MMA(0, w)
    __builtin_amdgcn_sched_barrier(0);
    if (GP_DBG < 3) __syncthreads();
    fread(S ^ 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    GP_MMA(1, x) GP_MMA(1, y) GP_MMA(1, z) GP_MMA(1, w)
    __builtin_amdgcn_sched_barrier(0);
  };
  gload(0, 0);
  gload(1, 1);
  __builtin_amdgcn_sched_barrier(0);
  lstore(0, 0, 0);
  gload(2, 0);
  __syncthreads();
  fread(0, 0);
  for (int kt = 0; kt < nkt; kt += 2) {
    step(kt, 0);
    if (kt + 1 >= nkt) break;
    step(kt + 1, 1);
  }
#undef GP_MMA
#endif

  // epilogue: lane (q, i) of accumulator (g, h) holds C[m][n .. n + 3], m = .. + i, n = .. + 4 q
  const bool split = nz > 1;
#pragma unroll
  for (int g = 0; g < WM; g++) {
    const int m = m0 + wm * (16 * WM) + 16 * g + (lane & 15);
#pragma unroll
    for (int h = 0; h < WN; h++) {
      const int n = n0 + wn * (16 * WN) + 16 * h + 4 * (lane >> 4);
      if (m < M && n < N) {
        const gp_f32x4 v = acc[g][h];
        if (split) {
          float *p = P + ((long long)z * M + m) * N + n;
          if (n + 3 < N) {
            gp_f4u o; o.v[0] = v[0]; o.v[1] = v[1]; o.v[2] = v[2]; o.v[3] = v[3];
            *reinterpret_cast<gp_f4u *>(p) = o;
          } else {
#pragma unroll
            for (int r = 0; r < 4; r++) if (n + r < N) p[r] = v[r];
          }
        } else {
          float *c = C + (long long)m * ldc + n;
#pragma unroll
          for (int r = 0; r < 4; r++) if (n + r < N) {
            float o = alpha * v[r];
            if (beta != 0.f) o += beta * c[r];
            c[r] = o;
          }
        }
      }
    }
  }
}

// ---- host side: configuration menu and launch -------------------------------------------------------------
struct GemmPCfg {
  int wm, wn;            // wave tile in 16-granules; block tile = (32 wm) x (32 wn)
  int nz, kchunk;        // k-chunks and their length (a multiple of GP_KT)
  int tiles_m, tiles_n;
};

static inline size_t gemm_p_lds_bytes(int wm, int wn) { return (size_t)GP_NB * 32 * (wm + wn) * GP_LD * sizeof(float); }

// the instantiations the library carries
#define GP_FOR_EACH_TILE(X) X(4, 4) X(4, 3) X(4, 2) X(2, 4) X(2, 3) X(2, 2) X(3, 3) X(3, 2)

typedef void (*gemm_p_kernel_t)(int, int, int, float, const float *, int, const float *, int, float, float *, int, int, int,
                                float *, int, int, int);
static inline gemm_p_kernel_t gemm_p_kernel(int wm, int wn) {
#define GP_SEL(a, b) if (wm == a && wn == b) return k_gemm_p<a, b>;
  GP_FOR_EACH_TILE(GP_SEL)
#undef GP_SEL
  return nullptr;
}

// Once per process and tile: allow more than 64 KB of dynamic LDS.
static inline bool gemm_p_prepare(int wm, int wn) {
  static bool done[8][8] = {{false}};
  if (wm < 0 || wm > 7 || wn < 0 || wn > 7) return false;
  if (done[wm][wn]) return true;
  gemm_p_kernel_t f = gemm_p_kernel(wm, wn);
  if (!f) return false;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(f), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)gemm_p_lds_bytes(wm, wn)) != hipSuccess)
    return false;
  done[wm][wn] = true;
  return true;
}

// kchunk for a split into (about) ns chunks: whole k-tiles, every chunk non-empty
static inline void gemm_p_chunks(int K, int ns, int *kchunk, int *nz) {
  const int kt = (K + GP_KT - 1) / GP_KT;
  const int per = (kt + ns - 1) / ns;
  *kchunk = per * GP_KT;
  *nz = (kt + per - 1) / per;
}

// Cost model (cycles of the busiest SIMD, roughly): workgroups go round-robin over ncu CUs, two of them share a
// CU's four SIMDs.  Per workgroup: k-tiles x (8 wm wn matrix instructions x 32 cycles + a barrier's skew) + fill
// + the tile's stores; plus what the consumer pays for reading nz slabs.
static inline double gemm_p_cost(int M, int N, int K, int wm, int wn, int ns, GemmPCfg *out) {
  const int ncu = 256;
  const int BM = 32 * wm, BN = 32 * wn;
  GemmPCfg c;
  c.wm = wm; c.wn = wn;
  c.tiles_m = (M + BM - 1) / BM; c.tiles_n = (N + BN - 1) / BN;
  gemm_p_chunks(K, ns, &c.kchunk, &c.nz);
  const long long G = (long long)c.tiles_m * c.tiles_n * c.nz;
  const long long per_cu = (G + ncu - 1) / ncu;
  const double ktile = 8.0 * wm * wn * 32.0 + 250.0;
  const double wg = (c.kchunk / GP_KT) * ktile + 2500.0 + 16.0 * wm * wn * 4.0;
  // slabs: written once, read once by the consumer (~4 B/clk/CU effective each way)
  const double slabs = c.nz > 1 ? 2.0 * c.nz * (double)M * N * 4.0 / (ncu * 8.0) : 0.0;
  if (out) *out = c;
  return per_cu * wg + slabs;
}

static inline GemmPCfg gemm_p_pick(int M, int N, int K, size_t ws_floats, int max_split) {
  GemmPCfg best = {0, 0, 0, 0, 0, 0};
  double bc = -1.0;
#define GP_TRY(a, b)                                                                      \
  for (int ns = 1; ns <= max_split; ns++) {                                               \
    GemmPCfg c;                                                                           \
    const double cost = gemm_p_cost(M, N, K, a, b, ns, &c);                               \
    if (c.nz > 1 && (size_t)c.nz * M * N > ws_floats) break;                              \
    if (c.nz < ns) continue;                                                              \
    if (bc < 0 || cost < bc) { bc = cost; best = c; }                                     \
  }
  GP_FOR_EACH_TILE(GP_TRY)
#undef GP_TRY
  return best;
}

// launch with a given configuration (P may be null when cfg.nz == 1)
static inline bool gemm_p_launch(const GemmPCfg &c, int M, int N, int K, float alpha, const float *A, int lda,
                                 const float *B, int ldb, float beta, float *C, int ldc, float *P, int xcd,
                                 hipStream_t s) {
  gemm_p_kernel_t f = gemm_p_kernel(c.wm, c.wn);
  if (!f || !gemm_p_prepare(c.wm, c.wn)) return false;
  const int tiles = c.tiles_m * c.tiles_n;
  hipLaunchKernelGGL(f, dim3((unsigned)(tiles * c.nz)), dim3(256), gemm_p_lds_bytes(c.wm, c.wn), s, M, N, K, alpha, A,
                     lda, B, ldb, beta, C, ldc, c.kchunk, c.nz, P, c.tiles_n, tiles, xcd);
  return true;
}
