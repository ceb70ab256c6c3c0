// aomarl_dev.h -- device-side description shared by the kernels of libaomarl_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/aomarl.h"

// Wave priority of the control / agent chain's kernels against the frame kernel they share the SIMDs with: the issue
// arbiter serves the oldest ready wave first, and the frame kernel's long-lived waves are always ready -- a chain
// kernel's waves starve beside them (a 19 us product takes 160).  CHAIN_PRIO > 0 raises the chain's waves.
#ifndef CHAIN_PRIO
#define CHAIN_PRIO 0
#endif
#define CHAIN_SETPRIO() do { if (CHAIN_PRIO) __builtin_amdgcn_s_setprio(CHAIN_PRIO); } while (0)
#ifndef ATM_PRIO
#define ATM_PRIO 0
#endif
#define ATM_SETPRIO() do { if (ATM_PRIO) __builtin_amdgcn_s_setprio(ATM_PRIO); } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// split-fp16 operand pairs (the opt-in fast mode): 2 / 8 halfs, and the matrix instruction they feed
typedef _Float16 hx2 __attribute__((ext_vector_type(2)));
typedef _Float16 hx8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_h(hx8 a, hx8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

struct DevLayer {
  int dim, ns;
  long long screen_off;          // floats from the env's screen base
  const uint32_t *istx, *isty;   // packed (x | y << 16) logical stencil coordinates
  const uint32_t *istT;          // istx with x and y exchanged: the x stencil of the TRANSPOSED screen (reset)
  const float *AB;               // [dim][ldab]: row r = [A[r][0..ns) | B[r][0..dim)]
  int ldab;
  const float *ABt;              // [ns + dim][ldt], ldt = dim rounded up to 64 (zero padded): the transpose, small screens only
  int ldt;
  float amp;
  float wxo, wyo, txo, tyo;      // float offsets
  int wox, woy, tox, toy;        // integer parts
};

struct DevDm {
  int type, dim, nact, ss;
  long long shape_off;           // floats from the env's dm_shape base
  int com_off;                   // first command index
  const float *influ;
  const int32_t *influpos, *ninflu, *influstart;
  float wxo, wyo, txo, tyo;
  int wox, woy, tox, toy;
  // stack-array fast path: every actuator shares one separable patch influ[a][b] = prof[a] prof[b]
  // and sits on a regular lattice (i1min + pitch*gx, j1min + pitch*gy); grid[gy*gw+gx] = actuator
  // index or -1.  0 = use the generic gather tables.
  int sep, pitch, i1min, j1min, gw, gh;
  const int32_t *grid;
  const float *prof;
};

struct DevSys {
  int n, pupdiam;
  const float *mpupil, *spupil;
  int nvalid, pdiam, nfft, npix, nrebin, nxsub;
  const int32_t *phasemap;
  const int32_t *sub_xy;         // [nvalid] packed (x0 | y0 << 16): top-left phase pixel
  const float *halfxy;
  const int32_t *binmap;
  const float *flux;
  const int32_t *validx, *validy;
  float nphot, wfs_inv_lambda, noise, cog_offset, cog_scale, subapd;
  int nlayers;
  DevLayer layers[AOMARL_MAX_LAYERS];
  int ndm;
  DevDm dms[AOMARL_MAX_DMS];
  float tar_inv_lambda;
  int npsf, hw;
  const float *psf_tw;           // [npsf][2] cos, sin of 2 pi j / npsf
  float ref_peak;
  int nactu, nslope;
  int wfs_all_int, tar_all_int;  // every offset of that path is an integer
  long long screen_stride, shape_stride;
  // ---- fused frame kernel (WFS + science path share every phase pixel: the 16x16 sub-aperture
  // tiles ARE the tiles of the pupil grid): available when the geometry lines up (see create)
  int fused_ok, ntiles;          // ntiles = pupdiam / 16 tiles per axis
  const int32_t *stripe_order;   // [ntiles] stripes by decreasing number of lit tiles
  const int32_t *tile_info;      // [ntiles][ntiles] (stripe, tile): sub-aperture | lit / full / has-sub bits
  const int32_t *lit_info;       // [ntiles][ntiles + 8]: the stripe's LIT tiles in x order, compact: tile_info | tile << 24;
                                 //   entries past the last one repeat it
  const int32_t *lit_count;      // [ntiles] lit tiles per stripe
  // the same walk in PAIRS of adjacent tiles (2 g, 2 g + 1) with at least one lit tile: [ntiles][(ntiles + 1) / 2 + 2][2]
  // entries (tile A, tile B) of the lit_info form (bit 16 clear: that tile of the pair is not lit; the tile index is
  // valid either way); entries past the last pair repeat it
  const int32_t *pair_info;
  const int32_t *pair_count;     // [ntiles] pairs per stripe
  const uint16_t *tile_mask;     // [pupdiam][ntiles]: bit b = spupil[y][16 t + b] != 0
  // stack-array DM phase from the command lattice inside the frame kernel (separable lattice whose
  // pitch divides the tile size): nodes per axis that reach a tile <= 4 otf_nb
  int otf_ok, otf_nb, otf_tpn;   // otf_tpn = lattice nodes per tile step (16 / pitch)
  int otf_gx0, otf_gy0;          // first node column / row that reaches tile 0 / stripe 0
  int otf_xoff, otf_yoff;        // DM pixel of the tile origin minus the position of that node
  int otf_latw;                  // lattice columns a stripe can touch
  // PSF operand of the frame kernel for lane (q, c) of tile t, 16 B each: column c of [cos k X (k = 1..8) |
  // sin k X (k = 1..8)], X = 16 t + 4 q + j, j = 0..3
  const void *psf_tw_h;          // [ntiles][64] x 8 halfs: [hi(j = 0..3) | lo(j = 0..3)]  (split-fp16 form)
  const void *psf_tw_f;          // [ntiles][64] x 4 floats                                  (fp32 form)
  const float *qf_tab;           // [64 lanes][8]: SpotQf of each lane (spot_qf_consts, computed once at create by k_fill_qf_tab)
  // tip-tilt planes in pupil coordinates, 4 pixels of a row at a time: [pupdiam][pupdiam / 4] x [x0 x1 x2 x3 | y0 y1 y2 y3]
  const float *tt_pk;
};

struct DevState {
  int nenv, ld_actu;
  float *screens;
  int32_t *origin;
  uint32_t *seeds, *ext_count;
  float *com, *com1, *com2, *err, *voltage, *slopes, *dm_shape, *bincube, *wfs_phase, *tar_phase;
  float *strehl, *le_img;
  uint32_t *frame;
  float *work;
  int32_t *origin_snap;   // frame pipeline: the kernels that advance a ring origin also write it here (or null)
};

// Sum of split-K slabs in their fixed order with B loads in flight at a time.  Written as a plain loop over a run-time
// count, the compiler emits load / wait / add per slab: every slab costs a full load latency (4 us of the 18 us
// scatter + gather launch inside the reset, one to two microseconds per slab in the chains' consumers beside the
// frame kernel).  at(z): the z-th slab's element (slabs past the count re-read the last one and are not added).
template <int B, typename F>
__device__ __forceinline__ float slab_sum(int nsplit, F at) {
  float acc = 0.f;
  for (int z0 = 0; z0 < nsplit; z0 += B) {
    float pz[B];
#pragma unroll
    for (int z = 0; z < B; z++) pz[z] = at(min(z0 + z, nsplit - 1));
#pragma unroll
    for (int z = 0; z < B; z++) acc = (z0 + z < nsplit) ? acc + pz[z] : acc;
  }
  return acc;
}

// ---------------------------------------------------------------- Philox4x32-10 + normals
// (same definition as the oracle: key = {seed, "AOMR"}, ctr = {block, counter_lo, counter_hi,
//  stream}; 4 outputs -> 4 uniforms or 2 Box-Muller pairs)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
  // both halves of a 32 x 32 product from ONE instruction (v_mad_u64_u32: 6.5 cycles of the SIMD against 5.2 + 5.5 for
  // v_mul_hi_u32 + v_mul_lo_u32 with three waves resident, tools/valubench.hip); the multiplier rides in a scalar
  // register (VOP3 takes no 32-bit literal on this target)
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint64_t m0, m1;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(m0) : "s"(M0), "v"(c0) : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(m1) : "s"(M1), "v"(c2) : "vcc");
    const uint32_t hi0 = (uint32_t)(m0 >> 32), lo0 = (uint32_t)m0, hi1 = (uint32_t)(m1 >> 32), lo1 = (uint32_t)m1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t x) {
  return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// elements 4 j .. 4 j + 3 of the normal stream (seed, stream, counter): the four of them come out of
// one Philox block and two Box-Muller pairs (same values as philox_normal element by element)
__device__ __forceinline__ void philox_normal4(uint32_t seed, uint32_t stream, uint32_t cnt_lo,
                                               uint32_t cnt_hi, uint32_t j, float out[4]) {
  uint32_t x[4];
  philox4x32_10(j, cnt_lo, cnt_hi, stream, seed, 0x414F4D52u, x);
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const float u0 = u01(x[2 * h]), u1 = u01(x[2 * h + 1]);
    const float r = sqrtf(-2.0f * logf(u0));
    const float a = 6.28318530717958647692f * u1;
    out[2 * h] = r * cosf(a);
    out[2 * h + 1] = r * sinf(a);
  }
}

// element `idx` of the normal stream (seed, stream, counter)
__device__ __forceinline__ float philox_normal(uint32_t seed, uint32_t stream, uint32_t cnt_lo,
                                               uint32_t cnt_hi, uint32_t idx) {
  uint32_t x[4];
  philox4x32_10(idx >> 2, cnt_lo, cnt_hi, stream, seed, 0x414F4D52u, x);
  int h = (idx >> 1) & 1;
  float u0 = u01(x[2 * h]), u1 = u01(x[2 * h + 1]);
  float r = sqrtf(-2.0f * logf(u0));
  float a = 6.28318530717958647692f * u1;
  return (idx & 1) ? r * sinf(a) : r * cosf(a);
}

__device__ __forceinline__ float philox_uniform(uint32_t seed, uint32_t stream, uint32_t cnt_lo,
                                                uint32_t cnt_hi, uint32_t idx) {
  uint32_t x[4];
  philox4x32_10(idx >> 2, cnt_lo, cnt_hi, stream, seed, 0x414F4D52u, x);
  return u01(x[idx & 3]);
}

// ring-buffered screen: logical (x, y) -> physical float index.  Physical rows are
// n + RING_PAD floats long: columns [n, n + RING_PAD) mirror columns [0, RING_PAD), so up to
// RING_PAD consecutive logical pixels starting at a physical column < n are consecutive floats:
// a 16-pixel tile row whose first pixel is in range needs no wrap test per lane (the one-pass
// frame kernel wraps once per tile -- once per PAIR of tiles, 32 pixels, when it fetches whole 128-byte row
// pieces -- on the scalar unit); the extrusion scatter keeps the mirror up to date.
// (56, not 32: with the screens of the production files -- 648 and 168 pixels -- the row pitch is then a multiple of
// 128 bytes, every row of a tile starts at the same offset inside its line; see k_reset_env)
#ifndef RING_PAD
#define RING_PAD 56
#endif
static_assert(RING_PAD >= 32 && RING_PAD % 4 == 0, "the frame kernel's pair fetch reads 32 consecutive pixels in 16-byte pieces");
#ifndef FW_ALIGN_ORIGIN
#define FW_ALIGN_ORIGIN 1  // 0: a reset starts the rings at origin (0, 0)
#endif
__device__ __forceinline__ int ring_idx(int x, int y, int ox, int oy, int n) {
  int px = x + ox;
  px -= (px >= n) ? n : 0;
  int py = y + oy;
  py -= (py >= n) ? n : 0;
  return py * (n + RING_PAD) + px;
}

// ---- cross-lane helpers on the VALU (DPP) instead of the LDS crossbar (ds_bpermute)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// v + v[lane ^ 1]
__device__ __forceinline__ float add_xor1(float v) { return v + dpp_f<0xB1>(v); }   // quad_perm [1,0,3,2]
// sum over the 64 lanes, result uniform (returned in every lane)
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_f<0x124>(v);     // row_ror:4
  v += dpp_f<0x128>(v);     // row_ror:8   -> every lane holds its 16-lane row sum
  return (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) +
          __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) +
          __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48)));
}

// sum over the 64 lanes, valid in LANE 63 ONLY: 6 DPP adds, no v_readlane (row_bcast:15 / :31 carry
// the row sums across the four 16-lane rows)
__device__ __forceinline__ float wave_sum_last(float v) {
  v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);     // row_half_mirror
  v += dpp_f<0x140>(v);     // row_mirror -> every lane holds its 16-lane row sum
  // rows 1, 3 += last lane of the row before; rows 2, 3 += last lane of row 1: ONE instruction each
  // (dst == src, the rows the mask leaves out keep their value) -- written through update_dpp with a
  // zero `old` the compiler needs v_mov 0 + v_mov_dpp + v_add per step
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
  return v;
}
