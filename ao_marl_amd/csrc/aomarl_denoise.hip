// aomarl_denoise.hip -- the WFS-image denoiser as ONE fused gfx950 kernel.
//
// Network (reference: DenoisingAutoencoderCNN2DSingleSubapeture, src/autoencoder/
// autoencoder_models.py:130-197), per 16x16 spot image:
//   conv3x3(1->16)+ReLU+pool2 -> conv3x3(16->32)+ReLU+pool2 -> conv3x3(32->64)+ReLU ->
//   convT4x4s2(64->32)+ReLU -> convT4x4s2(32->16)+ReLU -> convT3x3s1(16->1)
// 1 712 128 MAC per image, 1200 images per environment per frame.
//
// One block (2 waves) = one image at a time, persistent over the images.  Every layer is a small
// GEMM on v_mfma_f32_16x16x4_f32 with M = 16 spatial positions, N = 16 output channels,
// K = (tap, 4 input channels):
//  * activations live in LDS, channel-last, zero-padded by one pixel (no border tests: every tap is
//    a compile-time immediate offset) with the channel stride padded by 4 floats (conflict-free
//    128-bit operand reads); two regions ping-pong between the layers;
//  * the K order inside a group of 16 channels is permuted so lane group q reads channels
//    4q..4q+3 with ONE ds_read_b128 and feeds four MFMAs; the weights are pre-arranged on the host
//    in exactly that order, one float4 per lane per (tap, channel group, channel tile), streamed
//    from L2 (they are shared by every image) and reused across the M tiles;
//  * pooling layers map the four pixels of a 2x2 window to the four accumulator registers of a
//    lane (m = 4 * window + r), so ReLU + max-pool is a max over registers;
//  * the transposed convolutions are split into their 4 output-parity classes, each a 2x2-tap
//    convolution of the input (no zero-stuffing);
//  * the last layer (16 -> 1) runs on the VALU.
// The network sees the image transposed ([x][y], the reference feeds COMPASS's first-index-fastest
// arrays); the transpose happens on the way into and out of LDS.
#include "aomarl_host.h"
#include <type_traits>
#include <vector>
#include <string.h>
#include <math.h>

typedef float f32x4d __attribute__((ext_vector_type(4)));

struct DenoiseW {
  const float *w1;        // [3][64]                 L1: tap 4i+q, channel c
  // fp32 B operands of L2, L3, D1, D2 in ONE allocation (k_denoise4 reads them through one buffer resource), one
  // float4 per lane per (tap, channel group, channel tile):
  //   L2 [9][1][2][64] at DN_WOFF2, L3 [9][2][4][64] at DN_WOFF3, D1 [4 classes][4 taps][4][2][64] at DN_WOFF4,
  //   D2 [4][4][2][1][64] at DN_WOFF5
  const float4 *wf;
  // k_denoise4's border table: [5][256] words, two 16-bit LDS byte offsets each (entry e of thread t in word
  // (e >> 1) * 256 + t, low half first): the zero stores of the six grids' borders, DN_DUMMY for a thread without one
  const unsigned *border;
  // split-fp16 B operands in 32-channel chunks (k_denoise4c): per (chunk, tile, lane) two 16-byte
  // words, hi then lo, of the lane's 8 consecutive K slots
  // in ONE allocation: L2 [5 tap pairs][2 tiles][64][2] at DC_WOFF2, L3 [9 taps][4 tiles][64][2] at DC_WOFF3,
  // D1 [4 classes][4 taps][2 halves][2 tiles][64][2] at DC_WOFF4, D2 [4 classes][4 taps][64][2] at DC_WOFF5
  const float4 *wc;
  const unsigned *border_c;   // k_denoise4c's border table (fp16 planes), same form as `border`
  const float *w6;        // [9][16]                 D3 (taps as a plain correlation)
  const float *b1, *b2, *b3, *b4, *b5;
  float b6;
  unsigned *ovf;           // device counter: images whose activations left the fp16 range (k_denoise4c)
};

#define DN_X 5184          // floats of region X (A1 2000, A3 2448, A5 5184)
#define DN_Y 3600          // floats of region Y (IN 324, A2 1296, A4 3600)
#define DN_S16 20          // padded channel strides
#define DN_S32 36
#define DN_S64 68

__device__ __forceinline__ f32x4d dn_mfma(float a, float b, f32x4d c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// zero the one-pixel border of an [R][R][S] channel-last grid (S floats per position, S % 4 == 0,
// or S == 1): 4R - 4 positions; the interior is overwritten by the layer that owns the grid (k_denoise4c;
// the fp32 kernel reads the same positions from a table)
template <int R, int S>
__device__ __forceinline__ void dn_border4(float *p, int tid) {
  constexpr int NP = 4 * R - 4;
  if (S == 1) {
    for (int i = tid; i < NP; i += 256) {
      const int row = i < R ? 0 : (i < 2 * R ? R - 1 : 1 + ((i - 2 * R) >> 1));
      const int col = i < R ? i : (i < 2 * R ? i - R : (((i - 2 * R) & 1) ? R - 1 : 0));
      p[row * R + col] = 0.f;
    }
  } else {
    constexpr int V = S / 4;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < NP * V; i += 256) {
      const int pi = i / V, k = i - pi * V;
      const int row = pi < R ? 0 : (pi < 2 * R ? R - 1 : 1 + ((pi - 2 * R) >> 1));
      const int col = pi < R ? pi : (pi < 2 * R ? pi - R : (((pi - 2 * R) & 1) ? R - 1 : 0));
      *reinterpret_cast<float4 *>(p + (row * R + col) * S + 4 * k) = z;
    }
  }
}

// four MFMAs: A = 4 consecutive channels of this lane's position, B = the matching weights
__device__ __forceinline__ f32x4d dn_quad(const float4 a, const float4 b, f32x4d acc) {
  acc = dn_mfma(a.x, b.x, acc);
  acc = dn_mfma(a.y, b.y, acc);
  acc = dn_mfma(a.z, b.z, acc);
  acc = dn_mfma(a.w, b.w, acc);
  return acc;
}

// ---------------------------------------------------------------------------------------------
// The all-fp32 kernel (aomarl_denoiser_apply_f32; the library's default arithmetic).  FOUR waves per image (four
// blocks = sixteen waves per CU; a two-wave-per-image version was measured at 6.1 ms per 307 200 images against
// 5.8 ms, profiles/r01h_*, and retired).
// Work split (wv = 0..3):
//   L1  M tiles 4 wv .. 4 wv + 3            L2  N tile wv & 1, M tiles 2 (wv >> 1), + 1
//   L3  N tile wv                           D1  N tile wv & 1, parity classes 2 (wv >> 1), + 1
//   D2  parity class wv (py = wv >> 1, px = wv & 1), all four M tiles       D3  a quad of lanes per pixel
// fp32 matrix and vector instructions share the SIMD's issue cycles on gfx950 (a matrix instruction is 32 of them,
// tools/mfmabench.hip): everything here that is not a matrix instruction is paid for in full, so
//  * the weights come through ONE buffer resource (all four layers in one allocation): lane offset in a register,
//    step offset a scalar -- no 64-bit vector address arithmetic, and no flat loads whose lgkmcnt the LDS reads
//    would wait for;
//  * the one-pixel zero borders of the six grids are a per-thread table of LDS offsets computed on the host (nine
//    entries, loaded once): a border costs an unpack and a store per entry, not 30 instructions of index arithmetic;
//  * the bias starts the accumulators;
//  * LDS banks: a 128-bit read is served in passes of 16 lanes over the 64 banks, so the 16 pixels of an M tile
//    must fall into 16 different 16-byte groups (mod 16).  With an odd number of groups per pixel (5, 9, 17) that
//    means 16 pixel indices distinct mod 16: the 4 x 4 grids use a row pitch of 12 (pitch 6 puts two pairs of pixels
//    16 apart), and the M tiles of the 8 x 8 grids are 8 rows x 2 columns (rows 10 apart: residues 0, 10, 4, 14, 8,
//    2, 12, 6), not 2 rows x 8 columns.  Every A-operand read is then one pass per 16 lanes.
// ---------------------------------------------------------------------------------------------
#define DN_P 12            // row pitch (pixels) of the 6-row grids A2, A3
#define DN_WOFF2 0         // float4 offsets of the four layers in DenoiseW::wf
#define DN_WOFF3 1152
#define DN_WOFF4 5760
#define DN_WOFF5 13952
#define DN_WTOTAL 16000
#define DN_NBORDER 9       // border table entries per thread: IN, A1, A2, A3 x 2, A4 x 2, A5 x 2
#define DN_DUMMY ((DN_X + DN_Y + 144) * 4)      // LDS byte offset of the 16 bytes idle table entries write to

__device__ __forceinline__ float4 dn_ldw(__amdgpu_buffer_rsrc_t rs, unsigned vo, unsigned so) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
}
__device__ __forceinline__ void dn_zero16(unsigned byte_off) {     // 16 zero bytes at an LDS byte offset
  extern __shared__ __attribute__((aligned(16))) float lds[];
  *reinterpret_cast<float4 *>(reinterpret_cast<char *>(lds) + byte_off) = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_denoise4(DenoiseW w, float *__restrict__ cube, int nimg) {
  constexpr int PF = 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *X = lds, *Y = lds + DN_X, *W6 = Y + DN_Y;          // W6: 144 weights of the last layer (+ the dummy slot)
  const int tid0 = threadIdx.x, lane0 = tid0 & 63, q0 = lane0 >> 4, c0 = lane0 & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int nt2 = wv & 1, hi2 = wv >> 1;
  for (int i = tid0; i < 144; i += 256) W6[i] = w.w6[i];
  float b1w[3];
  int t1off[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    b1w[i] = w.w1[i * 64 + lane0];
    const int tap = 4 * i + q0;
    t1off[i] = tap < 9 ? (tap / 3 - 1) * 18 + (tap % 3 - 1) : 0;
  }
  const float bias1 = w.b1[c0], bias2 = w.b2[16 * nt2 + c0], bias3 = w.b3[16 * wv + c0],
              bias4 = w.b4[16 * nt2 + c0], bias5 = w.b5[c0];
  unsigned bt[(DN_NBORDER + 1) / 2];     // border table: two 16-bit LDS byte offsets per register
#pragma unroll
  for (int i = 0; i < (DN_NBORDER + 1) / 2; i++) bt[i] = w.border[i * 256 + tid0];
  auto border = [&](int e) { return (e & 1) ? bt[e >> 1] >> 16 : bt[e >> 1] & 0xffffu; };
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(w.wf), 0, DN_WTOTAL * 16, 0x00020000);
  __syncthreads();
  float cur = 0.f;                       // this image's pixel of this thread, prefetched
  if ((int)blockIdx.x < nimg) cur = cube[(long long)blockIdx.x * 256 + tid0];
  for (int img = blockIdx.x; img < nimg; img += gridDim.x) {
    // (the per-lane LDS addresses of all six layers live in registers across the loop: a few of them spilled, and still
    // faster than recomputing them per image -- 7.8 against 8.1 ms; the split-fp16 kernel is the other way round.
    // Weight prefetch depth 2 / 3 / 4: 7.74 / 7.79 / 7.86 ms; three waves per SIMD without spills: 7.81)
    const int tid = tid0, q = q0, c = c0;
    const unsigned wvo = 16u * lane0;
    float *tile = cube + (long long)img * 256;
    const unsigned so2 = 16u * (DN_WOFF2 + nt2 * 64);        // step tap: + tap * 2 * 64 float4
    float4 rb2[PF];
#pragma unroll
    for (int s = 0; s < PF; s++) rb2[s] = dn_ldw(wrs, wvo, so2 + s * 2048);
    // ================= input (transposed) -> IN = Y[18][18]
    *reinterpret_cast<float *>(reinterpret_cast<char *>(lds) + border(0)) = 0.f;
    Y[((tid & 15) + 1) * 18 + ((tid >> 4) + 1)] = cur;       // tile[ty][tx] -> net row tx, col ty
    {
      const int nxt = img + gridDim.x;
      if (nxt < nimg) cur = cube[(long long)nxt * 256 + tid];
    }
    __syncthreads();
    // ================= L1: conv3x3 1->16, ReLU, pool -> A1 = X [10][10][20]
    dn_zero16(border(1));
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int mt = 4 * wv + k;
      const int win = 4 * mt + (c >> 2), r = c & 3;
      const int py = 2 * (win >> 3) + (r >> 1), px = 2 * (win & 7) + (r & 1);
      const float *in = Y + (py + 1) * 18 + (px + 1);
      f32x4d acc = {bias1, bias1, bias1, bias1};
#pragma unroll
      for (int i = 0; i < 3; i++) acc = dn_mfma(in[t1off[i]], b1w[i], acc);
      const float v = fmaxf(fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])), 0.f);
      const int wo = 4 * mt + q;
      X[(((wo >> 3) + 1) * 10 + (wo & 7) + 1) * DN_S16 + c] = v;
    }
    __syncthreads();
    // ================= L2: conv3x3 16->32 on 8x8, ReLU, pool -> A2 = Y [6][12][36]
    dn_zero16(border(2));
    const unsigned so3 = 16u * (DN_WOFF3 + wv * 64);         // step s = tap * 2 + g: + s * 4 * 64 float4
    float4 rb3[PF];
    {
      f32x4d acc[2] = {{bias2, bias2, bias2, bias2}, {bias2, bias2, bias2, bias2}};
      int abase[2];
#pragma unroll
      for (int m = 0; m < 2; m++) {
        // M tile mt = the four pooling windows of window COLUMN mt (8 pixel rows x 2 columns):
        // m = c -> window row c >> 2, pixel r = c & 3 of the window
        const int mt = 2 * hi2 + m, r = c & 3;
        const int py = 2 * (c >> 2) + (r >> 1), px = 2 * mt + (r & 1);
        abase[m] = ((py + 1) * 10 + (px + 1)) * DN_S16 + 4 * q;
      }
#pragma unroll
      for (int tap = 0; tap < 9; tap++) {
        const float4 b = rb2[tap % PF];
        if (tap + PF < 9) rb2[tap % PF] = dn_ldw(wrs, wvo, so2 + (tap + PF) * 2048);
        const int toff = ((tap / 3 - 1) * 10 + (tap % 3 - 1)) * DN_S16;
#pragma unroll
        for (int m = 0; m < 2; m++) {
          const float4 a = *reinterpret_cast<const float4 *>(X + abase[m] + toff);
          acc[m] = dn_quad(a, b, acc[m]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < PF; s++) rb3[s] = dn_ldw(wrs, wvo, so3 + s * 4096);
#pragma unroll
      for (int m = 0; m < 2; m++) {
        const int mt = 2 * hi2 + m;
        const float v = fmaxf(fmaxf(fmaxf(acc[m][0], acc[m][1]), fmaxf(acc[m][2], acc[m][3])), 0.f);
        // D: lane group q = window row q of column mt in the 4x4 pooled grid
        Y[((q + 1) * DN_P + mt + 1) * DN_S32 + 16 * nt2 + c] = v;
      }
    }
    __syncthreads();
    // ================= L3: conv3x3 32->64 on 4x4, ReLU -> A3 = X [6][12][68]; wave = channel tile wv
    dn_zero16(border(3));
    dn_zero16(border(4));
    // D1: step t = (clsl * 4 + tap) * 4 + g of this wave's two classes: + ((2 hi2) * 16 + t) * 2 * 64 float4
    const unsigned so4 = 16u * (DN_WOFF4 + (2 * hi2 * 16) * 128 + nt2 * 64);
    float4 rb4[PF];
    {
      f32x4d acc = {bias3, bias3, bias3, bias3};
      const int abase = (((c >> 2) + 1) * DN_P + (c & 3) + 1) * DN_S32 + 4 * q;
#pragma unroll
      for (int s = 0; s < 18; s++) {
        const int tap = s >> 1, g = s & 1;
        const int toff = ((tap / 3 - 1) * DN_P + (tap % 3 - 1)) * DN_S32;
        const float4 a = *reinterpret_cast<const float4 *>(Y + abase + toff + 16 * g);
        const float4 b = rb3[s % PF];
        if (s + PF < 18) rb3[s % PF] = dn_ldw(wrs, wvo, so3 + (s + PF) * 4096);
        acc = dn_quad(a, b, acc);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < PF; s++) rb4[s] = dn_ldw(wrs, wvo, so4 + s * 2048);
#pragma unroll
      for (int r = 0; r < 4; r++)                            // D: m = 4q + r -> pixel (q, r)
        X[((q + 1) * DN_P + r + 1) * DN_S64 + 16 * wv + c] = fmaxf(acc[r], 0.f);
    }
    __syncthreads();
    // ================= D1: convT4x4s2 64->32, 4x4 -> 8x8, ReLU -> A4 = Y [10][10][36]
    dn_zero16(border(5));
    dn_zero16(border(6));
    const unsigned so5 = 16u * (DN_WOFF5 + wv * 8 * 64);     // D2: class wv, step s = tap * 2 + g: + s * 64 float4
    float4 rb5[PF];
    // the output-row parity of this wave (py) is made a compile-time constant by a wave-uniform
    // branch: every tap offset stays an immediate of the LDS instruction (as run-time values they
    // became ~100 hoisted address registers)
    auto d1_body = [&](auto PYc) {
      constexpr int py = decltype(PYc)::value;
      const int a0 = c >> 2, b0 = c & 3;                     // A operand: m = c -> input pixel (a0, b0)
      const int abase = ((a0 + 1) * DN_P + b0 + 1) * DN_S64 + 4 * q;
      f32x4d acc = {bias4, bias4, bias4, bias4};
#pragma unroll
      for (int s = 0; s < 32; s++) {
        const int clsl = s >> 4, tap = (s >> 2) & 3, g = s & 3;
        const int px = clsl;                                 // class 2 py + clsl
        const int ty = tap >> 1, tx = tap & 1;
        const int dy = ty == 0 ? 0 : (py == 0 ? -1 : 1), dx = tx == 0 ? 0 : (px == 0 ? -1 : 1);
        const int toff = (dy * DN_P + dx) * DN_S64;
        const float4 a = *reinterpret_cast<const float4 *>(X + abase + toff + 16 * g);
        const float4 b = rb4[s % PF];
        if (s + PF < 32) rb4[s % PF] = dn_ldw(wrs, wvo, so4 + (s + PF) * 2048);
        acc = dn_quad(a, b, acc);
        if ((s & 15) == 15) {
          if (s == 31) {
#pragma unroll
            for (int t = 0; t < PF; t++) rb5[t] = dn_ldw(wrs, wvo, so5 + t * 1024);
          }
          // D: m = 4q + r -> input pixel (q, r) -> output pixel (2q + py, 2r + px)
#pragma unroll
          for (int r = 0; r < 4; r++)
            Y[((2 * q + py + 1) * 10 + 2 * r + px + 1) * DN_S32 + 16 * nt2 + c] = fmaxf(acc[r], 0.f);
          acc = f32x4d{bias4, bias4, bias4, bias4};
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (hi2 == 0) d1_body(std::integral_constant<int, 0>{}); else d1_body(std::integral_constant<int, 1>{});
    __syncthreads();
    // ================= D2: convT4x4s2 32->16, 8x8 -> 16x16, ReLU -> A5 = X [18][18][16]; wave = class
    dn_zero16(border(7));
    dn_zero16(border(8));
    auto d2_body = [&](auto PYc, auto PXc) {
      constexpr int py = decltype(PYc)::value, px = decltype(PXc)::value;
      int abase[4];
#pragma unroll
      for (int mt = 0; mt < 4; mt++)
        abase[mt] = (((c >> 1) + 1) * 10 + 2 * mt + (c & 1) + 1) * DN_S32 + 4 * q;   // M tile = columns 2mt, 2mt + 1
      f32x4d acc[4];
#pragma unroll
      for (int mt = 0; mt < 4; mt++) acc[mt] = f32x4d{bias5, bias5, bias5, bias5};
#pragma unroll
      for (int tap = 0; tap < 4; tap++) {
        const int ty = tap >> 1, tx = tap & 1;
        const int toff = ((ty == 0 ? 0 : (2 * py - 1)) * 10 + (tx == 0 ? 0 : (2 * px - 1))) * DN_S32;
#pragma unroll
        for (int g = 0; g < 2; g++) {
          const int s = tap * 2 + g;
          const float4 b = rb5[s % PF];
          if (s + PF < 8) rb5[s % PF] = dn_ldw(wrs, wvo, so5 + (s + PF) * 1024);
#pragma unroll
          for (int mt = 0; mt < 4; mt++) {
            const float4 a = *reinterpret_cast<const float4 *>(Y + abase[mt] + toff + 16 * g);
            acc[mt] = dn_quad(a, b, acc[mt]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // D: m = 4q + r -> input pixel (row 2q + (r >> 1), column 2 mt + (r & 1))
#pragma unroll
      for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int a = 2 * q + (r >> 1), b = 2 * mt + (r & 1);
          X[((2 * a + py + 1) * 18 + 2 * b + px + 1) * 16 + c] = fmaxf(acc[mt][r], 0.f);
        }
    };
    {
      std::integral_constant<int, 0> k0; std::integral_constant<int, 1> k1;
      if (wv == 0) d2_body(k0, k0); else if (wv == 1) d2_body(k0, k1); else if (wv == 2) d2_body(k1, k0); else d2_body(k1, k1);
    }
    __syncthreads();
    // ================= D3: 3x3 correlation 16 -> 1 on the vector lanes, write back transposed
    // Thread t = (pixel group t >> 2, channel group g = t & 3): the four lanes of a quad read the
    // four 16-byte channel groups of ONE pixel (64 contiguous bytes), a 16-lane pass of the LDS
    // 256 contiguous bytes -- one thread per pixel reading its 64 bytes was a 4-way bank conflict
    // on every read.  Each thread accumulates 4 pixels (p = (t >> 2) + 64 j) of its group, the quad
    // is summed with two DPP adds and lane g writes pixel j = g.
    {
      typedef float f32x2d __attribute__((ext_vector_type(2)));
      const int g = tid & 3, pg = tid >> 2;                  // pg: 0..63 -> pixels pg + 64 j
      float res = 0.f;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int p = pg + 64 * j;                           // net pixel (row p >> 4, col p & 15)
        const float *in = X + ((p >> 4) * 18 + (p & 15)) * 16 + 4 * g;
        f32x2d s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
          const float4 a = *reinterpret_cast<const float4 *>(in + ((tap / 3) * 18 + tap % 3) * 16);
          const float4 wg = *reinterpret_cast<const float4 *>(W6 + tap * 16 + 4 * g);
          s01 += f32x2d{a.x, a.y} * f32x2d{wg.x, wg.y};
          s23 += f32x2d{a.z, a.w} * f32x2d{wg.z, wg.w};
        }
        const f32x2d s2 = s01 + s23;
        float sum = s2[0] + s2[1];
        sum += dpp_f<0xB1>(sum);                             // quad_perm [1,0,3,2]
        sum += dpp_f<0x4E>(sum);                             // quad_perm [2,3,0,1]
        if (j == g) res = sum;
        __builtin_amdgcn_sched_barrier(0);
      }
      const int p = pg + 64 * g;
      tile[(p & 15) * 16 + (p >> 4)] = res + w.b6;           // tile[ty = net col][tx = net row]
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Split-fp16 kernel in 32-channel chunks.  The activations A1..A4 live in LDS as two fp16 planes
// (hi, lo), channel-last: an A operand of v_mfma_f32_16x16x32_f16 is 8 consecutive channels of one
// pixel = one ds_read_b128 per plane, a B operand 8 consecutive K slots of one output channel = one
// 16-byte load per part, and a chunk of 32 real channels costs THREE instructions
//     acc += A_hi B_hi + A_lo B_hi + A_hi B_lo          (lo x lo, 2^-22 of the product, is dropped)
// against four for 16-channel quads (the 32 slots filled with 16 channels twice; retired), with no operand
// duplication moves at all.  A chunk is one tap x 32 channels (Cin = 32, 64) or two taps x 16
// channels (Cin = 16: the second tap of the fifth pair does not exist -- zero weights).
// LDS: X = A1 / A3 planes or A5 (fp32), Y = IN (fp32) or A2 / A4 planes.
// LDS banks: a 128-bit read is served in passes of 16 lanes over the 64 banks, so the 16 pixels of an
// M tile must fall into 16 different 16-byte groups (mod 16).  With an odd pixel stride that means 16
// pixel indices distinct mod 16: the 4 x 4 grids use a row pitch of 12 (pitch 6 puts two pairs of
// pixels 16 apart), and the M tiles of the 8 x 8 grids are 8 rows x 2 columns (rows 10 apart: residues
// 0, 10, 4, 14, 8, 2, 12, 6), not 2 rows x 8 columns.  Every A-operand read is then one pass per 16 lanes.
// ---------------------------------------------------------------------------------------------
#define DC_S1 24           // halfs per pixel: 16 channels + 8 pad
#define DC_S2 40           // 32 + 8
#define DC_S3 72           // 64 + 8
#define DC_P 12            // row pitch (pixels) of the 6-row grids A2, A3 (see the bank note below)
#define DC_PLX 5184        // halfs: lo plane offset in X (A3: 6 x 12 x 72)
#define DC_PLY 4000        // halfs: lo plane offset in Y (A4: 100 x 40)
#define DC_Y 4000          // floats of region Y (2 x 4000 halfs)

struct DcB { hx8 h, l; };
__device__ __forceinline__ DcB dc_ldw(__amdgpu_buffer_rsrc_t rs, unsigned vo, unsigned so) {
  DcB r;                                 // vo = 32 lane: the lane's hi word, its lo word 16 bytes further
  r.h = __builtin_bit_cast(hx8, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
  r.l = __builtin_bit_cast(hx8, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 16u, so, 0));
  return r;
}
__device__ __forceinline__ f32x4d dc_chunk(const _Float16 *__restrict__ hi, const _Float16 *__restrict__ lo,
                                           int off, const DcB b, f32x4d acc) {
  const hx8 ah = *reinterpret_cast<const hx8 *>(hi + off), al = *reinterpret_cast<const hx8 *>(lo + off);
  acc = mfma_h(ah, b.h, acc);
  acc = mfma_h(al, b.h, acc);
  return mfma_h(ah, b.l, acc);
}
__device__ __forceinline__ void dc_store(_Float16 *hi, _Float16 *lo, int off, float v, float &vmax) {
  vmax = fmaxf(vmax, v);                 // activations are >= 0 (ReLU); checked against the fp16 range at the end
  const _Float16 h = (_Float16)v;
  hi[off] = h;
  lo[off] = (_Float16)(v - (float)h);
}

#define DC_WOFF2 0         // float4 offsets of the four layers in DenoiseW::wc
#define DC_WOFF3 1280
#define DC_WOFF4 5888
#define DC_WOFF5 14080
#define DC_WTOTAL 16128
#define DC_DUMMY ((DN_X + DC_Y + 144) * 4)      // LDS byte offset of the 16 bytes idle border-table entries write to

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_denoise4c(DenoiseW w, float *__restrict__ cube, int nimg) {
  constexpr int PF = 3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *X = lds, *Y = lds + DN_X, *W6 = Y + DC_Y;          // W6: 144 weights of the last layer
  _Float16 *XH = reinterpret_cast<_Float16 *>(X), *XL = XH + DC_PLX;
  _Float16 *YH = reinterpret_cast<_Float16 *>(Y), *YL = YH + DC_PLY;
  const int tid0 = threadIdx.x, lane0 = tid0 & 63, q0 = lane0 >> 4, c0 = lane0 & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int nt2 = wv & 1, hi2 = wv >> 1;
  for (int i = tid0; i < 144; i += 256) W6[i] = w.w6[i];
  float b1w[3];
  int t1off[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    b1w[i] = w.w1[i * 64 + lane0];
    const int tap = 4 * i + q0;
    t1off[i] = tap < 9 ? (tap / 3 - 1) * 18 + (tap % 3 - 1) : 0;
  }
  const float bias1 = w.b1[c0], bias2 = w.b2[16 * nt2 + c0], bias3 = w.b3[16 * wv + c0],
              bias4 = w.b4[16 * nt2 + c0], bias5 = w.b5[c0];
  unsigned bt[(DN_NBORDER + 1) / 2];     // border table: two 16-bit LDS byte offsets per register (see k_denoise4)
#pragma unroll
  for (int i = 0; i < (DN_NBORDER + 1) / 2; i++) bt[i] = w.border_c[i * 256 + tid0];
  auto border = [&](int e) { return (e & 1) ? bt[e >> 1] >> 16 : bt[e >> 1] & 0xffffu; };
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(w.wc), 0, DC_WTOTAL * 16, 0x00020000);
  __syncthreads();
  float cur = 0.f;                       // this image's pixel of this thread, prefetched
  float vmax = 0.f;                      // largest activation this thread stored as an fp16 pair
  if ((int)blockIdx.x < nimg) cur = cube[(long long)blockIdx.x * 256 + tid0];
  for (int img = blockIdx.x; img < nimg; img += gridDim.x) {
    // per-lane indices laundered once per image: every LDS address below is recomputed where it is used instead of
    // living in a register across the whole loop (~80 of them otherwise)
    int tid = tid0, q = q0, c = c0;
    asm volatile("" : "+v"(tid), "+v"(q), "+v"(c));
    const unsigned wvo = 32u * (tid & 63);
    float *tile = cube + (long long)img * 256;
    const unsigned so2 = 16u * DC_WOFF2 + 32u * (nt2 * 64);  // chunk j: + j * 2 tiles * 64 (x 2 words)
    DcB rb2[PF];
#pragma unroll
    for (int s = 0; s < PF; s++) rb2[s] = dc_ldw(wrs, wvo, so2 + s * 4096);
    // ================= input (transposed) -> IN = Y[18][18] (fp32)
    *reinterpret_cast<float *>(reinterpret_cast<char *>(lds) + border(0)) = 0.f;
    Y[((tid & 15) + 1) * 18 + ((tid >> 4) + 1)] = cur;       // tile[ty][tx] -> net row tx, col ty
    {
      const int nxt = img + gridDim.x;
      if (nxt < nimg) cur = cube[(long long)nxt * 256 + tid];
    }
    __syncthreads();
    // ================= L1: conv3x3 1->16, ReLU, pool -> A1 = X planes [10][10][24]
    dn_zero16(border(1));
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int mt = 4 * wv + k;
      const int win = 4 * mt + (c >> 2), r = c & 3;
      const int py = 2 * (win >> 3) + (r >> 1), px = 2 * (win & 7) + (r & 1);
      const float *in = Y + (py + 1) * 18 + (px + 1);
      f32x4d acc = {bias1, bias1, bias1, bias1};
#pragma unroll
      for (int i = 0; i < 3; i++) acc = dn_mfma(in[t1off[i]], b1w[i], acc);
      const float v = fmaxf(fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])), 0.f);
      const int wo = 4 * mt + q;
      dc_store(XH, XL, (((wo >> 3) + 1) * 10 + (wo & 7) + 1) * DC_S1 + c, v, vmax);
    }
    __syncthreads();
    // ================= L2: conv3x3 16->32 on 8x8, ReLU, pool -> A2 = Y planes [6][6][40]
    dn_zero16(border(2));
    const unsigned so3 = 16u * DC_WOFF3 + 32u * (wv * 64);   // chunk tap: + tap * 4 tiles * 64 (x 2)
    DcB rb3[PF];
    {
      f32x4d acc[2] = {{bias2, bias2, bias2, bias2}, {bias2, bias2, bias2, bias2}};
      int abase[2];
#pragma unroll
      for (int m = 0; m < 2; m++) {
        // M tile mt = the four pooling windows of window COLUMN mt (8 pixel rows x 2 columns):
        // m = c -> window row c >> 2, pixel r = c & 3 of the window
        const int mt = 2 * hi2 + m, r = c & 3;
        const int py = 2 * (c >> 2) + (r >> 1), px = 2 * mt + (r & 1);
        abase[m] = ((py + 1) * 10 + (px + 1)) * DC_S1 + 8 * (q & 1);
      }
      const int qt = q >> 1;                                 // which tap of the pair this lane group reads
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const DcB b = rb2[j % PF];
        if (j + PF < 5) rb2[j % PF] = dc_ldw(wrs, wvo, so2 + (j + PF) * 4096);
        const int ta = 2 * j, tb = (2 * j + 1 < 9) ? 2 * j + 1 : 2 * j;
        const int offa = ((ta / 3 - 1) * 10 + (ta % 3 - 1)) * DC_S1, offb = ((tb / 3 - 1) * 10 + (tb % 3 - 1)) * DC_S1;
        const int toff = qt ? offb : offa;
#pragma unroll
        for (int m = 0; m < 2; m++) acc[m] = dc_chunk(XH, XL, abase[m] + toff, b, acc[m]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < PF; s++) rb3[s] = dc_ldw(wrs, wvo, so3 + s * 8192);
#pragma unroll
      for (int m = 0; m < 2; m++) {
        const int mt = 2 * hi2 + m;
        const float v = fmaxf(fmaxf(fmaxf(acc[m][0], acc[m][1]), fmaxf(acc[m][2], acc[m][3])), 0.f);
        // D: lane group q = window row q of column mt in the 4x4 pooled grid
        dc_store(YH, YL, ((q + 1) * DC_P + mt + 1) * DC_S2 + 16 * nt2 + c, v, vmax);
      }
    }
    __syncthreads();
    // ================= L3: conv3x3 32->64 on 4x4, ReLU -> A3 = X planes [6][6][72]; wave = channel tile wv
    dn_zero16(border(3));
    dn_zero16(border(4));
    // D1: chunk t = (clsl * 4 + tap) * 2 + h of this wave's two classes: + ((2 hi2) * 8 + t) * 2 tiles * 64
    const unsigned so4 = 16u * DC_WOFF4 + 32u * ((2 * hi2 * 8) * 128 + nt2 * 64);
    DcB rb4[PF];
    {
      f32x4d acc = {bias3, bias3, bias3, bias3};
      const int abase = (((c >> 2) + 1) * DC_P + (c & 3) + 1) * DC_S2 + 8 * q;
#pragma unroll
      for (int tap = 0; tap < 9; tap++) {
        const DcB b = rb3[tap % PF];
        if (tap + PF < 9) rb3[tap % PF] = dc_ldw(wrs, wvo, so3 + (tap + PF) * 8192);
        acc = dc_chunk(YH, YL, abase + ((tap / 3 - 1) * DC_P + (tap % 3 - 1)) * DC_S2, b, acc);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < PF; s++) rb4[s] = dc_ldw(wrs, wvo, so4 + s * 4096);
#pragma unroll
      for (int r = 0; r < 4; r++)                            // D: m = 4q + r -> pixel (q, r)
        dc_store(XH, XL, ((q + 1) * DC_P + r + 1) * DC_S3 + 16 * wv + c, fmaxf(acc[r], 0.f), vmax);
    }
    __syncthreads();
    // ================= D1: convT4x4s2 64->32, 4x4 -> 8x8, ReLU -> A4 = Y planes [10][10][40]
    dn_zero16(border(5));
    dn_zero16(border(6));
    const unsigned so5 = 16u * DC_WOFF5 + 32u * (wv * 4 * 64);   // D2: class wv, chunk tap: + tap * 64 (x 2)
    DcB rb5[PF];
    auto d1_body = [&](auto PYc) {
      constexpr int py = decltype(PYc)::value;
      const int abase = (((c >> 2) + 1) * DC_P + (c & 3) + 1) * DC_S3 + 8 * q;   // m = c -> input pixel
      f32x4d acc = {bias4, bias4, bias4, bias4};
#pragma unroll
      for (int s = 0; s < 16; s++) {
        const int clsl = s >> 3, tap = (s >> 1) & 3, h = s & 1;
        const int px = clsl;                                 // class 2 py + clsl
        const int ty = tap >> 1, tx = tap & 1;
        const int dy = ty == 0 ? 0 : (py == 0 ? -1 : 1), dx = tx == 0 ? 0 : (px == 0 ? -1 : 1);
        const DcB b = rb4[s % PF];
        if (s + PF < 16) rb4[s % PF] = dc_ldw(wrs, wvo, so4 + (s + PF) * 4096);
        acc = dc_chunk(XH, XL, abase + (dy * DC_P + dx) * DC_S3 + 32 * h, b, acc);
        if ((s & 7) == 7) {
          if (s == 15) {
#pragma unroll
            for (int t = 0; t < PF; t++) rb5[t] = dc_ldw(wrs, wvo, so5 + t * 2048);
          }
          // D: m = 4q + r -> input pixel (q, r) -> output pixel (2q + py, 2r + px)
#pragma unroll
          for (int r = 0; r < 4; r++)
            dc_store(YH, YL, ((2 * q + py + 1) * 10 + 2 * r + px + 1) * DC_S2 + 16 * nt2 + c,
                     fmaxf(acc[r], 0.f), vmax);
          acc = f32x4d{bias4, bias4, bias4, bias4};
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (hi2 == 0) d1_body(std::integral_constant<int, 0>{}); else d1_body(std::integral_constant<int, 1>{});
    __syncthreads();
    // ================= D2: convT4x4s2 32->16, 8x8 -> 16x16, ReLU -> A5 = X [18][18][16] fp32; wave = class
    dn_zero16(border(7));
    dn_zero16(border(8));
    auto d2_body = [&](auto PYc, auto PXc) {
      constexpr int py = decltype(PYc)::value, px = decltype(PXc)::value;
      int abase[4];
#pragma unroll
      for (int mt = 0; mt < 4; mt++)
        abase[mt] = (((c >> 1) + 1) * 10 + 2 * mt + (c & 1) + 1) * DC_S2 + 8 * q;   // M tile = columns 2mt, 2mt + 1
      f32x4d acc[4];
#pragma unroll
      for (int mt = 0; mt < 4; mt++) acc[mt] = f32x4d{bias5, bias5, bias5, bias5};
#pragma unroll
      for (int tap = 0; tap < 4; tap++) {
        const int ty = tap >> 1, tx = tap & 1;
        const int toff = ((ty == 0 ? 0 : (2 * py - 1)) * 10 + (tx == 0 ? 0 : (2 * px - 1))) * DC_S2;
        const DcB b = rb5[tap % PF];
        if (tap + PF < 4) rb5[tap % PF] = dc_ldw(wrs, wvo, so5 + (tap + PF) * 2048);
#pragma unroll
        for (int mt = 0; mt < 4; mt++) acc[mt] = dc_chunk(YH, YL, abase[mt] + toff, b, acc[mt]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // D: m = 4q + r -> input pixel (row 2q + (r >> 1), column 2 mt + (r & 1))
#pragma unroll
      for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int a = 2 * q + (r >> 1), b = 2 * mt + (r & 1);
          X[((2 * a + py + 1) * 18 + 2 * b + px + 1) * 16 + c] = fmaxf(acc[mt][r], 0.f);
        }
    };
    {
      std::integral_constant<int, 0> k0; std::integral_constant<int, 1> k1;
      if (wv == 0) d2_body(k0, k0); else if (wv == 1) d2_body(k0, k1); else if (wv == 2) d2_body(k1, k0); else d2_body(k1, k1);
    }
    __syncthreads();
    // ================= D3: 3x3 correlation 16 -> 1 on the VALU, write back transposed
    // Thread t = (pixel group t >> 2, channel group g = t & 3): the four lanes of a quad read the
    // four 16-byte channel groups of ONE pixel (64 contiguous bytes), a 16-lane pass of the LDS
    // 256 contiguous bytes -- one thread per pixel reading its 64 bytes was a 4-way bank conflict
    // on every read.  Each thread accumulates 4 pixels (p = (t >> 2) + 64 j) of its group, the quad
    // is summed with two DPP adds and lane g writes pixel j = g.
    {
      typedef float f32x2d __attribute__((ext_vector_type(2)));
      const int g = tid & 3, pg = tid >> 2;                  // pg: 0..63 -> pixels pg + 64 j
      float res = 0.f;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int p = pg + 64 * j;                           // net pixel (row p >> 4, col p & 15)
        const float *in = X + ((p >> 4) * 18 + (p & 15)) * 16 + 4 * g;
        f32x2d s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
          const float4 a = *reinterpret_cast<const float4 *>(in + ((tap / 3) * 18 + tap % 3) * 16);
          const float4 wg = *reinterpret_cast<const float4 *>(W6 + tap * 16 + 4 * g);
          s01 += f32x2d{a.x, a.y} * f32x2d{wg.x, wg.y};
          s23 += f32x2d{a.z, a.w} * f32x2d{wg.z, wg.w};
        }
        const f32x2d s2 = s01 + s23;
        float sum = s2[0] + s2[1];
        sum += dpp_f<0xB1>(sum);                             // quad_perm [1,0,3,2]
        sum += dpp_f<0x4E>(sum);                             // quad_perm [2,3,0,1]
        if (j == g) res = sum;
        __builtin_amdgcn_sched_barrier(0);
      }
      const int p = pg + 64 * g;
      tile[(p & 15) * 16 + (p >> 4)] = res + w.b6;           // tile[ty = net col][tx = net row]
    }
  }
  // fp16 pairs saturate silently above 65504 (v_cvt rounds to the largest finite value): count it,
  // the host refuses to go on (aomarl_denoiser_overflow).  !(x <= limit) also catches NaN.
  if (!(vmax <= 65000.f)) atomicAdd(w.ovf, 1u);
}

// ------------------------------------------------------------------------------------ host side
struct aomarl_denoiser {
  DenoiseW w;
  std::vector<void *> owned;
};

template <typename T>
static int dn_upload(aomarl_denoiser *d, const std::vector<T> &h, const T **dev) {
  void *p = nullptr;
  if (hipMalloc(&p, h.size() * sizeof(T)) != hipSuccess) return fail("denoiser: hipMalloc failed");
  d->owned.push_back(p);
  if (hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
    return fail("denoiser: hipMemcpy failed");
  *dev = reinterpret_cast<const T *>(p);
  return 0;
}

int aomarl_denoiser_destroy(aomarl_denoiser *d) {
  if (!d) return 0;
  for (void *p : d->owned) (void)hipFree(p);
  delete d;
  return 0;
}

// weights / biases: 6 host arrays each, PyTorch layouts of the reference checkpoint
// (encoder1..3: Conv2d [Cout][Cin][3][3]; decoder1, 2: ConvTranspose2d [Cin][Cout][4][4];
//  decoder3: ConvTranspose2d [16][1][3][3])
int aomarl_denoiser_create(const float *const *wt, const float *const *bs, aomarl_denoiser **out) {
  if (!wt || !bs || !out) return fail("denoiser_create: null argument");
  for (int i = 0; i < 6; i++) if (!wt[i] || !bs[i]) return fail("denoiser_create: null layer %d", i);
  aomarl_denoiser *d = new aomarl_denoiser();
  int rc = 0;
  // value for lane (q, c), slot j of a (tap, group, tile) quad: input channel 16 g + 4 q + j,
  // output channel 16 nt + c
  auto lane_q = [](int lane) { return lane >> 4; };
  auto lane_c = [](int lane) { return lane & 15; };
  {  // L1: [3][64]: tap 4 i + q, output channel c
    std::vector<float> h(3 * 64, 0.f);
    for (int i = 0; i < 3; i++)
      for (int lane = 0; lane < 64; lane++) {
        const int tap = 4 * i + lane_q(lane), co = lane_c(lane);
        if (tap < 9) h[i * 64 + lane] = wt[0][(co * 1 + 0) * 9 + tap];
      }
    rc = dn_upload<float>(d, h, &d->w.w1);
  }
  auto conv_pack = [&](const float *W, int Cout, int Cin, std::vector<float4> &h) {
    const int G = Cin / 16, NT = Cout / 16;
    h.assign((size_t)9 * G * NT * 64, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int tap = 0; tap < 9; tap++)
      for (int g = 0; g < G; g++)
        for (int nt = 0; nt < NT; nt++)
          for (int lane = 0; lane < 64; lane++) {
            float v[4];
            for (int j = 0; j < 4; j++) {
              const int ci = 16 * g + 4 * lane_q(lane) + j, co = 16 * nt + lane_c(lane);
              v[j] = W[((size_t)co * Cin + ci) * 9 + tap];
            }
            h[((size_t)(tap * G + g) * NT + nt) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
          }
  };
  auto convT_pack = [&](const float *W, int Cin, int Cout, std::vector<float4> &h) {
    const int G = Cin / 16, NT = Cout / 16;
    h.assign((size_t)16 * G * NT * 64, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int cls = 0; cls < 4; cls++)
      for (int tap = 0; tap < 4; tap++) {
        const int py = cls >> 1, px = cls & 1, ty = tap >> 1, tx = tap & 1;
        // oy = 2 iy - 1 + ky: even rows take ky = 1 (same input row) / 3 (row above),
        // odd rows ky = 2 (same) / 0 (row below); columns alike
        const int ky = py == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 2 : 0);
        const int kx = px == 0 ? (tx == 0 ? 1 : 3) : (tx == 0 ? 2 : 0);
        for (int g = 0; g < G; g++)
          for (int nt = 0; nt < NT; nt++)
            for (int lane = 0; lane < 64; lane++) {
              float v[4];
              for (int j = 0; j < 4; j++) {
                const int ci = 16 * g + 4 * lane_q(lane) + j, co = 16 * nt + lane_c(lane);
                v[j] = W[(((size_t)ci * Cout + co) * 4 + ky) * 4 + kx];
              }
              h[((size_t)((cls * 4 + tap) * G + g) * NT + nt) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
            }
      }
  };
  {
    std::vector<float4> h4, all;
    conv_pack(wt[1], 32, 16, h4); all.insert(all.end(), h4.begin(), h4.end());
    if (all.size() != DN_WOFF3) rc = fail("denoiser: weight layout");
    conv_pack(wt[2], 64, 32, h4); all.insert(all.end(), h4.begin(), h4.end());
    if (all.size() != DN_WOFF4) rc = fail("denoiser: weight layout");
    convT_pack(wt[3], 64, 32, h4); all.insert(all.end(), h4.begin(), h4.end());
    if (all.size() != DN_WOFF5) rc = fail("denoiser: weight layout");
    convT_pack(wt[4], 32, 16, h4); all.insert(all.end(), h4.begin(), h4.end());
    if (all.size() != DN_WTOTAL) rc = fail("denoiser: weight layout");
    if (!rc) rc = dn_upload<float4>(d, all, &d->w.wf);
  }
  // border tables (LDS byte offsets; X at 0, Y behind it): per grid the planes' bases, rows, row pitch (pixels), pixel
  // stride and the bytes to zero per border pixel (4: one float per store, else 16-byte stores)
  struct Grid { int base[2], nplanes, rows, pitch, stride, bytes; };
  auto border_table = [&](const Grid (&grids)[6], unsigned dummy, const unsigned **dev) -> int {
    const int nent[6] = {1, 1, 1, 2, 2, 2};
    std::vector<unsigned> tab((size_t)((DN_NBORDER + 1) / 2) * 256, 0u);
    int e0 = 0;
    for (int gi = 0; gi < 6; gi++) {
      const Grid &g = grids[gi];
      std::vector<unsigned> off;                             // byte offsets of this grid's zero stores
      for (int pl = 0; pl < g.nplanes; pl++)
        for (int r = 0; r < g.rows; r++)
          for (int cc = 0; cc < g.rows; cc++) {
            if (r != 0 && r != g.rows - 1 && cc != 0 && cc != g.rows - 1) continue;
            const int px = g.base[pl] + (r * g.pitch + cc) * g.stride;
            for (int k = 0; k < g.bytes; k += 16) off.push_back((unsigned)(px + k));
          }
      if ((int)off.size() > 256 * nent[gi]) return fail("denoiser: border table");
      for (int e = 0; e < nent[gi]; e++)
        for (int t = 0; t < 256; t++) {
          const size_t i = (size_t)e * 256 + t;
          const unsigned v = i < off.size() ? off[i] : dummy;
          if (v > 0xffffu || (g.bytes >= 16 && (v & 15))) return fail("denoiser: border offset");
          const int ent = e0 + e;
          tab[(size_t)(ent >> 1) * 256 + t] |= v << (16 * (ent & 1));
        }
      e0 += nent[gi];
    }
    return dn_upload<unsigned>(d, tab, dev);
  };
  if (!rc) {  // k_denoise4: fp32 grids
    const Grid grids[6] = {{{4 * DN_X, 0}, 1, 18, 18, 4, 4},                 // IN
                           {{0, 0}, 1, 10, 10, 4 * DN_S16, 64},              // A1
                           {{4 * DN_X, 0}, 1, 6, DN_P, 4 * DN_S32, 128},     // A2
                           {{0, 0}, 1, 6, DN_P, 4 * DN_S64, 256},            // A3
                           {{4 * DN_X, 0}, 1, 10, 10, 4 * DN_S32, 128},      // A4
                           {{0, 0}, 1, 18, 18, 64, 64}};                     // A5
    rc = border_table(grids, DN_DUMMY, &d->w.border);
  }
  if (!rc) {  // k_denoise4c: A1..A4 as two fp16 planes (hi, lo), IN and A5 fp32
    const int Y0 = 4 * DN_X;
    const Grid grids[6] = {{{Y0, 0}, 1, 18, 18, 4, 4},                                   // IN
                           {{0, 2 * DC_PLX}, 2, 10, 10, 2 * DC_S1, 32},                  // A1
                           {{Y0, Y0 + 2 * DC_PLY}, 2, 6, DC_P, 2 * DC_S2, 64},           // A2
                           {{0, 2 * DC_PLX}, 2, 6, DC_P, 2 * DC_S3, 128},                // A3
                           {{Y0, Y0 + 2 * DC_PLY}, 2, 10, 10, 2 * DC_S2, 64},            // A4
                           {{0, 0}, 1, 18, 18, 64, 64}};                                 // A5
    rc = border_table(grids, DC_DUMMY, &d->w.border_c);
  }
  // ---- 32-channel chunk operands of k_denoise4c: lane (n = lane & 15, kg = lane >> 4) holds K slots
  //      8 kg .. 8 kg + 7 of output channel 16 nt + n, as 8 hi halfs then 8 lo halfs
  auto put_chunk = [](std::vector<float4> &h, size_t idx, const float v[8]) {
    _Float16 hi[8], lo[8];
    for (int i = 0; i < 8; i++) { hi[i] = (_Float16)v[i]; lo[i] = (_Float16)(v[i] - (float)hi[i]); }
    memcpy(&h[2 * idx], hi, 16);
    memcpy(&h[2 * idx + 1], lo, 16);
  };
  auto convT_k = [](int py, int ty) { return py == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 2 : 0); };
  std::vector<float4> allc;
  if (!rc) {  // L2: chunk j = taps 2j, 2j + 1 x 16 channels; slot: tap 2j + (kg >> 1), channel 8 (kg & 1) + i
    std::vector<float4> h((size_t)5 * 2 * 64 * 2);
    for (int j = 0; j < 5; j++)
      for (int nt = 0; nt < 2; nt++)
        for (int lane = 0; lane < 64; lane++) {
          const int kg = lane >> 4, co = 16 * nt + (lane & 15), tap = 2 * j + (kg >> 1);
          float v[8];
          for (int i = 0; i < 8; i++) v[i] = tap < 9 ? wt[1][((size_t)co * 16 + 8 * (kg & 1) + i) * 9 + tap] : 0.f;
          put_chunk(h, ((size_t)j * 2 + nt) * 64 + lane, v);
        }
    allc.insert(allc.end(), h.begin(), h.end());
    if (allc.size() != DC_WOFF3) rc = fail("denoiser: weight layout");
  }
  if (!rc) {  // L3: chunk = tap x 32 channels; slot: channel 8 kg + i
    std::vector<float4> h((size_t)9 * 4 * 64 * 2);
    for (int tap = 0; tap < 9; tap++)
      for (int nt = 0; nt < 4; nt++)
        for (int lane = 0; lane < 64; lane++) {
          const int kg = lane >> 4, co = 16 * nt + (lane & 15);
          float v[8];
          for (int i = 0; i < 8; i++) v[i] = wt[2][((size_t)co * 32 + 8 * kg + i) * 9 + tap];
          put_chunk(h, ((size_t)tap * 4 + nt) * 64 + lane, v);
        }
    allc.insert(allc.end(), h.begin(), h.end());
    if (allc.size() != DC_WOFF4) rc = fail("denoiser: weight layout");
  }
  if (!rc) {  // D1: chunk = (class, tap, half of the 64 channels); ConvTranspose2d weight [Cin][Cout][4][4]
    std::vector<float4> h((size_t)4 * 4 * 2 * 2 * 64 * 2);
    for (int cls = 0; cls < 4; cls++)
      for (int tap = 0; tap < 4; tap++)
        for (int hf = 0; hf < 2; hf++)
          for (int nt = 0; nt < 2; nt++)
            for (int lane = 0; lane < 64; lane++) {
              const int kg = lane >> 4, co = 16 * nt + (lane & 15);
              const int ky = convT_k(cls >> 1, tap >> 1), kx = convT_k(cls & 1, tap & 1);
              float v[8];
              for (int i = 0; i < 8; i++) v[i] = wt[3][(((size_t)(32 * hf + 8 * kg + i) * 32 + co) * 4 + ky) * 4 + kx];
              put_chunk(h, ((((size_t)cls * 4 + tap) * 2 + hf) * 2 + nt) * 64 + lane, v);
            }
    allc.insert(allc.end(), h.begin(), h.end());
    if (allc.size() != DC_WOFF5) rc = fail("denoiser: weight layout");
  }
  if (!rc) {  // D2: chunk = (class, tap) x 32 channels; weight [32][16][4][4]
    std::vector<float4> h((size_t)4 * 4 * 64 * 2);
    for (int cls = 0; cls < 4; cls++)
      for (int tap = 0; tap < 4; tap++)
        for (int lane = 0; lane < 64; lane++) {
          const int kg = lane >> 4, co = lane & 15;
          const int ky = convT_k(cls >> 1, tap >> 1), kx = convT_k(cls & 1, tap & 1);
          float v[8];
          for (int i = 0; i < 8; i++) v[i] = wt[4][(((size_t)(8 * kg + i) * 16 + co) * 4 + ky) * 4 + kx];
          put_chunk(h, ((size_t)cls * 4 + tap) * 64 + lane, v);
        }
    allc.insert(allc.end(), h.begin(), h.end());
    if (allc.size() != DC_WTOTAL) rc = fail("denoiser: weight layout");
    if (!rc) rc = dn_upload<float4>(d, allc, &d->w.wc);
  }
  if (!rc) {  // D3: out[oy][ox] = sum in[oy + 1 - ky][ox + 1 - kx] w[ci][0][ky][kx]: tap (ty, tx) = (2 - ky, 2 - kx)
    std::vector<float> h(9 * 16);
    for (int ty = 0; ty < 3; ty++)
      for (int tx = 0; tx < 3; tx++)
        for (int ci = 0; ci < 16; ci++) h[(ty * 3 + tx) * 16 + ci] = wt[5][(ci * 9) + (2 - ty) * 3 + (2 - tx)];
    rc = dn_upload<float>(d, h, &d->w.w6);
  }
  const int nb[5] = {16, 32, 64, 32, 16};
  const float **bdev[5] = {&d->w.b1, &d->w.b2, &d->w.b3, &d->w.b4, &d->w.b5};
  for (int i = 0; i < 5 && !rc; i++) {
    std::vector<float> h(bs[i], bs[i] + nb[i]);
    rc = dn_upload<float>(d, h, bdev[i]);
  }
  d->w.b6 = bs[5][0];
  if (!rc) {
    void *fl = nullptr;
    if (hipMalloc(&fl, sizeof(unsigned)) != hipSuccess || hipMemset(fl, 0, sizeof(unsigned)) != hipSuccess)
      rc = fail("denoiser: hipMalloc failed");
    else { d->owned.push_back(fl); d->w.ovf = reinterpret_cast<unsigned *>(fl); }
  }
  if (rc) { aomarl_denoiser_destroy(d); return rc; }
  *out = d;
  return 0;
}

static int denoiser_launch(aomarl_denoiser *d, float *cube, long long nimg, bool f32, void *stream) {
  if (!d || !cube) return fail("denoiser_apply: null argument");
  if (nimg <= 0) return 0;
  if (nimg > 0x7fffffffLL) return fail("denoiser_apply: too many images");
  const int blocks = (int)std::min<long long>(nimg, 256 * 4 * 4);
  if (!f32) {
    // split-fp16 operands in 32-channel chunks, four waves per image
    const size_t smc = sizeof(float) * (DN_X + DC_Y + 144 + 4);
    hipLaunchKernelGGL(k_denoise4c, dim3(blocks), dim3(256), smc, (hipStream_t)stream, d->w, cube, (int)nimg);
    g_arith[AR_DENOISE_SPLIT]++;
  } else {
    const size_t smem = sizeof(float) * (DN_X + DN_Y + 144 + 4);
    hipLaunchKernelGGL(k_denoise4, dim3(blocks), dim3(256), smem, (hipStream_t)stream, d->w, cube, (int)nimg);
    g_arith[AR_DENOISE_F32]++;
  }
  LAUNCHCHK();
  return 0;
}

int aomarl_denoiser_apply(aomarl_denoiser *d, float *cube, long long nimg, void *stream) {
  return denoiser_launch(d, cube, nimg, g_precision == 0, stream);      // the library's precision mode
}

int aomarl_denoiser_apply_split_f16(aomarl_denoiser *d, float *cube, long long nimg, void *stream) {
  return denoiser_launch(d, cube, nimg, false, stream);
}

int aomarl_denoiser_apply_f32(aomarl_denoiser *d, float *cube, long long nimg, void *stream) {
  return denoiser_launch(d, cube, nimg, true, stream);
}

int aomarl_denoiser_overflow(aomarl_denoiser *d, unsigned *count, void *stream) {
  if (!d || !count) return fail("denoiser_overflow: null argument");
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(count, d->w.ovf, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (*count) HIPCHK(hipMemsetAsync(d->w.ovf, 0, sizeof(unsigned), s));
  return 0;
}
