// aomarl_sac.hip -- one soft actor-critic update of EVERY agent as a fixed sequence of gfx950 kernels
// ("replay + SAC update").  A translation unit of its own (aomarl_host.h).
//
// Reference (per agent, one process + one torch autograd graph each):
//   SAC.update_critic   src/reinforcement_learning/rpc_training/train_rpc.py:985-1038
//   SAC.update_actor    train_rpc.py:1044-1064        SAC.update_alpha  train_rpc.py:1070-1084
//   soft_update         train_rpc.py:1128-1129        networks          model_rpc.py:60-160
// Here the A agents are the batch dimension of every GEMM, Q1 | Q2 and mean | log_std share one
// weight matrix along the output dimension (so dL/d(input) of the twin critics sums itself inside
// one GEMM), and the backward passes are written out by hand:
//
//   gather   XA = [x | a], XP = [x | .], X2 = [x' | .], r, mask        one kernel, replay rows -> HBM
//   critic   h' = policy(x') -> a', log pi(a')                          L + 1 GEMMs + sample
//            target = r + mask gamma (min Qt(x', a') - alpha log pi)    1 GEMM + head kernel
//            dq = 2 (Q(x, a) - target) / B                              1 GEMM + (same) head kernel
//            dWout, dbout, dbin, dh (ReLU)                              one reduction kernel
//            dWin = XA^T dh                                             1 GEMM
//            Adam + target <- (1 - tau) target + tau critic             one kernel over the flat buffer
//   actor    pi = policy(x), sampled; Q(x, pi) with the new critic      L + 2 GEMMs + sample
//            dq = -[argmin] / B, dh (ReLU)                              head kernel
//            dpi = dh Win[action rows]^T                                1 GEMM
//            d(mean | log_std) through tanh / log-prob / clamp          one kernel
//            policy backward                                            2 L + 1 GEMMs + L + 1 column sums
//            Adam                                                       one kernel
//   alpha    dlog_alpha = -mean(log pi + target entropy), Adam, losses  one kernel
//
// Round 5: the products run on k_gemm_g (aomarl_gemm_g.h: k_gemm_p's inner loop, the agent as the group, operands
// in either orientation so that forward, input gradient and weight gradient all read the tensors as they lie); the
// policy runs ONCE on the stacked rows [x' ; x] (the critic phase's pi(x') and the actor phase's pi(x) use the same
// weights: one product of 2 B rows per layer instead of two of B); bias + ReLU, the ReLU mask of the input gradients
// and the bias gradients (column sums) are epilogues of the products; products that do not depend on each other share
// a launch (k_gemm_g_multi) and the temperature update that of the policy's Adam: ONE stream, no events, 10 product
// launches + 8 small kernels for the reference's two-layer actor, no host synchronisation, no allocation; everything between the replay ring and the updated parameters stays in HBM.  Shapes whose rows are not
// 16-byte aligned (hidden or 2 x act_max not a multiple of 4) take round 1's general kernel, k_gemm_batched_gen.
#include "aomarl_host.h"
#include "aomarl_gemm_g.h"
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>

#define SAC_MAX_HIDDEN 8
#define SAC_EPSILON 1e-5f                 // model_rpc.py:8

struct aomarl_sac {
  aomarl_sac_desc d;
  int A, B, I, Na, H, Hc, L, NA, ldx, ldhd, ldna;
  long long poff[2 * SAC_MAX_HIDDEN + 2], coff[4], plen, clen;
  int32_t *sg = nullptr, *ag = nullptr, *nact = nullptr;
  float *te = nullptr;
  // XS = [x' ; x] stacked per agent ([A][2 B][ldx]: rows 0 .. B-1 the critic phase's next states, rows B .. 2B-1 the
  // actor phase's states; the sampled actions land in their action columns); act / HD: the policy's activations on XS
  float *XA, *XS, *act[SAC_MAX_HIDDEN], *HD, *HQ, *HT, *DQ, *R, *MK, *LP2,
      *LPI, *SQ, *PL, *DPI, *DHD, *dA[SAC_MAX_HIDDEN], *gP, *gC, *gLA;   // gP / gC / gLA: the caller's buffers
  bool aligned = false;                   // every product's rows start on 16 bytes: k_gemm_g
  std::vector<void *> owned;
};

static long long sac_up4(long long v) { return (v + 3) & ~3LL; }

int aomarl_sac_layout(const aomarl_sac_desc *d, long long *policy_off, long long *critic_off,
                      long long *policy_len, long long *critic_len) {
  if (!d) return fail("sac_layout: null descriptor");
  if (d->n_agents < 1 || d->batch < 1 || d->in_max < 1 || d->act_max < 1 || d->hidden < 1 || d->hidden_critic < 1)
    return fail("sac_layout: sizes must be positive");
  if (d->n_hidden < 1 || d->n_hidden > SAC_MAX_HIDDEN) return fail("sac_layout: 1..%d hidden layers", SAC_MAX_HIDDEN);
  const long long A = d->n_agents, I = d->in_max, Na = d->act_max, H = d->hidden, Hc = d->hidden_critic;
  long long o = 0;
  int n = 0;
  auto put = [&](long long *dst, long long len) { if (dst) dst[n] = o; n++; o = sac_up4(o + len); };
  put(policy_off, A * I * H);
  put(policy_off, A * H);
  for (int l = 1; l < d->n_hidden; l++) { put(policy_off, A * H * H); put(policy_off, A * H); }
  put(policy_off, A * H * 2 * Na);
  put(policy_off, A * 2 * Na);
  if (policy_len) *policy_len = o;
  o = 0; n = 0;
  put(critic_off, A * (I + Na) * 2 * Hc);
  put(critic_off, A * 2 * Hc);
  put(critic_off, A * 2 * Hc);
  put(critic_off, A * 2);
  if (critic_len) *critic_len = o;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sac_gather(
    int B, int I, int Na, int ldx, int S, int AD, int A, const int32_t *__restrict__ sg,
    const int32_t *__restrict__ ag, const float *__restrict__ state, const float *__restrict__ next_state,
    const float *__restrict__ action, const float *__restrict__ reward, const float *__restrict__ mask,
    long long rows, const int64_t *__restrict__ idx, uint32_t seed, uint32_t counter,
    float *__restrict__ XA, float *__restrict__ XS, float *__restrict__ R, float *__restrict__ MK) {
  const int b = blockIdx.x, a = blockIdx.y;
  long long r;
  if (idx) {
    r = idx[(long long)a * B + b];
  } else {
    uint32_t x[4];
    philox4x32_10((uint32_t)(a * B + b), counter, 0u, 11u, seed, 0x414F4D52u, x);
    r = (long long)(((unsigned long long)x[0] * (unsigned long long)rows) >> 32);
  }
  r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
  const float *s = state + r * S, *s2 = next_state + r * S, *ac = action + r * AD;
  const long long o = ((long long)a * B + b) * ldx;
  const long long o2 = ((long long)a * 2 * B + b) * ldx, op = o2 + (long long)B * ldx;
  for (int c = threadIdx.x; c < I; c += 256) {
    const int g = sg[a * I + c];
    const bool on = g >= 0 && g < S;
    const float v = on ? s[g] : 0.f;
    XA[o + c] = v;
    XS[op + c] = v;
    XS[o2 + c] = on ? s2[g] : 0.f;
  }
  for (int c = threadIdx.x; c < Na; c += 256) {
    const int g = ag[a * Na + c];
    XA[o + I + c] = (g >= 0 && g < AD) ? ac[g] : 0.f;
  }
  if (threadIdx.x == 0) {
    R[(long long)a * B + b] = reward[r * A + a];
    MK[(long long)a * B + b] = mask[r];
  }
}

// GaussianPolicy.sample (model_rpc.py:140-160) on the stacked rows [x' ; x]: one wave per (agent, row); rows of the
// first half draw from random stream 12 / eps_next and log into LP2, rows of the second half from stream 13 / eps_pi
// into LPI; the action goes straight into the action columns of the row (the critics' input).
__global__ __launch_bounds__(256) void k_sac_sample(
    int B, int I, int Na, int ldx, int ldhd, const int32_t *__restrict__ nact, const float *__restrict__ HD,
    const float *__restrict__ eps_next, const float *__restrict__ eps_pi, uint32_t seed, uint32_t counter,
    float ls_min, float ls_max, float scale, float bias, float *__restrict__ XS, float *__restrict__ LP2,
    float *__restrict__ LPI) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6), a = blockIdx.y;
  if (row >= 2 * B) return;
  const int half = row >= B ? 1 : 0;
  const int na = nact[a];
  const long long ar = (long long)a * B + (row - half * B), sr = (long long)a * 2 * B + row;
  const float *hd = HD + sr * ldhd;
  const float *eps_in = half ? eps_pi : eps_next;
  const uint32_t stream = half ? 13u : 12u;
  float lp = 0.f;
  for (int j = lane; j < Na; j += 64) {
    float act = 0.f;
    if (j < na) {
      const float m = hd[j], ls = fminf(fmaxf(hd[Na + j], ls_min), ls_max);
      const float eps = eps_in ? eps_in[ar * Na + j] : philox_normal(seed, stream, counter, (uint32_t)ar, (uint32_t)j);
      const float y = tanhf(m + expf(ls) * eps);
      act = y * scale + bias;
      lp += -0.5f * eps * eps - ls - 0.91893853320467274178f -
            logf(scale * (1.f - fminf(y * y, 1.f)) + SAC_EPSILON);
    }
    XS[sr * ldx + I + j] = act;
  }
  lp = wave_sum(lp);
  if (lane == 0) (half ? LPI : LP2)[ar] = lp;
}

// q_k = h[kH:(k+1)H] . Wout_k + bout_k for one row; valid in every lane
__device__ __forceinline__ void sac_q_pair(const float *__restrict__ h, const float *__restrict__ w,
                                           const float *__restrict__ bo, int H, int lane, float &q0,
                                           float &q1) {
  float s0 = 0.f, s1 = 0.f;
  for (int j = lane; j < H; j += 64) {
    s0 += h[j] * w[j];
    s1 += h[H + j] * w[H + j];
  }
  q0 = wave_sum(s0) + bo[0];
  q1 = wave_sum(s1) + bo[1];
}

// critic loss head (train_rpc.py:1000-1023): target from the target critic's hidden layer HT, the
// critic's own outputs from HQ, dq = d(mse_1 + mse_2)/dq and the squared errors for the log.
__global__ __launch_bounds__(256) void k_sac_critic_head(
    int B, int H, const float *__restrict__ HQ, const float *__restrict__ HT, const float *__restrict__ Wout,
    const float *__restrict__ bout, const float *__restrict__ WoutT, const float *__restrict__ boutT,
    const float *__restrict__ R, const float *__restrict__ MK, const float *__restrict__ LP2,
    const float *__restrict__ alpha, float gamma, float *__restrict__ DQ, float *__restrict__ SQ) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6), a = blockIdx.y;
  if (row >= B) return;
  const long long ar = (long long)a * B + row;
  float t0, t1, q0, q1;
  sac_q_pair(HT + ar * 2 * H, WoutT + (long long)a * 2 * H, boutT + a * 2, H, lane, t0, t1);
  sac_q_pair(HQ + ar * 2 * H, Wout + (long long)a * 2 * H, bout + a * 2, H, lane, q0, q1);
  if (lane == 0) {
    const float target = R[ar] + MK[ar] * gamma * (fminf(t0, t1) - alpha[a] * LP2[ar]);
    const float d0 = q0 - target, d1 = q1 - target, s = 2.f / (float)B;
    DQ[ar * 2] = s * d0; DQ[ar * 2 + 1] = s * d1;
    SQ[ar * 2] = d0 * d0; SQ[ar * 2 + 1] = d1 * d1;
  }
}

// output-layer backward of the twin critics: 32 columns x 32 row groups per block.
// in : HQ = relu hidden [A][B][2H], DQ [A][B][2]
// out: HQ <- dL/d(pre-activation) = dq_k Wout_k [h > 0];  gWout, gbout, gbin
__global__ __launch_bounds__(1024) void k_sac_q_out_bwd(int B, int H, float *__restrict__ HQ,
                                                        const float *__restrict__ DQ,
                                                        const float *__restrict__ Wout,
                                                        float *__restrict__ gWout, float *__restrict__ gbout,
                                                        float *__restrict__ gbin) {
  __shared__ float sW[32][33], sB[32][33], sO[32][33];
  const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5, a = blockIdx.y;
  const int col = blockIdx.x * 32 + cx;
  const bool valid = col < 2 * H;
  const int k = (valid && col >= H) ? 1 : 0, j = col - k * H;
  const float w = valid ? Wout[(long long)a * 2 * H + col] : 0.f;
  float accW = 0.f, accB = 0.f, accO = 0.f;
  if (valid)
    for (int b = rg; b < B; b += 32) {
      const long long ar = (long long)a * B + b;
      const float hv = HQ[ar * 2 * H + col], dq = DQ[ar * 2 + k];
      accW += hv * dq;
      const float g = hv > 0.f ? dq * w : 0.f;
      HQ[ar * 2 * H + col] = g;
      accB += g;
      if (j == 0) accO += dq;
    }
  sW[rg][cx] = accW; sB[rg][cx] = accB; sO[rg][cx] = accO;
  __syncthreads();
  if (rg == 0 && valid) {
    float tw = 0.f, tb = 0.f, to = 0.f;
    for (int r = 0; r < 32; r++) { tw += sW[r][cx]; tb += sB[r][cx]; to += sO[r][cx]; }
    gWout[(long long)a * 2 * H + col] = tw;
    gbin[(long long)a * 2 * H + col] = tb;
    if (j == 0) gbout[a * 2 + k] = to;
  }
}

// actor loss head (train_rpc.py:1049-1055): loss = mean(alpha log pi - min(Q1, Q2)); the gradient of
// torch.min goes to the smaller output (half each on a tie); HQ <- dL/d(pre-activation)
__global__ __launch_bounds__(256) void k_sac_actor_head(int B, int H, float *__restrict__ HQ,
                                                        const float *__restrict__ Wout,
                                                        const float *__restrict__ bout,
                                                        const float *__restrict__ LPI,
                                                        const float *__restrict__ alpha, float *__restrict__ PL) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6), a = blockIdx.y;
  if (row >= B) return;
  const long long ar = (long long)a * B + row;
  float *h = HQ + ar * 2 * H;
  const float *w = Wout + (long long)a * 2 * H;
  float q0, q1;
  sac_q_pair(h, w, bout + a * 2, H, lane, q0, q1);
  const float invB = 1.f / (float)B;
  const float d0 = -(q0 < q1 ? 1.f : (q0 > q1 ? 0.f : 0.5f)) * invB;
  const float d1 = -(q1 < q0 ? 1.f : (q1 > q0 ? 0.f : 0.5f)) * invB;
  if (lane == 0) PL[ar] = alpha[a] * LPI[ar] - fminf(q0, q1);
  for (int c = lane; c < 2 * H; c += 64) {
    const float hv = h[c];
    h[c] = hv > 0.f ? (c < H ? d0 : d1) * w[c] : 0.f;
  }
}

// backward of GaussianPolicy.sample + the alpha log pi term of the actor loss.
// DPI = dL/d(action) from the critics; out DHD = dL/d(mean | log_std) (pre-clamp)
__global__ __launch_bounds__(256) void k_sac_sample_bwd(
    int B, int Na, int ldhd, long long sHD, int ldna, const int32_t *__restrict__ nact, const float *__restrict__ HD,
    const float *__restrict__ DPI, const float *__restrict__ eps_in, uint32_t seed, uint32_t counter,
    uint32_t stream, const float *__restrict__ alpha, float ls_min, float ls_max, float scale,
    float *__restrict__ DHD) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6), a = blockIdx.y;
  if (row >= B) return;
  const int na = nact[a];
  const long long ar = (long long)a * B + row;
  const float *hd = HD + (long long)a * sHD + (long long)row * ldhd;
  const float aB = alpha[a] / (float)B;
  for (int j = lane; j < Na; j += 64) {
    float dm = 0.f, dls = 0.f;
    if (j < na) {
      const float m = hd[j], lr = hd[Na + j], ls = fminf(fmaxf(lr, ls_min), ls_max);
      const float eps = eps_in ? eps_in[ar * Na + j] : philox_normal(seed, stream, counter, (uint32_t)ar, (uint32_t)j);
      const float sd = expf(ls), y = tanhf(m + sd * eps);
      const float y2 = y * y, om = 1.f - y2;             // d tanh / dx
      // action = scale y + bias;  log pi has -log(scale (1 - clamp(y^2, 0, 1)) + EPS)
      float gx = DPI[ar * ldna + j] * scale * om;
      if (y2 <= 1.f) gx += aB * (2.f * scale * y * om) / (scale * om + SAC_EPSILON);
      dm = gx;
      dls = (lr >= ls_min && lr <= ls_max) ? gx * sd * eps - aB : 0.f;
    }
    DHD[ar * ldhd + j] = dm;
    DHD[ar * ldhd + Na + j] = dls;
  }
}

// out[a][c] = sum_b X[a][b][c]      (bias gradients)
__global__ __launch_bounds__(1024) void k_sac_colsum(int B, int N, int ld, long long sX,
                                                     const float *__restrict__ X, float *__restrict__ out,
                                                     long long sOut) {
  __shared__ float sm[32][33];
  const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5, a = blockIdx.y;
  const int col = blockIdx.x * 32 + cx;
  float acc = 0.f;
  if (col < N)
    for (int b = rg; b < B; b += 32) acc += X[(long long)a * sX + (long long)b * ld + col];
  sm[rg][cx] = acc;
  __syncthreads();
  if (rg == 0 && col < N) {
    float t = 0.f;
    for (int r = 0; r < 32; r++) t += sm[r][cx];
    out[(long long)a * sOut + col] = t;
  }
}

struct AdamK { float step_size, inv_sqrt_bc2, b1, b2, eps; };

__device__ __forceinline__ float adam_one(float p, float g, float &m, float &v, const AdamK k) {
  // torch.optim.Adam (amsgrad off, no weight decay): _single_tensor_adam
  m += (g - m) * (1.f - k.b1);
  v = v * k.b2 + (1.f - k.b2) * g * g;
  const float denom = sqrtf(v) * k.inv_sqrt_bc2 + k.eps;
  return p - k.step_size * (m / denom);
}

// update_alpha (train_rpc.py:1070-1084) + the per-agent loss log; one block per agent
__device__ __forceinline__ void sac_alpha_block(int a, int B, int A, const float *__restrict__ LPI,
                                                const float *__restrict__ SQ, const float *__restrict__ PL,
                                                const float *__restrict__ te, float *__restrict__ la,
                                                float *__restrict__ la_m, float *__restrict__ la_v,
                                                float *__restrict__ alpha, float *__restrict__ gla, AdamK k,
                                                int tune, float *__restrict__ losses) {
  __shared__ float sm[4][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < B; b += 256) {
    const long long ar = (long long)a * B + b;
    s[0] += LPI[ar]; s[1] += SQ[ar * 2]; s[2] += SQ[ar * 2 + 1]; s[3] += PL[ar];
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    s[i] = wave_sum(s[i]);
    if (lane == 0) sm[wv][i] = s[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[4];
    for (int i = 0; i < 4; i++) t[i] = (sm[0][i] + sm[1][i]) + (sm[2][i] + sm[3][i]);
    const float invB = 1.f / (float)B;
    float al = 0.f;
    if (tune) {
      const float ent = t[0] * invB + te[a];         // mean(log pi + target entropy)
      const float l = la[a];
      al = -l * ent;
      const float g = -ent;
      float m = la_m[a], v = la_v[a];
      const float ln = adam_one(l, g, m, v, k);
      la[a] = ln; la_m[a] = m; la_v[a] = v; gla[a] = g;
      alpha[a] = expf(ln);
    }
    if (losses) {
      losses[a] = t[1] * invB;
      losses[A + a] = t[2] * invB;
      losses[2 * A + a] = t[3] * invB;
      losses[3 * A + a] = al;
      losses[4 * A + a] = alpha[a];
    }
  }
}

struct SacAlphaArgs {
  int on, B, A, tune;
  const float *LPI, *SQ, *PL, *te;
  float *la, *la_m, *la_v, *alpha, *gla, *losses;
};

// torch.optim.Adam over a flat parameter buffer (+ soft_update of the target, train_rpc.py:1128-1129); with
// `al.on` the grid carries A more blocks that run the temperature update (nothing of it depends on this Adam).
__global__ __launch_bounds__(256) void k_sac_adam(long long n4, float4 *__restrict__ p,
                                                  const float4 *__restrict__ g, float4 *__restrict__ m,
                                                  float4 *__restrict__ v, float4 *__restrict__ tgt, AdamK k,
                                                  float tau, unsigned nb, SacAlphaArgs al) {
  if (blockIdx.x >= nb) {                            // block-uniform
    if (al.on)
      sac_alpha_block((int)(blockIdx.x - nb), al.B, al.A, al.LPI, al.SQ, al.PL, al.te, al.la, al.la_m, al.la_v,
                      al.alpha, al.gla, k, al.tune, al.losses);
    return;
  }
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 P = p[i], M = m[i], V = v[i];
  const float4 G = g[i];
  P.x = adam_one(P.x, G.x, M.x, V.x, k); P.y = adam_one(P.y, G.y, M.y, V.y, k);
  P.z = adam_one(P.z, G.z, M.z, V.z, k); P.w = adam_one(P.w, G.w, M.w, V.w, k);
  p[i] = P; m[i] = M; v[i] = V;
  if (tgt) {                                         // soft_update, train_rpc.py:1128-1129
    float4 T = tgt[i];
    T.x = T.x * (1.f - tau) + P.x * tau; T.y = T.y * (1.f - tau) + P.y * tau;
    T.z = T.z * (1.f - tau) + P.z * tau; T.w = T.w * (1.f - tau) + P.w * tau;
    tgt[i] = T;
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int aomarl_sac_destroy(aomarl_sac *s) {
  if (!s) return 0;
  for (void *p : s->owned) (void)hipFree(p);
  delete s;
  return 0;
}

int aomarl_sac_create(const aomarl_sac_desc *d, aomarl_sac **out) {
  if (!d || !out) return fail("sac_create: null pointer");
  *out = nullptr;
  aomarl_sac *s = new aomarl_sac;
  s->d = *d;
  if (aomarl_sac_layout(d, s->poff, s->coff, &s->plen, &s->clen)) { delete s; return 1; }
  if (!d->state_gather || !d->action_gather || !d->n_act || !d->target_entropy || !d->policy ||
      !d->policy_m || !d->policy_v || !d->critic || !d->critic_m || !d->critic_v || !d->critic_target ||
      !d->log_alpha || !d->log_alpha_m || !d->log_alpha_v || !d->alpha || !d->policy_grad ||
      !d->critic_grad || !d->log_alpha_grad) {
    delete s;
    return fail("sac_create: null buffer in the descriptor");
  }
  if (d->state_dim < 1 || d->action_dim < 1) { delete s; return fail("sac_create: replay row sizes must be positive"); }
  for (const float *p : {d->policy, d->policy_m, d->policy_v, d->policy_grad, d->critic, d->critic_m, d->critic_v,
                         d->critic_grad, d->critic_target})
    if ((uintptr_t)p & 15) { delete s; return fail("sac_create: parameter buffers must be 16-byte aligned"); }
  const int A = s->A = d->n_agents, B = s->B = d->batch, I = s->I = d->in_max, Na = s->Na = d->act_max;
  const int H = s->H = d->hidden, Hc = s->Hc = d->hidden_critic;
  s->L = d->n_hidden;
  s->NA = I + Na;
  s->ldx = (int)sac_up4(s->NA); s->ldhd = (int)sac_up4(2 * Na); s->ldna = (int)sac_up4(Na);
  for (int a = 0; a < A; a++)
    if (d->n_act[a] < 0 || d->n_act[a] > Na) { delete s; return fail("sac_create: n_act[%d] outside [0, act_max]", a); }
  bool ok = true;
  auto alloc = [&](size_t n) -> float * {
    void *p = nullptr;
    if (hipMalloc(&p, n * sizeof(float)) != hipSuccess) { ok = false; return nullptr; }
    s->owned.push_back(p);
    if (hipMemset(p, 0, n * sizeof(float)) != hipSuccess) ok = false;
    return (float *)p;
  };
  auto upload = [&](const void *h, size_t bytes) -> void * {
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { ok = false; return nullptr; }
    s->owned.push_back(p);
    if (hipMemcpy(p, h, bytes, hipMemcpyHostToDevice) != hipSuccess) ok = false;
    return p;
  };
  s->sg = (int32_t *)upload(d->state_gather, sizeof(int32_t) * A * I);
  s->ag = (int32_t *)upload(d->action_gather, sizeof(int32_t) * A * Na);
  s->nact = (int32_t *)upload(d->n_act, sizeof(int32_t) * A);
  s->te = (float *)upload(d->target_entropy, sizeof(float) * A);
  const size_t AB = (size_t)A * B;
  s->XA = alloc(AB * s->ldx); s->XS = alloc(2 * AB * s->ldx);
  for (int l = 0; l < SAC_MAX_HIDDEN; l++) {
    s->act[l] = l < s->L ? alloc(2 * AB * H) : nullptr;
    s->dA[l] = l < s->L ? alloc(AB * H) : nullptr;
  }
  s->HD = alloc(2 * AB * s->ldhd); s->DHD = alloc(AB * s->ldhd);
  // k_gemm_g wants every operand row on a 16-byte boundary: the weights' rows are H, 2 Na and 2 Hc floats long
  s->aligned = !(H & 3) && !((2 * Na) & 3) && !((2 * Hc) & 3);
  { const char *e = getenv("AOMARL_SAC_GEMM"); if (e && e[0] == 'g' && e[1] == 'e') s->aligned = false; }   // "gen": round 1's kernel (A/B)
  s->HQ = alloc(AB * 2 * Hc); s->HT = alloc(AB * 2 * Hc);
  s->DQ = alloc(AB * 2); s->SQ = alloc(AB * 2);
  s->R = alloc(AB); s->MK = alloc(AB); s->LP2 = alloc(AB); s->LPI = alloc(AB); s->PL = alloc(AB);
  s->DPI = alloc(AB * s->ldna);
  s->gP = d->policy_grad; s->gC = d->critic_grad; s->gLA = d->log_alpha_grad;
  if (!ok) { aomarl_sac_destroy(s); return fail("sac_create: device allocation failed"); }
  *out = s;
  return 0;
}

static AdamK sac_adam_consts(const aomarl_sac_desc &d, int step) {
  const double bc1 = 1.0 - pow((double)d.beta1, step), bc2 = 1.0 - pow((double)d.beta2, step);
  AdamK k;
  k.step_size = (float)((double)d.lr / bc1);
  k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  k.b1 = d.beta1; k.b2 = d.beta2; k.eps = d.adam_eps;
  return k;
}

// One product of the update for every agent: C[a] = opA(A[a]) . opB(B[a]) (+ bias, ReLU | masked by `mask` > 0), and
// optionally colsum[a][n] = sum_k B[a][k][n] (the bias gradient beside a weight gradient).  tA: A is [K][M]; tB: B is
// [N][K].
struct SacProb { GemmGArgs g; bool ak, bk; };
static SacProb sac_prob(int tA, int tB, int M, int N, int K, const float *A, int lda, long long sA, const float *B,
                        int ldb, long long sB, const float *bias, long long sBias, float *C, int ldc, long long sC,
                        int relu, const float *mask, int ldm, long long sM, float *colsum, long long sCs) {
  SacProb p;
  memset(&p, 0, sizeof(p));
  GemmGArgs &g = p.g;
  g.M = M; g.N = N; g.K = K;
  g.A = A; g.lda = lda; g.sA = sA; g.B = B; g.ldb = ldb; g.sB = sB; g.C = C; g.ldc = ldc; g.sC = sC;
  g.bias = bias; g.sBias = sBias; g.relu = relu; g.mask = mask; g.ldm = ldm; g.sM = sM;
  g.colsum = colsum; g.sCs = sCs;
  p.ak = !tA; p.bk = tB != 0;
  return p;
}

// n independent products: one launch of k_gemm_g_multi; rows that are not 16-byte aligned: round 1's kernel, one
// launch per product (+ the column sums)
static int sac_run(aomarl_sac *s, int n, const SacProb *pr, hipStream_t st) {
  if (s->aligned) {
    GemmGArgs g[GG_MAXP];
    bool ak[GG_MAXP], bk[GG_MAXP];
    for (int i = 0; i < n; i++) { g[i] = pr[i].g; ak[i] = pr[i].ak; bk[i] = pr[i].bk; }
    if (gemm_g_launch_multi(s->A, n, g, ak, bk, st)) return fail("sac_update: k_gemm_g_multi launch failed");
    return 0;
  }
  for (int i = 0; i < n; i++) {
    const GemmGArgs &g = pr[i].g;
    if (gemm_batched_launch(s->A, !pr[i].ak, !pr[i].bk, g.M, g.N, g.K, g.A, g.lda, g.sA, g.B, g.ldb, g.sB, g.bias, g.sBias,
                            g.C, g.ldc, g.sC, g.relu, 0, g.mask, g.ldm, g.sM, st)) return 1;
    if (g.colsum) {
      hipLaunchKernelGGL(k_sac_colsum, dim3((g.N + 31) / 32, s->A), dim3(1024), 0, st, g.K, g.N, g.ldb, g.sB, g.B,
                         g.colsum, g.sCs);
      LAUNCHCHK();
    }
  }
  return 0;
}

// hidden layer of the twin critics on B rows per agent: Hout = relu(X Win + bin), [A][B][2H]
static SacProb sac_critic_hidden(aomarl_sac *s, const float *X, long long sX, const float *C, float *Hout) {
  const int B = s->B, H = s->Hc, NA = s->NA;
  return sac_prob(0, 0, B, 2 * H, NA, X, s->ldx, sX, C + s->coff[0], 2 * H, (long long)NA * 2 * H, C + s->coff[1], 2 * H,
                  Hout, 2 * H, (long long)B * 2 * H, 1, nullptr, 0, 0, nullptr, 0);
}

int aomarl_sac_update(aomarl_sac *s, const float *state, const float *next_state, const float *action,
                      const float *reward, const float *mask, long long replay_rows, const int64_t *idx,
                      const float *eps_next, const float *eps_pi, uint32_t seed, uint32_t counter,
                      int adam_step, int flags, float *losses, void *stream) {
  if (!s) return fail("sac_update: null handle");
  if (!state || !next_state || !action || !reward || !mask) return fail("sac_update: null replay pointer");
  if (replay_rows < 1) return fail("sac_update: empty replay memory");
  if (adam_step < 1) return fail("sac_update: adam_step is 1-based");
  hipStream_t st = (hipStream_t)stream;
  const aomarl_sac_desc &d = s->d;
  const int A = s->A, B = s->B, B2 = 2 * s->B, I = s->I, Na = s->Na, H = s->H, Hc = s->Hc, L = s->L, NA = s->NA;
  const dim3 rows4((B + 3) / 4, A), rows8((B2 + 3) / 4, A), w256(256);
  const AdamK ak = sac_adam_consts(d, adam_step);
  float *C = d.critic, *P = d.policy;
  const long long sH = (long long)B * H, sH2 = 2 * sH;           // dA rows; the stacked activations' agent stride
  const long long sXS = (long long)B2 * s->ldx, sHD2 = (long long)B2 * s->ldhd;
  const float *XP = s->XS + (long long)B * s->ldx;               // the actor phase's half of the stacked rows
  const float *HDP = s->HD + (long long)B * s->ldhd;
  SacAlphaArgs no_alpha;
  memset(&no_alpha, 0, sizeof(no_alpha));
  SacProb pr[GG_MAXP];

  // ONE stream, no events: products that do not depend on each other share a launch (k_gemm_g_multi).
  hipLaunchKernelGGL(k_sac_gather, dim3(B, A), w256, 0, st, B, I, Na, s->ldx, d.state_dim, d.action_dim, A,
                     s->sg, s->ag, state, next_state, action, reward, mask, replay_rows, idx, seed, counter,
                     s->XA, s->XS, s->R, s->MK);
  LAUNCHCHK();
  // ---------------- critic ----------------
  // pi(x') and pi(x) in one pass on the stacked rows (the policy does not change before the end of the update);
  // beside its first layer: the critics' hidden layer on (x, a)
  pr[0] = sac_prob(0, 0, B2, H, I, s->XS, s->ldx, sXS, P + s->poff[0], H, (long long)I * H, P + s->poff[1], H, s->act[0],
                   H, sH2, 1, nullptr, 0, 0, nullptr, 0);
  pr[1] = sac_critic_hidden(s, s->XA, (long long)B * s->ldx, C, s->HQ);
  if (sac_run(s, 2, pr, st)) return 1;
  for (int l = 1; l < L; l++) {
    pr[0] = sac_prob(0, 0, B2, H, H, s->act[l - 1], H, sH2, P + s->poff[2 * l], H, (long long)H * H, P + s->poff[2 * l + 1],
                     H, s->act[l], H, sH2, 1, nullptr, 0, 0, nullptr, 0);
    if (sac_run(s, 1, pr, st)) return 1;
  }
  pr[0] = sac_prob(0, 0, B2, 2 * Na, H, s->act[L - 1], H, sH2, P + s->poff[2 * L], 2 * Na, (long long)H * 2 * Na,
                   P + s->poff[2 * L + 1], 2 * Na, s->HD, s->ldhd, sHD2, 0, nullptr, 0, 0, nullptr, 0);
  if (sac_run(s, 1, pr, st)) return 1;
  hipLaunchKernelGGL(k_sac_sample, rows8, w256, 0, st, B, I, Na, s->ldx, s->ldhd, s->nact, s->HD, eps_next, eps_pi,
                     seed, counter, d.log_sig_min, d.log_sig_max, d.action_scale, d.action_bias, s->XS, s->LP2, s->LPI);
  LAUNCHCHK();
  pr[0] = sac_critic_hidden(s, s->XS, sXS, d.critic_target, s->HT);
  if (sac_run(s, 1, pr, st)) return 1;
  hipLaunchKernelGGL(k_sac_critic_head, rows4, w256, 0, st, B, Hc, s->HQ, s->HT, C + s->coff[2], C + s->coff[3],
                     d.critic_target + s->coff[2], d.critic_target + s->coff[3], s->R, s->MK, s->LP2, d.alpha,
                     d.gamma, s->DQ, s->SQ);
  LAUNCHCHK();
  hipLaunchKernelGGL(k_sac_q_out_bwd, dim3((2 * Hc + 31) / 32, A), dim3(1024), 0, st, B, Hc, s->HQ, s->DQ,
                     C + s->coff[2], s->gC + s->coff[2], s->gC + s->coff[3], s->gC + s->coff[1]);
  LAUNCHCHK();
  // dWin = XA^T dh
  pr[0] = sac_prob(1, 0, NA, 2 * Hc, B, s->XA, s->ldx, (long long)B * s->ldx, s->HQ, 2 * Hc, (long long)B * 2 * Hc, nullptr,
                   0, s->gC + s->coff[0], 2 * Hc, (long long)NA * 2 * Hc, 0, nullptr, 0, 0, nullptr, 0);
  if (sac_run(s, 1, pr, st)) return 1;
  {
    const unsigned nb = (unsigned)((s->clen / 4 + 255) / 256);
    hipLaunchKernelGGL(k_sac_adam, dim3(nb), w256, 0, st, s->clen / 4, (float4 *)C, (const float4 *)s->gC,
                       (float4 *)d.critic_m, (float4 *)d.critic_v,
                       (flags & AOMARL_SAC_SOFT_UPDATE) ? (float4 *)d.critic_target : (float4 *)nullptr, ak, d.tau, nb,
                       no_alpha);
    LAUNCHCHK();
  }
  // ---------------- actor ----------------
  pr[0] = sac_critic_hidden(s, XP, sXS, C, s->HQ);
  if (sac_run(s, 1, pr, st)) return 1;
  hipLaunchKernelGGL(k_sac_actor_head, rows4, w256, 0, st, B, Hc, s->HQ, C + s->coff[2], C + s->coff[3], s->LPI,
                     d.alpha, s->PL);
  LAUNCHCHK();
  // dpi = dh Win[action rows]^T   (sums the two critics: K runs over both hidden halves)
  pr[0] = sac_prob(0, 1, B, Na, 2 * Hc, s->HQ, 2 * Hc, (long long)B * 2 * Hc, C + s->coff[0] + (long long)I * 2 * Hc, 2 * Hc,
                   (long long)NA * 2 * Hc, nullptr, 0, s->DPI, s->ldna, (long long)B * s->ldna, 0, nullptr, 0, 0, nullptr, 0);
  if (sac_run(s, 1, pr, st)) return 1;
  hipLaunchKernelGGL(k_sac_sample_bwd, rows4, w256, 0, st, B, Na, s->ldhd, sHD2, s->ldna, s->nact, HDP, s->DPI,
                     eps_pi, seed, counter, 13u, d.alpha, d.log_sig_min, d.log_sig_max, d.action_scale, s->DHD);
  LAUNCHCHK();
  // policy backward: a layer's input gradient (ReLU-masked) and its weight + bias gradient share a launch
  const float *actP[SAC_MAX_HIDDEN];                              // the x half of the stacked activations
  for (int l = 0; l < L; l++) actP[l] = s->act[l] + sH;
  pr[0] = sac_prob(0, 1, B, H, 2 * Na, s->DHD, s->ldhd, (long long)B * s->ldhd, P + s->poff[2 * L], 2 * Na,
                   (long long)H * 2 * Na, nullptr, 0, s->dA[L - 1], H, sH, 0, actP[L - 1], H, sH2, nullptr, 0);
  pr[1] = sac_prob(1, 0, H, 2 * Na, B, actP[L - 1], H, sH2, s->DHD, s->ldhd, (long long)B * s->ldhd, nullptr, 0,
                   s->gP + s->poff[2 * L], 2 * Na, (long long)H * 2 * Na, 0, nullptr, 0, 0, s->gP + s->poff[2 * L + 1],
                   2 * Na);
  if (sac_run(s, 2, pr, st)) return 1;
  for (int l = L - 1; l >= 1; l--) {
    pr[0] = sac_prob(0, 1, B, H, H, s->dA[l], H, sH, P + s->poff[2 * l], H, (long long)H * H, nullptr, 0, s->dA[l - 1], H,
                     sH, 0, actP[l - 1], H, sH2, nullptr, 0);
    pr[1] = sac_prob(1, 0, H, H, B, actP[l - 1], H, sH2, s->dA[l], H, sH, nullptr, 0, s->gP + s->poff[2 * l], H,
                     (long long)H * H, 0, nullptr, 0, 0, s->gP + s->poff[2 * l + 1], H);
    if (sac_run(s, 2, pr, st)) return 1;
  }
  pr[0] = sac_prob(1, 0, I, H, B, XP, s->ldx, sXS, s->dA[0], H, sH, nullptr, 0, s->gP + s->poff[0], H, (long long)I * H, 0,
                   nullptr, 0, 0, s->gP + s->poff[1], H);
  if (sac_run(s, 1, pr, st)) return 1;
  // ---------------- policy Adam; temperature + log in the same launch ----------------
  {
    const unsigned nb = (unsigned)((s->plen / 4 + 255) / 256);
    SacAlphaArgs al;
    al.on = 1; al.B = B; al.A = A; al.tune = (flags & AOMARL_SAC_TUNE_ALPHA) ? 1 : 0;
    al.LPI = s->LPI; al.SQ = s->SQ; al.PL = s->PL; al.te = s->te;
    al.la = d.log_alpha; al.la_m = d.log_alpha_m; al.la_v = d.log_alpha_v; al.alpha = d.alpha; al.gla = s->gLA;
    al.losses = losses;
    hipLaunchKernelGGL(k_sac_adam, dim3(nb + (unsigned)A), w256, 0, st, s->plen / 4, (float4 *)P, (const float4 *)s->gP,
                       (float4 *)d.policy_m, (float4 *)d.policy_v, (float4 *)nullptr, ak, 0.f, nb, al);
    LAUNCHCHK();
  }
  return 0;
}

// ---------------------------------------------------------------- the grouped kernel for the C ABI's batched products
int gemm_g_batched(int batch, bool ak, bool bk, int M, int N, int K, const float *A, int lda, long long sA,
                   const float *B, int ldb, long long sB, const float *bias, long long sBias, float *C, int ldc,
                   long long sC, int relu, hipStream_t s) {
  GemmGArgs g;
  memset(&g, 0, sizeof(g));
  g.M = M; g.N = N; g.K = K;
  g.A = A; g.lda = lda; g.sA = sA; g.B = B; g.ldb = ldb; g.sB = sB; g.C = C; g.ldc = ldc; g.sC = sC;
  g.bias = bias; g.sBias = sBias; g.relu = relu;
  if (gemm_g_launch(batch, ak, bk, g, 0, 0, s)) return fail("gemm_batched: k_gemm_g launch failed");
  return 0;
}
