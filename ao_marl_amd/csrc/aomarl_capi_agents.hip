// aomarl_capi_agents.hip -- part of the C ABI implementation (included by aomarl_capi.hip, one translation unit):
// agent-side glue: state split / assembly, policy sampling, per-agent rewards, the fused actor (A12 - A15).
// ---------------------------------------------------------------- agent-side glue (A12 - A15)
// The reference does these in NumPy / torch on the host, a handful of tiny operations per agent
// per step; on the device each of them would be its own ~5 us launch, so the chains are fused.
__global__ void k_split_states(int nenv, int state_dim, int in_max, const int32_t *__restrict__ gather,
                               const float *__restrict__ state, float *__restrict__ out) {
  // out[a][e][k] = state[e][gather[a][k]]  (gather == state_dim -> 0: padding)
  const int a = blockIdx.z, e = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= in_max) return;
  const int g = gather[a * in_max + k];
  out[((long long)a * nenv + e) * in_max + k] = g < state_dim ? state[(long long)e * state_dim + g] : 0.f;
}

__global__ void k_policy_sample(int nenv, int act_max, int action_dim, const float *__restrict__ head,
                                float ls_min, float ls_max, float scale, float bias,
                                const int32_t *__restrict__ sc_agent, const int32_t *__restrict__ sc_local,
                                const float *__restrict__ eps_in, uint32_t seed, uint32_t counter,
                                float *__restrict__ action, float *__restrict__ mean) {
  const int e = blockIdx.y, g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= action_dim) return;
  const int a = sc_agent[g], l = sc_local[g];
  const float *h = head + ((long long)a * nenv + e) * (2 * act_max);
  const float m = h[l];
  const float ls = fminf(fmaxf(h[act_max + l], ls_min), ls_max);
  const float eps = eps_in ? eps_in[(long long)e * action_dim + g]
                           : philox_normal(seed, 7u, counter, (uint32_t)e, (uint32_t)g);
  const float x = m + expf(ls) * eps;
  action[(long long)e * action_dim + g] = tanhf(x) * scale + bias;
  mean[(long long)e * action_dim + g] = tanhf(m) * scale + bias;
}

// ---- the whole actor in one launch -----------------------------------------------------------
// One workgroup = one agent x 16 environments: gather the agent's state columns into LDS, run the
// Linear + ReLU stack and the merged head with the activations staying in LDS (fp32 matrix
// instructions, 16 x 16 x 4), then clamp / exp / sample / tanh / scatter.  Replaces k_split_states +
// (n_hidden + 1) k_gemm_nt_batched2 + k_policy_sample: launch-latency-bound kernels of ~0.3-1 GFLOP.
//
// Every workgroup streams its agent's ~1 MB of weights from L2 (each agent's workgroups sit on one
// XCD, so HBM sees them once); what bounds the kernel is the number of cache lines a load instruction
// touches, so the weights come PRE-TILED in the operand order of the matrix instruction
// (aomarl_actor_tile_weights): tile (n, s) = rows 16 n .. 16 n + 15, columns 16 s .. 16 s + 15, stored as
// 64 x float4 with lane l = (row l & 15, columns 4 (l >> 4) .. + 3) -- one 1 KB contiguous read per
// wave and k step.  Read row-major, the same loads touch 64 lines instead of 8 and the kernel runs at
// half the speed.  The activations use the same tiling in LDS (conflict-free 128-bit reads).
struct ActorArgs {
  int A, nenv, state_dim, in_max, act_max, H, n_hidden, action_dim;
  const int32_t *gather;
  const float *W1, *b1, *Wh[8], *bh[8], *Whead, *bhead;      // W*: tiled
  const int32_t *sc_agent, *sc_local;
  float ls_min, ls_max, scale, bias;
  const float *state, *eps;
  uint32_t seed, counter;
  float *action, *mean;
};

__global__ void k_actor_tile_weights(int N, int K, int ntile, int ksteps, const float *__restrict__ src,
                                     float *__restrict__ dst) {
  // dst[a][n][s][lane][j] = src[a][16 n + (lane & 15)][16 s + 4 (lane >> 4) + j], zero outside N x K
  const long long per = (long long)ntile * ksteps * 256;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per) return;
  const int a = blockIdx.y;
  const int jj = (int)(i & 3), lane = (int)((i >> 2) & 63);
  const long long t = i >> 8;
  const int sidx = (int)(t % ksteps), n = (int)(t / ksteps);
  const int row = 16 * n + (lane & 15), col = 16 * sidx + 4 * (lane >> 4) + jj;
  dst[(long long)a * per + i] = (row < N && col < K) ? src[((long long)a * N + row) * K + col] : 0.f;
}

// position of element (row, col) of a 16-row activation tile in its LDS image
__device__ __forceinline__ int af_at(int row, int col) {
  return (((col >> 4) * 64 + ((col >> 2) & 3) * 16 + row) << 2) + (col & 3);
}

#ifndef AF_D
#define AF_D 2
#endif
// Out[16][N] = act(Xs[16][K] . W^T + b) on tiled images; ksteps = ceil(K / 16), W has ntile row tiles.
__device__ __forceinline__ void af_layer(const float *__restrict__ Xs, int ksteps, const float *__restrict__ W,
                                         const float *__restrict__ b, int N, bool relu, float *__restrict__ Out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kk = lane >> 4;
  const int ntile = (N + 15) / 16, npairs = (ntile + 1) / 2;
  constexpr int D = AF_D;
  for (int pair = wave; pair < npairs; pair += 8) {
    const float4 *wa = reinterpret_cast<const float4 *>(W) + (long long)(2 * pair) * ksteps * 64 + lane;
    const float4 *wb = reinterpret_cast<const float4 *>(W) + (long long)min(2 * pair + 1, ntile - 1) * ksteps * 64 + lane;
    const float4 *xs = reinterpret_cast<const float4 *>(Xs) + lane;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    auto fma8 = [&](const float4 x, const float4 a, const float4 c) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, a.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, c.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, a.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, c.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, a.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, c.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, a.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, c.w, acc1, 0, 0, 0);
    };
    // two register sets of D steps each, filled and drained in turn.  No branch around a load, no
    // select on its result, no rotation of the sets: each of those makes the compiler wait for the data
    // where it is loaded; and scheduling barriers, or it sinks every load to just before its use.
    float4 ra[D], rb[D], qa[D], qb[D];
    auto fill = [&](float4 (&a)[D], float4 (&c)[D], int s0) {
#pragma unroll
      for (int u = 0; u < D; u++) {
        const int st = min(s0 + u, ksteps - 1) * 64;       // wave-uniform; beyond the end: any tile, unused
        a[u] = wa[st]; c[u] = wb[st];
      }
    };
    auto drain = [&](const float4 (&a)[D], const float4 (&c)[D], int s0) {
#pragma unroll
      for (int u = 0; u < D; u++) fma8(xs[(s0 + u) * 64], a[u], c[u]);
    };
    fill(ra, rb, 0);
    int s = 0;
    for (; s + 2 * D <= ksteps; s += 2 * D) {
      fill(qa, qb, s + D);
      __builtin_amdgcn_sched_barrier(0);
      drain(ra, rb, s);
      __builtin_amdgcn_sched_barrier(0);
      fill(ra, rb, s + 2 * D);
      __builtin_amdgcn_sched_barrier(0);
      drain(qa, qb, s + D);
      __builtin_amdgcn_sched_barrier(0);
    }
    fill(qa, qb, s + D);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < D; u++)
      if (s + u < ksteps) fma8(xs[(s + u) * 64], ra[u], rb[u]);
#pragma unroll
    for (int u = 0; u < D; u++)
      if (s + D + u < ksteps) fma8(xs[(s + D + u) * 64], qa[u], qb[u]);
    // C layout: register t of lane l = row 4 (l >> 4) + t, column l & 15
    const int ca = 32 * pair + r, cb = ca + 16;
    const float ba = (b && ca < N) ? b[ca] : 0.f, bb = (b && cb < N) ? b[cb] : 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      float va = acc0[t] + ba, vb = acc1[t] + bb;
      if (relu) { va = fmaxf(va, 0.f); vb = fmaxf(vb, 0.f); }
      if (ca < N) Out[af_at(4 * kk + t, ca)] = va;
      if (cb < N) Out[af_at(4 * kk + t, cb)] = vb;
    }
  }
}

__global__ __launch_bounds__(512) void k_actor_fused(ActorArgs p) {
  CHAIN_SETPRIO();
  extern __shared__ __attribute__((aligned(16))) float af_lds[];
  const int tiles = (p.nenv + 15) / 16;
  const int q = blockIdx.x & 7, idx = blockIdx.x >> 3;          // q: the XCD this workgroup lands on
  const int a = q + 8 * (idx / tiles), e0 = (idx % tiles) * 16;
  if (a >= p.A) return;
  const int tid = threadIdx.x;
  const int H = p.H, no = 2 * p.act_max;
  const int ks1 = (p.in_max + 15) / 16, ksh = H / 16, nth = H / 16, nto = (no + 15) / 16;
  const int img1 = 256 * max(ksh, nto), img0 = max(256 * ks1, img1);   // floats of the two activation images
  float *R0 = af_lds, *R1 = af_lds + img0;
  int *alist = reinterpret_cast<int *>(af_lds + img0 + img1);
  if (tid == 0) alist[0] = 0;
  __syncthreads();
  // this agent's entries of the global action vector (any order; the loads of one thread are independent)
  for (int g = tid; g < p.action_dim; g += 512)
    if (p.sc_agent[g] == a) alist[1 + atomicAdd(&alist[0], 1)] = g;
  // the gather index of a column does not depend on the row: one index load, 16 independent state loads
  for (int k = tid; k < 16 * ks1; k += 512) {
    const int g = k < p.in_max ? p.gather[a * p.in_max + k] : p.state_dim;
    const bool col = g < p.state_dim;
    const float *src = p.state + (col ? g : 0);
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = src[(long long)min(e0 + r, p.nenv - 1) * p.state_dim];
#pragma unroll
    for (int r = 0; r < 16; r++) R0[af_at(r, k)] = (col && e0 + r < p.nenv) ? v[r] : 0.f;
  }
  __syncthreads();
  af_layer(R0, ks1, p.W1 + (long long)a * nth * ks1 * 256, p.b1 + (long long)a * H, H, true, R1);
  __syncthreads();
  float *cur = R1, *nxt = R0;
  for (int l = 0; l + 1 < p.n_hidden; l++) {
    af_layer(cur, ksh, p.Wh[l] + (long long)a * nth * ksh * 256, p.bh[l] + (long long)a * H, H, true, nxt);
    __syncthreads();
    float *t = cur; cur = nxt; nxt = t;
  }
  af_layer(cur, ksh, p.Whead + (long long)a * nto * ksh * 256, p.bhead + (long long)a * no, no, false, nxt);
  __syncthreads();
  // k_policy_sample on the rows at hand: thread = (row, one in 32 of the agent's actions)
  const int r = tid & 15, e = e0 + r;
  if (e >= p.nenv) return;
  const int nact = min(alist[0], p.act_max);
  for (int i = tid >> 4; i < nact; i += 32) {
    const int g = alist[1 + i];
    const int l = p.sc_local[g];
    const float m = nxt[af_at(r, l)];
    const float ls = fminf(fmaxf(nxt[af_at(r, p.act_max + l)], p.ls_min), p.ls_max);
    const float eps = p.eps ? p.eps[(long long)e * p.action_dim + g]
                            : philox_normal(p.seed, 7u, p.counter, (uint32_t)e, (uint32_t)g);
    const float x = m + expf(ls) * eps;
    p.action[(long long)e * p.action_dim + g] = tanhf(x) * p.scale + p.bias;
    p.mean[(long long)e * p.action_dim + g] = tanhf(m) * p.scale + p.bias;
  }
}

struct StateBlocks {
  const float *src[8], *mean[8], *std[8];
  int ld[8], dim[8], off[8];
  int nblocks, total;
  const int32_t *sel;           // optional: column sel[i] of the source instead of column i (every block)
  // optional: the LAST block's source is still split-K partial tiles part[z][nenv][pn]: its value is
  // alpha * sum_z part[z] (k_gemm_reduce's expression), also written in full to sum_out[nenv][pn]
  const float *part; int nsplit, pn; float alpha; float *sum_out;
};

__global__ void k_assemble_state(int nenv, StateBlocks sb, float *__restrict__ out) {
  CHAIN_SETPRIO();
  const int e = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  auto psum = [&](int col) {
    return sb.alpha * slab_sum<4>(sb.nsplit, [&](int z) { return sb.part[((long long)z * nenv + e) * sb.pn + col]; });
  };
  if (j >= sb.total) {                       // extra threads: the reduced matrix itself
    const int col = j - sb.total;
    if (sb.part && col < sb.pn) sb.sum_out[(long long)e * sb.pn + col] = psum(col);
    return;
  }
  int b = 0;
#pragma unroll
  for (int k = 1; k < 8; k++) if (k < sb.nblocks && j >= sb.off[k]) b = k;
  const int i = j - sb.off[b];
  const int col = sb.sel ? sb.sel[i] : i;
  float v = (sb.part && b == sb.nblocks - 1) ? psum(col) : sb.src[b][(long long)e * sb.ld[b] + col];
  if (sb.mean[b]) v = (v - sb.mean[b][i]) / sb.std[b][i];
  out[(long long)e * sb.total + j] = v;
}

__global__ void k_agent_rewards(int nenv, int nmodes, int n_agents, const float *__restrict__ res, int ld,
                                const int32_t *__restrict__ lohi, float factor, float *__restrict__ out) {
  // out[e][a] = -factor * mean(res[e][lo:hi]^2); one wave per (env, agent)
  const int e = blockIdx.y, a = blockIdx.x, lane = threadIdx.x;
  const int lo = lohi[2 * a], hi = lohi[2 * a + 1];
  float s = 0.f;
  for (int m = lo + lane; m < hi; m += 64) { const float v = res[(long long)e * ld + m]; s += v * v; }
  s = wave_sum(s);
  if (lane == 0) out[(long long)e * n_agents + a] = -factor * s / (float)(hi - lo);
}

int aomarl_split_states(int nenv, int state_dim, int n_agents, int in_max, const int32_t *gather,
                        const float *state, float *out, void *stream) {
  if (!gather || !state || !out) return fail("split_states: null pointer");
  if (nenv <= 0 || n_agents <= 0 || in_max <= 0) return 0;
  hipLaunchKernelGGL(k_split_states, dim3((in_max + 255) / 256, nenv, n_agents), dim3(256), 0,
                     (hipStream_t)stream, nenv, state_dim, in_max, gather, state, out);
  LAUNCHCHK();
  return 0;
}

int aomarl_policy_sample(int nenv, int act_max, int action_dim, const float *head, float log_sig_min,
                         float log_sig_max, float scale, float bias, const int32_t *sc_agent,
                         const int32_t *sc_local, const float *eps, uint32_t seed, uint32_t counter,
                         float *action, float *mean, void *stream) {
  if (!head || !sc_agent || !sc_local || !action || !mean) return fail("policy_sample: null pointer");
  if (nenv <= 0 || action_dim <= 0) return 0;
  hipLaunchKernelGGL(k_policy_sample, dim3((action_dim + 255) / 256, nenv), dim3(256), 0,
                     (hipStream_t)stream, nenv, act_max, action_dim, head, log_sig_min, log_sig_max, scale,
                     bias, sc_agent, sc_local, eps, seed, counter, action, mean);
  LAUNCHCHK();
  return 0;
}

struct AssemblePart { const float *part; int nsplit, pn; float alpha; float *sum_out; };
static int assemble_state_impl(int nenv, int nblocks, const float *const *src, const int32_t *ld,
                               const int32_t *dim, const float *const *mean, const float *const *std_,
                               const int32_t *sel, float *out, void *stream, const AssemblePart *pt = nullptr);

int aomarl_assemble_state(int nenv, int nblocks, const float *const *src, const int32_t *ld,
                          const int32_t *dim, const float *const *mean, const float *const *std_,
                          float *out, void *stream) {
  return assemble_state_impl(nenv, nblocks, src, ld, dim, mean, std_, nullptr, out, stream);
}

int aomarl_assemble_state_cols(int nenv, int nblocks, const float *const *src, const int32_t *ld,
                               const int32_t *dim, const float *const *mean, const float *const *std_,
                               const int32_t *sel, float *out, void *stream) {
  return assemble_state_impl(nenv, nblocks, src, ld, dim, mean, std_, sel, out, stream);
}

static int assemble_state_impl(int nenv, int nblocks, const float *const *src, const int32_t *ld,
                               const int32_t *dim, const float *const *mean, const float *const *std_,
                               const int32_t *sel, float *out, void *stream, const AssemblePart *pt) {
  if (!src || !ld || !dim || !out) return fail("assemble_state: null pointer");
  if (nblocks < 1 || nblocks > 8) return fail("assemble_state: 1..8 blocks");
  StateBlocks sb;
  int off = 0;
  for (int k = 0; k < 8; k++) {
    const bool on = k < nblocks;
    sb.src[k] = on ? src[k] : nullptr; sb.ld[k] = on ? ld[k] : 0; sb.dim[k] = on ? dim[k] : 0;
    sb.mean[k] = (on && mean) ? mean[k] : nullptr; sb.std[k] = (on && std_) ? std_[k] : nullptr;
    sb.off[k] = off;
    if (on) {
      if (!src[k] || dim[k] <= 0 || (!sel && ld[k] < dim[k])) return fail("assemble_state: bad block %d", k);
      if ((sb.mean[k] == nullptr) != (sb.std[k] == nullptr)) return fail("assemble_state: mean/std must come together");
      off += dim[k];
    }
  }
  sb.nblocks = nblocks; sb.total = off; sb.sel = sel;
  sb.part = nullptr; sb.nsplit = 0; sb.pn = 0; sb.alpha = 1.f; sb.sum_out = nullptr;
  int extra = 0;
  if (pt && pt->part && pt->nsplit > 0) {
    sb.part = pt->part; sb.nsplit = pt->nsplit; sb.pn = pt->pn; sb.alpha = pt->alpha; sb.sum_out = pt->sum_out;
    extra = pt->pn;
  }
  if (nenv <= 0) return 0;
  hipLaunchKernelGGL(k_assemble_state, dim3((off + extra + 255) / 256, nenv), dim3(256), 0, (hipStream_t)stream,
                     nenv, sb, out);
  LAUNCHCHK();
  return 0;
}

int aomarl_agent_rewards(int nenv, int nmodes, int n_agents, const float *res_modes, int ld,
                         const int32_t *lohi, float factor, float *out, void *stream) {
  if (!res_modes || !lohi || !out) return fail("agent_rewards: null pointer");
  if (ld < nmodes) return fail("agent_rewards: ld < nmodes");
  if (nenv <= 0 || n_agents <= 0) return 0;
  hipLaunchKernelGGL(k_agent_rewards, dim3(n_agents, nenv), dim3(64), 0, (hipStream_t)stream, nenv, nmodes,
                     n_agents, res_modes, ld, lohi, factor, out);
  LAUNCHCHK();
  return 0;
}
