// aomarl_capi_step.hip -- part of the C ABI implementation (included by aomarl_capi.hip, one translation unit):
// aomarl_env_step: the fused step, the small-system chain, the frame pipeline, the HIP-graph form.
// ---------------------------------------------------------------- one call per half of a training step
// The per-step host work of the reference is a chain of ~25 tiny operations; here each of them is a
// native launch already, but issuing them one by one from Python costs ~10 us each -- more than the
// kernels themselves at small batch sizes.  These two entry points issue the same launches, in the
// same order, from C.
long long aomarl_actor_tiled_floats(int n_agents, int N, int K) {
  return (long long)n_agents * ((N + 15) / 16) * ((K + 15) / 16) * 256;
}

int aomarl_actor_tile_weights(int n_agents, int N, int K, const float *src, float *dst, void *stream) {
  if (!src || !dst) return fail("actor_tile_weights: null pointer");
  if (n_agents <= 0 || N <= 0 || K <= 0) return fail("actor_tile_weights: bad sizes");
  const int ntile = (N + 15) / 16, ksteps = (K + 15) / 16;
  const long long per = (long long)ntile * ksteps * 256;
  hipLaunchKernelGGL(k_actor_tile_weights, dim3((unsigned)((per + 255) / 256), n_agents), dim3(256), 0,
                     (hipStream_t)stream, N, K, ntile, ksteps, src, dst);
  LAUNCHCHK();
  return 0;
}

int aomarl_actor_forward(const aomarl_actor_desc *d, const float *state, const float *eps, uint32_t seed,
                         uint32_t counter, float *action, float *mean, void *stream) {
  if (!d || !state || !action || !mean) return fail("actor_forward: null argument");
  if (d->n_hidden < 1 || d->n_hidden > 8) return fail("actor_forward: 1..8 hidden layers");
  const int A = d->n_agents, n = d->nenv, H = d->hidden;
  if (A <= 0 || n <= 0) return 0;
  if (!(d->flags & AOMARL_ACTOR_LAYER_BY_LAYER) && d->W1_tiled && d->Whead_tiled && H % 16 == 0) {
    // one launch: pre-tiled weights at hand and the activations of 16 environments fit in LDS
    const int ks1 = (d->in_max + 15) / 16, nto = (2 * d->act_max + 15) / 16;
    const size_t img1 = (size_t)256 * std::max(H / 16, nto), img0 = std::max((size_t)256 * ks1, img1);
    const size_t lds = (img0 + img1 + d->act_max + 4) * sizeof(float);
    static bool big_lds = false;
    if (!big_lds && lds > 64 * 1024 && lds <= 128 * 1024) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_actor_fused), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
      big_lds = true;
    }
    if (lds <= 128 * 1024) {
      ActorArgs p;
      p.A = A; p.nenv = n; p.state_dim = d->state_dim; p.in_max = d->in_max; p.act_max = d->act_max; p.H = H;
      p.n_hidden = d->n_hidden; p.action_dim = d->action_dim;
      p.gather = d->gather; p.W1 = d->W1_tiled; p.b1 = d->b1;
      for (int l = 0; l < 8; l++) {
        p.Wh[l] = l + 1 < d->n_hidden ? d->Wh_tiled[l] : nullptr;
        p.bh[l] = l + 1 < d->n_hidden ? d->bh[l] : nullptr;
        if (l + 1 < d->n_hidden && !p.Wh[l]) return fail("actor_forward: tiled hidden weights missing");
      }
      if (((uintptr_t)p.W1 | (uintptr_t)d->Whead_tiled) & 15) return fail("actor_forward: tiled weights must be 16-byte aligned");
      p.Whead = d->Whead_tiled; p.bhead = d->bhead; p.sc_agent = d->sc_agent; p.sc_local = d->sc_local;
      p.ls_min = d->log_sig_min; p.ls_max = d->log_sig_max; p.scale = d->scale; p.bias = d->bias;
      p.state = state; p.eps = eps; p.seed = seed; p.counter = counter; p.action = action; p.mean = mean;
      const int tiles = (n + 15) / 16, groups = (A + 7) / 8;
      hipLaunchKernelGGL(k_actor_fused, dim3(8 * tiles * groups), dim3(512), lds, (hipStream_t)stream, p);
      g_arith[AR_ACTOR_F32]++;
      LAUNCHCHK();
      return 0;
    }
  }
  if (!d->x || !d->h0 || !d->h1 || !d->head) return fail("actor_forward: the layer-by-layer path needs its scratch buffers");
  int rc = aomarl_split_states(n, d->state_dim, A, d->in_max, d->gather, state, d->x, stream);
  if (rc) return rc;
  rc = aomarl_gemm_nt_batched(A, n, H, d->in_max, d->x, d->in_max, (long long)n * d->in_max, d->W1, d->in_max,
                              (long long)H * d->in_max, d->b1, H, d->h0, H, (long long)n * H, 1, stream);
  if (rc) return rc;
  float *cur = d->h0, *nxt = d->h1;
  for (int l = 0; l + 1 < d->n_hidden; l++) {
    rc = aomarl_gemm_nt_batched(A, n, H, H, cur, H, (long long)n * H, d->Wh[l], H, (long long)H * H, d->bh[l], H,
                                nxt, H, (long long)n * H, 1, stream);
    if (rc) return rc;
    std::swap(cur, nxt);
  }
  const int no = 2 * d->act_max;
  rc = aomarl_gemm_nt_batched(A, n, no, H, cur, H, (long long)n * H, d->Whead, H, (long long)no * H, d->bhead, no,
                              d->head, no, (long long)n * no, 0, stream);
  if (rc) return rc;
  return aomarl_policy_sample(n, d->act_max, d->action_dim, d->head, d->log_sig_min, d->log_sig_max, d->scale,
                              d->bias, d->sc_agent, d->sc_local, eps, seed, counter, action, mean, stream);
}

static int env_step_validate(aomarl_ctx *c, aomarl_state *st, aomarl_env_glue *g, const float *action, float *state_out,
                             float *reward_out) {
  if (!c || !st || !g || !state_out) return fail("env_step: null argument");
  if (g->nhist < 0 || g->nhist > 5) return fail("env_step: 0..5 command histories");
  const int n = st->nenv, nm = g->nmodes, R = g->nhist + 1;
  if (g->ring_pos < 0 || g->ring_pos >= R) return fail("env_step: ring position out of range");
  int rc = check_range(c, st, 0, n);
  if (rc) return rc;
  if (!c->v2m || !c->m2v) return fail("env_step: no modal basis (aomarl_set_modal)");
  if (nm != c->nmodes) return fail("env_step: glue has %d modes, the basis %d", nm, c->nmodes);
  if (action && c->nact <= 0) return fail("env_step: no action modes set");
  if (!g->modes_ring || !g->res_modes) return fail("env_step: glue->modes_ring / glue->res_modes are null");
  if (reward_out && (g->n_agents <= 0 || !g->lohi)) return fail("env_step: reward_out needs glue->n_agents > 0 and glue->lohi");
  if (g->dm_dim <= 0 || g->dm_dim > nm || (!g->sel && g->dm_dim != nm))
    return fail("env_step: glue->dm_dim = %d does not fit %d modes%s", g->dm_dim, nm, g->sel ? "" : " (no column selection given)");
  if ((g->mean_dm || g->std_dm || g->mean_res || g->std_res) && !(g->mean_dm && g->std_dm && g->mean_res && g->std_res))
    return fail("env_step: standardisation needs all of mean_dm, std_dm, mean_res, std_res (or none)");
  if (c->env_gain)
    return fail("env_step: per-environment integrator gains are set on this context (aomarl_set_env_gains); env_step "
                "takes ONE scalar gain -- clear them (aomarl_set_env_gains(ctx, NULL, 0)) or step call by call");
  if (g->sel && (c->sel_checked != g->sel || c->sel_checked_n != g->dm_dim || c->sel_checked_nm != nm)) {
    // column selection of the state blocks: validated once per (pointer, size) -- a synchronous copy of
    // dm_dim indices, never again in the steady state
    // (every stream first: a selection just written by a kernel of a non-blocking stream is not ordered
    // with a synchronous copy -- seen as garbage indices under bench.py's own stream)
    std::vector<int32_t> h((size_t)g->dm_dim);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(h.data(), g->sel, sizeof(int32_t) * h.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); i++)
      if (h[i] < 0 || h[i] >= nm) return fail("env_step: glue->sel[%zu] = %d is outside the %d modes", i, h[i], nm);
    c->sel_checked = g->sel; c->sel_checked_n = g->dm_dim; c->sel_checked_nm = nm;
  }
  return 0;
}

// ---------------------------------------------------------------- small systems: the chain in two kernels
// With <= 512 actuators / modes and <= 1024 slopes (the 10x10 files: 90 / 85 / 152) the three products of the
// control chain are a few thousand multiply-adds per environment: as GEMMs they are three launches + their
// neighbours (compose, delay line, Strehl commit; integrate, state assembly), 8 launches of ~5 us on a step that is
// bound by launches (configs[1]).  One workgroup per environment does each half of the chain by itself:
//   k_small_head: Btt compose (+ per-agent rewards), m2v product, delay line, tip-tilt slot, Strehl commit
//   k_small_tail: -cmat . s, integrator, v2m . err, the state blocks
// Same formulas as the kernels they stand for; the sums of the products run in one thread each, in index order
// (the split-K GEMM sums tiles): fp32 round-off apart, the same numbers ("small_chain" = 0: the general chain).
constexpr int SMALL_NM = 512, SMALL_NA = 512, SMALL_NSL = 1024;
// y[o] = sum_k x[k] W[o][k] for o < no, x in LDS, by the whole block: FOUR threads per output (they read 16
// consecutive bytes of the row per step, two accumulators each, then two xor-shuffles), blockDim.x / 4 outputs per pass.
// done(o, y) runs in the first thread of each quad.
template <class F>
__device__ __forceinline__ void small_gemv(const float *__restrict__ W, int ldw, int no, int K, const float *xs, F done) {
  const int tid = threadIdx.x, q = tid & 3, per = blockDim.x >> 2;
  for (int o0 = 0; o0 < no; o0 += per) {
    const int o = o0 + (tid >> 2);
    float a0 = 0.f, a1 = 0.f;
    if (o < no) {
      const float *row = W + (long long)o * ldw;
      int k = q;
      // eight row elements in flight per thread (same accumulators, same order: the plain loop below, which the
      // compiler runs one load pair at a time -- K / 8 load latencies in a row)
      for (; k + 28 < K; k += 32) {
        float r[8];
#pragma unroll
        for (int u = 0; u < 8; u++) r[u] = row[k + 4 * u];
#pragma unroll
        for (int u = 0; u < 8; u += 2) { a0 = fmaf(xs[k + 4 * u], r[u], a0); a1 = fmaf(xs[k + 4 * u + 4], r[u + 1], a1); }
      }
      for (; k + 4 < K; k += 8) { a0 = fmaf(xs[k], row[k], a0); a1 = fmaf(xs[k + 4], row[k + 4], a1); }
      if (k < K) a0 = fmaf(xs[k], row[k], a0);
    }
    float y = a0 + a1;
    y += __shfl_xor(y, 1);
    y += __shfl_xor(y, 2);
    if (o < no && q == 0) done(o, y);
  }
}
struct SmallHead {
  int nm, na, nact, n_agents, ld_m2v, ld_actu, ktt, do_strehl;
  float gain, factor, wa, wb, wc;
  const float *m0, *m1, *action, *freedom, *m2v, *PEND;
  const int32_t *amode_inv, *lohi;
  float *modes_out, *rew;
};
__global__ __launch_bounds__(256) void k_small_head(DevSys sys, DevState st, SmallHead p) {
  __shared__ float sm[SMALL_NM], sv[SMALL_NA];
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int m = tid; m < p.nm; m += 256) {
    float v = p.m0[(long long)e * p.nm + m] + p.gain * p.m1[(long long)e * p.nm + m];
    if (p.action) {
      const int j = p.amode_inv[m];
      if (j >= 0) v += p.action[(long long)e * p.nact + j] * p.freedom[m];
    }
    sm[m] = v;
    if (p.modes_out) p.modes_out[(long long)e * p.nm + m] = v;
  }
  if (p.rew)
    for (int a = wv; a < p.n_agents; a += 4) {
      const int lo = p.lohi[2 * a], hi = p.lohi[2 * a + 1];
      float q = 0.f;
      for (int m = lo + lane; m < hi; m += 64) { const float v = p.m1[(long long)e * p.nm + m]; q += v * v; }
      q = wave_sum(q);
      if (lane == 0) p.rew[(long long)e * p.n_agents + a] = -p.factor * q / (float)(hi - lo);
    }
  __syncthreads();
  small_gemv(p.m2v, p.ld_m2v, p.na, p.nm, sm, [&](int a, float c0) {
    const long long o = (long long)e * p.ld_actu + a;
    const float c1 = st.com1[o], c2 = st.com2[o];
    const float v = p.wa * c0 + p.wb * c1 + p.wc * c2;
    st.com[o] = c0; st.voltage[o] = v; st.com2[o] = c1; st.com1[o] = c0;
    sv[a] = v;
  });
  __syncthreads();
  if (p.ktt >= 0 && tid < 3) {                   // dm_shape_tt_body on the voltages just formed
    const DevDm &D = sys.dms[p.ktt];
    float *shape = st.dm_shape + (long long)e * sys.shape_stride + D.shape_off;
    if (tid < 2) shape[tid] = sv[D.com_off + tid];
    if (tid == 2 && sys.fused_ok) {
      const DevDm &Z = sys.dms[0];
      const int half = sys.pupdiam / 2, zp = (half + Z.toy) * Z.dim + half + Z.tox;
      const int ss2 = Z.ss * Z.ss, s0 = Z.influstart[zp], cn = Z.ninflu[zp];
      float acc = 0.f;
      for (int t = 0; t < cn; t++) {
        const int pos = Z.influpos[s0 + t];
        acc += Z.influ[pos] * sv[Z.com_off + pos / ss2];
      }
      shape[2] = acc;
    }
  }
  if (p.do_strehl) strehl_commit_body(sys, st, 0, e, p.PEND);
}

struct SmallTail {
  int nsl, na, nm, ld_cmat, ld_v2m, ld_actu;
  float gain;
  const float *cmat, *v2m;
  float *res_modes;
};
__global__ __launch_bounds__(256) void k_small_tail(DevState st, SmallTail p, StateBlocks sb, float *__restrict__ out) {
  __shared__ float ss[SMALL_NSL], se[SMALL_NA], sr[SMALL_NM];
  const int e = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < p.nsl; k += 256) ss[k] = st.slopes[(long long)e * p.nsl + k];
  __syncthreads();
  small_gemv(p.cmat, p.ld_cmat, p.na, p.nsl, ss, [&](int a, float acc) {
    const float v = -acc;
    const long long o = (long long)e * p.ld_actu + a;
    st.err[o] = v;
    st.com[o] += p.gain * v;
    se[a] = v;
  });
  __syncthreads();
  small_gemv(p.v2m, p.ld_v2m, p.nm, p.na, se, [&](int m, float acc) {
    p.res_modes[(long long)e * p.nm + m] = acc;
    sr[m] = acc;
  });
  __syncthreads();
  for (int j = tid; j < sb.total; j += 256) {
    int b = 0;
#pragma unroll
    for (int k = 1; k < 8; k++) if (k < sb.nblocks && j >= sb.off[k]) b = k;
    const int i = j - sb.off[b];
    const int col = sb.sel ? sb.sel[i] : i;
    float v = (b == sb.nblocks - 1) ? sr[col] : sb.src[b][(long long)e * sb.ld[b] + col];
    if (sb.mean[b]) v = (v - sb.mean[b][i]) / sb.std[b][i];
    out[(long long)e * sb.total + j] = v;
  }
}

static bool small_chain_ok(const aomarl_ctx *c, const aomarl_env_glue *g) {
  return c->small_chain && c->sys.nactu <= SMALL_NA && c->sys.nslope <= SMALL_NSL && g->nmodes <= SMALL_NM &&
         (long long)c->sys.nactu * c->sys.nslope <= 65536 && c->cmat && !c->env_gain;
}

// may the chain run in its fused form?  (ktt: index of the tip-tilt mirror)
static bool env_step_fusable(aomarl_ctx *c, const aomarl_env_glue *g, int *ktt_out) {
  int ktt = -1, ntt = 0, nother = 0;
  for (int k = 0; k < c->ndm; k++) {
    if (c->sys.dms[k].type == AOMARL_DM_TT) { ktt = k; ntt++; } else nother++;
  }
  const bool defer = c->defer_dm_shape && aomarl_dm_from_voltage_available(c);
  if (ktt_out) *ktt_out = ktt;
  return !(g->flags & AOMARL_ENV_STEP_UNFUSED) && ntt == 1 && (defer || nother == 0);
}

// ---- AoEnv.rl_step, fused form: Btt correction from the coordinates at hand (+ the per-agent rewards of the
// residual measured before this action reaches the DM), delay line, tip-tilt shape, Strehl commit.
// `stv`: the state whose voltage / dm_shape / pending PSF window this call writes and commits (st itself, or
// the frame pipeline's view of the parity the NEXT frame uses).  ahead: the delay line is evaluated one frame
// ahead (the voltages of the frame that follows the one in flight): weights shifted by one command.
static int env_step_head_fused(aomarl_ctx *c, aomarl_state *st, aomarl_state *stv, aomarl_env_glue *g, const float *action,
                               float gain, float *reward_out, int ktt, bool ahead, hipEvent_t psf_ev, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const int n = st->nenv, nm = g->nmodes, R = g->nhist + 1, na = c->sys.nactu;
  const size_t slot = (size_t)n * nm;
  float *newest = g->modes_ring + (size_t)g->ring_pos * slot;
  float *mnew = g->modes_ring + (size_t)((g->ring_pos + 1) % R) * slot;
  Work w = work_layout(c, st->nenv);
  DevState dsv = dev_state(stv);
  if (small_chain_ok(c, g)) {
    const float d = c->delay;
    SmallHead p;
    if (d <= 1.f) { p.wa = 1.f - d; p.wb = d; p.wc = 0.f; } else { p.wa = 0.f; p.wb = 2.f - d; p.wc = d - 1.f; }
    if (ahead) { p.wa = p.wb; p.wb = p.wc; p.wc = 0.f; }
    p.nm = nm; p.na = na; p.nact = c->nact; p.n_agents = reward_out ? g->n_agents : 0; p.ld_m2v = c->ld_m2v;
    p.ld_actu = st->ld_actu; p.ktt = ktt; p.do_strehl = 1; p.gain = gain; p.factor = g->reward_factor;
    p.m0 = newest; p.m1 = g->res_modes; p.action = action; p.freedom = c->freedom; p.m2v = c->m2v;
    p.PEND = stv->work + w.PEND; p.amode_inv = c->amode_inv; p.lohi = g->lohi; p.modes_out = mnew; p.rew = reward_out;
    if (psf_ev) HIPCHK(hipStreamWaitEvent(s, psf_ev, 0));
    else if (stv == st) { int rc = psf_wait_pending(c, stream); if (rc) return rc; }
    // ONE kernel wrote the voltages and committed the pending window, behind the wait for that parity's PSF finish:
    // the release of the frame stream covers all three (no separate commit event, no wait of its own on the frame
    // stream); the event rides on the dispatch
    hipExtLaunchKernelGGL(k_small_head, dim3(n), dim3(256), 0, s, nullptr, ahead ? c->pipe.ev_cmd : nullptr, 0, c->sys, dsv, p);
    LAUNCHCHK();
    if (ahead) c->pipe.cmd_covers_commit = true;
    return 0;
  }
  float *modes = st->work + w.MODES;
  const int cx = (nm + 255) / 256;
  // (frame pipeline: the wait for that parity's last PSF finish -- two frames old, long over -- HERE, in front of the
  // chain's first kernel, so that the command event covers it and the frame stream has one cross-queue wait less
  // between two frame kernels)
  bool psf_waited = false;
  if (ahead && psf_ev) { HIPCHK(hipStreamWaitEvent(s, psf_ev, 0)); psf_waited = true; }
  hipLaunchKernelGGL(k_compose_rewards, dim3(cx + (reward_out ? g->n_agents : 0), n), dim3(256), 0, s, nm, newest,
                     g->res_modes, gain, action, c->nact, c->amode_inv, c->freedom, modes, w.ldm, mnew, cx,
                     g->n_agents, g->lohi, g->reward_factor, reward_out);
  LAUNCHCHK();
  int nsp = 0;
  float alpha = 1.f;
  launch_gemm_nt(n, na, nm, 1.0f, modes, w.ldm, c->m2v, c->ld_m2v, 0.0f, st->com, st->ld_actu, s,
                 st->work + w.GEMM, w.gemm_floats, nullptr, &nsp, true, 16.f, c->m2v_scale, &alpha, 288);
  LAUNCHCHK();
  const float d = c->delay;
  float wa, wb, wc;
  if (d <= 1.f) { wa = 1.f - d; wb = d; wc = 0.f; } else { wa = 0.f; wb = 2.f - d; wc = d - 1.f; }
  if (ahead) {
    // delay == 1 (the pipeline's condition): v(t+1) = c(t); the tip-tilt slot in the same launch, and the frame
    // stream released right behind it -- the Strehl commit below is not on the frame kernel's path
    // (the command event rides on the dispatch, like the frame kernel's and the move's: no marker packet behind it)
    hipExtLaunchKernelGGL(k_delay_ahead, dim3((na + 255) / 256, 2 * n), dim3(256), 0, s, nullptr, c->pipe.ev_cmd, 0, c->sys, dsv, na,
                          st->ld_actu, n, nsp > 0 ? st->work + w.GEMM : nullptr, nsp, alpha, ktt);
    LAUNCHCHK();
    c->pipe.cmd_covers_psf = psf_waited;
  } else {
    if (nsp > 0)
      hipLaunchKernelGGL(k_delay_sum, dim3((na + 255) / 256, n), dim3(256), 0, s, dsv, na, st->ld_actu, wa, wb, wc, 0, 1,
                         st->work + w.GEMM, nsp, alpha, n);
    else
      hipLaunchKernelGGL(k_delay, dim3((na + 255) / 256, n), dim3(256), 0, s, dsv, na, st->ld_actu, wa, wb, wc, 0, 1);
    LAUNCHCHK();
  }
  if (psf_ev) { if (!psf_waited) HIPCHK(hipStreamWaitEvent(s, psf_ev, 0)); }
  else if (stv == st) { int rc = psf_wait_pending(c, stream); if (rc) return rc; }
  hipLaunchKernelGGL(k_post_delay, dim3(ahead ? n : 2 * n), dim3(256), 0, s, c->sys, dsv, 0, n, stv->work + w.PEND, 1, ktt,
                     stv->voltage, st->ld_actu);
  LAUNCHCHK();
  if (ahead) { HIPCHK(hipEventRecord(c->pipe.ev_commit, s)); c->pipe.cmd_covers_commit = false; }   // the PSF finish of the frame about to be launched overwrites that window
  return 0;
}

// the residual shortcut applies: the matrix is there for these modes and the integrator gain is one scalar
static bool env_step_shortcut(const aomarl_ctx *c, const aomarl_env_glue *g) {
  return c->residual_shortcut && c->s2m && c->s2m_nmodes == g->nmodes && !c->env_gain;
}

// ---- the rest of AoEnv.linear_step behind do_control: v2m . err, the state blocks
// s2m_view: the residual shortcut -- the caller has NOT run do_control; the residual modes come from that state's slopes
// in one product with -(v2m . cmat) (the integrator lives in the Btt coordinates, the next head rebuilds the command)
static int env_step_tail(aomarl_ctx *c, aomarl_state *st, aomarl_env_glue *g, bool fused, float *state_out, void *stream,
                         const aomarl_state *slopes_view = nullptr, const aomarl_state *s2m_view = nullptr) {
  hipStream_t s = (hipStream_t)stream;
  const int n = st->nenv, nm = g->nmodes, R = g->nhist + 1, na = c->sys.nactu;
  const size_t slot = (size_t)n * nm;
  const int nxt = (g->ring_pos + 1) % R;
  float *mnew = g->modes_ring + (size_t)nxt * slot;
  Work w = work_layout(c, st->nenv);
  int rc = 0;
  if (slopes_view) {
    // small systems: do_control, v2m . err and the state blocks in ONE kernel (the caller has NOT run do_control)
    const float *src[8], *mean[8], *sd[8];
    int32_t ld[8], dim[8];
    int nb = 0;
    for (int h = g->nhist; h >= 1; h--) {
      src[nb] = g->modes_ring + (size_t)((nxt - h + R * 8) % R) * slot;
      mean[nb] = g->mean_dm; sd[nb] = g->std_dm; ld[nb] = nm; dim[nb] = g->dm_dim; nb++;
    }
    src[nb] = mnew; mean[nb] = g->mean_dm; sd[nb] = g->std_dm; ld[nb] = nm; dim[nb] = g->dm_dim; nb++;
    src[nb] = g->res_modes; mean[nb] = g->mean_res; sd[nb] = g->std_res; ld[nb] = nm; dim[nb] = g->dm_dim; nb++;
    const bool norm = g->mean_dm && g->std_dm && g->mean_res && g->std_res;
    StateBlocks sb;
    int off = 0;
    for (int k = 0; k < 8; k++) {
      const bool on = k < nb;
      sb.src[k] = on ? src[k] : nullptr; sb.ld[k] = on ? ld[k] : 0; sb.dim[k] = on ? dim[k] : 0;
      sb.mean[k] = (on && norm) ? mean[k] : nullptr; sb.std[k] = (on && norm) ? sd[k] : nullptr;
      sb.off[k] = off;
      if (on) off += dim[k];
    }
    sb.nblocks = nb; sb.total = off; sb.sel = g->sel;
    sb.part = nullptr; sb.nsplit = 0; sb.pn = 0; sb.alpha = 1.f; sb.sum_out = nullptr;
    SmallTail p;
    p.nsl = c->sys.nslope; p.na = na; p.nm = nm; p.ld_cmat = c->ld_cmat; p.ld_v2m = c->ld_v2m; p.ld_actu = st->ld_actu;
    p.gain = c->gain; p.cmat = c->cmat; p.v2m = c->v2m; p.res_modes = g->res_modes;
    hipLaunchKernelGGL(k_small_tail, dim3(n), dim3(256), 0, s, dev_state(slopes_view), p, sb, state_out);
    LAUNCHCHK();
    g->ring_pos = nxt;
    return 0;
  }
  AssemblePart part = {nullptr, 0, 0, 1.f, nullptr};
  if (s2m_view) {
    int nsp = 0;
    float alpha = -1.f;
    const int nsl = c->sys.nslope;
    launch_gemm_nt(n, nm, nsl, -1.0f, s2m_view->slopes, nsl, c->s2m, c->ld_cmat, 0.0f, g->res_modes, nm, s,
                   st->work + w.GEMM, w.gemm_floats, nullptr, &nsp, /* slopes (arcsec), unscaled */ true, 1.f, c->s2m_scale,
                   &alpha, 288);
    LAUNCHCHK();
    if (nsp > 0) { part.part = st->work + w.GEMM; part.nsplit = nsp; part.pn = nm; part.alpha = alpha; part.sum_out = g->res_modes; }
  } else if (fused) {
    int nsp = 0;
    float alpha = 1.f;
    launch_gemm_nt(n, nm, na, 1.0f, st->err, st->ld_actu, c->v2m, c->ld_v2m, 0.0f, g->res_modes, nm, s,
                   st->work + w.GEMM, w.gemm_floats, nullptr, &nsp, /* volts */ true, 1.f, c->v2m_scale, &alpha, 288);
    LAUNCHCHK();
    if (nsp > 0) { part.part = st->work + w.GEMM; part.nsplit = nsp; part.pn = nm; part.alpha = alpha; part.sum_out = g->res_modes; }
  } else {
    rc = aomarl_volts2modes(c, st, n, st->err, st->ld_actu, g->res_modes, stream);
    if (rc) return rc;
  }
  const float *src[8], *mean[8], *sd[8];
  int32_t ld[8], dim[8];
  int nb = 0;
  for (int h = g->nhist; h >= 1; h--) {                       // oldest first
    src[nb] = g->modes_ring + (size_t)((nxt - h + R * 8) % R) * slot;
    mean[nb] = g->mean_dm; sd[nb] = g->std_dm; ld[nb] = nm; dim[nb] = g->dm_dim; nb++;
  }
  src[nb] = mnew; mean[nb] = g->mean_dm; sd[nb] = g->std_dm; ld[nb] = nm; dim[nb] = g->dm_dim; nb++;
  src[nb] = g->res_modes; mean[nb] = g->mean_res; sd[nb] = g->std_res; ld[nb] = nm; dim[nb] = g->dm_dim; nb++;
  const bool norm = g->mean_dm && g->std_dm && g->mean_res && g->std_res;
  rc = assemble_state_impl(n, nb, src, ld, dim, norm ? mean : nullptr, norm ? sd : nullptr, g->sel, state_out, stream,
                           &part);
  if (rc) return rc;
  g->ring_pos = nxt;
  return 0;
}

static int env_step_body(aomarl_ctx *c, aomarl_state *st, aomarl_env_glue *g, const float *action, float gain,
                         float *accumx, float *accumy, float *state_out, float *reward_out, void *stream) {
  int rc = env_step_validate(c, st, g, action, state_out, reward_out);
  if (rc) return rc;
  const int n = st->nenv, nm = g->nmodes, R = g->nhist + 1;
  const size_t slot = (size_t)n * nm;
  float *newest = g->modes_ring + (size_t)g->ring_pos * slot;
  float *mnew = g->modes_ring + (size_t)((g->ring_pos + 1) % R) * slot;
  // Fused form of the chain (same arithmetic, same order of every sum -- the results are bit for bit
  // those of the entry points called one by one): every split-K reduction happens in the kernel that
  // consumes the product, independent small kernels share a launch.  10 launches per step on the
  // main stream instead of 14.
  int ktt = -1;
  const bool fused = env_step_fusable(c, g, &ktt);
  if (fused) {
    rc = env_step_head_fused(c, st, st, g, action, gain, reward_out, ktt, false, nullptr, stream);
    if (rc) return rc;
  } else {
    // ---- AoEnv.rl_step: Btt correction from the coordinates at hand, delay line, Strehl
    rc = aomarl_rl_control_modes(c, st, 0, n, newest, g->res_modes, gain, action, mnew, stream);
    if (rc) return rc;
    rc = aomarl_apply_control(c, st, 0, n, AOMARL_APPLY_COMP_VOLTAGE | (c->defer_dm_shape ? AOMARL_APPLY_DEFER_STACK_SHAPE : 0), stream);
    if (rc) return rc;
    rc = aomarl_comp_strehl(c, st, 0, n, stream);
    if (rc) return rc;
    // ---- per-agent rewards from the residual measured before this action reached the DM
    if (reward_out) {
      rc = aomarl_agent_rewards(n, nm, g->n_agents, g->res_modes, nm, g->lohi, g->reward_factor, reward_out, stream);
      if (rc) return rc;
    }
  }
  // ---- AoEnv.linear_step
  if (g->denoiser) {
    // rlSupervisor.py:975-984: image -> autoencoder -> centroids -> do_control, the cube staying on the device
    if (!st->bincube) return fail("env_step: the denoiser needs st->bincube");
    if (!aomarl_frame_fused_available(c)) return fail("env_step: denoiser branch needs the one-pass frame kernel");
    rc = aomarl_move_atmos(c, st, 0, n, accumx, accumy, stream);
    if (rc) return rc;
    const bool defer = c->defer_dm_shape && aomarl_dm_from_voltage_available(c);
    rc = aomarl_frame_fused(c, st, 0, n, AOMARL_IMG_NOISE | AOMARL_IMG_WRITE_BINCUBE | (defer ? AOMARL_IMG_DM_FROM_VOLTAGE : 0), stream);
    if (rc) return rc;
    const long long nimg = (long long)n * c->sys.nvalid;
    rc = g->denoiser_f32 ? aomarl_denoiser_apply_f32((aomarl_denoiser *)g->denoiser, st->bincube, nimg, stream)
                         : aomarl_denoiser_apply_split_f16((aomarl_denoiser *)g->denoiser, st->bincube, nimg, stream);
    if (rc) return rc;
    // the next frame's extrusions go beside centroids / control / agents, not beside the denoiser: that
    // kernel fills the GPU by itself and small kernels next to it only stretch both
    if (c->prefetch_atmos && !c->premoved) {
      rc = aomarl_prefetch_atmos(c, st, 0, n, accumx, accumy, stream);
      if (rc) return rc;
    }
    rc = aomarl_do_centroids(c, st, 0, n, stream);
    if (rc) return rc;
    rc = aomarl_do_control(c, st, 0, n, stream);
  } else {
    const bool small = fused && small_chain_ok(c, g);
    const bool shortcut = !small && fused && env_step_shortcut(c, g);
    c->skip_do_control = small || shortcut;      // the small chain's tail kernel does it; the shortcut needs none
    rc = aomarl_next_part_one(c, st, 0, n, accumx, accumy, 0, stream);
    c->skip_do_control = false;
    if (rc) return rc;
    if (small) return env_step_tail(c, st, g, fused, state_out, stream, st);
    if (shortcut) return env_step_tail(c, st, g, fused, state_out, stream, nullptr, st);
  }
  if (rc) return rc;
  return env_step_tail(c, st, g, fused, state_out, stream);
}

// ---------------------------------------------------------------- frame pipeline (aomarl_set_frame_pipeline)
// Step t of the plain order:  head(a_t) -> v_t | frame_t | do_control_t, tail -> state_{t+1}.  With delay == 1
// v_{t+1} = c_t is known after head(a_t), so frame_{t+1} is launched by the call of step t, on the frame stream,
// BEFORE that call reduces frame_t: the frame kernels run back to back, the control / agent chain of frame t
// (do_control_t .. actor .. head(a_{t+1})) runs beside frame_{t+1}, the move for frame t+2 beside it too.
//   buffers: parity 0 = st's slopes / voltage / dm_shape / work (PSF rows, pending window), parity 1 = the twin's;
//   ring origins: per-parity snapshots (the live origins move with the prefetched atmosphere).
static aomarl_state pipe_view(aomarl_ctx *c, const aomarl_state *st, int par) {
  aomarl_state v = *st;
  if (par) {
    v.slopes = c->pipe.twin.slopes; v.voltage = c->pipe.twin.voltage; v.dm_shape = c->pipe.twin.dm_shape;
    v.work = c->pipe.twin.work;
  }
  return v;
}

static bool pipe_eligible(aomarl_ctx *c, const aomarl_state *st, const aomarl_env_glue *g, const float *accumx,
                          const float *accumy) {
  const auto &P = c->pipe;
  return P.have_twin && c->pipe_enabled && P.owner_screens == st->screens && !c->graph_step && !c->capturing && c->prefetch_atmos &&
         c->delay == 1.f && c->sys.noise < 0.f && !g->denoiser && accumx && accumy && !c->subpixel_flow &&
         aomarl_frame_fused_available(c) && c->defer_dm_shape && aomarl_dm_from_voltage_available(c) &&
         env_step_fusable(c, g, nullptr);
}

static int pipe_init(aomarl_ctx *c, const aomarl_state *st) {
  auto &P = c->pipe;
  int rc = side_stream(c);
  if (rc) return rc;
  if (!P.fstream) {
    // normal priority, every CU: a low-priority frame stream (0.76 against 0.60 ms per step), CUs reserved for the
    // chains through a CU mask (0.59 - 1.03) and high-priority chain streams (-1 %) were measured and dropped
    HIPCHK(hipStreamCreateWithFlags(&P.fstream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&P.ev_cmd, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&P.ev_commit, hipEventDisableTiming));
    for (int k = 0; k < 2; k++) {
      HIPCHK(hipEventCreateWithFlags(&P.ev_done[k], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&P.ev_psf[k], hipEventDisableTiming));
    }
  }
  const size_t ints = (size_t)st->nenv * c->nlayers * 2;
  if (P.snap_ints < ints) {
    for (int k = 0; k < 2; k++) {
      if (P.snap[k]) (void)hipFree(P.snap[k]);
      P.snap[k] = nullptr;
      HIPCHK(hipMalloc((void **)&P.snap[k], sizeof(int32_t) * ints));
    }
    P.snap_ints = ints;
  }
  return 0;
}

// the frame of parity q on the frame stream: behind everything issued on `stream` so far (that parity's voltages,
// tip-tilt shape and committed PSF window) and behind the prefetched move, whose origins are in snap[q]
static int pipe_launch_frame(aomarl_ctx *c, aomarl_state *st, int q, void *stream) {
  auto &P = c->pipe;
  (void)stream;
  HIPCHK(hipStreamWaitEvent(P.fstream, P.ev_cmd, 0));       // recorded behind the kernel that wrote that parity's voltages / tip-tilt slot
  if (!c->premoved || c->pre_screens != st->screens || c->pre_b != 0 || c->pre_n != st->nenv)
    return fail("frame pipeline: no prefetched atmosphere frame of the whole batch is pending");
  HIPCHK(hipStreamWaitEvent(P.fstream, c->ev_moved, 0));
  c->premoved = false;
  aomarl_state v = pipe_view(c, st, q);
  v.origin = P.snap[q];
  return frame_fused_impl(c, &v, 0, st->nenv, AOMARL_IMG_COG | AOMARL_IMG_NOISE | AOMARL_IMG_DM_FROM_VOLTAGE,
                          (void *)P.fstream, q);
}

// the move for the frame after the newest one in flight, on the atmosphere stream: beside the newest frame when
// the plan allows (run_plan), behind the older one in any case; then the origins that frame will use
static int pipe_prefetch(aomarl_ctx *c, aomarl_state *st, float *accumx, float *accumy, int older, int newest) {
  auto &P = c->pipe;
  if (c->premoved) return fail("frame pipeline: a prefetched frame is already pending");
  c->side_joined = false;
  c->ev_frame_prev = P.ev_done_cur[older]; c->need_prev = true;
  c->ev_frame_cur = P.ev_done_cur[newest]; c->frame_wait_pending = true;
  // the kernels that advance the ring origins write them into that frame's snapshot as well (behind the same wait
  // as their ring writes); a copy of all origins only when some ring did not move at all
  c->snap_target = P.snap[older]; c->snap_complete = true;
  c->ride_ev = c->ev_moved; c->rode = false;
  int rc = move_atmos_now(c, st, 0, st->nenv, accumx, accumy, (void *)c->atm_stream);
  c->ride_ev = nullptr;
  const bool complete = c->snap_complete;
  c->snap_target = nullptr;
  if (!rc && !complete && c->need_prev && c->frame_wait_pending)   // nothing written yet: the copy overwrites what the older frame reads
    rc = hipStreamWaitEvent(c->atm_stream, c->ev_frame_prev, 0) == hipSuccess ? 0 : fail("frame pipeline: hipStreamWaitEvent failed");
  c->ev_frame_prev = nullptr; c->need_prev = false; c->frame_wait_pending = false; c->group_overlap = false;
  if (rc) return rc;
  if (!complete)
    HIPCHK(hipMemcpyAsync(P.snap[older], st->origin, sizeof(int32_t) * (size_t)st->nenv * c->nlayers * 2,
                          hipMemcpyDeviceToDevice, c->atm_stream));
  c->screens_dirty_main = false;
  if (!(c->rode && complete)) HIPCHK(hipEventRecord(c->ev_moved, c->atm_stream));     // (else the move's one launch carried it)
  c->premoved = true; c->pre_screens = st->screens; c->pre_b = 0; c->pre_n = st->nenv;
  return 0;
}

static int env_step_pipelined(aomarl_ctx *c, aomarl_state *st, aomarl_env_glue *g, const float *action, float gain,
                              float *accumx, float *accumy, float *state_out, float *reward_out, void *stream) {
  auto &P = c->pipe;
  hipStream_t s = (hipStream_t)stream;
  const int n = st->nenv;
  int ktt = -1;
  env_step_fusable(c, g, &ktt);
  if (!P.active) {
    // ---- first step: the plain order, then the next frame ahead
    int rc = env_step_body(c, st, g, action, gain, accumx, accumy, state_out, reward_out, stream);
    if (rc) return rc;
    if (!c->premoved || !c->psf_side) return 0;      // (the plain step did not leave the steady state behind: stay plain)
    rc = pipe_init(c, st);
    if (rc) return rc;
    P.ev_done_cur[0] = c->ev_frame_cur;              // the plain frame used st's buffers: parity 0
    HIPCHK(hipEventRecord(P.ev_psf[0], c->psf_stream));
    P.psf_out[0] = true; P.psf_out[1] = false;
    c->psf_side = false;
    // v(t+1) = c(t), the newest entry of the delay line after its shift; tip-tilt shape from it
    aomarl_state v1 = pipe_view(c, st, 1);
    HIPCHK(hipMemcpyAsync(v1.voltage, st->com1, sizeof(float) * (size_t)n * st->ld_actu, hipMemcpyDeviceToDevice, s));
    Work w = work_layout(c, st->nenv);
    hipLaunchKernelGGL(k_post_delay, dim3(2 * n), dim3(256), 0, s, c->sys, dev_state(&v1), 0, n, v1.work + w.PEND, 0, ktt,
                       v1.voltage, st->ld_actu);
    LAUNCHCHK();
    HIPCHK(hipEventRecord(P.ev_cmd, s));
    HIPCHK(hipEventRecord(P.ev_commit, s));
    P.cmd_covers_commit = false;
    // the origins of the move the plain step prefetched
    HIPCHK(hipMemcpyAsync(P.snap[1], st->origin, sizeof(int32_t) * (size_t)n * c->nlayers * 2, hipMemcpyDeviceToDevice,
                          c->atm_stream));
    HIPCHK(hipEventRecord(c->ev_moved, c->atm_stream));
    c->pipe_internal = true;
    rc = pipe_launch_frame(c, st, 1, stream);
    if (!rc) rc = pipe_prefetch(c, st, accumx, accumy, 0, 1);
    c->pipe_internal = false;
    if (rc) return rc;
    P.active = true; P.par = 1;
    return 0;
  }
  // ---- steady state: the frame of parity p is in flight
  const int p = P.par, q = 1 - p;
  c->pipe_internal = true;
  int rc = env_step_validate(c, st, g, action, state_out, reward_out);
  aomarl_state vq = pipe_view(c, st, q), vp = pipe_view(c, st, p);
  hipEvent_t pe = P.psf_out[q] ? P.ev_psf[q] : nullptr;     // the PSF finish of the last frame of parity q
  if (!rc) rc = env_step_head_fused(c, st, &vq, g, action, gain, reward_out, ktt, true, pe, stream);
  // the frame stream is released behind k_delay_ahead, in front of the Strehl commit.  The frame kernel overwrites the
  // PSF rows that parity's last PSF finish reads: the chain waited for that finish in front of its first kernel
  // (cmd_covers_psf; the small chain's one kernel waits for it too: cmd_covers_commit), else the frame stream does
  if (!rc && pe && !P.cmd_covers_commit && !P.cmd_covers_psf && hipStreamWaitEvent(P.fstream, pe, 0) != hipSuccess) rc = fail("frame pipeline: hipStreamWaitEvent failed");
  if (!rc) { P.psf_out[q] = false; P.cmd_covers_psf = false; rc = pipe_launch_frame(c, st, q, stream); }
  if (!rc) rc = pipe_prefetch(c, st, accumx, accumy, p, q);
  // ---- reduce frame p
  if (!rc && hipStreamWaitEvent(s, P.ev_done_cur[p], 0) != hipSuccess) rc = fail("frame pipeline: hipStreamWaitEvent failed");
  if (!rc && small_chain_ok(c, g)) rc = env_step_tail(c, st, g, true, state_out, stream, &vp);
  else if (!rc && env_step_shortcut(c, g)) rc = env_step_tail(c, st, g, true, state_out, stream, nullptr, &vp);
  else {
    if (!rc) rc = aomarl_do_control(c, &vp, 0, n, stream);
    if (!rc) rc = env_step_tail(c, st, g, true, state_out, stream);
  }
  c->pipe_internal = false;
  if (rc) return rc;
  P.par = q; P.steps++;
  return 0;
}

// everything the pipeline has in flight joins `stream`; the frame in flight is dropped (full-range reset)
static int pipe_drop(aomarl_ctx *c, void *stream) {
  auto &P = c->pipe;
  hipStream_t s = (hipStream_t)stream;
  if (P.ev_done_cur[P.par]) HIPCHK(hipStreamWaitEvent(s, P.ev_done_cur[P.par], 0));
  for (int k = 0; k < 2; k++)
    if (P.psf_out[k]) { HIPCHK(hipStreamWaitEvent(s, P.ev_psf[k], 0)); P.psf_out[k] = false; }
  P.active = false; P.par = 0;
  return 0;
}

int aomarl_set_frame_pipeline(aomarl_ctx *c, const aomarl_state *st, const aomarl_state *twin) {
  if (!c) return fail("set_frame_pipeline: null context");
  auto &P = c->pipe;
  if (P.active) return fail("set_frame_pipeline: a frame is in flight (reset first)");
  if (!twin) { P.have_twin = false; P.owner_screens = nullptr; return 0; }
  if (!st) return fail("set_frame_pipeline: null state");
  int rc = check_range(c, st, 0, st->nenv);
  if (rc) return rc;
  if (twin->nenv != st->nenv || twin->ld_actu != st->ld_actu) return fail("set_frame_pipeline: the twin's nenv / ld_actu differ");
  if (twin->screens != st->screens || twin->origin != st->origin || twin->seeds != st->seeds || twin->ext_count != st->ext_count ||
      twin->com != st->com || twin->com1 != st->com1 || twin->com2 != st->com2 || twin->err != st->err ||
      twin->strehl != st->strehl || twin->le_img != st->le_img || twin->frame != st->frame)
    return fail("set_frame_pipeline: the twin must share every buffer of the state except slopes, voltage, dm_shape, work");
  if (!twin->slopes || !twin->voltage || !twin->dm_shape || !twin->work || twin->slopes == st->slopes ||
      twin->voltage == st->voltage || twin->dm_shape == st->dm_shape || twin->work == st->work)
    return fail("set_frame_pipeline: the twin needs slopes, voltage, dm_shape and work buffers of its own");
  P.twin = *twin; P.have_twin = true; P.owner_screens = st->screens;
  return 0;
}

int aomarl_frame_pipeline_state(aomarl_ctx *c, int *in_flight, int *consumed_in_twin, unsigned long long *steps,
                                unsigned long long *overlapped) {
  if (!c) return fail("frame_pipeline_state: null context");
  if (in_flight) *in_flight = c->pipe.active ? 1 : 0;
  if (consumed_in_twin) *consumed_in_twin = c->pipe.active ? (1 - c->pipe.par) : 0;
  if (steps) *steps = c->pipe.steps;
  if (overlapped) *overlapped = c->pipe.overlapped;
  return 0;
}



// ---------------------------------------------------------------- aomarl_env_step as a HIP graph ("graph_step")
// The launch sequence of one step depends on three things the host decides: the extrusion plan of the prefetched
// move (how many lines each layer moves this frame: 2 values per layer and axis), the position of the command ring,
// and the addresses of the caller's buffers.  One graph per distinct combination, captured from the very code path
// the plain call takes (env_step_body) the first time it occurs, replayed afterwards.  Inside a graph the side
// streams fork from the caller's stream in front of the frame kernel and join it again at the end: the next step's
// head (compose .. Strehl commit) therefore starts after this step's extrusions -- a dependency the plain path
// does not have (there the extrusion chain runs on beside the next step's head), which is why this mode is for
// the launch-bound regime (small batches: 10x10 / 64 environments is host-bound at ~0.16 ms per step) and off by
// default.  Results are identical: same kernels, same arguments, same order per stream.
static bool step_plan_uniform(const aomarl_ctx *c, int n, const float *accumx, const float *accumy, Plan &p) {
  const int nl = c->nlayers;
  for (int e = 0; e < n; e++)
    for (int l = 0; l < nl; l++) {
      const int kx = (int)(accumx[(size_t)e * nl + l] + c->deltax[l]), ky = (int)(accumy[(size_t)e * nl + l] + c->deltay[l]);
      if (e == 0) { p.kx[l] = kx; p.ky[l] = ky; }
      else if (p.kx[l] != kx || p.ky[l] != ky) return false;
    }
  return true;
}

// choose_action + env_step from C: one host round trip instead of two.  (A workgroup-per-environment kernel that ran
// the actors as matrix-vector products inside the small systems' head launch was built and measured in round 5: 64
// workgroups streaming the same megabyte of weights in lock-step took 130 us against 23 + 11 us for k_actor_fused +
// k_small_head -- the 16-environment matrix tiles of k_actor_fused read every weight four times, not sixty-four;
// LAB_NOTEBOOK section 11.)
int aomarl_policy_env_step(aomarl_ctx *c, aomarl_state *st, aomarl_env_glue *g, const aomarl_actor_desc *d,
                           const float *state, const float *eps, uint32_t seed, uint32_t counter, float gain,
                           float *accumx, float *accumy, float *action, float *mean, float *state_out,
                           float *reward_out, void *stream) {
  if (!c || !st || !g || !d || !state || !action || !mean || !state_out) return fail("policy_env_step: null argument");
  if (d->nenv != st->nenv) return fail("policy_env_step: the actors are set up for %d environments, the state has %d", d->nenv, st->nenv);
  int rc = aomarl_actor_forward(d, state, eps, seed, counter, action, mean, stream);
  if (rc) return rc;
  return aomarl_env_step(c, st, g, action, gain, accumx, accumy, state_out, reward_out, stream);
}

int aomarl_do_control_reduced(aomarl_ctx *c, aomarl_state *st, void *stream) {
  if (!c || !st) return fail("do_control_reduced: null argument");
  auto &P = c->pipe;
  if (!(P.active && P.owner_screens == st->screens)) return aomarl_do_control(c, st, 0, st->nenv, stream);
  aomarl_state v = pipe_view(c, st, 1 - P.par);       // the frame the last call reduced (the other parity is in flight)
  c->pipe_internal = true;
  const int rc = aomarl_do_control(c, &v, 0, st->nenv, stream);
  c->pipe_internal = false;
  return rc;
}

int aomarl_env_step_shortcut(aomarl_ctx *c, const aomarl_env_glue *g) {
  if (!c || !g) return 0;
  return (!g->denoiser && env_step_fusable(c, g, nullptr) && !small_chain_ok(c, g) && env_step_shortcut(c, g)) ? 1 : 0;
}

int aomarl_env_step(aomarl_ctx *c, aomarl_state *st, aomarl_env_glue *g, const float *action, float gain,
                    float *accumx, float *accumy, float *state_out, float *reward_out, void *stream) {
  if (!c || !st || !g || !state_out) return fail("env_step: null argument");
  if (pipe_eligible(c, st, g, accumx, accumy))
    return env_step_pipelined(c, st, g, action, gain, accumx, accumy, state_out, reward_out, stream);
  if (c->pipe.active && c->pipe.owner_screens == st->screens)
    return fail("env_step: a pipelined frame is in flight but this call is not eligible for the frame pipeline "
                "(options, glue or arguments changed within an episode): reset first");
  if (!c->graph_step || c->capturing) return env_step_body(c, st, g, action, gain, accumx, accumy, state_out, reward_out, stream);
  const int n = st->nenv, nl = c->nlayers;
  // the steady state only: a prefetched move of exactly this batch is pending, the glue has been validated by a
  // plain call, every environment moves by the same plan
  Plan plan;
  // (the null stream cannot be captured: a caller on it gets the plain path)
  // (with "prefetch_atmos" off the whole step is ONE stream: a linear graph, no fork / join -- the form that replays
  // cheaply on this runtime, tools/graphbench.hip: 13 small kernels 29 us per replay against 37 us launched one by one)
  const bool pf = c->prefetch_atmos;
  const bool steady = stream && aomarl_frame_fused_available(c) && accumx && accumy &&
                      (pf ? (c->premoved && c->pre_screens == st->screens && c->pre_b == 0 && c->pre_n == n) : !c->premoved) &&
                      (!g->sel || (c->sel_checked == g->sel && c->sel_checked_n == g->dm_dim && c->sel_checked_nm == g->nmodes)) &&
                      g->nhist >= 0 && g->nhist <= 5 && g->ring_pos >= 0 && g->ring_pos <= g->nhist &&
                      step_plan_uniform(c, n, accumx, accumy, plan);
  if (!steady) return env_step_body(c, st, g, action, gain, accumx, accumy, state_out, reward_out, stream);
  hipStream_t s = (hipStream_t)stream;
  int rc = 0;
  if (pf) {
    rc = side_stream(c);
    if (rc) return rc;
    if (!c->side_joined) {        // work issued on the side streams by plain calls: wait for it OUTSIDE the graph
      HIPCHK(hipStreamWaitEvent(s, c->ev_moved, 0));
      if (c->psf_side) HIPCHK(hipStreamWaitEvent(s, c->ev_psf, 0));
      c->side_joined = true;
    }
  } else if (c->psf_side) {       // a PSF finish left on the side stream by an earlier call with the prefetch on
    rc = psf_wait_pending(c, stream);
    if (rc) return rc;
  }
  std::vector<long long> key;
  auto kp = [&](const void *p) { key.push_back((long long)(uintptr_t)p); };
  for (int l = 0; l < nl; l++) { key.push_back(plan.kx[l]); key.push_back(plan.ky[l]); }
  key.push_back(g->ring_pos); key.push_back(g->nhist); key.push_back(g->nmodes); key.push_back(g->dm_dim);
  key.push_back(g->n_agents); key.push_back(g->flags); key.push_back(g->denoiser_f32);
  { int gi; memcpy(&gi, &gain, sizeof(gi)); key.push_back(gi); memcpy(&gi, &g->reward_factor, sizeof(gi)); key.push_back(gi); }
  kp(st); kp(st->screens); kp(st->com); kp(st->voltage); kp(st->slopes); kp(st->work); kp(st->bincube); kp(st->strehl);
  kp(action); kp(state_out); kp(reward_out); kp(stream);
  kp(g->sel); kp(g->mean_dm); kp(g->std_dm); kp(g->mean_res); kp(g->std_res); kp(g->lohi); kp(g->modes_ring); kp(g->res_modes);
  kp(g->denoiser); kp(c->cmat); kp(c->v2m); kp(c->m2v); kp(c->freedom); kp(c->amode_inv);
  key.push_back(n); key.push_back(g_precision); key.push_back(g_gemm_split_f16 ? 1 : 0); key.push_back(c->dft_mode);
  key.push_back(pf ? 1 : 0); key.push_back(c->small_move); key.push_back(c->small_chain); key.push_back(c->residual_shortcut);
  key.push_back(c->defer_dm_shape ? 1 : 0); key.push_back(g_gemm_target_blocks); key.push_back(c->fused_debug);
  { int gi; memcpy(&gi, &c->gain, sizeof(gi)); key.push_back(gi); }
  key.push_back((long long)c->cfg_epoch); key.push_back((long long)g_cfg_epoch);
  aomarl_ctx::StepGraph *hit = nullptr;
  for (auto &sg : c->graphs)
    if (sg.key == key) { hit = &sg; break; }
  if (hit) {
    HIPCHK(hipGraphLaunch(hit->exec, s));
    // the host bookkeeping the body does: wind accumulators, ring position, what is pending where
    for (int e = 0; e < n; e++)
      for (int l = 0; l < nl; l++) {
        const float ax = accumx[(size_t)e * nl + l] + c->deltax[l], ay = accumy[(size_t)e * nl + l] + c->deltay[l];
        accumx[(size_t)e * nl + l] = ax - (float)(int)ax;
        accumy[(size_t)e * nl + l] = ay - (float)(int)ay;
        if (e == 0) { c->frac_x[l] = ax - (float)(int)ax; c->frac_y[l] = ay - (float)(int)ay; }
      }
    g->ring_pos = (g->ring_pos + 1) % (g->nhist + 1);
    if (pf) {
      c->premoved = true; c->psf_side = true; c->side_joined = true;
      c->frame_marked = true; c->frame_wait_pending = false; c->screens_dirty_main = false;
    } else {
      c->frame_marked = false; c->screens_dirty_main = true;
    }
    for (int i = 0; i < AR_N; i++) g_arith[i] += hit->arith[i];
    memcpy(c->fw_variant, hit->fw_variant, sizeof(c->fw_variant));
    c->graph_hits++;
    return 0;
  }
  // ---- capture
  if (g_gemm_split_f16) (void)gemm_sat_counter();          // nothing may allocate during the capture
  if (c->graphs.size() >= 256) {                            // a caller that cycles through many buffers: start over
    for (auto &sg : c->graphs) { (void)hipGraphExecDestroy(sg.exec); (void)hipGraphDestroy(sg.graph); }
    c->graphs.clear();
  }
  unsigned long long before[AR_N];
  for (int i = 0; i < AR_N; i++) before[i] = g_arith[i];
  // the body advances host bookkeeping while it is being RECORDED (no kernel runs): kept, so that a capture that
  // fails leaves the host where the device still is
  struct Snap {
    std::vector<float> ax, ay; int ring_pos; bool premoved, psf_side, side_joined, frame_marked, frame_wait_pending, screens_dirty_main;
    float fx[AOMARL_MAX_LAYERS], fy[AOMARL_MAX_LAYERS];
  } snap;
  snap.ax.assign(accumx, accumx + (size_t)n * nl); snap.ay.assign(accumy, accumy + (size_t)n * nl);
  snap.ring_pos = g->ring_pos; snap.premoved = c->premoved; snap.psf_side = c->psf_side; snap.side_joined = c->side_joined;
  snap.frame_marked = c->frame_marked; snap.frame_wait_pending = c->frame_wait_pending; snap.screens_dirty_main = c->screens_dirty_main;
  memcpy(snap.fx, c->frac_x, sizeof(snap.fx)); memcpy(snap.fy, c->frac_y, sizeof(snap.fy));
  auto restore = [&]() {
    memcpy(accumx, snap.ax.data(), sizeof(float) * snap.ax.size()); memcpy(accumy, snap.ay.data(), sizeof(float) * snap.ay.size());
    g->ring_pos = snap.ring_pos; c->premoved = snap.premoved; c->psf_side = snap.psf_side; c->side_joined = snap.side_joined;
    c->frame_marked = snap.frame_marked; c->frame_wait_pending = snap.frame_wait_pending; c->screens_dirty_main = snap.screens_dirty_main;
    memcpy(c->frac_x, snap.fx, sizeof(snap.fx)); memcpy(c->frac_y, snap.fy, sizeof(snap.fy));
    for (int i = 0; i < AR_N; i++) g_arith[i] = before[i];
  };
  HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
  c->capturing = true; c->fork_recorded = false;
  rc = env_step_body(c, st, g, action, gain, accumx, accumy, state_out, reward_out, stream);
  hipError_t je = hipSuccess;
  if (!rc && pf) {                                          // the side streams join the caller's stream again
    if (c->psf_side) je = hipStreamWaitEvent(s, c->ev_psf, 0);
    if (je == hipSuccess && c->premoved) je = hipStreamWaitEvent(s, c->ev_moved, 0);
  }
  c->capturing = false;
  hipGraph_t graph = nullptr;
  const hipError_t ee = hipStreamEndCapture(s, &graph);
  if (rc) { if (graph) (void)hipGraphDestroy(graph); restore(); return rc; }
  if (je != hipSuccess || ee != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    restore();
    return fail("env_step: graph capture failed: %s", hipGetErrorString(je != hipSuccess ? je : ee));
  }
  if (pf) c->side_joined = true;
  aomarl_ctx::StepGraph sg;
  sg.key = key; sg.graph = graph; sg.exec = nullptr;
  for (int i = 0; i < AR_N; i++) sg.arith[i] = g_arith[i] - before[i];
  memcpy(sg.fw_variant, c->fw_variant, sizeof(sg.fw_variant));
  {
    const hipError_t ie = hipGraphInstantiate(&sg.exec, graph, nullptr, nullptr, 0);
    if (ie != hipSuccess) {
      (void)hipGraphDestroy(graph);
      restore();
      return fail("env_step: hipGraphInstantiate failed: %s", hipGetErrorString(ie));
    }
  }
  {
    const hipError_t le = hipGraphLaunch(sg.exec, s);
    if (le != hipSuccess) {
      (void)hipGraphExecDestroy(sg.exec); (void)hipGraphDestroy(graph);
      restore();
      return fail("env_step: hipGraphLaunch failed: %s", hipGetErrorString(le));
    }
  }
  c->graphs.push_back(sg);
  c->graph_captures++;
  return 0;
}

int aomarl_graph_stats(aomarl_ctx *c, unsigned long long *captures, unsigned long long *replays) {
  if (!c || !captures || !replays) return fail("graph_stats: null argument");
  *captures = c->graph_captures; *replays = c->graph_hits;
  return 0;
}
