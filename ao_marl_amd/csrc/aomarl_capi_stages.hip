// aomarl_capi_stages.hip -- part of the C ABI implementation (included by aomarl_capi.hip, one translation unit):
// the per-stage entry points: DM shapes, raytraces, WFS image / centroids, controller, target PSF / Strehl (A3 - A10).
// ---------------------------------------------------------------- DMs
// skip_stack: leave the stack-array planes alone (their phase will be evaluated from st->voltage
// inside the one-pass frame kernel); the tip-tilt slot (commands + pivot) is always refreshed
static int dm_shape_impl(aomarl_ctx *c, aomarl_state *st, int b, int n, const float *volts,
                         bool skip_stack, void *stream) {
  const float *v = volts ? volts : st->voltage + (size_t)b * st->ld_actu;
  const int ldv = volts ? c->sys.nactu : st->ld_actu;
  DevState ds = dev_state(st);
  for (int k = 0; k < c->ndm; k++) {
    const DevDm &D = c->sys.dms[k];
    const int np = D.dim * D.dim;
    if (D.type == AOMARL_DM_TT)
      hipLaunchKernelGGL(k_dm_shape, dim3(1, n), dim3(64), 0, (hipStream_t)stream, c->sys, ds, b, k, v, ldv);
    else if (skip_stack)
      continue;
    else if (D.sep && !c->force_generic_dm)
      hipLaunchKernelGGL(k_dm_shape_sep, dim3((D.dim + DMS_TX - 1) / DMS_TX, (D.dim + DMS_TY - 1) / DMS_TY, n),
                         dim3(256), 0, (hipStream_t)stream, c->sys, ds, b, k, v, ldv);
    else
      hipLaunchKernelGGL(k_dm_shape, dim3((np + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, c->sys, ds, b, k, v, ldv);
    LAUNCHCHK();
  }
  return 0;
}

int aomarl_comp_dm_shape(aomarl_ctx *c, aomarl_state *st, int b, int n, const float *volts, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (n == 0) return 0;
  return dm_shape_impl(c, st, b, n, volts, false, stream);
}

int aomarl_dm_from_voltage_available(aomarl_ctx *c) {
  return c && c->sys.fused_ok && c->sys.otf_ok && !c->force_unfused_frame ? 1 : 0;
}

int aomarl_get_dm_shape(aomarl_ctx *c, aomarl_state *st, int b, int n, int k, float *dst, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (k < 0 || k >= c->ndm || !dst) return fail("get_dm_shape: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_get_dm_shape, dim3(256, n), dim3(256), 0, (hipStream_t)stream, c->sys, dev_state(st), b, k, dst);
  LAUNCHCHK();
  return 0;
}

int aomarl_set_option(aomarl_ctx *c, const char *name, int value) {
  if (!name) return fail("set_option: null argument");
  g_cfg_epoch++;
  if (c) c->cfg_epoch++;
  if (!strcmp(name, "gemm_kgroups")) {          // process-wide, no context needed
    if (value != 0 && value != 1 && value != 2 && value != 4) return fail("gemm_kgroups: 0, 1, 2 or 4");
    g_gemm_kgroups = value;
    return 0;
  }
  if (!strcmp(name, "gemm_xcd_map")) { g_gemm_xcd = value != 0; return 0; }   // process-wide
  if (!strcmp(name, "gemm_target_blocks")) { g_gemm_target_blocks = value > 0 ? value : 0; return 0; }   // process-wide
  if (!strcmp(name, "gemm_split_f16")) { g_gemm_split_f16 = value != 0; return 0; }          // process-wide
  if (!strcmp(name, "precision")) return aomarl_set_precision(value);                        // process-wide
  if (!c) return fail("set_option: null context");
  if (!strcmp(name, "force_generic_dm")) { c->force_generic_dm = value != 0; return 0; }
  if (!strcmp(name, "force_valu_target")) { c->force_valu_target = value != 0; return 0; }
  if (!strcmp(name, "defer_dm_shape")) { c->defer_dm_shape = value != 0; return 0; }
  if (!strcmp(name, "frame_pipeline")) {
    if (c->pipe.active) return fail("frame_pipeline: a frame is in flight (reset first)");
    c->pipe_enabled = value != 0; return 0;
  }
  if (!strcmp(name, "small_move")) { c->small_move = value != 0; return 0; }
  if (!strcmp(name, "small_chain")) { c->small_chain = value != 0; return 0; }
  if (!strcmp(name, "renew_frame_stream")) {
    // the frame pipeline's stream, anew: the runtime deals its hardware queues out as streams come, and a frame stream
    // that shares one with another stream of the step serialises the pipelined order (VecAoEnv's probe asks for this)
    auto &P = c->pipe;
    if (P.active) return fail("renew_frame_stream: a frame is in flight (reset first)");
    if (P.fstream) {
      HIPCHK(hipStreamSynchronize(P.fstream));
      HIPCHK(hipStreamDestroy(P.fstream));
      P.fstream = nullptr;
      HIPCHK(hipStreamCreateWithFlags(&P.fstream, hipStreamNonBlocking));
    }
    return 0;
  }
  if (!strcmp(name, "residual_shortcut")) {
    if (value && (!c->s2m || c->s2m_nmodes < 1)) return fail("residual_shortcut: no v2m . cmat matrix (aomarl_set_slopes2modes)");
    c->residual_shortcut = value != 0;
    return 0;
  }
  if (!strcmp(name, "reset_streams")) { c->reset_streams = value < 1 ? 1 : (value > 4 ? 4 : value); return 0; }
  if (!strcmp(name, "reset_prefetch_whole")) { c->reset_prefetch_whole = value != 0; return 0; }
  if (!strcmp(name, "extrude_unfused")) { c->no_extrude_sg = value != 0; return 0; }
  if (!strcmp(name, "reset_untransposed")) { c->reset_untransposed = value != 0; return 0; }
  if (!strcmp(name, "time_frame_kernel")) {
    // value = number of launches to keep event pairs for (0: off)
    c->time_fw = value > 0;
    { const int rrc = fw_ev_rewind(c); if (rrc) return rrc; }
    while (c->fw_ev.size() < 2 * (size_t)std::max(value, 0)) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      c->fw_ev.push_back(e);
    }
    return 0;
  }
  if (!strcmp(name, "prefetch_atmos")) { c->prefetch_atmos = value != 0; return 0; }
  if (!strcmp(name, "subpixel_flow")) { c->subpixel_flow = value != 0; return 0; }
  if (!strcmp(name, "graph_step")) { c->graph_step = value != 0; return 0; }
  if (!strcmp(name, "fused_debug")) { c->fused_debug = value; return 0; }
  if (!strcmp(name, "force_f32_dft")) { c->dft_mode = value < 0 ? -1 : (value != 0 ? 0 : 1); return 0; }
  if (!strcmp(name, "force_unfused_frame")) { c->force_unfused_frame = value != 0; return 0; }
  if (!strcmp(name, "force_generic_spot")) { c->force_generic_spot = value != 0; return 0; }
  if (!strcmp(name, "force_generic_target")) { c->force_generic_target = value != 0; return 0; }
  return fail("set_option: unknown option %s", name);
}

// ---------------------------------------------------------------- raytrace (unfused API)
// the static description with the layer windows moved by the wind accumulators' remainder ("subpixel_flow")
static DevSys traced_sys(const aomarl_ctx *c) {
  DevSys sy = c->sys;
  if (c->subpixel_flow)
    for (int l = 0; l < c->nlayers; l++) {
      sy.layers[l].wxo += c->frac_x[l]; sy.layers[l].txo += c->frac_x[l];
      sy.layers[l].wyo += c->frac_y[l]; sy.layers[l].tyo += c->frac_y[l];
    }
  return sy;
}

int aomarl_raytrace_wfs(aomarl_ctx *c, aomarl_state *st, int b, int n, int flags, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  rc = atmos_wait_pending(c, stream);
  if (rc) return rc;
  if (!st->wfs_phase) return fail("raytrace_wfs needs st->wfs_phase");
  if (n == 0) return 0;
  const int np = c->sys.n * c->sys.n;
  hipLaunchKernelGGL(k_raytrace<false>, dim3((np + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, traced_sys(c), dev_state(st), b, flags);
  LAUNCHCHK();
  return 0;
}

int aomarl_raytrace_target(aomarl_ctx *c, aomarl_state *st, int b, int n, int flags, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  rc = atmos_wait_pending(c, stream);
  if (rc) return rc;
  if (!st->tar_phase) return fail("raytrace_target needs st->tar_phase");
  if (n == 0) return 0;
  const int np = c->sys.pupdiam * c->sys.pupdiam;
  hipLaunchKernelGGL(k_raytrace<true>, dim3((np + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, traced_sys(c), dev_state(st), b, flags);
  LAUNCHCHK();
  return 0;
}

// ---------------------------------------------------------------- WFS
__global__ void k_inc_u32(uint32_t *p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1u;
}

int aomarl_comp_image(aomarl_ctx *c, aomarl_state *st, int b, int n, int flags, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  rc = atmos_wait_pending(c, stream);
  if (rc) return rc;
  if (n == 0) return 0;
  const bool from_buf = flags & AOMARL_IMG_FROM_PHASE_BUFFER;
  const bool noise = (flags & AOMARL_IMG_NOISE) && c->sys.noise >= 0.f;
  const bool cube = flags & AOMARL_IMG_WRITE_BINCUBE;
  const int cog = (flags & AOMARL_IMG_COG) ? 1 : 0;
  if (from_buf && !st->wfs_phase) return fail("comp_image: FROM_PHASE_BUFFER needs st->wfs_phase");
  if (!from_buf && !c->sys.wfs_all_int)
    return fail("comp_image: fused raytrace needs integer layer offsets; use raytrace_wfs + FROM_PHASE_BUFFER");
  if (cube && !st->bincube) return fail("comp_image: WRITE_BINCUBE needs st->bincube");
  if (!cube && !cog) return fail("comp_image: nothing to produce (neither bincube nor slopes)");
  const int na = (flags & AOMARL_IMG_NO_ATMOS) ? 1 : 0, nd = (flags & AOMARL_IMG_NO_DMS) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  DevState ds = dev_state(st);
  // persistent waves: enough blocks per environment to fill the chip ~2x (256 CUs x 32 waves)
  int gx = (16384 + 4 * n - 1) / (4 * n);
  gx = std::max(1, std::min(gx, (c->sys.nvalid + 3) / 4));
  dim3 grid(gx, n), blk(256);
#define SPOT(FB, NZ, WC) hipLaunchKernelGGL((k_wfs_spot<FB, NZ, WC>), grid, blk, 0, s, c->sys, ds, b, na, nd, cog)
#define FAST(NL, NZ, WC) hipLaunchKernelGGL((k_wfs_spot_fast<NL, NZ, WC>), grid, blk, 0, s, c->sys, ds, b, cog)
  const bool fast_ok = !from_buf && !na && !nd && !c->force_generic_spot && c->ndm == 2 &&
                       c->sys.dms[0].type == AOMARL_DM_PZT && c->sys.dms[1].type == AOMARL_DM_TT &&
                       (c->nlayers == 1 || c->nlayers == 3);
  if (fast_ok) {
    if (c->nlayers == 1) {
      if (noise) { if (cube) FAST(1, true, true); else FAST(1, true, false); }
      else { if (cube) FAST(1, false, true); else FAST(1, false, false); }
    } else {
      if (noise) { if (cube) FAST(3, true, true); else FAST(3, true, false); }
      else { if (cube) FAST(3, false, true); else FAST(3, false, false); }
    }
  } else if (from_buf) {
    if (noise) { if (cube) SPOT(true, true, true); else SPOT(true, true, false); }
    else { if (cube) SPOT(true, false, true); else SPOT(true, false, false); }
  } else {
    if (noise) { if (cube) SPOT(false, true, true); else SPOT(false, true, false); }
    else { if (cube) SPOT(false, false, true); else SPOT(false, false, false); }
  }
#undef SPOT
#undef FAST
  LAUNCHCHK();
  hipLaunchKernelGGL(k_inc_u32, dim3((n + 255) / 256), dim3(256), 0, s, st->frame + b, n);
  LAUNCHCHK();
  return 0;
}

int aomarl_do_centroids(aomarl_ctx *c, aomarl_state *st, int b, int n, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (!st->bincube) return fail("do_centroids needs st->bincube");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_cog, dim3((c->sys.nvalid + 3) / 4, n), dim3(256), 0, (hipStream_t)stream, c->sys, dev_state(st), b);
  LAUNCHCHK();
  return 0;
}

int aomarl_slopes_geom(aomarl_ctx *c, aomarl_state *st, int b, int n, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (!st->wfs_phase) return fail("slopes_geom needs st->wfs_phase");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_slopes_geom, dim3((c->sys.nvalid + 3) / 4, n), dim3(256), 0, (hipStream_t)stream, c->sys, dev_state(st), b);
  LAUNCHCHK();
  return 0;
}

// ---------------------------------------------------------------- controller
int aomarl_do_control(aomarl_ctx *c, aomarl_state *st, int b, int n, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (!c->cmat) return fail("do_control: no command matrix (aomarl_set_cmat)");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int na = c->sys.nactu, nsl = c->sys.nslope;
  // err[env][a] = - sum_s slopes[env][s] cmat[a][s]
  Work w = work_layout(c, st->nenv);
  GemmEpi ep = {};
  if (c->env_gain && c->env_gain_n != st->nenv)
    return fail("do_control: %d per-environment gains set, the state has %d environments", c->env_gain_n, st->nenv);
  ep.mode = 1; ep.com = st->com + (size_t)b * st->ld_actu; ep.ldcom = st->ld_actu; ep.gain = c->gain;
  ep.gain_row = c->env_gain ? c->env_gain + b : nullptr;
  const bool fused = launch_gemm_nt(n, na, nsl, -1.0f, st->slopes + (size_t)b * nsl, nsl, c->cmat, c->ld_cmat, 0.0f,
                                    st->err + (size_t)b * st->ld_actu, st->ld_actu, s, st->work + w.GEMM,
                                    w.gemm_floats, &ep, nullptr, /* slopes (arcsec): unscaled, saturation only beyond 65504" */ true, 1.f, c->cmat_scale, nullptr, 288);
  LAUNCHCHK();
  if (!fused) {
    hipLaunchKernelGGL(k_integrate, dim3((na + 255) / 256, n), dim3(256), 0, s, st->com, st->err, na, st->ld_actu, c->gain, b, c->env_gain);
    LAUNCHCHK();
  }
  return 0;
}

int aomarl_set_com(aomarl_ctx *c, aomarl_state *st, int b, int n, const float *com, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (!com) return fail("set_com: null command");
  if (n == 0) return 0;
  const int na = c->sys.nactu;
  hipLaunchKernelGGL(k_copy_rows, dim3((na + 255) / 256, n), dim3(256), 0, (hipStream_t)stream,
                     st->com + (size_t)b * st->ld_actu, st->ld_actu, com, na, na);
  LAUNCHCHK();
  return 0;
}

int aomarl_volts2modes(aomarl_ctx *c, aomarl_state *st, int nrows, const float *vec, int ldvec,
                       float *modes, void *stream) {
  if (!c || !c->v2m) return fail("volts2modes: no modal basis (aomarl_set_modal)");
  if (!vec || !modes) return fail("volts2modes: null argument");
  if (ldvec < c->sys.nactu) return fail("volts2modes: ldvec < nactu");
  float *ws = nullptr;
  size_t wsn = 0;
  if (st && st->work) { Work w = work_layout(c, st->nenv); ws = st->work + w.GEMM; wsn = w.gemm_floats; }
  launch_gemm_nt(nrows, c->nmodes, c->sys.nactu, 1.0f, vec, ldvec, c->v2m, c->ld_v2m, 0.0f,
                 modes, c->nmodes, (hipStream_t)stream, ws, wsn, nullptr, nullptr, /* volts */ true, 1.f, c->v2m_scale, nullptr, 288);
  LAUNCHCHK();
  return 0;
}

int aomarl_slopes2modes(aomarl_ctx *c, aomarl_state *st, int b, int n, float *modes, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (!c->s2m || c->s2m_nmodes < 1) return fail("slopes2modes: no matrix (aomarl_set_slopes2modes)");
  if (!modes) return fail("slopes2modes: null output");
  if (n == 0) return 0;
  Work w = work_layout(c, st->nenv);
  const int nsl = c->sys.nslope, ld = (nsl + 3) & ~3;
  // residual modes = v2m . err = -(v2m . cmat) . slopes
  launch_gemm_nt(n, c->s2m_nmodes, nsl, -1.0f, st->slopes + (size_t)b * nsl, nsl, c->s2m, ld, 0.0f, modes,
                 c->s2m_nmodes, (hipStream_t)stream, st->work + w.GEMM, w.gemm_floats, nullptr, nullptr,
                 /* slopes (arcsec), unscaled */ true, 1.f, c->s2m_scale, nullptr, 288);
  LAUNCHCHK();
  return 0;
}

int aomarl_rl_control(aomarl_ctx *c, aomarl_state *st, int b, int n, const float *action, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (!c->v2m || !c->m2v) return fail("rl_control: no modal basis (aomarl_set_modal)");
  if (c->nact <= 0) return fail("rl_control: no action modes set");
  if (!action) return fail("rl_control: null action");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  Work w = work_layout(c, st->nenv);
  float *modes = st->work + w.MODES;
  const int na = c->sys.nactu, nm = c->nmodes;
  float *com = st->com + (size_t)b * st->ld_actu;
  GemmEpi ep = {};
  ep.mode = 2; ep.action = action; ep.nact = c->nact; ep.amode_inv = c->amode_inv; ep.freedom = c->freedom;
  const bool fused = launch_gemm_nt(n, nm, na, 1.0f, com, st->ld_actu, c->v2m, c->ld_v2m, 0.0f, modes, w.ldm, s,
                                    st->work + w.GEMM, w.gemm_floats, &ep, nullptr, true, 1.f, c->v2m_scale, nullptr, 288);
  LAUNCHCHK();
  if (!fused) {
    hipLaunchKernelGGL(k_modal_add, dim3((c->nact + 255) / 256, n), dim3(256), 0, s, modes, w.ldm, action, c->nact, c->amodes, c->freedom);
    LAUNCHCHK();
  }
  launch_gemm_nt(n, na, nm, 1.0f, modes, w.ldm, c->m2v, c->ld_m2v, 0.0f, com, st->ld_actu, s, st->work + w.GEMM, w.gemm_floats,
                 nullptr, nullptr, /* Btt coordinates x 2^4 */ true, 16.f, c->m2v_scale, nullptr, 288);
  LAUNCHCHK();
  return 0;
}

// modes = m0 + g * m1 (+ action on the action modes), written to the GEMM operand and to modes_out
__global__ void k_modal_compose(int nm, const float *__restrict__ m0, const float *__restrict__ m1,
                                float g, const float *__restrict__ action, int nact,
                                const int32_t *__restrict__ amode_inv,
                                const float *__restrict__ freedom, float *__restrict__ modes, int ldm,
                                float *__restrict__ modes_out) {
  const int r = blockIdx.y, m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= nm) return;
  float v = m0[(long long)r * nm + m] + g * m1[(long long)r * nm + m];
  if (action) {
    const int j = amode_inv[m];
    if (j >= 0) v += action[(long long)r * nact + j] * freedom[m];
  }
  modes[(long long)r * ldm + m] = v;
  if (modes_out) modes_out[(long long)r * nm + m] = v;
}

// k_modal_compose and k_agent_rewards side by side in one launch (blocks beyond the compose range: one
// per agent, first wave): both read the residual modes, neither reads what the other writes
__global__ __launch_bounds__(256) void k_compose_rewards(int nm, const float *__restrict__ m0, const float *__restrict__ m1,
                                                         float g, const float *__restrict__ action, int nact,
                                                         const int32_t *__restrict__ amode_inv,
                                                         const float *__restrict__ freedom, float *__restrict__ modes, int ldm,
                                                         float *__restrict__ modes_out, int cx, int n_agents,
                                                         const int32_t *__restrict__ lohi, float factor,
                                                         float *__restrict__ rew) {
  CHAIN_SETPRIO();
  const int r = blockIdx.y;
  if ((int)blockIdx.x >= cx) {
    if (threadIdx.x >= 64) return;
    const int a = blockIdx.x - cx, lane = threadIdx.x;
    const int lo = lohi[2 * a], hi = lohi[2 * a + 1];
    float s = 0.f;
    for (int m = lo + lane; m < hi; m += 64) { const float v = m1[(long long)r * nm + m]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) rew[(long long)r * n_agents + a] = -factor * s / (float)(hi - lo);
    return;
  }
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= nm) return;
  float v = m0[(long long)r * nm + m] + g * m1[(long long)r * nm + m];
  if (action) {
    const int j = amode_inv[m];
    if (j >= 0) v += action[(long long)r * nact + j] * freedom[m];
  }
  modes[(long long)r * ldm + m] = v;
  if (modes_out) modes_out[(long long)r * nm + m] = v;
}

int aomarl_rl_control_modes(aomarl_ctx *c, aomarl_state *st, int b, int n, const float *m0,
                            const float *m1, float g, const float *action, float *modes_out,
                            void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (!c->v2m || !c->m2v) return fail("rl_control_modes: no modal basis (aomarl_set_modal)");
  if (!m0 || !m1) return fail("rl_control_modes: null modal vectors");
  if (action && c->nact <= 0) return fail("rl_control_modes: no action modes set");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  Work w = work_layout(c, st->nenv);
  float *modes = st->work + w.MODES;
  const int na = c->sys.nactu, nm = c->nmodes;
  hipLaunchKernelGGL(k_modal_compose, dim3((nm + 255) / 256, n), dim3(256), 0, s, nm, m0, m1, g, action,
                     c->nact, c->amode_inv, c->freedom, modes, w.ldm, modes_out);
  LAUNCHCHK();
  launch_gemm_nt(n, na, nm, 1.0f, modes, w.ldm, c->m2v, c->ld_m2v, 0.0f,
                 st->com + (size_t)b * st->ld_actu, st->ld_actu, s, st->work + w.GEMM, w.gemm_floats,
                 nullptr, nullptr, /* Btt coordinates x 2^4 */ true, 16.f, c->m2v_scale, nullptr, 288);
  LAUNCHCHK();
  return 0;
}

int aomarl_apply_control(aomarl_ctx *c, aomarl_state *st, int b, int n, int comp_voltage, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (n == 0) return 0;
  const float d = c->delay;
  float wa, wb, wc;
  if (d <= 1.f) { wa = 1.f - d; wb = d; wc = 0.f; } else { wa = 0.f; wb = 2.f - d; wc = d - 1.f; }
  const int na = c->sys.nactu;
  hipLaunchKernelGGL(k_delay, dim3((na + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, dev_state(st), na, st->ld_actu, wa, wb, wc, b, comp_voltage & AOMARL_APPLY_COMP_VOLTAGE);
  LAUNCHCHK();
  const bool defer = (comp_voltage & AOMARL_APPLY_DEFER_STACK_SHAPE) && aomarl_dm_from_voltage_available(c);
  return dm_shape_impl(c, st, b, n, nullptr, defer, stream);
}

// ---------------------------------------------------------------- target
static int target_psf_impl(aomarl_ctx *c, aomarl_state *st, int b, int n, bool from_buf, void *stream) {
  if (!from_buf && atmos_wait_pending(c, stream)) return 1;
  if (psf_wait_pending(c, stream)) return 1;
  hipStream_t s = (hipStream_t)stream;
  Work w = work_layout(c, st->nenv);
  const int W = 2 * c->sys.hw, RB = 256 / W;
  float *TR = st->work + w.TR + (size_t)b * c->sys.pupdiam * W * 2;
  float *TP = st->work + w.TPART + (size_t)b * w.nblk * 4;
  float *PEND = st->work + w.PEND + (size_t)b * (W * W + 4);
  DevState ds = dev_state(st);
  const bool tfast = c->sys.hw == 8 && !c->force_valu_target && !c->force_generic_target && !from_buf &&
                     c->ndm == 2 && c->sys.dms[0].type == AOMARL_DM_PZT && c->sys.dms[1].type == AOMARL_DM_TT &&
                     (c->nlayers == 1 || c->nlayers == 3);
  if (tfast) {
    size_t smm = sizeof(float) * (4 * 16 * 65 + 4 * 2 * 256) + (c->sys.npsf <= 4096 ? sizeof(float) * 2 * c->sys.npsf : 0);
    if (c->nlayers == 1)
      hipLaunchKernelGGL(k_target_rows_fast<1>, dim3(w.nblk, n), dim3(256), smm, s, c->sys, ds, b, TR, TP, w.nblk);
    else
      hipLaunchKernelGGL(k_target_rows_fast<3>, dim3(w.nblk, n), dim3(256), smm, s, c->sys, ds, b, TR, TP, w.nblk);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_target_finish_mfma, dim3(n), dim3(256), 0, s, c->sys, TR, TP, w.nblk, PEND, (uint32_t *)nullptr);
    LAUNCHCHK();
    return 0;
  }
  if (c->sys.hw == 8 && !c->force_valu_target) {
    size_t smm = sizeof(float) * (2 * 16 * 65 + 4 * 2 * 256) + (c->sys.npsf <= 4096 ? sizeof(float) * 2 * c->sys.npsf : 0);
    if (from_buf)
      hipLaunchKernelGGL(k_target_rows_mfma<true>, dim3(w.nblk, n), dim3(256), smm, s, c->sys, ds, b, TR, TP, w.nblk);
    else
      hipLaunchKernelGGL(k_target_rows_mfma<false>, dim3(w.nblk, n), dim3(256), smm, s, c->sys, ds, b, TR, TP, w.nblk);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_target_finish_mfma, dim3(n), dim3(256), 0, s, c->sys, TR, TP, w.nblk, PEND, (uint32_t *)nullptr);
    LAUNCHCHK();
    return 0;
  }
  size_t sm = sizeof(float) * (2 * RB * TGT_XC + 3 * 256) + (c->sys.npsf <= 4096 ? sizeof(float) * 2 * c->sys.npsf : 0);
  if (from_buf)
    hipLaunchKernelGGL(k_target_rows<true>, dim3(w.nblk, n), dim3(256), sm, s, c->sys, ds, b, TR, TP, w.nblk);
  else
    hipLaunchKernelGGL(k_target_rows<false>, dim3(w.nblk, n), dim3(256), sm, s, c->sys, ds, b, TR, TP, w.nblk);
  LAUNCHCHK();
  hipLaunchKernelGGL(k_target_finish, dim3(n), dim3(256), 0, s, c->sys, TR, TP, w.nblk, PEND);
  LAUNCHCHK();
  return 0;
}

int aomarl_target_psf(aomarl_ctx *c, aomarl_state *st, int b, int n, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!c->sys.tar_all_int) {
    rc = aomarl_raytrace_target(c, st, b, n, AOMARL_TRACE_ATMOS | AOMARL_TRACE_DMS | AOMARL_TRACE_RESET, stream);
    if (rc) return rc;
    return target_psf_impl(c, st, b, n, true, stream);
  }
  return target_psf_impl(c, st, b, n, false, stream);
}

int aomarl_comp_strehl(aomarl_ctx *c, aomarl_state *st, int b, int n, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (n == 0) return 0;
  rc = psf_wait_pending(c, stream);
  if (rc) return rc;
  Work w = work_layout(c, st->nenv);
  const int W = 2 * c->sys.hw;
  float *PEND = st->work + w.PEND + (size_t)b * (W * W + 4);
  hipLaunchKernelGGL(k_strehl_commit, dim3(n), dim3(256), 0, (hipStream_t)stream, c->sys, dev_state(st), b, PEND);
  LAUNCHCHK();
  return 0;
}

int aomarl_strehl_fit(aomarl_ctx *c, aomarl_state *st, int b, int n, void *stream) {
  if (!c || !st) return fail("strehl_fit: null ctx/state");
  if (b < 0 || n < 0 || b + n > st->nenv || !st->strehl || !st->le_img) return fail("strehl_fit: bad range / state");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_strehl_fit_le, dim3(n), dim3(64), 0, (hipStream_t)stream, c->sys, dev_state(st), b);
  LAUNCHCHK();
  return 0;
}
