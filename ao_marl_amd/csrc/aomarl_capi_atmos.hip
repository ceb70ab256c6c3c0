// aomarl_capi_atmos.hip -- part of the C ABI implementation (included by aomarl_capi.hip, one translation unit):
// side streams, Fried-Clark extrusion rounds, move / prefetch of the atmosphere, reset and prefetched reset (A1, A2).
// ---------------------------------------------------------------- side stream
// Two streams of the library run beside the control / agent chain: the next frame's extrusions
// (aomarl_prefetch_atmos) and, at the lowest priority, the second axis of the PSF window
// (k_target_finish_mfma, whose result nobody reads before the end-of-step Strehl commit).
static int side_stream(aomarl_ctx *c) {
  if (!c->atm_stream) {
    // lowest priority: this work has a whole control / agent chain of slack, the kernels of that
    // chain should not queue behind it
    int prio_lo = 0, prio_hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    // the extrusions: with the control / agent chain down to ten launches they are as long as that chain,
    // i.e. on the critical path themselves (the next frame kernel waits for them) -- normal priority,
    // and nothing in front of them; the PSF finish (needed at the end of the step) has its own stream
    // (high / normal / low priority for it: +-0.5 %, measured)
    // ONE pair of side streams per device for every context of the process: the runtime multiplexes streams
    // onto four hardware queues, and two contexts with a pair each (a training and an evaluation
    // environment, say) ran at 0.89 ms per step instead of 0.56 (round-2 script two_sims.py, since removed)
    static hipStream_t g_atm[64] = {nullptr}, g_psf[64] = {nullptr};
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail("side_stream: device ordinal %d", dev);
    if (!g_atm[dev]) {
      HIPCHK(hipStreamCreateWithPriority(&g_atm[dev], hipStreamNonBlocking, 0));
      HIPCHK(hipStreamCreateWithPriority(&g_psf[dev], hipStreamNonBlocking, prio_lo));
    }
    c->atm_stream = g_atm[dev]; c->psf_stream = g_psf[dev];
    HIPCHK(hipEventCreateWithFlags(&c->ev_frame, hipEventDisableTiming));
    c->ev_frame_cur = c->ev_frame;
    HIPCHK(hipEventCreateWithFlags(&c->ev_moved, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_psf, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  }
  return 0;
}
// everything that reads (or overwrites) the pending PSF window on `stream` waits for a finish kernel
// that may still be running on the side stream
static int psf_wait_pending(aomarl_ctx *c, void *stream) {
  if (c->psf_side && !c->side_joined) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->ev_psf, 0));
  c->psf_side = false;
  return 0;
}

// ---------------------------------------------------------------- atmosphere
// A prefetched move_atmos may still be running on the side stream: everything that touches the
// screens on `stream` waits for it first.
static int atmos_wait_pending(aomarl_ctx *c, void *stream) {
  if (c->premoved && !c->side_joined) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->ev_moved, 0));
  return 0;
}

// the first kernel of a prefetched move that WRITES ring lines waits for the readers of the screens
static int first_write_wait(aomarl_ctx *c, hipStream_t s) {
  if (s != c->atm_stream) return 0;
  if (c->group_overlap && c->ev_frame_prev) {
    if (c->need_prev) { HIPCHK(hipStreamWaitEvent(s, c->ev_frame_prev, 0)); c->need_prev = false; }
  } else if (c->frame_wait_pending) {
    HIPCHK(hipStreamWaitEvent(s, c->ev_frame_cur, 0));
    c->frame_wait_pending = false; c->need_prev = false;
  }
  return 0;
}

// the round behind `a` as k_extrude_sg wants it; false when `b` holds a layer that `a` does not (never the case for
// the rounds of a frame or of a reset: a layer's operations fill consecutive rounds from the first one on)
static bool next_round(const RoundOps &a, const RoundOps &b, RoundNext &nx) {
  nx.nops = b.nops;
  for (int i = 0; i < AOMARL_MAX_LAYERS; i++) { nx.idx[i] = -1; nx.dir[i] = 0; nx.tflag[i] = 0; }
  for (int j = 0; j < b.nops; j++) {
    int at = -1;
    for (int i = 0; i < a.nops; i++) if (a.layer[i] == b.layer[j]) at = i;
    if (at < 0) return false;
    nx.idx[at] = j; nx.dir[j] = b.dir[j]; nx.tflag[j] = b.tflag[j];
  }
#ifdef SG_SAME_ONLY                              // (A/B builds: fuse only two rounds with the same operations, as before)
  if (a.nops != b.nops) return false;
  for (int i = 0; i < a.nops; i++)
    if (a.layer[i] != b.layer[i] || a.dir[i] != b.dir[i] || a.tflag[i] != b.tflag[i]) return false;
#endif
  return b.nops > 0;
}

// A sequence of extrusion rounds (round = at most one operation per layer).  Per round: stencil gather +
// normals -> Z, Z . [A|B]^T (split-K tiles), new line -> ring.  Two consecutive rounds share a launch for the
// scatter of the first and the gather of the second (k_extrude_sg; the second may hold fewer layers and other
// directions): 2 launches per round instead of 3 -- every round of a reset (1296 of them) and of a frame.
struct ExtrudeRun {          // one range of environments walking through a sequence of rounds on one stream
  aomarl_ctx *c; aomarl_state *st; int b, n; hipStream_t s; bool ordered;
  Work w; DevState ds; float *Zb[2], *NEWL, *ZREFb[2], *WS; size_t ws_floats; bool gathered; int par;
  int pick_n = 0;            // > 0: the products take the tile and k split a range of pick_n environments would get
  // ordered = false: the caller has ordered the stream behind every reader of the screens (reset)
  ExtrudeRun(aomarl_ctx *c_, aomarl_state *st_, int b_, int n_, void *stream, bool ordered_ = true)
      : c(c_), st(st_), b(b_), n(n_), s((hipStream_t)stream), ordered(ordered_), gathered(false), par(0) {
    w = work_layout(c, st->nenv);
    ds = dev_state(st);
    if (ordered) ds.origin_snap = c->snap_target;
    // the range's own part of every work area (columns are numbered from the range's first environment):
    // two ranges may run side by side on two streams
    const size_t col0 = (size_t)b * (c->nlayers > 0 ? c->nlayers : 1), ncols = (size_t)n * (c->nlayers > 0 ? c->nlayers : 1);
    Zb[0] = st->work + w.Z + col0 * w.ldz; Zb[1] = st->work + w.Z2 + col0 * w.ldz;
    NEWL = st->work + w.NEWL + col0 * w.ldn;
    ZREFb[0] = st->work + w.ZREF + col0; ZREFb[1] = st->work + w.ZREF2 + col0;
    WS = st->work + w.GEMM_ATM + 8 * col0 * w.ldn; ws_floats = 8 * ncols * w.ldn;
  }
  int step(const RoundOps *rounds, int r, int nrounds) {
    // one sub-round per [A|B] class
    for (int cls = 0; cls < c->nclass; cls++) {
      RoundOps ops;
      ops.nops = 0;
      int ref = -1;
      for (int i = 0; i < rounds[r].nops; i++)
        if (c->abclass[rounds[r].layer[i]] == cls) {
          ops.layer[ops.nops] = rounds[r].layer[i]; ops.dir[ops.nops] = rounds[r].dir[i];
          ops.tflag[ops.nops] = rounds[r].tflag[i]; ops.nops++; ref = rounds[r].layer[i];
        }
      if (ops.nops == 0) continue;
      const int dimc = c->dim[ref], nsc = c->ns[ref], K = dimc + nsc;
      const int ncol = n * ops.nops;
      // fusing across rounds only when the round is ONE sub-round (one class) and the next round repeats it
      const bool single = ops.nops == rounds[r].nops;
      RoundNext nx;
      // (k_extrude_sg keeps a column's whole index list in registers: four entries per thread of its 512)
      bool fuse_next = single && !c->no_extrude_sg && nsc <= SG_U * SG_THREADS && dimc <= SG_MAX_N && r + 1 < nrounds && next_round(rounds[r], rounds[r + 1], nx);
      if (fuse_next)                             // (the next round must be one sub-round too: same class)
        for (int i = 0; i < rounds[r + 1].nops; i++) fuse_next = fuse_next && c->abclass[rounds[r + 1].layer[i]] == cls;
      float *Z = Zb[par], *ZREF = ZREFb[par];
      if (!(gathered && single)) {
        hipLaunchKernelGGL(k_extrude_gather, dim3(ncol, (nsc + (dimc + 3) / 4 + 255) / 256), dim3(256), 0, s, c->sys, ds, b,
                           ops, Z, w.ldz, ZREF);
        LAUNCHCHK();
      }
      int nsp = 0;
      float pscale = 1.f;
      launch_gemm_nt(ncol, dimc, K, 1.0f, Z, w.ldz, c->sys.layers[ref].AB, c->sys.layers[ref].ldab,
                     0.0f, NEWL, w.ldn, s, WS, ws_floats, nullptr, &nsp,
                     /* split-f16: stencil values (um) and N(0,1) draws x 2^8 */ true, 256.f, c->ab_scale[cls], &pscale,
                     128, pick_n > 0 && pick_n < n ? pick_n * ops.nops : 0);
      LAUNCHCHK();
      if (ordered) {
        int wrc = first_write_wait(c, s);
        if (wrc) return wrc;
        if (s != c->atm_stream) c->screens_dirty_main = true;
      }
      if (fuse_next) {
        hipLaunchKernelGGL(k_extrude_sg, dim3(ncol), dim3(SG_THREADS), 0, s, c->sys, ds, b, ops, NEWL, w.ldn, ZREF,
                           WS, nsp, ncol, dimc, pscale, Zb[par ^ 1], w.ldz, ZREFb[par ^ 1], nx);
        par ^= 1;
        gathered = true;
      } else {
        // (the last launch of a whole-batch move carries the caller's "moved" event on its dispatch: no marker packet
        // of its own between the move and the frame kernel that waits for it)
        hipEvent_t ride = (c->ride_ev && r + 1 == nrounds && cls + 1 == c->nclass && b == 0 && n == st->nenv && !c->capturing)
                              ? c->ride_ev : nullptr;
        hipExtLaunchKernelGGL(k_extrude_scatter, dim3(ncol), dim3(256), 0, s, nullptr, ride, 0, c->sys, ds, b, ops, NEWL, w.ldn,
                              ZREF, WS, nsp, ncol, dimc, pscale);
        if (ride) c->rode = true;
        gathered = false;
      }
      LAUNCHCHK();
    }
    return 0;
  }
};

static int extrude_rounds(aomarl_ctx *c, aomarl_state *st, int b, int n, const RoundOps *rounds, int nrounds,
                          void *stream) {
  if (n == 0 || nrounds == 0) return 0;
  ExtrudeRun run(c, st, b, n, stream);
  for (int r = 0; r < nrounds; r++) {
    int rc = run.step(rounds, r, nrounds);
    if (rc) return rc;
  }
  return 0;
}

int aomarl_extrude(aomarl_ctx *c, aomarl_state *st, int b, int n, int nops, const int32_t *layer,
                   const int32_t *dir, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (n == 0 || nops == 0) return 0;
  if (nops < 0 || nops > c->nlayers) return fail("nops out of range");
  if ((hipStream_t)stream != c->atm_stream || !c->atm_stream) { rc = atmos_wait_pending(c, stream); if (rc) return rc; }
  RoundOps ops;
  ops.nops = nops;
  for (int i = 0; i < nops; i++) {
    if (layer[i] < 0 || layer[i] >= c->nlayers) return fail("extrude: bad layer");
    if (!(dir[i] == 1 || dir[i] == -1 || dir[i] == 2 || dir[i] == -2)) return fail("extrude: bad direction");
    for (int j = 0; j < i; j++)
      if (layer[j] == layer[i]) return fail("extrude: a layer appears twice in one round");
    ops.layer[i] = layer[i]; ops.dir[i] = dir[i]; ops.tflag[i] = 0;
  }
  return extrude_rounds(c, st, b, n, &ops, 1, stream);
}

// plan of one env: signed pixel shifts per layer after adding the per-frame deltas
struct Plan { int kx[AOMARL_MAX_LAYERS], ky[AOMARL_MAX_LAYERS]; };

static bool plan_eq(const Plan &a, const Plan &b, int nl) {
  for (int l = 0; l < nl; l++)
    if (a.kx[l] != b.kx[l] || a.ky[l] != b.ky[l]) return false;
  return true;
}

static int run_plan(aomarl_ctx *c, aomarl_state *st, int b, int n, const Plan &p, void *stream) {
  // layer l's queue: |kx| x-extrusions then |ky| y-extrusions; round r = r-th op of each layer
  int maxr = 0;
  for (int l = 0; l < c->nlayers; l++) {
    int len = abs(p.kx[l]) + abs(p.ky[l]);
    if (len > maxr) maxr = len;
  }
  for (int l = 0; l < c->nlayers; l++)
    if (p.kx[l] == 0 && p.ky[l] == 0) c->snap_complete = false;   // a ring that does not move this frame: nobody writes its snapshot entry
  if (maxr == 0) return 0;
  if ((hipStream_t)stream != c->atm_stream || !c->atm_stream) { int rc = atmos_wait_pending(c, stream); if (rc) return rc; }
  // frame pipeline: may this move run beside the older frame in flight?  The extrusions rewrite the |kx| oldest
  // columns / |ky| oldest rows of each ring (logical 0.. for a positive shift, dim-1.. for a negative one):
  // outside every window the one-pass frame kernel reads  <=>  within the margins around the pupil
  c->group_overlap = false;
  if (c->ev_frame_prev) {
    bool fits = true;
    for (int l = 0; l < c->nlayers; l++) {
      const DevLayer &L = c->sys.layers[l];
      const int lox = L.tox, hix = L.dim - L.tox - c->sys.pupdiam, loy = L.toy, hiy = L.dim - L.toy - c->sys.pupdiam;
      if ((p.kx[l] > 0 ? p.kx[l] > lox : -p.kx[l] > hix) || (p.ky[l] > 0 ? p.ky[l] > loy : -p.ky[l] > hiy)) fits = false;
    }
    c->group_overlap = fits;
    if (fits) c->pipe.overlapped++; else c->pipe.behind++;
  }
  if (c->small_ok && c->small_move) {            // small screens: the whole move of these environments in one launch
    hipStream_t s = (hipStream_t)stream;
    MovePlan mp;
    for (int l = 0; l < AOMARL_MAX_LAYERS; l++) { mp.kx[l] = l < c->nlayers ? p.kx[l] : 0; mp.ky[l] = l < c->nlayers ? p.ky[l] : 0; }
    { int wrc = first_write_wait(c, s); if (wrc) return wrc; }      // it reads AND writes the rings: behind their readers
    if (s != c->atm_stream) c->screens_dirty_main = true;
    DevState dsm = dev_state(st);
    dsm.origin_snap = c->snap_target;
    // (the whole batch in this one launch: the caller's "moved" event rides on the dispatch)
    hipEvent_t ride = (c->ride_ev && b == 0 && n == st->nenv && !c->capturing) ? c->ride_ev : nullptr;
    hipExtLaunchKernelGGL(k_move_small, dim3(n, c->nlayers), dim3(MOVE_SMALL_T), 0, s, nullptr, ride, 0, c->sys, dsm, b, mp);
    LAUNCHCHK();
    if (ride) c->rode = true;
    return 0;
  }
  std::vector<RoundOps> rounds((size_t)maxr);
  for (int r = 0; r < maxr; r++) {
    RoundOps &o = rounds[r];
    o.nops = 0;
    for (int l = 0; l < c->nlayers; l++) {
      int ax = abs(p.kx[l]), ay = abs(p.ky[l]);
      if (r < ax) { o.layer[o.nops] = l; o.dir[o.nops] = p.kx[l] > 0 ? 1 : -1; o.tflag[o.nops] = 0; o.nops++; }
      else if (r < ax + ay) { o.layer[o.nops] = l; o.dir[o.nops] = p.ky[l] > 0 ? 2 : -2; o.tflag[o.nops] = 0; o.nops++; }
    }
  }
  return extrude_rounds(c, st, b, n, rounds.data(), maxr, stream);
}

static int move_atmos_now(aomarl_ctx *c, aomarl_state *st, int b, int n, float *accumx, float *accumy,
                          void *stream);

int aomarl_move_atmos(aomarl_ctx *c, aomarl_state *st, int b, int n, float *accumx, float *accumy,
                      void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (!accumx || !accumy) return fail("move_atmos: null accumulators");
  if (c->premoved) {
    rc = atmos_wait_pending(c, stream);
    if (rc) return rc;
    if (c->pre_screens == st->screens && c->pre_b == b && c->pre_n == n) {   // this frame's move is done
      c->premoved = false;
      return 0;
    }
  }
  return move_atmos_now(c, st, b, n, accumx, accumy, stream);
}

static int prefetch_atmos_impl(aomarl_ctx *c, aomarl_state *st, int b, int n, float *accumx, float *accumy,
                               void *stream, bool frame_marked) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (!accumx || !accumy) return fail("prefetch_atmos: null accumulators");
  if (c->premoved) return fail("prefetch_atmos: a prefetched frame is already pending");
  rc = side_stream(c);
  if (rc) return rc;
  if (!frame_marked) {                           // readers of the screens are done
    HIPCHK(hipEventRecord(c->ev_frame, (hipStream_t)stream));
    c->ev_frame_cur = c->ev_frame;
  }
  // The stencil gather and the GEMM of the first round only READ the screens (like the frame kernel the
  // caller has just launched): they need not wait for it.  The first kernel that writes a ring line
  // does (extrude_rounds).  Only in the steady state, though: if the screens were last written on the
  // caller's stream (reset, set_screen, an un-prefetched move), those writes are ordered before this
  // point of that stream only, so the side stream waits for it right away.
  c->side_joined = false;
  if (c->capturing) {      // the side stream enters the capture at the fork recorded in front of the frame kernel
    if (!c->fork_recorded) { HIPCHK(hipEventRecord(c->ev_fork, (hipStream_t)stream)); c->fork_recorded = true; }
    HIPCHK(hipStreamWaitEvent(c->atm_stream, c->ev_fork, 0));
  }
  if (c->screens_dirty_main) {
    HIPCHK(hipStreamWaitEvent(c->atm_stream, c->ev_frame_cur, 0));
    c->frame_wait_pending = false;
  } else {
    c->frame_wait_pending = true;
  }
  c->ride_ev = c->ev_moved; c->rode = false;
  rc = move_atmos_now(c, st, b, n, accumx, accumy, (void *)c->atm_stream);
  c->ride_ev = nullptr;
  if (rc) return rc;
  if (c->frame_wait_pending) {            // nothing was extruded this frame: still order the marker behind the readers
    HIPCHK(hipStreamWaitEvent(c->atm_stream, c->ev_frame_cur, 0));
    c->frame_wait_pending = false;
  }
  c->screens_dirty_main = false;
  if (!c->rode) HIPCHK(hipEventRecord(c->ev_moved, c->atm_stream));      // (else the move's one launch carried it)
  c->premoved = true; c->pre_screens = st->screens; c->pre_b = b; c->pre_n = n;
  return 0;
}

int aomarl_prefetch_atmos(aomarl_ctx *c, aomarl_state *st, int b, int n, float *accumx, float *accumy,
                          void *stream) {
  return prefetch_atmos_impl(c, st, b, n, accumx, accumy, stream, false);
}

static int move_atmos_now(aomarl_ctx *c, aomarl_state *st, int b, int n, float *accumx, float *accumy,
                          void *stream) {
  int rc = 0;
  const int nl = c->nlayers;
  int g0 = b;
  Plan cur;
  for (int e = b; e <= b + n; e++) {
    Plan p;
    if (e < b + n) {
      for (int l = 0; l < nl; l++) {
        float ax = accumx[(size_t)e * nl + l] + c->deltax[l];
        float ay = accumy[(size_t)e * nl + l] + c->deltay[l];
        int kx = (int)ax, ky = (int)ay;
        p.kx[l] = kx; p.ky[l] = ky;
        accumx[(size_t)e * nl + l] = ax - (float)kx;
        accumy[(size_t)e * nl + l] = ay - (float)ky;
        if (e == b) { c->frac_x[l] = ax - (float)kx; c->frac_y[l] = ay - (float)ky; }   // "subpixel_flow": one remainder for the range
      }
    }
    if (e == b) { cur = p; continue; }
    if (e == b + n || !plan_eq(p, cur, nl)) {
      rc = run_plan(c, st, g0, e - g0, cur, stream);
      if (rc) return rc;
      g0 = e; cur = p;
    }
  }
  return 0;
}

int aomarl_reset_strehl(aomarl_ctx *c, aomarl_state *st, int b, int n, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_strehl_reset, dim3(n), dim3(256), 0, (hipStream_t)stream, c->sys, dev_state(st), b);
  LAUNCHCHK();
  return 0;
}

// The rounds of a reset (refresh_screen: 2*dim extrusions along x, sign of deltax, atmosCompass.py:141-145).
// The 2 n extrusions of a reset all run along x: every new line is a COLUMN of the row-major ring
// (648 scattered 4-byte writes per environment and layer, and the stencil's full first column 648
// scattered reads: one 64-byte sector each).  Done on the TRANSPOSED screen they are row
// operations -- the x stencil with its coordinates exchanged, the zero screen is its own
// transpose -- and one in-place transposition at the end gives the same screen, bit for bit.
static void reset_rounds_plan(const aomarl_ctx *c, std::vector<RoundOps> &rounds) {
  int maxr = 0;
  for (int l = 0; l < c->nlayers; l++) if (2 * c->dim[l] > maxr) maxr = 2 * c->dim[l];
  rounds.assign((size_t)maxr, RoundOps());
  const bool tr = !c->reset_untransposed;
  for (int r = 0; r < maxr; r++) {
    RoundOps &o = rounds[r];
    o.nops = 0;
    for (int l = 0; l < c->nlayers; l++)
      if (r < 2 * c->dim[l]) {
        const int dx = c->deltax[l] > 0.f ? 1 : -1;
        o.layer[o.nops] = l; o.dir[o.nops] = tr ? 2 * dx : dx; o.tflag[o.nops] = tr ? 1 : 0; o.nops++;
      }
  }
}
// In how many parts a reset of n environments walks its rounds (each part's products have its own columns: the
// partition fixes the split-K order of every sum, so the prefetched reset uses the plain one's)
static int reset_parts(const aomarl_ctx *c, int n) {
  if (c->reset_streams > 1 && c->prefetch_atmos && n >= 16 * c->reset_streams && !c->capturing)
    return c->reset_streams > 4 ? 4 : c->reset_streams;
  return 1;
}
// the screens' last step: back from the transposed form, mirror columns
static int reset_screens_finish(aomarl_ctx *c, aomarl_state *st, int b, int n, hipStream_t s) {
  if (c->reset_untransposed) return 0;
  DevState ds = dev_state(st);
  for (int l = 0; l < c->nlayers; l++) {
    const int T = (c->dim[l] + 31) / 32;
    hipLaunchKernelGGL(k_transpose_ring, dim3(T * (T + 1) / 2, n), dim3(256), 0, s, c->sys, ds, b, l, T);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_refresh_mirror, dim3((c->dim[l] * RING_PAD + 255) / 256, n), dim3(256), 0, s, c->sys, ds, b, l);
    LAUNCHCHK();
  }
  return 0;
}
// everything of a reset but the screens: seeds, ring origins, counters, integrator vectors, DM shapes, slopes, Strehl
static int reset_small(aomarl_ctx *c, aomarl_state *st, int b, int n, const uint32_t *seeds, float *accumx, float *accumy,
                       uint32_t *&stage, int &stage_n, hipStream_t s, bool whole_state) {
  DevState ds = dev_state(st);
  if (stage_n < n) {
    if (stage) (void)hipFree(stage);
    HIPCHK(hipMalloc((void **)&stage, sizeof(uint32_t) * (size_t)st->nenv));
    stage_n = st->nenv;
  }
  HIPCHK(hipMemcpyAsync(stage, seeds, sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_reset_env, dim3(n), dim3(256), 0, s, c->sys, ds, b, n, stage, st->ld_actu,
                     c->reset_untransposed ? 0 : 1);
  LAUNCHCHK();
  if (!whole_state) return 0;
  hipLaunchKernelGGL(k_fill_f32, dim3(2048), dim3(256), 0, s, st->dm_shape + (size_t)b * c->sys.shape_stride,
                     (long long)n * c->sys.shape_stride, 0.f);
  LAUNCHCHK();
  hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, s, st->slopes + (size_t)b * c->sys.nslope,
                     (long long)n * c->sys.nslope, 0.f);
  LAUNCHCHK();
  int rc = aomarl_reset_strehl(c, st, b, n, (void *)s);
  if (rc) return rc;
  for (int e = b; e < b + n; e++)
    for (int l = 0; l < c->nlayers; l++) { accumx[(size_t)e * c->nlayers + l] = 0.f; accumy[(size_t)e * c->nlayers + l] = 0.f; }
  return 0;
}
// what a reset checks and drops first: a pipelined frame in flight, a prefetched atmosphere frame
static int reset_prologue(aomarl_ctx *c, aomarl_state *st, int b, int n, void *stream) {
  if (c && st && c->pipe.active && st->screens == c->pipe.owner_screens) {
    if (b != 0 || n != st->nenv) return fail("reset of environments [%d, %d) while a pipelined frame of the whole batch is in flight", b, b + n);
    int prc = pipe_drop(c, stream);
    if (prc) return prc;
  }
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (n == 0) return 0;
  rc = atmos_wait_pending(c, stream);
  if (rc) return rc;
  if (c->premoved && c->pre_screens == st->screens) {
    // a prefetched frame is pending on these screens.  A reset of (at least) the prefetched range
    // drops it -- the episode is over.  A reset of a part of it cannot: the other environments'
    // screens and accumulators have already advanced, dropping the flag would make the next
    // move_atmos advance them a second time (they would silently skip an atmosphere frame).
    const bool covers = b <= c->pre_b && b + n >= c->pre_b + c->pre_n;
    const bool disjoint = b + n <= c->pre_b || b >= c->pre_b + c->pre_n;
    if (covers) c->premoved = false;
    else if (!disjoint)
      return fail("reset of environments [%d, %d) while the prefetched atmosphere frame of [%d, %d) is pending: "
                  "reset the whole prefetched range, or call aomarl_move_atmos on it first",
                  b, b + n, c->pre_b, c->pre_b + c->pre_n);
  }
  return 0;
}

int aomarl_reset(aomarl_ctx *c, aomarl_state *st, int b, int n, const uint32_t *seeds, float *accumx,
                 float *accumy, void *stream) {
  int rc = reset_prologue(c, st, b, n, stream);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!seeds || !accumx || !accumy) return fail("reset: null argument");
  hipStream_t s = (hipStream_t)stream;
  c->screens_dirty_main = true;
  rc = reset_small(c, st, b, n, seeds, accumx, accumy, c->seed_stage, c->seed_stage_n, s, true);
  if (rc) return rc;
  hipLaunchKernelGGL(k_fill_f32, dim3(2048), dim3(256), 0, s, st->screens + (size_t)b * c->sys.screen_stride,
                     (long long)n * c->sys.screen_stride, 0.f);
  LAUNCHCHK();
  std::vector<RoundOps> rounds;
  reset_rounds_plan(c, rounds);
  const int maxr = (int)rounds.size();
  const int parts = reset_parts(c, n);
  if (parts > 1 && side_stream(c) == 0) {
    // The batch in parts side by side, one stream each (the caller's, the extrusion stream, two more of the
    // process): a round is gather | GEMM | scatter + gather, 45 us of which 15 are latency (launch, first operand
    // lines, the dependent loads of the stencil gather) that one part's kernels hide for the others' -- 1296
    // dependent rounds.  Same kernels on the same columns; the split-K rule sees a part's columns per product.
    static hipStream_t g_rst[64][2] = {{nullptr}};
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    hipStream_t str[4] = {s, c->atm_stream, nullptr, nullptr};
    for (int k = 2; k < parts; k++) {
      if (dev < 0 || dev >= 64) return fail("reset: device ordinal %d", dev);
      if (!g_rst[dev][k - 2]) HIPCHK(hipStreamCreateWithFlags(&g_rst[dev][k - 2], hipStreamNonBlocking));
      str[k] = g_rst[dev][k - 2];
    }
    if (!c->ev_reset) {
      HIPCHK(hipEventCreateWithFlags(&c->ev_reset, hipEventDisableTiming));
      for (int k = 0; k < 3; k++) HIPCHK(hipEventCreateWithFlags(&c->ev_reset2[k], hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(c->ev_reset, s));
    std::vector<ExtrudeRun> runs;
    int e0 = b;
    for (int k = 0; k < parts; k++) {
      const int nk = (b + n - e0) / (parts - k);
      if (k) HIPCHK(hipStreamWaitEvent(str[k], c->ev_reset, 0));
      runs.emplace_back(c, st, e0, nk, (void *)str[k], false);
      e0 += nk;
    }
    for (int r = 0; r < maxr; r++)
      for (auto &run : runs) {
        rc = run.step(rounds.data(), r, maxr);
        if (rc) return rc;
      }
    for (int k = 1; k < parts; k++) {
      HIPCHK(hipEventRecord(c->ev_reset2[k - 1], str[k]));
      HIPCHK(hipStreamWaitEvent(s, c->ev_reset2[k - 1], 0));
    }
  } else {
    rc = extrude_rounds(c, st, b, n, rounds.data(), maxr, stream);
    if (rc) return rc;
  }
  rc = reset_screens_finish(c, st, b, n, s);
  if (rc) return rc;
  // pending PSF of the fresh atmosphere with flat DMs: comp_strehl before the first
  // next_part_one is well defined
  if (!c->sys.tar_all_int && !st->tar_phase) return 0;
  return aomarl_target_psf(c, st, b, n, stream);
}

// ---------------------------------------------------------------- prefetched reset
// The seeds of the next episode are known while this one runs (train_rpc.py:486-487: seed += 1 per episode), and a
// reset is 2 x 648 DEPENDENT extrusion rounds per layer -- 45 ms for 256 environments, mostly latency.  So the next
// episode's screens are grown in a SHADOW state (own screens, ring origins, counters, seeds, workspace) on a stream
// of the caller's, a few rounds per step of the running episode, beside its kernels; aomarl_reset_adopt then
// copies them in (1.3 GB device to device: < 1 ms) and does the rest of the reset.  Same kernels, same columns,
// same split-K order as aomarl_reset (whose partition of the batch fixes it): the same screens, bit for bit.
struct ResetPrefetch {
  aomarl_state shadow;                 // a copy of the caller's struct (its buffers stay the caller's)
  int b = 0, n = 0, next_round = 0;
  std::vector<RoundOps> rounds;
  std::vector<ExtrudeRun> runs;
  std::vector<uint32_t> seeds;
  uint32_t *stage = nullptr; int stage_n = 0;
  hipEvent_t ev = nullptr, ev_copied = nullptr;
  bool finished = false, copied = false;
};
// stream == NULL: the library's own low-priority side stream (the one the PSF finish runs on: no further hardware queue)
static int rp_stream(aomarl_ctx *c, void *stream, hipStream_t *out) {
  if (stream) { *out = (hipStream_t)stream; return 0; }
  int rc = side_stream(c);
  if (rc) return rc;
  *out = c->psf_stream;
  return 0;
}

static void rp_free(aomarl_ctx *c) {
  if (!c->rp) return;
  if (c->rp->stage) (void)hipFree(c->rp->stage);
  if (c->rp->ev) (void)hipEventDestroy(c->rp->ev);
  if (c->rp->ev_copied) (void)hipEventDestroy(c->rp->ev_copied);
  delete c->rp;
  c->rp = nullptr;
}

int aomarl_reset_prefetch_begin(aomarl_ctx *c, const aomarl_state *shadow, int b, int n, const uint32_t *seeds, void *stream) {
  if (!c || !shadow || !seeds) return fail("reset_prefetch_begin: null argument");
  if (c->pipe.active && shadow->screens == c->pipe.owner_screens) return fail("reset_prefetch_begin: the shadow must not be the live state");
  int rc = check_range(c, shadow, b, n);
  if (rc) return rc;
  if (n == 0) return fail("reset_prefetch_begin: empty range");
  ResetPrefetch *rp = c->rp;
  if (!rp) {
    rp = c->rp = new ResetPrefetch();
    HIPCHK(hipEventCreateWithFlags(&rp->ev, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&rp->ev_copied, hipEventDisableTiming));
  }
  rp->shadow = *shadow; rp->b = b; rp->n = n; rp->next_round = 0; rp->finished = false;
  rp->seeds.assign(seeds, seeds + n);
  hipStream_t s = nullptr;
  rc = rp_stream(c, stream, &s);
  if (rc) return rc;
  stream = (void *)s;
  if (rp->copied) HIPCHK(hipStreamWaitEvent(s, rp->ev_copied, 0));      // the last adoption has read the shadow
  // (reset_small with whole_state = false: seeds, origins, counters and the SHADOW's small vectors only)
  rc = reset_small(c, &rp->shadow, b, n, seeds, nullptr, nullptr, rp->stage, rp->stage_n, s, false);
  if (rc) return rc;
  hipLaunchKernelGGL(k_fill_f32, dim3(2048), dim3(256), 0, s, rp->shadow.screens + (size_t)b * c->sys.screen_stride,
                     (long long)n * c->sys.screen_stride, 0.f);
  LAUNCHCHK();
  reset_rounds_plan(c, rp->rounds);
  rp->runs.clear();
  const int parts = reset_parts(c, n);
  if (c->reset_prefetch_whole && parts > 1 && n % parts == 0) {
    // One range whose products take the tile and the k split of a part's (launch_gemm_nt's pick_M): every sum in the
    // plain reset's order, so the same screens bit for bit, with half the launches beside the running episode
    // (256 environments: the rounds alone 47 ms instead of 60, the step beside them 2.7 % shorter).
    rp->runs.emplace_back(c, &rp->shadow, b, n, stream, false);
    rp->runs.back().pick_n = n / parts;
  } else {
    int e0 = b;
    for (int k = 0; k < parts; k++) {         // the plain reset's partition, all parts on the one stream
      const int nk = (b + n - e0) / (parts - k);
      rp->runs.emplace_back(c, &rp->shadow, e0, nk, stream, false);
      e0 += nk;
    }
  }
  HIPCHK(hipEventRecord(rp->ev, s));
  return 0;
}

int aomarl_reset_prefetch_advance(aomarl_ctx *c, int nrounds, void *stream, int *remaining) {
  if (!c || !c->rp) return fail("reset_prefetch_advance: no prefetch has begun");
  ResetPrefetch *rp = c->rp;
  hipStream_t s = nullptr;
  { int src = rp_stream(c, stream, &s); if (src) return src; }
  const int maxr = (int)rp->rounds.size();
  if (!rp->finished) {
    for (auto &run : rp->runs) run.s = s;
    const int end = nrounds < 0 ? maxr : std::min(maxr, rp->next_round + nrounds);
    for (; rp->next_round < end; rp->next_round++)
      for (auto &run : rp->runs) {
        int rc = run.step(rp->rounds.data(), rp->next_round, maxr);
        if (rc) return rc;
      }
    if (rp->next_round >= maxr) {
      int rc = reset_screens_finish(c, &rp->shadow, rp->b, rp->n, s);
      if (rc) return rc;
      rp->finished = true;
    }
    HIPCHK(hipEventRecord(rp->ev, s));
  }
  if (remaining) *remaining = maxr - rp->next_round;
  return 0;
}

int aomarl_reset_prefetch_cancel(aomarl_ctx *c) {
  if (!c) return fail("reset_prefetch_cancel: null ctx");
  if (c->rp) { c->rp->runs.clear(); c->rp->finished = false; c->rp->n = 0; }
  return 0;
}

int aomarl_reset_adopt(aomarl_ctx *c, aomarl_state *st, int b, int n, const uint32_t *seeds, float *accumx, float *accumy,
                       void *prefetch_stream, void *stream) {
  if (!c || !c->rp || c->rp->n == 0) return fail("reset_adopt: no prefetched reset");
  ResetPrefetch *rp = c->rp;
  if (!seeds || !accumx || !accumy) return fail("reset_adopt: null argument");
  if (rp->b != b || rp->n != n) return fail("reset_adopt: prefetched environments [%d, %d), asked for [%d, %d)", rp->b, rp->b + rp->n, b, b + n);
  for (int i = 0; i < n; i++)
    if (rp->seeds[i] != seeds[i]) return fail("reset_adopt: the prefetched reset was begun with other seeds");
  if (st->screens == rp->shadow.screens) return fail("reset_adopt: the shadow is the state itself");
  int rc = reset_prologue(c, st, b, n, stream);
  if (rc) return rc;
  if (!rp->finished) {                      // what is left of the rounds, now
    rc = aomarl_reset_prefetch_advance(c, -1, prefetch_stream, nullptr);
    if (rc) return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  c->screens_dirty_main = true;
  rc = reset_small(c, st, b, n, seeds, accumx, accumy, c->seed_stage, c->seed_stage_n, s, true);
  if (rc) return rc;
  HIPCHK(hipStreamWaitEvent(s, rp->ev, 0));
  const size_t so = (size_t)b * c->sys.screen_stride, nl = (size_t)c->nlayers;
  HIPCHK(hipMemcpyAsync(st->screens + so, rp->shadow.screens + so, sizeof(float) * (size_t)n * c->sys.screen_stride, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(st->origin + (size_t)b * nl * 2, rp->shadow.origin + (size_t)b * nl * 2, sizeof(int32_t) * (size_t)n * nl * 2, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(st->ext_count + (size_t)b * nl, rp->shadow.ext_count + (size_t)b * nl, sizeof(uint32_t) * (size_t)n * nl, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipEventRecord(rp->ev_copied, s));
  rp->copied = true;
  rp->n = 0; rp->finished = false; rp->runs.clear();          // consumed
  if (!c->sys.tar_all_int && !st->tar_phase) return 0;
  return aomarl_target_psf(c, st, b, n, stream);
}

int aomarl_set_screen(aomarl_ctx *c, aomarl_state *st, int b, int n, int layer, const float *src, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  rc = atmos_wait_pending(c, stream);
  if (rc) return rc;
  if (layer < 0 || layer >= c->nlayers || !src) return fail("set_screen: bad argument");
  if (n == 0) return 0;
  c->screens_dirty_main = true;
  hipLaunchKernelGGL(k_set_screen, dim3(256, n), dim3(256), 0, (hipStream_t)stream, c->sys, dev_state(st), b, layer, src);
  LAUNCHCHK();
  return 0;
}

int aomarl_get_screen(aomarl_ctx *c, aomarl_state *st, int b, int n, int layer, float *dst, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  rc = atmos_wait_pending(c, stream);
  if (rc) return rc;
  if (layer < 0 || layer >= c->nlayers || !dst) return fail("get_screen: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_get_screen, dim3(256, n), dim3(256), 0, (hipStream_t)stream, c->sys, dev_state(st), b, layer, dst);
  LAUNCHCHK();
  return 0;
}

// ---------------------------------------------------------------- run-time wind and r0 (atmosCompass.py:79-135)
// Host values of the context (deltax / deltay drive the planning of every later move and the direction of a reset's
// rounds; amp rides in the kernels' DevSys argument) + the stencil lists on the device.  Rare calls (the trainer's
// non-stationary experiments change the atmosphere once, train_rpc.py:429-450): they synchronise the device instead of
// ordering themselves into the streams.
static int stencil_rewrite(aomarl_ctx *c, int l, int axis, const uint32_t *flat /* null: mirror what is there */, int n) {
  DevLayer &D = c->sys.layers[l];
  const int dim = D.dim, ns = D.ns;
  if (flat && n != ns) return fail("set_stencil: layer %d has %d stencil points, got %d", l, ns, n);
  uint32_t *dst = const_cast<uint32_t *>(axis == 0 ? D.istx : D.isty);
  std::vector<uint32_t> pk((size_t)ns);
  HIPCHK(hipDeviceSynchronize());
  if (flat) {
    for (int k = 0; k < ns; k++) {
      if (flat[k] >= (uint32_t)(dim * dim)) return fail("set_stencil: index out of range");
      pk[k] = (flat[k] % dim) | ((flat[k] / dim) << 16);
    }
  } else {
    HIPCHK(hipMemcpy(pk.data(), dst, sizeof(uint32_t) * ns, hipMemcpyDeviceToHost));
    for (int k = 0; k < ns; k++) {                 // n * n - 1 - (y * n + x) = (n - 1 - y) * n + (n - 1 - x)
      const uint32_t x = pk[k] & 0xFFFFu, y = pk[k] >> 16;
      pk[k] = (uint32_t)(dim - 1 - (int)x) | ((uint32_t)(dim - 1 - (int)y) << 16);
    }
  }
  HIPCHK(hipMemcpy(dst, pk.data(), sizeof(uint32_t) * ns, hipMemcpyHostToDevice));
  if (axis == 0) {                                 // the reset's rounds read the x stencil with x and y exchanged
    for (int k = 0; k < ns; k++) pk[k] = (pk[k] >> 16) | (pk[k] << 16);
    HIPCHK(hipMemcpy(const_cast<uint32_t *>(D.istT), pk.data(), sizeof(uint32_t) * ns, hipMemcpyHostToDevice));
  }
  return 0;
}

static int atmos_change_ok(aomarl_ctx *c, const char *who) {
  if (c->capturing) return fail("%s: inside a graph capture", who);
  if (c->rp && c->rp->n > 0)
    return fail("%s: a prefetched reset is in flight (its rounds were planned with the old atmosphere): "
                "aomarl_reset_prefetch_cancel first", who);
  return 0;
}

int aomarl_set_wind(aomarl_ctx *c, int layer, float deltax, float deltay, int mirror_stencils) {
  if (!c) return fail("set_wind: null ctx");
  if (layer < 0 || layer >= c->nlayers) return fail("set_wind: layer %d of %d", layer, c->nlayers);
  if (!(fabsf(deltax) < (float)c->dim[layer]) || !(fabsf(deltay) < (float)c->dim[layer]))
    return fail("set_wind: layer %d would move by more than its %d pixels per frame (or the value is not finite)", layer, c->dim[layer]);
  int rc = atmos_change_ok(c, "set_wind");
  if (rc) return rc;
  const float oldx = c->deltax[layer], oldy = c->deltay[layer];
  if (mirror_stencils) {
    if (oldx * deltax < 0.f && (rc = stencil_rewrite(c, layer, 0, nullptr, 0))) return rc;
    if (oldy * deltay < 0.f && (rc = stencil_rewrite(c, layer, 1, nullptr, 0))) return rc;
  }
  HIPCHK(hipDeviceSynchronize());
  c->deltax[layer] = deltax; c->deltay[layer] = deltay;
  g_cfg_epoch++; c->cfg_epoch++;                   // captured step graphs hold plans of the old wind
  return 0;
}

int aomarl_set_stencil(aomarl_ctx *c, int layer, int axis, const uint32_t *istencil, int n) {
  if (!c || !istencil) return fail("set_stencil: null argument");
  if (layer < 0 || layer >= c->nlayers || (axis != 0 && axis != 1)) return fail("set_stencil: layer %d, axis %d", layer, axis);
  int rc = atmos_change_ok(c, "set_stencil");
  if (rc) return rc;
  g_cfg_epoch++; c->cfg_epoch++;
  return stencil_rewrite(c, layer, axis, istencil, n);
}

int aomarl_set_r0(aomarl_ctx *c, const float *amplitude, int nlayers) {
  if (!c || !amplitude) return fail("set_r0: null argument");
  if (nlayers != c->nlayers) return fail("set_r0: %d amplitudes for %d layers", nlayers, c->nlayers);
  for (int l = 0; l < nlayers; l++)
    if (!(amplitude[l] >= 0.f) || !(amplitude[l] < 1e30f)) return fail("set_r0: amplitude %d is not a finite non-negative number", l);
  int rc = atmos_change_ok(c, "set_r0");
  if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());
  for (int l = 0; l < nlayers; l++) c->sys.layers[l].amp = amplitude[l];
  g_cfg_epoch++; c->cfg_epoch++;                   // DevSys rides by value in every captured kernel node
  return 0;
}

int aomarl_get_layer(const aomarl_ctx *c, int layer, float *deltax, float *deltay, float *amplitude) {
  if (!c) return fail("get_layer: null ctx");
  if (layer < 0 || layer >= c->nlayers) return fail("get_layer: layer %d of %d", layer, c->nlayers);
  if (deltax) *deltax = c->deltax[layer];
  if (deltay) *deltay = c->deltay[layer];
  if (amplitude) *amplitude = c->sys.layers[layer].amp;
  return 0;
}
