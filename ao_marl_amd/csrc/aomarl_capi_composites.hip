// aomarl_capi_composites.hip -- part of the C ABI implementation (included by aomarl_capi.hip, one translation unit):
// composites (frame_fused, next_part_one / two), the stand-alone GEMM entry points, the SAC update.
// ---------------------------------------------------------------- composites
const char *aomarl_frame_kernel_name(aomarl_ctx *c) {
  if (!c || !c->fw_variant[0]) return "";
  snprintf(c->fw_name, sizeof(c->fw_name), "k_frame_wave<%d, %d, %s, %s, %s, %s>", c->fw_variant[0], c->fw_variant[1],
           c->fw_variant[2] ? "true" : "false", c->fw_variant[3] ? "true" : "false",
           c->fw_variant[4] ? "true" : "false", c->fw_variant[5] ? "true" : "false");
  return c->fw_name;
}

int aomarl_frame_kernel_time(aomarl_ctx *c, double *total_ms, int *launches) {
  if (!c || !total_ms || !launches) return fail("frame_kernel_time: null argument");
  double tot = 0.0;
  int n = 0;
  for (size_t i = 0; i + 1 < c->fw_ev_used; i += 2) {
    float ms = 0.f;
    HIPCHK(hipEventSynchronize(c->fw_ev[i + 1]));
    HIPCHK(hipEventElapsedTime(&ms, c->fw_ev[i], c->fw_ev[i + 1]));
    tot += ms; n++;
  }
  { const int rrc = fw_ev_rewind(c); if (rrc) return rrc; }
  *total_ms = tot; *launches = n;
  return 0;
}

int aomarl_frame_fused_available(aomarl_ctx *c) {
  return c && c->sys.fused_ok && !c->force_unfused_frame ? 1 : 0;
}

// science-path PSF (pending, like aomarl_target_psf) + WFS image / slopes (like aomarl_comp_image
// without the NO_ATMOS / NO_DMS / FROM_PHASE_BUFFER variants) from one pass over the phase
// `slot` (frame pipeline): the launch goes to `stream` = the frame stream with parity slot's buffers in `st` (a
// view), carries ev_done[slot] (or a timing event) and its PSF finish records ev_psf[slot]; the caller has ordered
// `stream` behind the atmosphere and the previous users of that parity's buffers.
static int frame_fused_impl(aomarl_ctx *c, aomarl_state *st, int b, int n, int flags, void *stream, int slot);

int aomarl_frame_fused(aomarl_ctx *c, aomarl_state *st, int b, int n, int flags, void *stream) {
  return frame_fused_impl(c, st, b, n, flags, stream, -1);
}

static int frame_fused_impl(aomarl_ctx *c, aomarl_state *st, int b, int n, int flags, void *stream, int slot) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (slot < 0) {
    rc = atmos_wait_pending(c, stream);
    if (rc) return rc;
    rc = psf_wait_pending(c, stream);
    if (rc) return rc;
  }
  if (!c->sys.fused_ok) return fail("frame_fused: geometry not eligible (see aomarl_frame_fused_available)");
  if (flags & (AOMARL_IMG_FROM_PHASE_BUFFER | AOMARL_IMG_NO_ATMOS | AOMARL_IMG_NO_DMS))
    return fail("frame_fused: FROM_PHASE_BUFFER / NO_ATMOS / NO_DMS are not supported here");
  const bool noise = (flags & AOMARL_IMG_NOISE) && c->sys.noise >= 0.f;
  const bool cube = flags & AOMARL_IMG_WRITE_BINCUBE;
  const bool otf = flags & AOMARL_IMG_DM_FROM_VOLTAGE;
  const int cog = ((flags & AOMARL_IMG_COG) ? 1 : 0) | (c->fused_debug << 8);
  if (otf && !c->sys.otf_ok) return fail("frame_fused: DM_FROM_VOLTAGE needs a separable stack-array lattice (see aomarl_dm_from_voltage_available)");
  if (cube && !st->bincube) return fail("frame_fused: WRITE_BINCUBE needs st->bincube");
  if (!cube && !(cog & 1)) return fail("frame_fused: nothing to produce (neither bincube nor slopes)");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  Work w = work_layout(c, st->nenv);
  const int W = 2 * c->sys.hw;
  float *TR = st->work + w.TR + (size_t)b * c->sys.pupdiam * W * 2;
  float *TP = st->work + w.TPART + (size_t)b * w.nblk * 4;
  float *PEND = st->work + w.PEND + (size_t)b * (W * W + 4);
  DevState ds = dev_state(st);
  if (w.nblk != c->sys.ntiles) return fail("frame_fused: internal stripe count mismatch");
  const int nb = otf ? c->sys.otf_nb : 1;
  const bool hp = c->dft_mode < 0 ? g_precision != 0 : c->dft_mode == 1;
  // twiddles + command lattice + the block's shared-data slots (two tiles x two parities with the pair walk) + one
  // image of a tile pair's layer rows per wave (FW_DMA: the stack-array-from-voltages instantiations)
  // ... + the moments of a stripe's sub-apertures per wave (the slopes-only fp32 instantiation: 16 bytes per tile)
  const bool qf = FW_QF && otf && !hp && !noise && !cube;
  const bool dma = FW_DMA && otf && (hp || FW_DMA_F32);
  const size_t smm = sizeof(float) * ((qf ? 0 : 2 * 128) + (otf ? 4 * 4 * nb * c->sys.otf_latw : 0)) + FW_SLOT_BYTES(dma) +
                     (dma ? 4 * FWD_WAVE(c->nlayers == 1 ? 1 : 3) : 128) +
                     (qf ? 4 * 16 * (size_t)c->sys.ntiles : 0);
  // Frames in flight beside the chains (frame pipeline, slot >= 0), slopes-only fp32 instantiation of a large system: the
  // workgroup asks for so much LDS that TWO of them fit a CU (3 x request > 160 KB) and what is left takes a workgroup of
  // the chains' products (46 KB) or of the actors (52.7 KB at 14 agents) at any time, instead of three frame workgroups
  // that leave 13 KB and a chain that starts where one of them retires.  Round 6, with the lighter frame kernel: the
  // frame kernel in the loop 0.39 -> 0.44 ms, the step 0.482 -> 0.473 ms (profiles/r06_overlap_experiments.txt; round 5
  // measured the opposite with the heavier kernel).  Not in the plain order (the frame kernel alone on the GPU wants its
  // three workgroups: 0.292 against 0.317 ms).  AOMARL_FW_LDS_PAD=<bytes> overrides (0: off).
  static const long lds_pad_env = [] { const char *e = getenv("AOMARL_FW_LDS_PAD"); return e ? atol(e) : -1L; }();
  size_t lds_pad = 0;
  if (qf && slot >= 0 && smm >= 40 * 1024) {
    const size_t two_per_cu = (160 * 1024) / 3 + 64;          // 3 x this does not fit a CU's 160 KB
    lds_pad = lds_pad_env >= 0 ? (size_t)lds_pad_env : (smm < two_per_cu ? (two_per_cu - smm + 63) / 64 * 64 : 0);
    if (smm + lds_pad > 64 * 1024) lds_pad = 0;               // (beyond 64 KB the launch would need the opt-in attribute)
  }
  dim3 grid((n + 3) / 4, c->sys.ntiles), blk(256);
// the events ride on the dispatch itself (its start / completion signal): no marker packets of their own
// on the queue in front of and behind the kernel
#define FW(NL, NB, OTF, NZ, WC, HP) hipExtLaunchKernelGGL((k_frame_wave<NL, NB, OTF, NZ, WC, HP>), grid, blk, smm + lds_pad, s, ev_start, ev_done, 0, c->sys, ds, b, n, cog, TR, TP, w.nblk)
#define FW_H(NL, NB, OTF, NZ, WC) do { if (hp) FW(NL, NB, OTF, NZ, WC, true); else FW(NL, NB, OTF, NZ, WC, false); } while (0)
#define FW_NC(NL, NB, OTF)                                                                     \
  do {                                                                                          \
    if (noise) { if (cube) FW_H(NL, NB, OTF, true, true); else FW_H(NL, NB, OTF, true, false); }     \
    else { if (cube) FW_H(NL, NB, OTF, false, true); else FW_H(NL, NB, OTF, false, false); }         \
  } while (0)
#define FW_L(NL)                                                          \
  do {                                                                    \
    if (!otf) FW_NC(NL, 1, false);                                        \
    else if (nb == 1) FW_NC(NL, 1, true);                                 \
    else FW_NC(NL, 2, true);                                              \
  } while (0)
  const bool timed = !c->capturing && c->time_fw && c->fw_ev_used + 2 <= c->fw_ev.size();
  // closing event: the "readers of the screens are done" mark the side streams wait for (the closing
  // event of a timed launch doubles as it); only with the prefetch on, which is what creates ev_frame
  hipEvent_t ev_start = nullptr, ev_done = nullptr;
  if (c->prefetch_atmos) {
    rc = side_stream(c);
    if (rc) return rc;
    ev_done = slot < 0 ? c->ev_frame : c->pipe.ev_done[slot];
  }
  if (timed) { ev_start = c->fw_ev[c->fw_ev_used]; ev_done = c->fw_ev[c->fw_ev_used + 1]; c->fw_ev_used += 2; }
  hipEvent_t ev_mark = nullptr;
  if (c->capturing) { ev_mark = ev_done; ev_start = nullptr; ev_done = nullptr; }   // a captured dispatch carries no events
  if (c->nlayers == 1) FW_L(1); else FW_L(3);
  if (ev_mark) { HIPCHK(hipEventRecord(ev_mark, s)); ev_done = ev_mark; }
  c->frame_marked = false;
  c->fw_variant[0] = c->nlayers == 1 ? 1 : 3; c->fw_variant[1] = otf ? nb : 1; c->fw_variant[2] = otf;
  c->fw_variant[3] = noise; c->fw_variant[4] = cube; c->fw_variant[5] = hp;
  g_arith[hp ? AR_FRAME_SPLIT : AR_FRAME_F32]++;
#undef FW_L
#undef FW_NC
#undef FW_H
#undef FW
  LAUNCHCHK();
  if (slot >= 0) {
    c->pipe.ev_done_cur[slot] = ev_done;
    HIPCHK(hipStreamWaitEvent(c->psf_stream, ev_done, 0));
    if (!c->pipe.cmd_covers_commit)
      HIPCHK(hipStreamWaitEvent(c->psf_stream, c->pipe.ev_commit, 0));   // that parity's pending window has been committed
    // (its completion event rides on the dispatch, like the frame kernel's: one runtime call less per step)
    hipExtLaunchKernelGGL(k_target_finish_mfma, dim3(n), dim3(256), 0, c->psf_stream, nullptr, c->pipe.ev_psf[slot], 0, c->sys, TR,
                          TP, w.nblk, PEND, st->frame + b);
    LAUNCHCHK();
    c->pipe.psf_out[slot] = true;
    return 0;
  }
  if (c->prefetch_atmos) {
    // second axis of the PSF window: off the critical path (read by aomarl_comp_strehl at the end of
    // the step), so it goes to the side stream, in front of the next frame's extrusions
    c->ev_frame_cur = ev_done; c->frame_marked = true;
    c->side_joined = false;
    HIPCHK(hipStreamWaitEvent(c->psf_stream, ev_done, 0));
    hipExtLaunchKernelGGL(k_target_finish_mfma, dim3(n), dim3(256), 0, c->psf_stream, nullptr, c->capturing ? nullptr : c->ev_psf, 0,
                          c->sys, TR, TP, w.nblk, PEND, st->frame + b);
    LAUNCHCHK();
    if (c->capturing) HIPCHK(hipEventRecord(c->ev_psf, c->psf_stream));       // (a captured dispatch carries no events)
    c->psf_side = true;
    return 0;
  }
  hipLaunchKernelGGL(k_target_finish_mfma, dim3(n), dim3(256), 0, s, c->sys, TR, TP, w.nblk, PEND, st->frame + b);
  LAUNCHCHK();
  return 0;
}

/* refresh the stack-array planes of st->dm_shape from st->voltage (after deferred apply_control) */
int aomarl_materialize_dm_shape(aomarl_ctx *c, aomarl_state *st, int b, int n, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (n == 0) return 0;
  return dm_shape_impl(c, st, b, n, nullptr, false, stream);
}

int aomarl_next_part_one(aomarl_ctx *c, aomarl_state *st, int b, int n, float *accumx, float *accumy,
                         int image_flags, void *stream) {
  int rc = aomarl_move_atmos(c, st, b, n, accumx, accumy, stream);
  if (rc) return rc;
  int fl = (image_flags | AOMARL_IMG_COG | AOMARL_IMG_NOISE) & ~(AOMARL_IMG_NO_ATMOS | AOMARL_IMG_NO_DMS);
  const bool defer = c->defer_dm_shape && aomarl_dm_from_voltage_available(c);
  if (aomarl_frame_fused_available(c) && !(fl & AOMARL_IMG_FROM_PHASE_BUFFER)) {
    if (c->capturing && c->prefetch_atmos) {       // where the extrusion stream forks from the caller's
      rc = side_stream(c);
      if (rc) return rc;
      HIPCHK(hipEventRecord(c->ev_fork, (hipStream_t)stream));
      c->fork_recorded = true;
    }
    rc = aomarl_frame_fused(c, st, b, n, fl | (defer ? AOMARL_IMG_DM_FROM_VOLTAGE : 0), stream);
    if (rc) return rc;
    if (c->prefetch_atmos && !c->premoved) {     // one frame ahead for ONE range at a time
      // nothing was launched on `stream` since the frame kernel: its mark stands for the screens' readers
      rc = prefetch_atmos_impl(c, st, b, n, accumx, accumy, stream, c->frame_marked);
      if (rc) return rc;
    }
    if (c->skip_do_control) return 0;
    return aomarl_do_control(c, st, b, n, stream);
  }
  rc = aomarl_target_psf(c, st, b, n, stream);
  if (rc) return rc;
  if (!c->sys.wfs_all_int) {
    rc = aomarl_raytrace_wfs(c, st, b, n, AOMARL_TRACE_ATMOS | AOMARL_TRACE_DMS | AOMARL_TRACE_RESET, stream);
    if (rc) return rc;
    fl |= AOMARL_IMG_FROM_PHASE_BUFFER;
  }
  rc = aomarl_comp_image(c, st, b, n, fl, stream);
  if (rc) return rc;
  if (c->prefetch_atmos && !c->premoved) {
    rc = aomarl_prefetch_atmos(c, st, b, n, accumx, accumy, stream);
    if (rc) return rc;
  }
  if (c->skip_do_control) return 0;
  return aomarl_do_control(c, st, b, n, stream);
}

int aomarl_next_part_two(aomarl_ctx *c, aomarl_state *st, int b, int n, const float *action, void *stream) {
  int rc;
  if (action) {
    rc = aomarl_rl_control(c, st, b, n, action, stream);
    if (rc) return rc;
  }
  rc = aomarl_apply_control(c, st, b, n, AOMARL_APPLY_COMP_VOLTAGE | (c->defer_dm_shape ? AOMARL_APPLY_DEFER_STACK_SHAPE : 0), stream);
  if (rc) return rc;
  return aomarl_comp_strehl(c, st, b, n, stream);
}

int aomarl_gemm_nt(int M, int N, int K, float alpha, const float *A, int lda, const float *B, int ldb,
                   float beta, float *C, int ldc, void *stream) {
  if (!A || !B || !C) return fail("gemm_nt: null pointer");
  if (M < 0 || N < 0 || K < 0 || lda < K || ldb < K || ldc < N) return fail("gemm_nt: bad sizes");
  launch_gemm_nt(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, (hipStream_t)stream);
  LAUNCHCHK();
  return 0;
}

int aomarl_gemm_nt_split(int M, int N, int K, float alpha, const float *A, int lda, const float *B, int ldb,
                         float beta, float *C, int ldc, float scale_a, float scale_b, float *work,
                         long long work_floats, void *stream) {
  if (!A || !B || !C) return fail("gemm_nt_split: null pointer");
  if (M < 0 || N < 0 || K < 0 || lda < K || ldb < K || ldc < N) return fail("gemm_nt_split: bad sizes");
  if ((lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15))
    return fail("gemm_nt_split: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  auto pow2 = [](float v) { int e; return v > 0.f && frexpf(v, &e) == 0.5f; };
  if (!pow2(scale_a) || !pow2(scale_b)) return fail("gemm_nt_split: scales must be powers of two");
  const bool keep = g_gemm_split_f16;
  g_gemm_split_f16 = true;
  launch_gemm_nt(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, (hipStream_t)stream, work, (size_t)std::max(0LL, work_floats),
                 nullptr, nullptr, true, scale_a, scale_b);
  g_gemm_split_f16 = keep;
  LAUNCHCHK();
  return 0;
}

template <bool TA, bool TB, int G>
static int gemm_batched_launch_g(dim3 grid, hipStream_t s, int M, int N, int K, const float *A, int lda,
                                 long long strideA, const float *B, int ldb, long long strideB,
                                 const float *bias, long long strideBias, float *C, int ldc, long long strideC,
                                 int relu, int accumulate, int vecA, int vecB, const float *mask, int ldm,
                                 long long strideM) {
  static bool attr_done = false;
  const size_t smem = (size_t)G * 4 * 64 * G2_LD * sizeof(float);
  if (!attr_done) {
    HIPCHK(hipFuncSetAttribute((const void *)k_gemm_batched_gen<TA, TB, G>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_done = true;
  }
  const int ntile = (int)(grid.x * grid.y * grid.z);
  hipLaunchKernelGGL((k_gemm_batched_gen<TA, TB, G>), dim3((ntile + 7) / 8 * 8), dim3(256 * G), smem, s, M, N, K, A,
                     lda, strideA, B, ldb, strideB, bias, strideBias, C, ldc, strideC, relu, accumulate, vecA, vecB,
                     mask, ldm, strideM, (int)grid.x, (int)grid.y, ntile);
  LAUNCHCHK();
  return 0;
}

int gemm_batched_launch(int batch, int transA, int transB, int M, int N, int K, const float *A, int lda,
                               long long strideA, const float *B, int ldb, long long strideB,
                               const float *bias, long long strideBias, float *C, int ldc, long long strideC,
                               int relu, int accumulate, const float *mask, int ldm, long long strideM,
                               hipStream_t s) {
  if (batch == 0 || M == 0 || N == 0) return 0;
  // 128-bit loads only where every row of every matrix of the batch starts on a 16-byte boundary
  const int vecA = !((uintptr_t)A & 15) && !(strideA & 3) && !(lda & 3);
  const int vecB = !((uintptr_t)B & 15) && !(strideB & 3) && !(ldb & 3);
  dim3 grid((N + 63) / 64, (M + 63) / 64, batch);
  const int nslab = (K + 31) / 32;
  // measured on the SAC update (tools/gemm_bench.py, tools/time_sac.py): 2 groups (74 KB of LDS, two
  // tiles per CU, so kernels of the update's two streams can share a CU) beat 1 and 4
  int G = g_gemm_kgroups ? g_gemm_kgroups : (nslab >= 2 ? 2 : 1);
#define GG(TA, TB, GN) gemm_batched_launch_g<TA, TB, GN>(grid, s, M, N, K, A, lda, strideA, B, ldb, strideB, bias, strideBias, C, ldc, strideC, relu, accumulate, vecA, vecB, mask, ldm, strideM)
#define GT(GN) (transA ? (transB ? GG(true, true, GN) : GG(true, false, GN)) : (transB ? GG(false, true, GN) : GG(false, false, GN)))
  return G == 4 ? GT(4) : (G == 2 ? GT(2) : GT(1));
#undef GT
#undef GG
}

// Every batched product of the C ABI: the grouped kernel of the learner (aomarl_gemm_g.h: operands in either
// orientation, bias / ReLU epilogue) when every operand row starts on 16 bytes, round 1's general kernel otherwise
// (odd leading dimensions, accumulation into C).  transA: A is [K][M]; transB: B is [K][N] (else [N][K]).
static int gemm_batched_any(int batch, int transA, int transB, int M, int N, int K, const float *A, int lda,
                            long long strideA, const float *B, int ldb, long long strideB, const float *bias,
                            long long strideBias, float *C, int ldc, long long strideC, int relu, int accumulate,
                            hipStream_t s) {
  if (batch == 0 || M == 0 || N == 0) return 0;
  const bool al = !((uintptr_t)A & 15) && !((uintptr_t)B & 15) && !(lda & 3) && !(ldb & 3) && !(strideA & 3) && !(strideB & 3);
  // k_gemm_g fetches 16-byte pieces along each operand's contiguous dimension: a row's last piece reaches up to three
  // floats past its K (M, N) elements -- inside the leading dimension, but past the END of a tightly sized buffer in the
  // last row of the last matrix.  The public entry points promise nothing beyond (rows - 1) * ld + length floats, so
  // only whole pieces go there; a ragged inner dimension takes the general kernel (element-wise tail).
  const bool whole = !((transA ? M : K) & 3) && !((transB ? N : K) & 3);
  if (al && whole && !accumulate && K > 0 && !g_gemm_kgroups) {
    return gemm_g_batched(batch, !transA, !transB, M, N, K, A, lda, strideA, B, ldb, strideB, bias, strideBias, C, ldc,
                          strideC, relu, s);
  }
  return gemm_batched_launch(batch, transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, bias, strideBias, C, ldc,
                             strideC, relu, accumulate, nullptr, 0, 0, s);
}

int aomarl_gemm_batched(int batch, int transA, int transB, int M, int N, int K, const float *A, int lda,
                        long long strideA, const float *B, int ldb, long long strideB, const float *bias,
                        long long strideBias, float *C, int ldc, long long strideC, int relu,
                        int accumulate, void *stream) {
  if (!A || !B || !C) return fail("gemm_batched: null pointer");
  if (batch < 0 || M < 0 || N < 0 || K < 0 || ldc < N) return fail("gemm_batched: bad sizes");
  if (lda < (transA ? M : K) || ldb < (transB ? N : K)) return fail("gemm_batched: leading dimension too small");
  return gemm_batched_any(batch, transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, bias, strideBias,
                          C, ldc, strideC, relu, accumulate, (hipStream_t)stream);
}

// C[b] = act(A[b] . B[b]^T + bias[b]): B in nn.Linear's [out][in] layout (the actors' layer-by-layer path)
int aomarl_gemm_nt_batched(int batch, int M, int N, int K, const float *A, int lda, long long strideA,
                           const float *B, int ldb, long long strideB, const float *bias,
                           long long strideBias, float *C, int ldc, long long strideC, int relu,
                           void *stream) {
  if (!A || !B || !C) return fail("gemm_nt_batched: null pointer");
  if (batch < 0 || M < 0 || N < 0 || K < 0 || lda < K || ldb < K || ldc < N) return fail("gemm_nt_batched: bad sizes");
  return gemm_batched_any(batch, 0, 0, M, N, K, A, lda, strideA, B, ldb, strideB, bias, strideBias, C, ldc, strideC,
                          relu, 0, (hipStream_t)stream);
}

// (the multi-agent SAC update, section 8f, and the grouped GEMM it runs on are a translation unit of their own:
// aomarl_sac.hip)
