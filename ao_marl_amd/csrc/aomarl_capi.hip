// aomarl_capi.hip -- C ABI (include/aomarl.h) over the gfx950 kernels in aomarl_kernels.hip: context, create / destroy,
// setters and options here; the entry points by concern in aomarl_capi_*.hip, included at the end (one translation unit).
// Host side only sequences kernel launches on the caller's stream; no device<->host copies after
// aomarl_create / aomarl_set_* (aomarl_reset uploads env_count seeds, 4 bytes each).
#include "aomarl_kernels.hip"

#include <math.h>
#include <stdlib.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <climits>
#include <string>
#include <vector>

static thread_local char g_err[512] = "";

int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}


struct aomarl_ctx {
  DevSys sys;
  std::vector<void *> owned;       // device allocations
  // host copies needed for scheduling
  int nlayers = 0, ndm = 0;
  int dim[AOMARL_MAX_LAYERS], ns[AOMARL_MAX_LAYERS], abclass[AOMARL_MAX_LAYERS];
  float deltax[AOMARL_MAX_LAYERS], deltay[AOMARL_MAX_LAYERS];
  int nclass = 0;
  int maxdim = 0, maxK = 0;
  float gain = 0.f, delay = 0.f;
  bool spot_fast = false;
  bool force_generic_dm = false, force_valu_target = false;
  bool force_generic_spot = false, force_generic_target = false, force_unfused_frame = false;
  int dft_mode = -1;                   // frame kernel DFTs: -1 follow the library's precision mode, 0 fp32 MFMAs, 1 split-fp16 ("force_f32_dft")
  bool reset_untransposed = false;     // "reset_untransposed": the reset's x extrusions on the row-major screen itself
  int small_chain = 1;                 // "small_chain": small systems run the control / agent chain of aomarl_env_step as two kernels
  hipEvent_t ride_ev = nullptr;        // frame pipeline: the event a whole-batch one-launch move may carry on its dispatch
  bool rode = false;                   // ... and whether it did
  int residual_shortcut = 0;           // "residual_shortcut": aomarl_env_step takes the residual modes from ONE product with v2m . cmat (aomarl_set_slopes2modes)
  bool skip_do_control = false;        // (aomarl_env_step's small chain: aomarl_next_part_one leaves do_control to its tail kernel)
  int small_move = 1;                  // "small_move": 1 = one k_move_small launch per frame's move where the screens allow it
  bool small_ok = false;               // every layer has dim <= MOVE_SMALL_DIM, ns + dim <= MOVE_SMALL_K (transposed [A|B] uploaded)
  int reset_streams = 2;               // "reset_streams": a batch reset in that many parts side by side, one stream each (1..4)
  bool reset_prefetch_whole = true;    // "reset_prefetch_whole": the prefetched reset as ONE range with the parts' tile and k split
  hipEvent_t ev_reset = nullptr, ev_reset2[3] = {nullptr, nullptr, nullptr};
  bool no_extrude_sg = false;          // "extrude_unfused": scatter and gather of consecutive rounds as separate launches
  bool defer_dm_shape = false;         // composites: stack-array phase from st->voltage on the fly
  int fused_debug = 0;                 // development switches of k_frame_fused (tools/fw_ab.py, tools/fw_pmc.py)
  // "prefetch_atmos": the composite moves the atmosphere of the NEXT frame on a side stream as soon
  // as this frame's image kernels are done, so the extrusion chain runs beside do_control / the
  // agents / next_part_two instead of in front of the next image
  bool prefetch_atmos = false, premoved = false;
  // prefetch: the side stream has not yet waited for the frame kernel that reads the screens (ev_frame): the
  // first kernel that WRITES them does (extrude_rounds); gather and GEMM of the first round run beside it
  bool frame_wait_pending = false;
  bool screens_dirty_main = true;       // the screens / origins were last written on the caller's stream
  // power-of-two scales of the static matrices for the split-f16 GEMM (gemm_scale)
  float cmat_scale = 1.f, v2m_scale = 1.f, m2v_scale = 1.f, s2m_scale = 1.f, ab_scale[AOMARL_MAX_LAYERS] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  // "subpixel_flow" (experiment, profiles/r03_subpixel_flow.txt): the fractional remainder of the wind
  // accumulators as a sub-pixel shift of the layer windows in the generic bilinear raytrace
  // (aomarl_raytrace_wfs / _target); 0 (default): integer-pixel frozen flow
  bool subpixel_flow = false;
  float frac_x[AOMARL_MAX_LAYERS] = {0, 0, 0, 0, 0, 0, 0, 0}, frac_y[AOMARL_MAX_LAYERS] = {0, 0, 0, 0, 0, 0, 0, 0};
  // "graph_step": aomarl_env_step as a HIP graph (captured once per distinct launch sequence -- extrusion plan,
  // ring position, buffer addresses --, replayed afterwards: one hipGraphLaunch instead of ~25 launches and ~8
  // event operations per step).  capturing: the body is being recorded (no timed / event-carrying dispatch, the
  // side streams fork from and join the caller's stream inside the graph).  side_joined: the caller's stream
  // has already waited for everything issued on the side streams (the waits for ev_moved / ev_psf are skipped)
  bool graph_step = false, capturing = false, side_joined = false, fork_recorded = false;
  hipEvent_t ev_fork = nullptr;
  struct StepGraph { std::vector<long long> key; hipGraphExec_t exec; hipGraph_t graph; unsigned long long arith[AR_N]; int fw_variant[6]; };
  std::vector<StepGraph> graphs;
  unsigned long long graph_hits = 0, graph_captures = 0;
  // bumped by every call that changes a host value captured graphs have baked into their kernel arguments (matrix
  // scales and leading dimensions, nact, gains, per-context options): part of the graph key, together with the
  // process-wide g_cfg_epoch -- a re-uploaded matrix usually lands at the SAME address, so the pointers alone
  // would replay a stale graph
  unsigned long long cfg_epoch = 0;
  const int32_t *sel_checked = nullptr;     // aomarl_env_step: the column selection last validated
  int sel_checked_n = 0, sel_checked_nm = 0;
  int fw_variant[6] = {0, 0, 0, 0, 0, 0};   // template arguments of the last k_frame_wave launch
  char fw_name[96] = {0};
  // "time_frame_kernel": a HIP event pair around every k_frame_wave launch (aomarl_frame_kernel_time)
  bool time_fw = false;
  std::vector<hipEvent_t> fw_ev;            // 2 per timed launch, created on demand
  size_t fw_ev_used = 0;
  std::vector<hipEvent_t> fw_ev_retired;    // timing events a frame in flight still carries as its "done" mark (fw_ev_rewind)
  hipStream_t atm_stream = nullptr, psf_stream = nullptr;
  hipEvent_t ev_frame = nullptr, ev_moved = nullptr, ev_psf = nullptr;
  // the event the screens' readers were last marked with on the caller's stream (ev_frame, or the closing
  // event of a timed frame launch: an event record is a barrier packet of its own on the queue, 3-5 us of
  // the step each -- one per frame, not three); frame_marked: recorded by aomarl_frame_fused and nothing
  // launched on that stream since (the composites' prefetch reuses it)
  hipEvent_t ev_frame_cur = nullptr;
  bool frame_marked = false;
  // frame pipeline (aomarl_set_frame_pipeline): frames on a stream of their own, one step ahead of the chains
  struct FramePipe {
    bool have_twin = false, active = false;
    const float *owner_screens = nullptr;          // the state the twin belongs to
    aomarl_state twin;                             // odd frames: slopes / voltage / dm_shape / work
    int par = 0;                                   // parity (0: st's buffers, 1: the twin's) of the frame in flight
    hipStream_t fstream = nullptr;
    hipEvent_t ev_cmd = nullptr, ev_commit = nullptr, ev_done[2] = {nullptr, nullptr}, ev_psf[2] = {nullptr, nullptr};
    hipEvent_t ev_done_cur[2] = {nullptr, nullptr};  // the event each parity's last frame launch carries (ev_done[], or a timing event)
    bool psf_out[2] = {false, false};              // a PSF finish of that parity may still run
    bool cmd_covers_commit = false;                // ev_cmd was recorded behind the Strehl commit and the PSF-finish wait as well
    bool cmd_covers_psf = false;                   // ev_cmd was recorded behind the PSF-finish wait (not the commit)
    int32_t *snap[2] = {nullptr, nullptr};         // ring origins as of each parity's frame
    size_t snap_ints = 0;
    unsigned long long steps = 0, overlapped = 0, behind = 0;
  } pipe;
  int32_t *snap_target = nullptr;        // a pipelined move: where the scatter kernels also write the origins they advance
  bool snap_complete = false;            // ... and whether every layer of every environment group was advanced by it
  bool pipe_enabled = true;              // "frame_pipeline": 0 = plain call order although a twin is set
  bool pipe_internal = false;            // check_range: the pipelined step itself is calling
  // first write of a prefetched move: behind the OLDER frame in flight (ev_frame_prev) when the lines the move
  // rewrites are outside the frame kernel's windows (group_overlap, decided per plan), else behind the newest
  hipEvent_t ev_frame_prev = nullptr;
  bool need_prev = false, group_overlap = false;
  bool psf_side = false;                // a k_target_finish_mfma launched on the side stream may still run
  const float *pre_screens = nullptr;
  int pre_b = 0, pre_n = 0;
  // controller matrices
  float *cmat = nullptr;           // [nactu][ld_s]
  int ld_cmat = 0;
  int nmodes = 0, nact = 0, ld_v2m = 0, ld_m2v = 0;
  float *v2m = nullptr, *m2v = nullptr, *freedom = nullptr;
  float *s2m = nullptr;            // [nmodes][ld_cmat]: v2m . cmat (aomarl_set_slopes2modes)
  int s2m_nmodes = 0;
  int32_t *amodes = nullptr, *amode_inv = nullptr;   // action modes and their inverse map [nmodes]
  float *env_gain = nullptr;       // per-environment integrator gains (aomarl_set_env_gains) or null
  int env_gain_n = 0;
  uint32_t *seed_stage = nullptr;  // device staging for reset seeds
  int seed_stage_n = 0;
  // aomarl_reset_prefetch_*: the NEXT episode's screens grown in a shadow state while this episode runs
  struct ResetPrefetch *rp = nullptr;
  // aomarl_target_image: DFT tables and per-environment scratch of the full-frame PSF (on demand)
  float *timg = nullptr;
  // geometric controller (aomarl_set_geo): host copies of the lattice tables it is built from,
  // projection operands on the device
  std::vector<int32_t> h_grid;     // [gh][gw] actuator index or -1 (stack-array DM 0)
  std::vector<float> h_prof, h_spupil, h_tt;
  float *geoW = nullptr, *geoUx = nullptr, *geoUy = nullptr, *geoPlanes = nullptr;
  int32_t *geoMap = nullptr;       // stack-array actuator -> j * gh + i of the lattice product
  int geo_ldw = 0, geo_gw = 0, geo_gh = 0, geo_npzt = 0, geo_ldr = 0;
};

static int pipe_drop(aomarl_ctx *c, void *stream);
static void rp_free(aomarl_ctx *c);
// Start the timing events over.  The closing event of a timed frame launch doubles as that frame's "readers are
// done" mark (pipe.ev_done_cur / ev_frame_cur / ev_frame_prev): one that a frame in flight still carries must not be
// re-recorded by a later launch, so it is retired (it stays valid for whoever waits on it) and replaced.
static int fw_ev_rewind(aomarl_ctx *c) {
  const hipEvent_t held[4] = {c->pipe.ev_done_cur[0], c->pipe.ev_done_cur[1], c->ev_frame_cur, c->ev_frame_prev};
  for (size_t i = 0; i < c->fw_ev.size(); i++)
    for (int k = 0; k < 4; k++)
      if (held[k] && c->fw_ev[i] == held[k]) {
        hipEvent_t ne;
        HIPCHK(hipEventCreate(&ne));
        c->fw_ev_retired.push_back(c->fw_ev[i]);
        c->fw_ev[i] = ne;
        break;
      }
  if (c->fw_ev_retired.size() > 64 && !c->pipe.active) {      // nothing in flight refers to them any more
    bool live = false;
    for (hipEvent_t e : c->fw_ev_retired) for (int k = 0; k < 4; k++) live = live || e == held[k];
    if (!live) { for (hipEvent_t e : c->fw_ev_retired) (void)hipEventDestroy(e); c->fw_ev_retired.clear(); }
  }
  c->fw_ev_used = 0;
  return 0;
}
static int frame_fused_impl(aomarl_ctx *c, aomarl_state *st, int b, int n, int flags, void *stream, int slot);

static unsigned long long g_cfg_epoch = 0;      // process-wide options / precision changes (see aomarl_ctx::cfg_epoch)
const char *aomarl_last_error(void) { return g_err; }
int aomarl_abi_version(void) { return AOMARL_ABI_VERSION; }

int aomarl_set_precision(int mode) {
  g_cfg_epoch++;
  if (mode != AOMARL_PRECISION_F32 && mode != AOMARL_PRECISION_SPLIT_F16) return fail("set_precision: unknown mode %d", mode);
  g_precision = mode;
  g_gemm_split_f16 = mode == AOMARL_PRECISION_SPLIT_F16;
  return 0;
}
int aomarl_get_precision(void) { return g_precision; }
int aomarl_gemm_saturated(unsigned *count, void *stream) {
  if (!count) return fail("gemm_saturated: null argument");
  *count = 0;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !g_gemm_sat[dev]) return 0;       // no split-fp16 GEMM ever ran on this device
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(count, g_gemm_sat[dev], sizeof(unsigned), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (*count) HIPCHK(hipMemsetAsync(g_gemm_sat[dev], 0, sizeof(unsigned), s));
  return 0;
}
int aomarl_arith_families(void) { return AR_N; }
const char *aomarl_arith_family_name(int i) { return (i >= 0 && i < AR_N) ? g_arith_name[i] : ""; }
unsigned long long aomarl_arith_launches(int i) { return (i >= 0 && i < AR_N) ? g_arith[i] : 0ULL; }
void aomarl_arith_reset(void) { for (int i = 0; i < AR_N; i++) g_arith[i] = 0ULL; }

template <typename T>
static int upload(aomarl_ctx *c, const T *host, size_t n, T **dev) {
  *dev = nullptr;
  if (n == 0) return 0;
  if (!host) return fail("null host array in descriptor");
  void *p = nullptr;
  HIPCHK(hipMalloc(&p, n * sizeof(T)));
  c->owned.push_back(p);
  HIPCHK(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
  *dev = (T *)p;
  return 0;
}

static bool is_int(float v) { return floorf(v) == v; }

int aomarl_create(const aomarl_desc *d, aomarl_ctx **out) {
  if (!d || !out) return fail("aomarl_create: null argument");
  if (d->abi_version != AOMARL_ABI_VERSION)
    return fail("aomarl_create: ABI version %d, library is %d", d->abi_version, AOMARL_ABI_VERSION);
  if (d->nlayers < 0 || d->nlayers > AOMARL_MAX_LAYERS) return fail("nlayers out of range");
  if (d->ndm < 1 || d->ndm > AOMARL_MAX_DMS) return fail("ndm out of range");
  if (d->n <= 0 || d->pupdiam <= 0 || d->n < d->pupdiam) return fail("bad pupil sizes");
  if (d->pupdiam % 4 || d->n % 4) return fail("pupil grid sizes must be multiples of 4");
  if (d->npsf & (d->npsf - 1)) return fail("npsf must be a power of two");
  if (!(d->strehl_halfwin == 4 || d->strehl_halfwin == 8 || d->strehl_halfwin == 16))
    return fail("strehl_halfwin must be 4, 8 or 16");
  aomarl_ctx *c = new aomarl_ctx();
  DevSys &s = c->sys;
  memset(&s, 0, sizeof(s));
  int rc = 0;
#define UP(T, host, n, dev)                                      \
  do {                                                           \
    T *_p;                                                       \
    rc = upload<T>(c, host, n, &_p);                             \
    if (rc) { aomarl_destroy(c); return rc; }                    \
    dev = _p;                                                    \
  } while (0)
  s.n = d->n; s.pupdiam = d->pupdiam;
  UP(float, d->mpupil, (size_t)d->n * d->n, s.mpupil);
  UP(float, d->spupil, (size_t)d->pupdiam * d->pupdiam, s.spupil);
  c->h_spupil.assign(d->spupil, d->spupil + (size_t)d->pupdiam * d->pupdiam);
  s.nvalid = d->nvalid; s.pdiam = d->pdiam; s.nfft = d->nfft; s.npix = d->npix;
  s.nrebin = d->nrebin; s.nxsub = d->nxsub;
  const int pd2 = d->pdiam * d->pdiam;
  UP(int32_t, d->phasemap, (size_t)pd2 * d->nvalid, s.phasemap);
  // every sub-aperture must be a contiguous pdiam x pdiam tile of the phase grid
  std::vector<int32_t> sub(d->nvalid);
  for (int i = 0; i < d->nvalid; i++) {
    int p0 = d->phasemap[i];
    int x0 = p0 % d->n, y0 = p0 / d->n;
    if (x0 + d->pdiam > d->n || y0 + d->pdiam > d->n) { aomarl_destroy(c); return fail("phasemap tile outside the phase grid"); }
    for (int k = 0; k < pd2; k++)
      if (d->phasemap[(size_t)k * d->nvalid + i] != p0 + (k % d->pdiam) + d->n * (k / d->pdiam)) {
        aomarl_destroy(c);
        return fail("phasemap of sub-aperture %d is not a contiguous tile", i);
      }
    sub[i] = x0 | (y0 << 16);
  }
  UP(int32_t, sub.data(), sub.size(), s.sub_xy);
  std::vector<float> hrev(pd2);
  for (int k = 0; k < pd2; k++) hrev[k] = (float)((double)d->halfxy[k] / (2.0 * M_PI));
  UP(float, hrev.data(), hrev.size(), s.halfxy);
  UP(int32_t, d->binmap, (size_t)d->nrebin * d->nrebin * d->npix * d->npix, s.binmap);
  UP(float, d->flux, (size_t)d->nvalid, s.flux);
  UP(int32_t, d->validsubsx, (size_t)d->nvalid, s.validx);
  UP(int32_t, d->validsubsy, (size_t)d->nvalid, s.validy);
  s.nphot = d->nphot; s.wfs_inv_lambda = 1.0f / d->wfs_lambda; s.noise = d->noise;
  s.cog_offset = d->cog_offset; s.cog_scale = d->cog_scale; s.subapd = d->subapd;
  c->spot_fast = (d->pdiam == 16 && d->nfft == 64 && d->nrebin == 2 && d->npix == 16);
  if (c->spot_fast) {
    // the kernel hard-codes the binmap of this sampling: LR (Y, X) <- HR rows/cols
    // (2Y-16 .. 2Y-15) mod 64 ; check the map handed in says the same
    for (int px = 0; px < 256 && c->spot_fast; px++) {
      int Y = px / 16, X = px % 16;
      bool seen[4] = {false, false, false, false};
      for (int r = 0; r < 4; r++) {
        int hr = d->binmap[r * 256 + px];
        int ky = hr / 64, kx = hr % 64;
        int dy = (ky - (2 * Y - 16) + 64) % 64, dx = (kx - (2 * X - 16) + 64) % 64;
        if (dy > 1 || dx > 1) { c->spot_fast = false; break; }
        seen[dy * 2 + dx] = true;
      }
      if (!(seen[0] && seen[1] && seen[2] && seen[3])) c->spot_fast = false;
    }
  }
  if (c->spot_fast) {
    // the kernel folds the half-pixel ramp into half-integer DFT frequencies: it must be the
    // reference's halfxy = pi (x + y) / Nfft (geom_init.py:689-692)
    for (int k = 0; k < pd2 && c->spot_fast; k++) {
      double want = M_PI * (double)((k % d->pdiam) + (k / d->pdiam)) / (double)d->nfft;
      if (fabs((double)d->halfxy[k] - want) > 2e-6) c->spot_fast = false;
    }
  }
  if (!c->spot_fast) {
    aomarl_destroy(c);
    return fail("unsupported WFS sampling (pdiam=%d nfft=%d nrebin=%d npix=%d, binmap, halfxy): the "
                "spot kernel is specialised for 16/64/2/16 with the standard half-pixel ramp",
                d->pdiam, d->nfft, d->nrebin, d->npix);
  }
  // layers
  c->nlayers = s.nlayers = d->nlayers;
  long long off = 0;
  const float *seenA[AOMARL_MAX_LAYERS]; const float *seenB[AOMARL_MAX_LAYERS];
  const float *devAB[AOMARL_MAX_LAYERS]; int ldab[AOMARL_MAX_LAYERS];
  const float *devABt[AOMARL_MAX_LAYERS]; int ldt[AOMARL_MAX_LAYERS];
  c->small_ok = true;
  s.wfs_all_int = 1; s.tar_all_int = 1;
  for (int l = 0; l < d->nlayers; l++) {
    const aomarl_layer_desc &L = d->layers[l];
    DevLayer &D = s.layers[l];
    if (L.dim <= 0 || L.dim > 65535 || L.nstencil <= 0) { aomarl_destroy(c); return fail("bad layer %d", l); }
    // the mirror columns repeat the row's first RING_PAD pixels: a narrower screen would mirror pixels it is
    // writing (k_refresh_mirror, k_set_screen, the scatter's px < RING_PAD)
    if (L.dim < RING_PAD) { aomarl_destroy(c); return fail("layer %d: a screen of %d pixels is narrower than the ring's %d mirror columns", l, L.dim, RING_PAD); }
    D.dim = L.dim; D.ns = L.nstencil; D.screen_off = off; off += (long long)L.dim * (L.dim + RING_PAD);
    c->dim[l] = L.dim; c->ns[l] = L.nstencil; c->deltax[l] = L.deltax; c->deltay[l] = L.deltay;
    if (L.dim > c->maxdim) c->maxdim = L.dim;
    if (L.dim + L.nstencil > c->maxK) c->maxK = L.dim + L.nstencil;
    std::vector<uint32_t> ix(L.nstencil), iy(L.nstencil);
    for (int k = 0; k < L.nstencil; k++) {
      if (L.istx[k] >= (uint32_t)(L.dim * L.dim) || L.isty[k] >= (uint32_t)(L.dim * L.dim)) { aomarl_destroy(c); return fail("stencil index out of range"); }
      ix[k] = (L.istx[k] % L.dim) | ((L.istx[k] / L.dim) << 16);
      iy[k] = (L.isty[k] % L.dim) | ((L.isty[k] / L.dim) << 16);
    }
    UP(uint32_t, ix.data(), ix.size(), D.istx);
    UP(uint32_t, iy.data(), iy.size(), D.isty);
    {
      std::vector<uint32_t> it(ix.size());
      for (size_t k = 0; k < ix.size(); k++) it[k] = (ix[k] >> 16) | (ix[k] << 16);
      UP(uint32_t, it.data(), it.size(), D.istT);
    }
    // [A | B] concatenated, shared between layers that were given the same host matrices
    int cls = -1;
    for (int m = 0; m < c->nclass; m++)
      if (seenA[m] == L.A && seenB[m] == L.B) cls = m;
    if (cls < 0) {
      cls = c->nclass++;
      seenA[cls] = L.A; seenB[cls] = L.B;
      int K = L.nstencil + L.dim;
      int ld = (K + 3) & ~3;
      std::vector<float> ab((size_t)L.dim * ld, 0.f);
      for (int r = 0; r < L.dim; r++) {
        memcpy(&ab[(size_t)r * ld], L.A + (size_t)r * L.nstencil, sizeof(float) * L.nstencil);
        memcpy(&ab[(size_t)r * ld + L.nstencil], L.B + (size_t)r * L.dim, sizeof(float) * L.dim);
      }
      float *p;
      UP(float, ab.data(), ab.size(), p);
      devAB[cls] = p; ldab[cls] = ld;
      devABt[cls] = nullptr; ldt[cls] = 0;
      if (L.dim <= MOVE_SMALL_DIM && K <= MOVE_SMALL_K) {          // small screens: the transpose for k_move_small
        const int lt = (L.dim + 63) & ~63;
        std::vector<float> abt((size_t)K * lt, 0.f);
        for (int r = 0; r < L.dim; r++)
          for (int j = 0; j < K; j++) abt[(size_t)j * lt + r] = ab[(size_t)r * ld + j];
        float *pt;
        UP(float, abt.data(), abt.size(), pt);
        devABt[cls] = pt; ldt[cls] = lt;
      }
      c->ab_scale[cls] = gemm_scale(ab.data(), ab.size());
    }
    c->abclass[l] = cls;
    D.AB = devAB[cls]; D.ldab = ldab[cls];
    D.ABt = devABt[cls]; D.ldt = ldt[cls];
    if (!D.ABt) c->small_ok = false;
    D.amp = L.amplitude;
    D.wxo = L.wfs_xoff; D.wyo = L.wfs_yoff; D.txo = L.tar_xoff; D.tyo = L.tar_yoff;
    D.wox = (int)L.wfs_xoff; D.woy = (int)L.wfs_yoff; D.tox = (int)L.tar_xoff; D.toy = (int)L.tar_yoff;
    if (!is_int(L.wfs_xoff) || !is_int(L.wfs_yoff)) s.wfs_all_int = 0;
    if (!is_int(L.tar_xoff) || !is_int(L.tar_yoff)) s.tar_all_int = 0;
    if (s.wfs_all_int && (D.wox < 0 || D.woy < 0 || D.wox + d->n > L.dim || D.woy + d->n > L.dim)) { aomarl_destroy(c); return fail("WFS window leaves screen %d", l); }
    if (s.tar_all_int && (D.tox < 0 || D.toy < 0 || D.tox + d->pupdiam > L.dim || D.toy + d->pupdiam > L.dim)) { aomarl_destroy(c); return fail("target window leaves screen %d", l); }
  }
  s.screen_stride = off;
  // DMs
  c->ndm = s.ndm = d->ndm;
  long long soff = 0; int coff = 0;
  for (int k = 0; k < d->ndm; k++) {
    const aomarl_dm_desc &M = d->dms[k];
    DevDm &D = s.dms[k];
    D.type = M.type; D.dim = M.dim; D.nact = M.nact; D.ss = M.influsize;
    D.shape_off = soff;
    soff += (M.type == AOMARL_DM_TT) ? 4 : (long long)M.dim * M.dim;   // TT: 2 commands (+pad)
    D.com_off = coff; coff += M.nact;
    if (M.type == AOMARL_DM_PZT) {
      UP(float, M.influ, (size_t)M.nact * M.influsize * M.influsize, D.influ);
      UP(int32_t, M.influpos, (size_t)M.ninflupos, D.influpos);
      UP(int32_t, M.ninflu, (size_t)M.dim * M.dim, D.ninflu);
      UP(int32_t, M.influstart, (size_t)M.dim * M.dim, D.influstart);
      // bounds of the gather tables (a bad table would fault on the device)
      long long tot = 0;
      for (long long p = 0; p < (long long)M.dim * M.dim; p++) {
        if (M.influstart[p] != tot || M.ninflu[p] < 0) { aomarl_destroy(c); return fail("DM %d: influstart/ninflu inconsistent", k); }
        tot += M.ninflu[p];
      }
      if (tot != M.ninflupos) { aomarl_destroy(c); return fail("DM %d: sum(ninflu) != len(influpos)", k); }
      for (long long q = 0; q < M.ninflupos; q++)
        if (M.influpos[q] < 0 || M.influpos[q] >= M.nact * M.influsize * M.influsize) { aomarl_destroy(c); return fail("DM %d: influpos out of range", k); }
      // ---- separable-lattice fast path: recover (i1, j1) of every actuator from the gather
      // tables (first pixel that references sample 0 of its patch), check that all patches are
      // one rank-1 profile and that the actuators sit on a regular lattice.
      {
        const int ss = M.influsize, ss2 = ss * ss, na = M.nact;
        std::vector<int> i1(na, INT32_MIN), j1(na, INT32_MIN);
        for (long long p = 0; p < (long long)M.dim * M.dim; p++)
          for (int t = 0; t < M.ninflu[p]; t++) {
            int pos = M.influpos[M.influstart[p] + t];
            int act = pos / ss2, rem = pos % ss2, a = rem / ss, b = rem % ss;   // b: x offset
            int x = (int)(p % M.dim) - b, y = (int)(p / M.dim) - a;
            if (i1[act] == INT32_MIN) { i1[act] = x; j1[act] = y; }
            else if (i1[act] != x || j1[act] != y) { i1[act] = INT32_MAX; }
          }
        bool ok = na > 0 && ss <= 128;
        for (int a = 0; a < na && ok; a++) ok = (i1[a] != INT32_MIN && i1[a] != INT32_MAX);
        // identical patches
        for (int a = 1; a < na && ok; a++)
          ok = memcmp(M.influ + (size_t)a * ss2, M.influ, sizeof(float) * ss2) == 0;
        std::vector<float> prof(ss, 0.f);
        if (ok) {
          int cdx = 0; float best = 0.f;
          for (int t = 0; t < ss; t++) if (M.influ[t * ss + t] > best) { best = M.influ[t * ss + t]; cdx = t; }
          ok = best > 0.f;
          if (ok) {
            const float sc = 1.0f / sqrtf(best);
            float mx = 0.f;
            for (int t = 0; t < ss; t++) prof[t] = M.influ[t * ss + cdx] * sc;
            for (int a = 0; a < ss; a++)
              for (int b = 0; b < ss; b++) mx = fmaxf(mx, fabsf(M.influ[a * ss + b] - prof[a] * prof[b]));
            ok = mx <= 2e-6f * best;
          }
        }
        int pitch = 0, imin = INT32_MAX, jmin = INT32_MAX, imax = INT32_MIN, jmax = INT32_MIN;
        if (ok) {
          auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
          for (int a = 0; a < na; a++) { imin = std::min(imin, i1[a]); jmin = std::min(jmin, j1[a]); imax = std::max(imax, i1[a]); jmax = std::max(jmax, j1[a]); }
          for (int a = 0; a < na; a++) { pitch = gcd(pitch, i1[a] - imin); pitch = gcd(pitch, j1[a] - jmin); }
          ok = pitch > 0 && ss <= 4 * pitch && (DMS_TX + ss - 1) / pitch + 2 <= DMS_GX && (DMS_TY + ss - 1) / pitch + 2 <= DMS_GY;
        }
        if (ok) {
          const int gw = (imax - imin) / pitch + 1, gh = (jmax - jmin) / pitch + 1;
          std::vector<int32_t> grid((size_t)gw * gh, -1);
          for (int a = 0; a < na && ok; a++) {
            int32_t &cell = grid[(size_t)((j1[a] - jmin) / pitch) * gw + (i1[a] - imin) / pitch];
            if (cell != -1) ok = false;
            cell = a;
          }
          if (ok) {
            D.sep = 1; D.pitch = pitch; D.i1min = imin; D.j1min = jmin; D.gw = gw; D.gh = gh;
            UP(int32_t, grid.data(), grid.size(), D.grid);
            UP(float, prof.data(), prof.size(), D.prof);
            if (k == 0) { c->h_grid = grid; c->h_prof = prof; }
          }
        }
      }
    } else if (M.type == AOMARL_DM_TT) {
      if (M.nact != 2) { aomarl_destroy(c); return fail("tip-tilt DM must have 2 actuators"); }
      UP(float, M.influ, (size_t)M.dim * M.dim * 2, D.influ);
      if (k == 1) c->h_tt.assign(M.influ, M.influ + (size_t)M.dim * M.dim * 2);
    } else {
      aomarl_destroy(c);
      return fail("DM %d: unknown type %d", k, M.type);
    }
    D.wxo = M.wfs_xoff; D.wyo = M.wfs_yoff; D.txo = M.tar_xoff; D.tyo = M.tar_yoff;
    D.wox = (int)M.wfs_xoff; D.woy = (int)M.wfs_yoff; D.tox = (int)M.tar_xoff; D.toy = (int)M.tar_yoff;
    if (!is_int(M.wfs_xoff) || !is_int(M.wfs_yoff)) s.wfs_all_int = 0;
    if (!is_int(M.tar_xoff) || !is_int(M.tar_yoff)) s.tar_all_int = 0;
    if (s.wfs_all_int && (D.wox < 0 || D.woy < 0 || D.wox + d->n > M.dim || D.woy + d->n > M.dim)) { aomarl_destroy(c); return fail("WFS window leaves DM %d", k); }
    if (s.tar_all_int && (D.tox < 0 || D.toy < 0 || D.tox + d->pupdiam > M.dim || D.toy + d->pupdiam > M.dim)) { aomarl_destroy(c); return fail("target window leaves DM %d", k); }
  }
  s.shape_stride = soff;
  if (coff != d->nactu) { aomarl_destroy(c); return fail("sum of DM actuators (%d) != nactu (%d)", coff, d->nactu); }
  if (d->nslope != 2 * d->nvalid) { aomarl_destroy(c); return fail("nslope must be 2*nvalid"); }
  s.nactu = d->nactu; s.nslope = d->nslope;
  // target
  s.tar_inv_lambda = 1.0f / d->tar_lambda; s.npsf = d->npsf; s.hw = d->strehl_halfwin;
  std::vector<float> tw((size_t)d->npsf * 2);
  for (int j = 0; j < d->npsf; j++) {
    double a = 2.0 * M_PI * (double)j / (double)d->npsf;
    tw[2 * j] = (float)cos(a); tw[2 * j + 1] = (float)sin(a);
  }
  UP(float, tw.data(), tw.size(), s.psf_tw);
  double sp = 0.;
  for (size_t p = 0; p < (size_t)d->pupdiam * d->pupdiam; p++) sp += d->spupil[p];
  s.ref_peak = (float)(sp * sp);
  c->gain = d->gain; c->delay = d->delay;
  if (c->delay < 0.f || c->delay > 2.f) { aomarl_destroy(c); return fail("delay must be in [0, 2]"); }
  // ---- fused frame kernel: the WFS tiles must BE the tiles of the pupil grid, at the same
  // screen / DM pixels, the pupil must be binary
  s.fused_ok = 0; s.ntiles = 0;
  {
    const int pd = d->pupdiam, pad = (d->n - pd) / 2;
    bool ok = s.wfs_all_int && s.tar_all_int && d->ndm == 2 && d->dms[0].type == AOMARL_DM_PZT &&
              d->dms[1].type == AOMARL_DM_TT && (d->nlayers == 1 || d->nlayers == 3) &&
              d->strehl_halfwin == 8 && pd % 16 == 0 && pad >= 0 && d->n == pd + 2 * pad && d->nvalid <= 0xFFFF &&
              (d->npsf & (d->npsf - 1)) == 0 && d->npsf <= 4096 && pd / 16 <= 256;
    for (int l = 0; l < d->nlayers && ok; l++)
      ok = s.layers[l].wox + pad == s.layers[l].tox && s.layers[l].woy + pad == s.layers[l].toy;
    for (int k = 0; k < d->ndm && ok; k++)
      ok = s.dms[k].wox + pad == s.dms[k].tox && s.dms[k].woy + pad == s.dms[k].toy;
    for (int y = 0; y < pd && ok; y++)
      for (int x = 0; x < pd && ok; x++) {
        const float m = d->spupil[(size_t)y * pd + x];
        ok = (m == 0.f || m == 1.f) && d->mpupil[(size_t)(y + pad) * d->n + x + pad] == m;
      }
    const int nt = pd / 16;
    std::vector<int32_t> tsub((size_t)std::max(nt * nt, 1), -1);
    std::vector<uint16_t> tmask((size_t)std::max(pd * nt, 1), 0);
    if (ok) {
      for (int y = 0; y < pd; y++)
        for (int x = 0; x < pd; x++)
          if (d->spupil[(size_t)y * pd + x] != 0.f) tmask[(size_t)y * nt + x / 16] |= (uint16_t)(1u << (x % 16));
      for (int r = 0; r < nt; r++)
        for (int t = 0; t < nt; t++) {
          bool lit = false;
          for (int yy = 0; yy < 16; yy++) lit = lit || tmask[(size_t)(16 * r + yy) * nt + t] != 0;
          if (!lit) tsub[(size_t)r * nt + t] = -2;
        }
      for (int i = 0; i < d->nvalid && ok; i++) {
        const int x0 = (sub[i] & 0xFFFF) - pad, y0 = (sub[i] >> 16) - pad;
        ok = x0 >= 0 && y0 >= 0 && x0 % 16 == 0 && y0 % 16 == 0 && x0 < pd && y0 < pd;
        if (ok) {
          int32_t &cell = tsub[(size_t)(y0 / 16) * nt + x0 / 16];
          ok = cell < 0;                 // one sub-aperture per tile (an unlit tile with a
          cell = i;                      // sub-aperture is still imaged: zero flux, zero slopes)
        }
      }
    }
    if (ok) {
      // tile_info: [15:0] sub-aperture, bit 16 lit, 17 every pixel lit, 18 valid sub-aperture
      std::vector<int32_t> tinfo(tsub.size(), 0);
      for (int r = 0; r < nt; r++)
        for (int t = 0; t < nt; t++) {
          const int32_t sb = tsub[(size_t)r * nt + t];
          bool full = true;
          for (int yy = 0; yy < 16; yy++) full = full && tmask[(size_t)(16 * r + yy) * nt + t] == 0xFFFF;
          int32_t v = 0;
          if (sb != -2) v |= 0x10000;
          if (full) v |= 0x20000;
          if (sb >= 0) { v |= 0x40000 | sb; v |= 0x10000; }     // an unlit tile with a sub-aperture is still imaged
          tinfo[(size_t)r * nt + t] = v;
        }
      UP(int32_t, tinfo.data(), tinfo.size(), s.tile_info);
      {
        if (nt > 127) { aomarl_destroy(c); return fail("pupil too wide for the frame kernel's tile list"); }
        std::vector<int32_t> linfo((size_t)nt * (nt + 8), 0), lcount(nt, 0);
        for (int r = 0; r < nt; r++) {
          int k = 0;
          for (int t = 0; t < nt; t++)
            if (tinfo[(size_t)r * nt + t] & 0x10000) linfo[(size_t)r * (nt + 8) + k++] = tinfo[(size_t)r * nt + t] | (t << 24);
          lcount[r] = k;
          for (int kk = k; kk < nt + 8; kk++) linfo[(size_t)r * (nt + 8) + kk] = k ? linfo[(size_t)r * (nt + 8) + k - 1] : 0;
        }
        UP(int32_t, linfo.data(), linfo.size(), s.lit_info);
        UP(int32_t, lcount.data(), lcount.size(), s.lit_count);
        // the same tiles in pairs (2 g, 2 g + 1): the frame kernel fetches the layer rows of a pair as whole
        // 128-byte pieces
        const int npm = (nt + 1) / 2 + 2;
        std::vector<int32_t> pinfo((size_t)nt * npm * 2, 0), pcount(nt, 0);
        for (int r = 0; r < nt; r++) {
          int k = 0;
          int32_t *row = &pinfo[(size_t)r * npm * 2];
          for (int g = 0; 2 * g < nt; g++) {
            const int ta = 2 * g, tb = 2 * g + 1 < nt ? 2 * g + 1 : 2 * g;
            const int32_t va = tinfo[(size_t)r * nt + ta], vb = 2 * g + 1 < nt ? tinfo[(size_t)r * nt + tb] : 0;
            if (!((va | vb) & 0x10000)) continue;
            row[2 * k] = ((va & 0x10000) ? va : 0) | (ta << 24);
            row[2 * k + 1] = ((vb & 0x10000) ? vb : 0) | (tb << 24);
            k++;
          }
          pcount[r] = k;
          for (int kk = k; kk < npm; kk++) {
            row[2 * kk] = k ? row[2 * (k - 1)] : 0;
            row[2 * kk + 1] = k ? row[2 * (k - 1) + 1] : 0;
          }
        }
        UP(int32_t, pinfo.data(), pinfo.size(), s.pair_info);
        UP(int32_t, pcount.data(), pcount.size(), s.pair_count);
      }
      {
        std::vector<int32_t> order(nt), work(nt, 0);
        for (int r = 0; r < nt; r++) {
          order[r] = r;
          for (int t = 0; t < nt; t++) {
            const int32_t v = tinfo[(size_t)r * nt + t];
            work[r] += (v & 0x10000) ? ((v & 0x40000) ? 4 : 1) : 0;   // a sub-aperture tile costs ~4x a bare one
          }
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return work[a] > work[b]; });
        UP(int32_t, order.data(), order.size(), s.stripe_order);
      }
      // PSF operand of the frame kernel for lane (q, c) of tile t: X = 16 t + 4 q + j (j = 0..3), column c of
      // [cos 2 pi k X / npsf (k = 1..8) | sin 2 pi k X / npsf (k = 1..8)]; fp32 and [hi | lo] split-fp16 form
      {
        std::vector<_Float16> tw((size_t)nt * 64 * 8);
        std::vector<float> twf((size_t)nt * 64 * 4);
        for (int t = 0; t < nt; t++)
          for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 4; j++) {
              const int x = 16 * t + 4 * (lane >> 4) + j, cc = lane & 15;
              const int k = cc < 8 ? cc + 1 : cc - 7;
              const double th = 2.0 * M_PI * (double)(((long long)k * x) % d->npsf) / (double)d->npsf;
              const float v = (float)(cc < 8 ? cos(th) : sin(th));
              const _Float16 hi = (_Float16)v;
              const _Float16 lo = (_Float16)(v - (float)hi);
              _Float16 *o = &tw[((size_t)t * 64 + lane) * 8];
              o[j] = hi; o[4 + j] = lo;
              twf[((size_t)t * 64 + lane) * 4 + j] = v;
            }
        const _Float16 *dev = nullptr;
        UP(_Float16, tw.data(), tw.size(), dev);
        s.psf_tw_h = dev;
        const float *devf = nullptr;
        UP(float, twf.data(), twf.size(), devf);
        s.psf_tw_f = devf;
      }
      {
        std::vector<float> z(64 * 8, 0.f);
        float *qt;
        UP(float, z.data(), z.size(), qt);
        hipLaunchKernelGGL(k_fill_qf_tab, dim3(1), dim3(128), 0, 0, qt);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { aomarl_destroy(c); return fail("create: quadratic-form constants"); }
        s.qf_tab = qt;
      }
      UP(uint16_t, tmask.data(), tmask.size(), s.tile_mask);
      {
        // tip-tilt planes as the frame kernel reads them (k_frame_wave: tvo): pupil pixel (y, 4 g + j)
        const DevDm &T = s.dms[1];
        std::vector<float> tp((size_t)pd * pd * 2);
        for (int y = 0; y < pd; y++)
          for (int g = 0; g < pd / 4; g++)
            for (int j = 0; j < 4; j++) {
              const size_t o = (size_t)(y + T.toy) * T.dim + T.tox + 4 * g + j;
              float *dst = &tp[((size_t)y * (pd / 4) + g) * 8];
              dst[j] = c->h_tt[2 * o]; dst[4 + j] = c->h_tt[2 * o + 1];
            }
        UP(float, tp.data(), tp.size(), s.tt_pk);
      }
      s.fused_ok = 1; s.ntiles = nt;
      // stack-array DM evaluated from the command lattice inside the frame kernel
      const DevDm &Z = s.dms[0];
      s.otf_ok = 0;
      if (Z.sep && Z.pitch > 0 && 16 % Z.pitch == 0) {
        auto first = [&](int p0, int pmin) {
          const int num = p0 - pmin - (Z.ss - 1);
          return num >= 0 ? (num + Z.pitch - 1) / Z.pitch : -((-num) / Z.pitch);
        };
        s.otf_gx0 = first(Z.tox, Z.i1min); s.otf_gy0 = first(Z.toy, Z.j1min);
        s.otf_xoff = Z.tox - (Z.i1min + Z.pitch * s.otf_gx0);
        s.otf_yoff = Z.toy - (Z.j1min + Z.pitch * s.otf_gy0);
        const int cnt = std::max((s.otf_xoff + 15) / Z.pitch + 1, (s.otf_yoff + 15) / Z.pitch + 1);
        s.otf_nb = (cnt + 3) / 4; s.otf_tpn = 16 / Z.pitch;
        s.otf_latw = (nt - 1) * s.otf_tpn + 4 * s.otf_nb;
        s.otf_ok = (s.otf_nb >= 1 && s.otf_nb <= 2 && s.otf_xoff >= 0 && s.otf_yoff >= 0) ? 1 : 0;
      }
    }
  }
#undef UP
  *out = c;
  return 0;
}

int aomarl_destroy(aomarl_ctx *c) {
  if (c)
    for (auto &sg : c->graphs) { (void)hipGraphExecDestroy(sg.exec); (void)hipGraphDestroy(sg.graph); }
  if (c) c->graphs.clear();
  if (!c) return 0;
  for (void *p : c->owned) (void)hipFree(p);
  for (int k = 0; k < 2; k++) {
    if (c->pipe.snap[k]) (void)hipFree(c->pipe.snap[k]);
    if (c->pipe.ev_done[k]) (void)hipEventDestroy(c->pipe.ev_done[k]);
    if (c->pipe.ev_psf[k]) (void)hipEventDestroy(c->pipe.ev_psf[k]);
  }
  if (c->ev_reset) (void)hipEventDestroy(c->ev_reset);
  for (int k = 0; k < 3; k++) if (c->ev_reset2[k]) (void)hipEventDestroy(c->ev_reset2[k]);
  if (c->pipe.ev_cmd) (void)hipEventDestroy(c->pipe.ev_cmd);
  if (c->pipe.ev_commit) (void)hipEventDestroy(c->pipe.ev_commit);
  if (c->pipe.fstream) { (void)hipStreamSynchronize(c->pipe.fstream); (void)hipStreamDestroy(c->pipe.fstream); }
  // the side streams belong to the process (side_stream): drained here, never destroyed
  if (c->atm_stream) (void)hipStreamSynchronize(c->atm_stream);
  if (c->psf_stream) (void)hipStreamSynchronize(c->psf_stream);
  if (c->ev_frame) (void)hipEventDestroy(c->ev_frame);
  if (c->ev_moved) (void)hipEventDestroy(c->ev_moved);
  if (c->ev_psf) (void)hipEventDestroy(c->ev_psf);
  if (c->env_gain) (void)hipFree(c->env_gain);
  if (c->seed_stage) (void)hipFree(c->seed_stage);
  rp_free(c);
  if (c->timg) (void)hipFree(c->timg);
  for (hipEvent_t e : c->fw_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->fw_ev_retired) (void)hipEventDestroy(e);
  delete c;
  return 0;
}

static int replace_dev(aomarl_ctx *c, float **slot, const std::vector<float> &h) {
  if (*slot) {
    for (size_t i = 0; i < c->owned.size(); i++)
      if (c->owned[i] == *slot) { c->owned.erase(c->owned.begin() + i); break; }
    (void)hipFree(*slot);
    *slot = nullptr;
  }
  return upload<float>(c, h.data(), h.size(), slot);
}

int aomarl_set_cmat(aomarl_ctx *c, const float *cmat) {
  if (c) c->cfg_epoch++;
  if (!c || !cmat) return fail("aomarl_set_cmat: null argument");
  const int na = c->sys.nactu, nsl = c->sys.nslope;
  const int ld = (nsl + 3) & ~3;
  std::vector<float> h((size_t)na * ld, 0.f);
  for (int r = 0; r < na; r++) memcpy(&h[(size_t)r * ld], cmat + (size_t)r * nsl, sizeof(float) * nsl);
  c->ld_cmat = ld;
  c->cmat_scale = gemm_scale(h.data(), h.size());
  return replace_dev(c, &c->cmat, h);
}

int aomarl_set_slopes2modes(aomarl_ctx *c, int nmodes, const float *s2m) {
  if (c) c->cfg_epoch++;
  if (!c) return fail("aomarl_set_slopes2modes: null ctx");
  if (!s2m) { c->s2m_nmodes = 0; return 0; }                 // dropped (cmat or basis changed)
  if (nmodes < 1) return fail("aomarl_set_slopes2modes: nmodes must be positive");
  const int nsl = c->sys.nslope, ld = (nsl + 3) & ~3;
  std::vector<float> h((size_t)nmodes * ld, 0.f);
  for (int r = 0; r < nmodes; r++) memcpy(&h[(size_t)r * ld], s2m + (size_t)r * nsl, sizeof(float) * nsl);
  int rc = replace_dev(c, &c->s2m, h);
  if (rc) return rc;
  c->s2m_scale = gemm_scale(h.data(), h.size());
  c->s2m_nmodes = nmodes;
  return 0;
}

int aomarl_set_gain(aomarl_ctx *c, float gain) {
  if (c) c->cfg_epoch++;
  if (!c) return fail("null ctx");
  c->gain = gain;
  return 0;
}

int aomarl_set_env_gains(aomarl_ctx *c, const float *gains, int nenv) {
  if (c) c->cfg_epoch++;
  if (!c) return fail("null ctx");
  if (!gains) {                      // back to the scalar gain
    if (c->env_gain) { HIPCHK(hipDeviceSynchronize()); (void)hipFree(c->env_gain); }
    c->env_gain = nullptr; c->env_gain_n = 0;
    return 0;
  }
  if (nenv < 1) return fail("set_env_gains: nenv must be positive");
  if (c->env_gain_n != nenv) {
    if (c->env_gain) { HIPCHK(hipDeviceSynchronize()); (void)hipFree(c->env_gain); c->env_gain = nullptr; }
    HIPCHK(hipMalloc((void **)&c->env_gain, sizeof(float) * (size_t)nenv));
    c->env_gain_n = nenv;
  }
  HIPCHK(hipMemcpy(c->env_gain, gains, sizeof(float) * (size_t)nenv, hipMemcpyHostToDevice));
  return 0;
}

int aomarl_set_modal(aomarl_ctx *c, int nmodes, const float *v2m, const float *m2v,
                     const float *freedom, int nact, const int32_t *amodes) {
  if (c) c->cfg_epoch++;
  if (!c || !v2m || !m2v) return fail("aomarl_set_modal: null argument");
  const int na = c->sys.nactu;
  if (nmodes <= 0 || nmodes > na) return fail("nmodes out of range");
  c->nmodes = nmodes;
  c->ld_v2m = (na + 3) & ~3;
  c->ld_m2v = (nmodes + 3) & ~3;
  std::vector<float> a((size_t)nmodes * c->ld_v2m, 0.f), b((size_t)na * c->ld_m2v, 0.f);
  for (int r = 0; r < nmodes; r++) memcpy(&a[(size_t)r * c->ld_v2m], v2m + (size_t)r * na, sizeof(float) * na);
  for (int r = 0; r < na; r++) memcpy(&b[(size_t)r * c->ld_m2v], m2v + (size_t)r * nmodes, sizeof(float) * nmodes);
  int rc = replace_dev(c, &c->v2m, a);
  if (rc) return rc;
  rc = replace_dev(c, &c->m2v, b);
  if (rc) return rc;
  c->v2m_scale = gemm_scale(a.data(), a.size());
  c->m2v_scale = gemm_scale(b.data(), b.size());
  std::vector<float> f(nmodes, 0.f);
  if (freedom) memcpy(f.data(), freedom, sizeof(float) * nmodes);
  rc = replace_dev(c, &c->freedom, f);
  if (rc) return rc;
  c->nact = 0;
  if (nact > 0) {
    if (!amodes) return fail("action_modes is null");
    for (int j = 0; j < nact; j++)
      if (amodes[j] < 0 || amodes[j] >= nmodes) return fail("action mode %d out of range", amodes[j]);
    if (c->amodes) {
      for (size_t i = 0; i < c->owned.size(); i++)
        if (c->owned[i] == c->amodes) { c->owned.erase(c->owned.begin() + i); break; }
      (void)hipFree(c->amodes);
      c->amodes = nullptr;
    }
    rc = upload<int32_t>(c, amodes, nact, &c->amodes);
    if (rc) return rc;
    if (c->amode_inv) {
      for (size_t i = 0; i < c->owned.size(); i++)
        if (c->owned[i] == c->amode_inv) { c->owned.erase(c->owned.begin() + i); break; }
      (void)hipFree(c->amode_inv);
      c->amode_inv = nullptr;
    }
    std::vector<int32_t> inv(nmodes, -1);
    for (int j = 0; j < nact; j++) {
      if (inv[amodes[j]] != -1) return fail("action mode %d listed twice", amodes[j]);
      inv[amodes[j]] = j;
    }
    rc = upload<int32_t>(c, inv.data(), inv.size(), &c->amode_inv);
    if (rc) return rc;
    c->nact = nact;
  }
  return 0;
}

// ---- workspace layout (floats)
struct Work {
  size_t Z, Z2, NEWL, ZREF, ZREF2, MODES, TR, TPART, PEND, GEMM, GEMM_ATM, gemm_floats, total;
  int ldz, ldn, ldm, nblk;
};

static Work work_layout(const aomarl_ctx *c, int nenv) {
  Work w;
  const DevSys &s = c->sys;
  const size_t ncol = (size_t)nenv * (c->nlayers > 0 ? c->nlayers : 1);
  w.ldz = (c->maxK + 3) & ~3;
  w.ldn = (c->maxdim + 3) & ~3;
  w.ldm = (s.nactu + 3) & ~3;
  const int W = 2 * s.hw, RB = 256 / W;
  w.nblk = (s.pupdiam + RB - 1) / RB;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~(size_t)3; return r; };
  w.Z = take(ncol * w.ldz);
  w.NEWL = take(ncol * w.ldn);
  w.ZREF = take(ncol);
  w.Z2 = take(ncol * w.ldz);       // the fused scatter + gather writes the NEXT round's operand while this round's is in use
  w.ZREF2 = take(ncol);
  w.MODES = take((size_t)nenv * w.ldm);
  w.TR = take((size_t)nenv * s.pupdiam * W * 2);
  w.TPART = take((size_t)nenv * w.nblk * 4);
  w.PEND = take((size_t)nenv * (W * W + 4));
  {
    size_t mn = std::max(ncol * (size_t)w.ldn, (size_t)nenv * (size_t)w.ldm);
    w.gemm_floats = 8 * mn;            // up to 8 partial tiles of the split-K GEMM
    w.GEMM = take(w.gemm_floats);
    w.GEMM_ATM = take(w.gemm_floats);  // the extrusion's own split-K workspace: it may run on the side stream
  }
  w.total = o;
  return w;
}

size_t aomarl_workspace_floats(const aomarl_ctx *c, int nenv) {
  return c ? work_layout(c, nenv).total : 0;
}
size_t aomarl_screen_stride(const aomarl_ctx *c) { return c ? (size_t)c->sys.screen_stride : 0; }
size_t aomarl_dmshape_stride(const aomarl_ctx *c) { return c ? (size_t)c->sys.shape_stride : 0; }

static int check_range(const aomarl_ctx *c, const aomarl_state *st, int b, int n) {
  if (!c || !st) return fail("null ctx/state");
  if (c->pipe.active && !c->pipe_internal && st->screens == c->pipe.owner_screens)
    return fail("a pipelined frame is in flight on this state (aomarl_set_frame_pipeline): only aomarl_env_step and a "
                "full-range aomarl_reset are accepted until the reset -- slopes / voltage of odd frames are in the twin, "
                "the screens a frame ahead");
  if (b < 0 || n < 0 || b + n > st->nenv) return fail("env range [%d, %d) outside [0, %d)", b, b + n, st->nenv);
  if (st->ld_actu < c->sys.nactu) return fail("ld_actu (%d) < nactu (%d)", st->ld_actu, c->sys.nactu);
  if (!st->screens || !st->origin || !st->seeds || !st->ext_count || !st->com || !st->com1 ||
      !st->com2 || !st->err || !st->voltage || !st->slopes || !st->dm_shape || !st->strehl ||
      !st->le_img || !st->frame || !st->work)
    return fail("aomarl_state has a null mandatory buffer");
  return 0;
}

static DevState dev_state(const aomarl_state *st) {
  DevState d;
  d.nenv = st->nenv; d.ld_actu = st->ld_actu;
  d.screens = st->screens; d.origin = st->origin; d.seeds = st->seeds; d.ext_count = st->ext_count;
  d.com = st->com; d.com1 = st->com1; d.com2 = st->com2; d.err = st->err; d.voltage = st->voltage;
  d.slopes = st->slopes; d.dm_shape = st->dm_shape; d.bincube = st->bincube;
  d.wfs_phase = st->wfs_phase; d.tar_phase = st->tar_phase; d.strehl = st->strehl;
  d.le_img = st->le_img; d.frame = st->frame; d.work = st->work;
  d.origin_snap = nullptr;
  return d;
}

#include "aomarl_capi_atmos.hip"
#include "aomarl_capi_stages.hip"
#include "aomarl_capi_agents.hip"
#include "aomarl_capi_step.hip"
#include "aomarl_capi_extras.hip"
#include "aomarl_capi_composites.hip"
