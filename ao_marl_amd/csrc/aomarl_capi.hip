// aomarl_capi.hip -- C ABI (include/aomarl.h) over the gfx950 kernels in aomarl_kernels.hip.
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

static int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

#define HIPCHK(x)                                                                            \
  do {                                                                                       \
    hipError_t _e = (x);                                                                     \
    if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e),    \
                                      __FILE__, __LINE__);                                   \
  } while (0)
#define LAUNCHCHK()                                                                          \
  do {                                                                                       \
    hipError_t _e = hipGetLastError();                                                       \
    if (_e != hipSuccess) return fail("kernel launch failed: %s (%s:%d)",                    \
                                      hipGetErrorString(_e), __FILE__, __LINE__);            \
  } while (0)

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
  bool skip_do_control = false;        // (aomarl_env_step's small chain: aomarl_next_part_one leaves do_control to its tail kernel)
  int small_move = 1;                  // "small_move": 1 = one k_move_small launch per frame's move where the screens allow it
  bool small_ok = false;               // every layer has dim <= MOVE_SMALL_DIM, ns + dim <= MOVE_SMALL_K (transposed [A|B] uploaded)
  int reset_streams = 2;               // "reset_streams": a batch reset in that many parts side by side, one stream each (1..4)
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
        std::vector<int32_t> linfo((size_t)nt * (nt + 4), 0), lcount(nt, 0);
        for (int r = 0; r < nt; r++) {
          int k = 0;
          for (int t = 0; t < nt; t++)
            if (tinfo[(size_t)r * nt + t] & 0x10000) linfo[(size_t)r * (nt + 4) + k++] = tinfo[(size_t)r * nt + t] | (t << 24);
          lcount[r] = k;
          for (int kk = k; kk < nt + 4; kk++) linfo[(size_t)r * (nt + 4) + kk] = k ? linfo[(size_t)r * (nt + 4) + k - 1] : 0;
        }
        UP(int32_t, linfo.data(), linfo.size(), s.lit_info);
        UP(int32_t, lcount.data(), lcount.size(), s.lit_count);
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
  size_t Z, NEWL, ZREF, MODES, TR, TPART, PEND, GEMM, GEMM_ATM, gemm_floats, total;
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

static bool same_round(const RoundOps &a, const RoundOps &b) {
  if (a.nops != b.nops) return false;
  for (int i = 0; i < a.nops; i++)
    if (a.layer[i] != b.layer[i] || a.dir[i] != b.dir[i] || a.tflag[i] != b.tflag[i]) return false;
  return true;
}

// A sequence of extrusion rounds (round = at most one operation per layer).  Per round: stencil gather +
// normals -> Z, Z . [A|B]^T (split-K tiles), new line -> ring.  Two consecutive rounds with the same
// operations share a launch for the scatter of the first and the gather of the second (k_extrude_sg):
// 2 launches per round instead of 3 -- every round of a reset (1296 of them), most rounds of a frame.
struct ExtrudeRun {          // one range of environments walking through a sequence of rounds on one stream
  aomarl_ctx *c; aomarl_state *st; int b, n; hipStream_t s; bool ordered;
  Work w; DevState ds; float *Z, *NEWL, *ZREF, *WS; size_t ws_floats; bool gathered;
  // ordered = false: the caller has ordered the stream behind every reader of the screens (reset)
  ExtrudeRun(aomarl_ctx *c_, aomarl_state *st_, int b_, int n_, void *stream, bool ordered_ = true)
      : c(c_), st(st_), b(b_), n(n_), s((hipStream_t)stream), ordered(ordered_), gathered(false) {
    w = work_layout(c, st->nenv);
    ds = dev_state(st);
    if (ordered) ds.origin_snap = c->snap_target;
    // the range's own part of every work area (columns are numbered from the range's first environment):
    // two ranges may run side by side on two streams
    const size_t col0 = (size_t)b * (c->nlayers > 0 ? c->nlayers : 1), ncols = (size_t)n * (c->nlayers > 0 ? c->nlayers : 1);
    Z = st->work + w.Z + col0 * w.ldz; NEWL = st->work + w.NEWL + col0 * w.ldn; ZREF = st->work + w.ZREF + col0;
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
      const bool fuse_next = single && !c->no_extrude_sg && r + 1 < nrounds && same_round(rounds[r], rounds[r + 1]);
      if (!(gathered && single)) {
        hipLaunchKernelGGL(k_extrude_gather, dim3(ncol, (nsc + (dimc + 3) / 4 + 255) / 256), dim3(256), 0, s, c->sys, ds, b,
                           ops, Z, w.ldz, ZREF);
        LAUNCHCHK();
      }
      int nsp = 0;
      float pscale = 1.f;
      launch_gemm_nt(ncol, dimc, K, 1.0f, Z, w.ldz, c->sys.layers[ref].AB, c->sys.layers[ref].ldab,
                     0.0f, NEWL, w.ldn, s, WS, ws_floats, nullptr, &nsp,
                     /* split-f16: stencil values (um) and N(0,1) draws x 2^8 */ true, 256.f, c->ab_scale[cls], &pscale);
      LAUNCHCHK();
      if (ordered) {
        int wrc = first_write_wait(c, s);
        if (wrc) return wrc;
        if (s != c->atm_stream) c->screens_dirty_main = true;
      }
      if (fuse_next) {
        hipLaunchKernelGGL(k_extrude_sg, dim3(ncol), dim3(512), 0, s, c->sys, ds, b, ops, NEWL, w.ldn, ZREF,
                           WS, nsp, ncol, dimc, pscale, Z, w.ldz);
        gathered = true;
      } else {
        hipLaunchKernelGGL(k_extrude_scatter, dim3(ncol), dim3(256), 0, s, c->sys, ds, b, ops, NEWL, w.ldn,
                           ZREF, WS, nsp, ncol, dimc, pscale);
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
    hipLaunchKernelGGL(k_move_small, dim3(n, c->nlayers), dim3(MOVE_SMALL_T), 0, s, c->sys, dsm, b, mp);
    LAUNCHCHK();
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
  rc = move_atmos_now(c, st, b, n, accumx, accumy, (void *)c->atm_stream);
  if (rc) return rc;
  if (c->frame_wait_pending) {            // nothing was extruded this frame: still order the marker behind the readers
    HIPCHK(hipStreamWaitEvent(c->atm_stream, c->ev_frame_cur, 0));
    c->frame_wait_pending = false;
  }
  c->screens_dirty_main = false;
  HIPCHK(hipEventRecord(c->ev_moved, c->atm_stream));
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
  hipLaunchKernelGGL(k_reset_env, dim3(n), dim3(256), 0, s, c->sys, ds, b, n, stage, st->ld_actu);
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
// copies them in (1.3 GB device to device: < 1 ms) and does the rest of the reset.  Same kernels, same partition
// of the batch, same columns, same split-K order as aomarl_reset: the same screens, bit for bit.
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
  int e0 = b;
  for (int k = 0; k < parts; k++) {         // the plain reset's partition, all parts on the one stream
    const int nk = (b + n - e0) / (parts - k);
    rp->runs.emplace_back(c, &rp->shadow, e0, nk, stream, false);
    e0 += nk;
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
  if (!strcmp(name, "gemm_balanced")) { g_gemm_p = value != 0; return 0; }    // process-wide
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
  if (!strcmp(name, "reset_streams")) { c->reset_streams = value < 1 ? 1 : (value > 4 ? 4 : value); return 0; }
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
    float s = 0.f;
    for (int z = 0; z < sb.nsplit; z++) s += sb.part[((long long)z * nenv + e) * sb.pn + col];
    return sb.alpha * s;
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
// y[o] = sum_k x[k] W[o][k] for o < no, x in LDS, by a 256-thread block: FOUR threads per output (they read 16
// consecutive bytes of the row per step, two accumulators each, then two xor-shuffles), 64 outputs per pass.
// done(o, y) runs in the first thread of each quad.
template <class F>
__device__ __forceinline__ void small_gemv(const float *__restrict__ W, int ldw, int no, int K, const float *xs, F done) {
  const int tid = threadIdx.x, q = tid & 3;
  for (int o0 = 0; o0 < no; o0 += 64) {
    const int o = o0 + (tid >> 2);
    float a0 = 0.f, a1 = 0.f;
    if (o < no) {
      const float *row = W + (long long)o * ldw;
      int k = q;
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
    hipLaunchKernelGGL(k_small_head, dim3(n), dim3(256), 0, s, c->sys, dsv, p);
    LAUNCHCHK();
    // ONE kernel wrote the voltages and committed the pending window, behind the wait for that parity's PSF finish:
    // the release of the frame stream covers all three (no separate commit event, no wait of its own on the frame stream)
    if (ahead) { HIPCHK(hipEventRecord(c->pipe.ev_cmd, s)); c->pipe.cmd_covers_commit = true; }
    return 0;
  }
  float *modes = st->work + w.MODES;
  const int cx = (nm + 255) / 256;
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
    hipLaunchKernelGGL(k_delay_ahead, dim3((na + 255) / 256, 2 * n), dim3(256), 0, s, c->sys, dsv, na, st->ld_actu, n,
                       nsp > 0 ? st->work + w.GEMM : nullptr, nsp, alpha, ktt);
    LAUNCHCHK();
    HIPCHK(hipEventRecord(c->pipe.ev_cmd, s));
  } else {
    if (nsp > 0)
      hipLaunchKernelGGL(k_delay_sum, dim3((na + 255) / 256, n), dim3(256), 0, s, dsv, na, st->ld_actu, wa, wb, wc, 0, 1,
                         st->work + w.GEMM, nsp, alpha, n);
    else
      hipLaunchKernelGGL(k_delay, dim3((na + 255) / 256, n), dim3(256), 0, s, dsv, na, st->ld_actu, wa, wb, wc, 0, 1);
    LAUNCHCHK();
  }
  if (psf_ev) HIPCHK(hipStreamWaitEvent(s, psf_ev, 0));
  else if (stv == st) { int rc = psf_wait_pending(c, stream); if (rc) return rc; }
  hipLaunchKernelGGL(k_post_delay, dim3(ahead ? n : 2 * n), dim3(256), 0, s, c->sys, dsv, 0, n, stv->work + w.PEND, 1, ktt,
                     stv->voltage, st->ld_actu);
  LAUNCHCHK();
  if (ahead) { HIPCHK(hipEventRecord(c->pipe.ev_commit, s)); c->pipe.cmd_covers_commit = false; }   // the PSF finish of the frame about to be launched overwrites that window
  return 0;
}

// ---- the rest of AoEnv.linear_step behind do_control: v2m . err, the state blocks
static int env_step_tail(aomarl_ctx *c, aomarl_state *st, aomarl_env_glue *g, bool fused, float *state_out, void *stream,
                         const aomarl_state *slopes_view = nullptr) {
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
  if (fused) {
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
    c->skip_do_control = small;                  // the small chain's tail kernel does it
    rc = aomarl_next_part_one(c, st, 0, n, accumx, accumy, 0, stream);
    c->skip_do_control = false;
    if (rc) return rc;
    if (small) return env_step_tail(c, st, g, fused, state_out, stream, st);
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
  int rc = move_atmos_now(c, st, 0, st->nenv, accumx, accumy, (void *)c->atm_stream);
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
  HIPCHK(hipEventRecord(c->ev_moved, c->atm_stream));
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
  // the frame stream is released behind k_delay_ahead, in front of the Strehl commit that waits for that finish:
  // the frame kernel overwrites the PSF rows it reads, so the frame stream waits for it itself
  if (!rc && pe && !P.cmd_covers_commit && hipStreamWaitEvent(P.fstream, pe, 0) != hipSuccess) rc = fail("frame pipeline: hipStreamWaitEvent failed");
  if (!rc) { P.psf_out[q] = false; rc = pipe_launch_frame(c, st, q, stream); }
  if (!rc) rc = pipe_prefetch(c, st, accumx, accumy, p, q);
  // ---- reduce frame p
  if (!rc && hipStreamWaitEvent(s, P.ev_done_cur[p], 0) != hipSuccess) rc = fail("frame pipeline: hipStreamWaitEvent failed");
  if (!rc && small_chain_ok(c, g)) rc = env_step_tail(c, st, g, true, state_out, stream, &vp);
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
  key.push_back(pf ? 1 : 0); key.push_back(c->small_move); key.push_back(c->small_chain);
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

// ---------------------------------------------------------------- WFS-image denoiser (A17)
#include "aomarl_denoise.hip"

// ---------------------------------------------------------------- geometric controller
__global__ void k_geo_assemble(int nactu, int npzt, int ldr, int gwgh, const int32_t *__restrict__ map,
                               const float *__restrict__ lat, const float *__restrict__ r3,
                               float *__restrict__ r) {
  const int e = blockIdx.y, a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= ldr) return;
  float v = 0.f;
  if (a < npzt) v = lat[(long long)e * gwgh + map[a]];
  else if (a <= nactu) v = r3[(long long)e * 4 + (a - npzt)];   // TT0, TT1, sum
  r[(long long)e * ldr + a] = v;
}

int aomarl_set_geo(aomarl_ctx *c, const float *W) {
  if (!c || !W) return fail("aomarl_set_geo: null argument");
  const DevSys &s = c->sys;
  if (!(s.ndm == 2 && s.dms[0].type == AOMARL_DM_PZT && s.dms[0].sep && s.dms[1].type == AOMARL_DM_TT &&
        s.tar_all_int))
    return fail("aomarl_set_geo: needs DMs = [separable stack array, tip-tilt] at integer target offsets");
  const DevDm &Z = s.dms[0], &T = s.dms[1];
  const int pd = s.pupdiam, na = s.nactu, npzt = Z.nact;
  const int gw = Z.gw, gh = Z.gh;
  // profile matrices over the pupil window: UxT[j][x] = u(x + tox - X_j), UyT[i][y] likewise
  std::vector<float> ux((size_t)gw * pd, 0.f), uy((size_t)gh * pd, 0.f);
  for (int j = 0; j < gw; j++)
    for (int x = 0; x < pd; x++) {
      const int a = x + Z.tox - (Z.i1min + Z.pitch * j);
      if (a >= 0 && a < Z.ss) ux[(size_t)j * pd + x] = c->h_prof[a];
    }
  for (int i = 0; i < gh; i++)
    for (int y = 0; y < pd; y++) {
      const int a = y + Z.toy - (Z.j1min + Z.pitch * i);
      if (a >= 0 && a < Z.ss) uy[(size_t)i * pd + y] = c->h_prof[a];
    }
  // planes: the two tip-tilt influence maps and the constant, over the pupil window (the phase
  // handed in is already masked, so the planes need no mask)
  std::vector<float> planes((size_t)3 * pd * pd);
  for (int y = 0; y < pd; y++)
    for (int x = 0; x < pd; x++) {
      const size_t o = (size_t)(y + T.toy) * T.dim + x + T.tox, p = (size_t)y * pd + x;
      planes[p] = c->h_tt[2 * o]; planes[(size_t)pd * pd + p] = c->h_tt[2 * o + 1];
      planes[(size_t)2 * pd * pd + p] = 1.0f;
    }
  std::vector<int32_t> map(npzt, -1);
  for (int i = 0; i < gh; i++)
    for (int j = 0; j < gw; j++) {
      const int a = c->h_grid[(size_t)i * gw + j];
      if (a >= 0) { if (a >= npzt) return fail("aomarl_set_geo: lattice table out of range"); map[a] = j * gh + i; }
    }
  for (int a = 0; a < npzt; a++) if (map[a] < 0) return fail("aomarl_set_geo: actuator %d is not on the lattice", a);
  c->geo_ldw = (na + 1 + 3) & ~3;
  std::vector<float> w((size_t)na * c->geo_ldw, 0.f);
  for (int r = 0; r < na; r++) memcpy(&w[(size_t)r * c->geo_ldw], W + (size_t)r * (na + 1), sizeof(float) * (na + 1));
  int rc = replace_dev(c, &c->geoW, w);
  if (!rc) rc = replace_dev(c, &c->geoUx, ux);
  if (!rc) rc = replace_dev(c, &c->geoUy, uy);
  if (!rc) rc = replace_dev(c, &c->geoPlanes, planes);
  if (rc) return rc;
  if (c->geoMap) {
    for (size_t i = 0; i < c->owned.size(); i++)
      if (c->owned[i] == c->geoMap) { c->owned.erase(c->owned.begin() + i); break; }
    (void)hipFree(c->geoMap);
    c->geoMap = nullptr;
  }
  rc = upload<int32_t>(c, map.data(), map.size(), &c->geoMap);
  if (rc) return rc;
  c->geo_gw = gw; c->geo_gh = gh; c->geo_npzt = npzt; c->geo_ldr = c->geo_ldw;
  return 0;
}

// floats of scratch aomarl_geo_control needs: row products [n][gw][pd], lattice products
// [n][gw][gh], plane products [n][4], right-hand sides [n][ldr], split-K workspace
struct GeoWork { size_t T, LAT, R3, R, GEMM, gemm_floats, total; };
static GeoWork geo_layout(aomarl_ctx *c, int n) {
  GeoWork g; size_t o = 0;
  auto take = [&](size_t k) { size_t at = o; o += (k + 63) & ~(size_t)63; return at; };
  g.T = take((size_t)n * c->geo_gw * c->sys.pupdiam);
  g.LAT = take((size_t)n * c->geo_gw * c->geo_gh);
  g.R3 = take((size_t)n * 4);
  g.R = take((size_t)n * c->geo_ldr);
  g.gemm_floats = (size_t)8 * n * std::max(c->sys.nactu, 4) + 4096;
  g.GEMM = take(g.gemm_floats);
  g.total = o;
  return g;
}

size_t aomarl_geo_workspace_floats(aomarl_ctx *c, int nenv) {
  if (!c || !c->geoW || nenv <= 0) return 0;
  return geo_layout(c, nenv).total;
}

int aomarl_geo_control(aomarl_ctx *c, aomarl_state *st, int b, int n, float *work, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (!c->geoW) return fail("geo_control: no projector (aomarl_set_geo)");
  if (!st->tar_phase) return fail("geo_control needs st->tar_phase (masked atmosphere phase of the target)");
  if (!work) return fail("geo_control: null workspace");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int pd = c->sys.pupdiam, gw = c->geo_gw, gh = c->geo_gh, na = c->sys.nactu;
  GeoWork g = geo_layout(c, n);
  const float *phi = st->tar_phase + (size_t)b * pd * pd;
  float *T = work + g.T, *LAT = work + g.LAT, *R3 = work + g.R3, *R = work + g.R;
  // T[e][j][y] = sum_x UxT[j][x] phi[e][y][x]
  hipLaunchKernelGGL(k_gemm_nt_batched2, dim3((pd + 63) / 64, (gw + 63) / 64, n), dim3(256), 0, s, gw, pd, pd,
                     c->geoUx, pd, (long long)0, phi, pd, (long long)pd * pd, (const float *)nullptr,
                     (long long)0, T, pd, (long long)gw * pd, 0);
  LAUNCHCHK();
  // LAT[e][j][i] = sum_y T[e][j][y] UyT[i][y]
  hipLaunchKernelGGL(k_gemm_nt_batched2, dim3((gh + 63) / 64, (gw + 63) / 64, n), dim3(256), 0, s, gw, gh, pd,
                     T, pd, (long long)gw * pd, c->geoUy, pd, (long long)0, (const float *)nullptr,
                     (long long)0, LAT, gh, (long long)gw * gh, 0);
  LAUNCHCHK();
  // R3[e][k] = sum_p phi[e][p] planes[k][p]   (TT0, TT1, 1)
  launch_gemm_nt(n, 3, pd * pd, 1.0f, phi, pd * pd, c->geoPlanes, pd * pd, 0.0f, R3, 4, s, work + g.GEMM,
                 g.gemm_floats);
  LAUNCHCHK();
  hipLaunchKernelGGL(k_geo_assemble, dim3((c->geo_ldr + 255) / 256, n), dim3(256), 0, s, na, c->geo_npzt,
                     c->geo_ldr, gw * gh, c->geoMap, LAT, R3, R);
  LAUNCHCHK();
  // com[e] = W . r[e]
  launch_gemm_nt(n, na, na + 1, 1.0f, R, c->geo_ldr, c->geoW, c->geo_ldw, 0.0f,
                 st->com + (size_t)b * st->ld_actu, st->ld_actu, s, work + g.GEMM, g.gemm_floats);
  LAUNCHCHK();
  return 0;
}

int aomarl_target_psf_buffer(aomarl_ctx *c, aomarl_state *st, int b, int n, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (!st->tar_phase) return fail("target_psf_buffer needs st->tar_phase");
  if (n == 0) return 0;
  return target_psf_impl(c, st, b, n, true, stream);
}

// ---------------------------------------------------------------- full-frame PSF (on demand)
// Target.get_tar_image(expo_type = "se") (targetCompass.py:71-92): the whole npsf x npsf short-exposure PSF,
// |FFT2(pupil . exp(2 pi i phase / lambda))|^2, centred (what fftshift returns).  The hot path only ever forms its
// central 16 x 16 window; three of the environment's reward branches read the full frame (ao_env.py:621-623,
// 654-656).  Two DFT passes as products on the library's fp32 GEMM: rows (pupdiam samples -> npsf frequencies), then
// columns; 28 GFLOP per environment at 40x40 -- an on-demand diagnostic, one environment at a time.
__global__ void k_timg_tables(float *__restrict__ W1, float *__restrict__ Wc, float *__restrict__ Ws, int pd, int npsf) {
  // W1 [2 npsf][2 pd]: row (re, k) = [cos | sin], row (im, k) = [-sin | cos] of theta = 2 pi (k - npsf/2) x / npsf
  // Wc, Ws [npsf][pd]: cos / sin of the same angle (second pass)
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)npsf * pd) return;
  const int k = (int)(i / pd), x = (int)(i - (long long)k * pd);
  const long long f = (((long long)(k - npsf / 2) * x) % npsf + npsf) % npsf;
  float sn, cs;
  sincospif(2.0f * (float)f / (float)npsf, &sn, &cs);
  Wc[i] = cs; Ws[i] = sn;
  float *re = W1 + (long long)k * 2 * pd, *im = W1 + (long long)(npsf + k) * 2 * pd;
  re[x] = cs; re[pd + x] = sn;
  im[x] = -sn; im[pd + x] = cs;
}
__global__ void k_timg_amp(const float *__restrict__ phase, const float *__restrict__ pupil, float inv_lambda,
                           float *__restrict__ amp, int pd) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pd * pd) return;
  const int y = p / pd, x = p - y * pd;
  const float m = pupil[p];
  float a = phase[p] * inv_lambda;
  a -= rintf(a);
  amp[(long long)y * 2 * pd + x] = m != 0.f ? m * __builtin_amdgcn_cosf(a) : 0.f;
  amp[(long long)y * 2 * pd + pd + x] = m != 0.f ? m * __builtin_amdgcn_sinf(a) : 0.f;
}
__global__ void k_timg_abs2(const float *__restrict__ yr, const float *__restrict__ yi, float *__restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = yr[i] * yr[i] + yi[i] * yi[i];
}

int aomarl_target_image(aomarl_ctx *c, aomarl_state *st, int b, int n, float *out, void *stream) {
  int rc = check_range(c, st, b, n);
  if (rc) return rc;
  if (!out) return fail("target_image: null output");
  if (n == 0) return 0;
  if (c->premoved && c->pre_screens == st->screens)
    return fail("target_image: the screens have already been moved to the next frame (aomarl_prefetch_atmos / "
                "\"prefetch_atmos\"): the image of THIS frame cannot be formed any more -- run with the prefetch off");
  rc = atmos_wait_pending(c, stream);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int pd = c->sys.pupdiam, np = c->sys.npsf;
  const size_t o_w1 = 0, o_wc = o_w1 + (size_t)4 * np * pd, o_ws = o_wc + (size_t)np * pd, o_ph = o_ws + (size_t)np * pd,
               o_amp = o_ph + (size_t)pd * pd, o_x = o_amp + (size_t)2 * pd * pd, o_yr = o_x + (size_t)2 * np * pd,
               o_yi = o_yr + (size_t)np * np, total = o_yi + (size_t)np * np;
  if (!c->timg) {
    HIPCHK(hipMalloc((void **)&c->timg, sizeof(float) * total));
    hipLaunchKernelGGL(k_timg_tables, dim3((unsigned)(((long long)np * pd + 255) / 256)), dim3(256), 0, s, c->timg + o_w1,
                       c->timg + o_wc, c->timg + o_ws, pd, np);
    LAUNCHCHK();
  }
  float *W1 = c->timg + o_w1, *Wc = c->timg + o_wc, *Ws = c->timg + o_ws, *ph = c->timg + o_ph, *amp = c->timg + o_amp;
  float *X = c->timg + o_x, *Yr = c->timg + o_yr, *Yi = c->timg + o_yi;
  if (c->defer_dm_shape) {                      // the stack-array shapes exist only as voltages: form them
    rc = dm_shape_impl(c, st, b, n, nullptr, false, stream);
    if (rc) return rc;
  }
  for (int e = b; e < b + n; e++) {
    DevState ds = dev_state(st);
    ds.tar_phase = ph - (long long)e * pd * pd;           // environment e of the kernel lands in the scratch
    hipLaunchKernelGGL(k_raytrace<true>, dim3((pd * pd + 255) / 256, 1), dim3(256), 0, s, traced_sys(c), ds, e,
                       AOMARL_TRACE_RESET | AOMARL_TRACE_ATMOS | AOMARL_TRACE_DMS);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_timg_amp, dim3((pd * pd + 255) / 256), dim3(256), 0, s, ph, c->sys.spupil, c->sys.tar_inv_lambda, amp, pd);
    LAUNCHCHK();
    // pass 1: X[(re | im, kx)][y] = W1 . amp^T
    launch_gemm_nt(2 * np, pd, 2 * pd, 1.0f, W1, 2 * pd, amp, 2 * pd, 0.0f, X, pd, s);
    // pass 2: Y[ky][kx]:  Yr = Wc Xr^T + Ws Xi^T,  Yi = Wc Xi^T - Ws Xr^T
    const float *Xr = X, *Xi = X + (size_t)np * pd;
    launch_gemm_nt(np, np, pd, 1.0f, Wc, pd, Xr, pd, 0.0f, Yr, np, s);
    launch_gemm_nt(np, np, pd, 1.0f, Ws, pd, Xi, pd, 1.0f, Yr, np, s);
    launch_gemm_nt(np, np, pd, 1.0f, Wc, pd, Xi, pd, 0.0f, Yi, np, s);
    launch_gemm_nt(np, np, pd, -1.0f, Ws, pd, Xr, pd, 1.0f, Yi, np, s);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_timg_abs2, dim3((unsigned)(((long long)np * np + 255) / 256)), dim3(256), 0, s, Yr, Yi,
                       out + (size_t)(e - b) * np * np, (long long)np * np);
    LAUNCHCHK();
  }
  return 0;
}

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
  const size_t smm = sizeof(float) * (2 * 128 + (otf ? 4 * 4 * nb * c->sys.otf_latw : 0)) + 8192 + 128;
  dim3 grid((n + 3) / 4, c->sys.ntiles), blk(256);
// the events ride on the dispatch itself (its start / completion signal): no marker packets of their own
// on the queue in front of and behind the kernel
#define FW(NL, NB, OTF, NZ, WC, HP) hipExtLaunchKernelGGL((k_frame_wave<NL, NB, OTF, NZ, WC, HP>), grid, blk, smm, s, ev_start, ev_done, 0, c->sys, ds, b, n, cog, TR, TP, w.nblk)
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
    hipLaunchKernelGGL(k_target_finish_mfma, dim3(n), dim3(256), 0, c->psf_stream, c->sys, TR, TP, w.nblk, PEND, st->frame + b);
    LAUNCHCHK();
    HIPCHK(hipEventRecord(c->pipe.ev_psf[slot], c->psf_stream));
    c->pipe.psf_out[slot] = true;
    return 0;
  }
  if (c->prefetch_atmos) {
    // second axis of the PSF window: off the critical path (read by aomarl_comp_strehl at the end of
    // the step), so it goes to the side stream, in front of the next frame's extrusions
    c->ev_frame_cur = ev_done; c->frame_marked = true;
    c->side_joined = false;
    HIPCHK(hipStreamWaitEvent(c->psf_stream, ev_done, 0));
    hipLaunchKernelGGL(k_target_finish_mfma, dim3(n), dim3(256), 0, c->psf_stream, c->sys, TR, TP, w.nblk, PEND, st->frame + b);
    LAUNCHCHK();
    HIPCHK(hipEventRecord(c->ev_psf, c->psf_stream));
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

int aomarl_gemm_nt_batched(int batch, int M, int N, int K, const float *A, int lda, long long strideA,
                           const float *B, int ldb, long long strideB, const float *bias,
                           long long strideBias, float *C, int ldc, long long strideC, int relu,
                           void *stream) {
  if (!A || !B || !C) return fail("gemm_nt_batched: null pointer");
  if (batch < 0 || M < 0 || N < 0 || K < 0 || lda < K || ldb < K || ldc < N) return fail("gemm_nt_batched: bad sizes");
  if (batch == 0 || M == 0 || N == 0) return 0;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return fail("gemm_nt_batched: A and B must be 16-byte aligned");
  const bool al = (lda % 4 == 0) && (ldb % 4 == 0) && (strideA % 4 == 0) && (strideB % 4 == 0);
  if (al)
    hipLaunchKernelGGL(k_gemm_nt_batched2, dim3((N + 63) / 64, (M + 63) / 64, batch), dim3(256), 0,
                       (hipStream_t)stream, M, N, K, A, lda, strideA, B, ldb, strideB, bias, strideBias,
                       C, ldc, strideC, relu);
  else
    hipLaunchKernelGGL(k_gemm_nt_batched, dim3((N + 63) / 64, (M + 63) / 64, batch), dim3(256), 0,
                       (hipStream_t)stream, M, N, K, A, lda, strideA, B, ldb, strideB, bias, strideBias,
                       C, ldc, strideC, relu);
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

static int gemm_batched_launch(int batch, int transA, int transB, int M, int N, int K, const float *A, int lda,
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

int aomarl_gemm_batched(int batch, int transA, int transB, int M, int N, int K, const float *A, int lda,
                        long long strideA, const float *B, int ldb, long long strideB, const float *bias,
                        long long strideBias, float *C, int ldc, long long strideC, int relu,
                        int accumulate, void *stream) {
  if (!A || !B || !C) return fail("gemm_batched: null pointer");
  if (batch < 0 || M < 0 || N < 0 || K < 0 || ldc < N) return fail("gemm_batched: bad sizes");
  if (lda < (transA ? M : K) || ldb < (transB ? N : K)) return fail("gemm_batched: leading dimension too small");
  return gemm_batched_launch(batch, transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, bias, strideBias,
                             C, ldc, strideC, relu, accumulate, nullptr, 0, 0, (hipStream_t)stream);
}

// ---------------------------------------------------------------- multi-agent SAC update (section 8f)
#include "aomarl_sac.hip"
