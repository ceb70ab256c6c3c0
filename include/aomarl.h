/*
 * aomarl.h -- C ABI of the MI355X-native AO environment hot path (libaomarl_hip.so).
 *
 * DROP-IN BOUNDARY.  The reference (Tomeu7/AO-MARL) reaches its per-frame arithmetic through the
 * pybind11 modules `sutraWrap` / `carmaWrap` of COMPASS (shesha/sutra_wrap.py:4-38,46-72); there
 * is no C header in its tree.  Each entry point below names the native call it replaces and the
 * reference call site that drives it.  Differences by design:
 *   - every call is BATCHED over environments [env_begin, env_begin+env_count) (independent
 *     atmosphere seeds); batch size 1 reproduces the reference's single simulation;
 *   - state lives in caller-owned device buffers (aomarl_state): no hidden device<->host copies
 *     (the reference copies ~8 arrays per frame through np.array(d_xxx), rtcCompass.py:114,310);
 *   - every call takes an explicit hipStream_t and returns an int status (0 = ok); the message of
 *     the last failure on the calling thread is aomarl_last_error().
 * Plain C types only: pointers + sizes, no torch/HIP types in signatures (stream is void*).
 *
 * Conventions: fp32; flat pixel index p = x + n*y (x fast); per-env vectors are rows [env][i].
 */
#ifndef AOMARL_H
#define AOMARL_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AOMARL_MAX_LAYERS 8
#define AOMARL_MAX_DMS 4
#define AOMARL_ABI_VERSION 2

enum { AOMARL_DM_PZT = 0, AOMARL_DM_TT = 1 };

/* flags of aomarl_raytrace_* (sourceCompass.py:54-85: tel/atm/dms/reset arguments) */
enum { AOMARL_TRACE_ATMOS = 1, AOMARL_TRACE_DMS = 2, AOMARL_TRACE_RESET = 4,
       AOMARL_TRACE_MASK = 8 /* multiply the result by the pupil (geometric controller input) */ };
/* flags of aomarl_comp_image */
enum {
  AOMARL_IMG_FROM_PHASE_BUFFER = 1, /* read st->wfs_phase (else: fused integer-offset raytrace) */
  AOMARL_IMG_NOISE = 2,             /* apply photon / read-out noise if desc.noise >= 0       */
  AOMARL_IMG_WRITE_BINCUBE = 4,     /* store the 16x16 spot images (st->bincube)              */
  AOMARL_IMG_COG = 8,               /* fused centre of gravity -> st->slopes                  */
  AOMARL_IMG_NO_ATMOS = 16,         /* fused raytrace: DMs only (interaction matrix)          */
  AOMARL_IMG_NO_DMS = 32,           /* fused raytrace: atmosphere only                        */
  AOMARL_IMG_DM_FROM_VOLTAGE = 64   /* aomarl_frame_fused only: evaluate the stack-array DM phase
                                       from st->voltage on the fly (st->dm_shape's stack-array
                                       planes are not read; see aomarl_dm_from_voltage_available) */
};
/* bits of aomarl_apply_control's comp_voltage argument */
enum {
  AOMARL_APPLY_COMP_VOLTAGE = 1,     /* run the delay line (Rtc.apply_control's compVoltage)     */
  AOMARL_APPLY_DEFER_STACK_SHAPE = 2 /* do not materialise the stack-array shapes now: the next
                                        aomarl_frame_fused(DM_FROM_VOLTAGE) evaluates them from
                                        st->voltage; ignored when that path is unavailable.
                                        aomarl_materialize_dm_shape refreshes them on demand     */
};

typedef struct {
  int32_t type;          /* AOMARL_DM_PZT | AOMARL_DM_TT                                   */
  int32_t dim;           /* support is dim x dim (dm_init.py:146-147,168)                  */
  int32_t nact;
  int32_t influsize;     /* pzt: side of one influence patch (dm_init.py:373)              */
  int64_t ninflupos;     /* pzt: len(influpos)                                             */
  const float *influ;    /* pzt: [nact][ss][ss] = influ.flatten('F') of the (ss,ss,nact) cube
                            tt : [dim*dim][2]   = C-order (dim,dim,2) cube (dm_init.py:661-694) */
  const int32_t *influpos;   /* dm_init.py:800 */
  const int32_t *ninflu;     /* [dim*dim]  dm_init.py:804 */
  const int32_t *influstart; /* [dim*dim]  dm_init.py:805 */
  float wfs_xoff, wfs_yoff;  /* wfs_init.py:196-204  */
  float tar_xoff, tar_yoff;  /* target_init.py:119-141 */
} aomarl_dm_desc;

typedef struct {
  int32_t dim;          /* screen is dim x dim (atmos_init.py:94-96)              */
  int32_t nstencil;     /* iterkolmo.py:76-96                                      */
  const float *A;       /* [dim][nstencil] row-major (iterkolmo.py:229)            */
  const float *B;       /* [dim][dim]                (iterkolmo.py:239)            */
  const uint32_t *istx; /* [nstencil] flat logical indices, mirrored if deltax < 0 */
  const uint32_t *isty;
  float deltax, deltay; /* pixels per frame (atmos_init.py:99-102)                 */
  float amplitude;      /* r0_layer^(-5/6) * 0.5/(2 pi): screens in microns        */
  float wfs_xoff, wfs_yoff; /* wfs_init.py:177-185   */
  float tar_xoff, tar_yoff; /* target_init.py:104-113 */
} aomarl_layer_desc;

typedef struct {
  int32_t abi_version; /* = AOMARL_ABI_VERSION */
  /* pupil (geom_init.py:813-868) */
  int32_t n, pupdiam;
  const float *mpupil; /* [n*n]            */
  const float *spupil; /* [pupdiam*pupdiam] */
  /* Shack-Hartmann WFS (geom_init.py:168-321, 622-810; Sensors ctor wfs_init.py:107-110) */
  int32_t nvalid, pdiam, nfft, npix, nrebin, nxsub;
  const int32_t *phasemap; /* [pdiam^2][nvalid] */
  const float *halfxy;     /* [pdiam^2]         */
  const int32_t *binmap;   /* [nrebin^2][npix^2] */
  const float *flux;       /* [nvalid] fluxPerSub of valid subaps (wfs_init.py:145) */
  const int32_t *validsubsx, *validsubsy; /* [nvalid] pixel coords in binimg */
  float nphot, wfs_lambda, noise, cog_offset, cog_scale, subapd;
  /* atmosphere */
  int32_t nlayers;
  aomarl_layer_desc layers[AOMARL_MAX_LAYERS];
  /* DMs of the controller, stack arrays first, tip-tilt last */
  int32_t ndm;
  aomarl_dm_desc dms[AOMARL_MAX_DMS];
  /* science target */
  float tar_lambda;
  int32_t npsf;           /* FFT support of the PSF the window is cut from */
  int32_t strehl_halfwin; /* PSF evaluated on frequencies [-hw, hw) in x and y */
  /* controller (rtc_init.py:385-389, 506-513) */
  int32_t nactu, nslope;
  float gain, delay;
} aomarl_desc;

/* Device buffers owned by the caller (e.g. torch tensors); all [nenv] leading. */
typedef struct {
  int32_t nenv;
  int32_t ld_actu;     /* row stride (floats) of com/com1/com2/err/voltage, >= nactu, % 4 == 0 */
  float *screens;      /* [nenv][aomarl_screen_stride] ring-buffered phase screens (microns);
                          private layout, use aomarl_get_screen / aomarl_set_screen            */
  int32_t *origin;     /* [nenv][nlayers][2]     ring origin (ox, oy)                       */
  uint32_t *seeds;     /* [nenv]                 atmosphere base seed (layer k uses seed+k) */
  uint32_t *ext_count; /* [nenv][nlayers]        extrusions drawn so far (RNG counter)      */
  float *com, *com1, *com2, *err, *voltage; /* [nenv][nactu]                               */
  float *slopes;       /* [nenv][nslope]         all x then all y (ao_env.py:665-666)       */
  float *dm_shape;     /* [nenv][aomarl_dmshape_stride]: dim^2 per stack array, 4 per tip-tilt */
  float *bincube;      /* [nenv][nvalid][npix^2] or NULL                                    */
  float *wfs_phase;    /* [nenv][n*n]            or NULL (only the unfused API needs it)    */
  float *tar_phase;    /* [nenv][pupdiam^2]      or NULL                                    */
  float *strehl;       /* [nenv][8]: se, le, phase_var, phase_var_sum, count, peak_on_edge  */
  float *le_img;       /* [nenv][(2hw)^2]        long-exposure PSF window                   */
  uint32_t *frame;     /* [nenv]                 WFS noise frame counter                    */
  float *work;         /* aomarl_workspace_floats(ctx, nenv) floats                         */
} aomarl_state;

typedef struct aomarl_ctx aomarl_ctx;

const char *aomarl_last_error(void);
int aomarl_abi_version(void);

/* Arithmetic of the library (process-wide).  The reference computes in fp32 throughout (Rtc_FFF,
 * shesha/sutra_wrap.py:49; np.float32 arrays, shesha/init/wfs_init.py:76-101), and AOMARL_PRECISION_F32 --
 * fp32 operands on fp32 matrix instructions, fp32 vector arithmetic in every kernel -- is the default.
 * AOMARL_PRECISION_SPLIT_F16 is the opt-in fast mode: operands carried as fp16 pairs (hi + lo, 22-bit
 * mantissa, fp32 accumulation) in the three kernel families that have such a form: the DFTs of the
 * one-pass frame kernel, the internal GEMMs (extrusion, command matrix, Btt projections; see
 * aomarl_gemm_nt_split) and the denoiser (aomarl_denoiser_apply).  The per-family switches of
 * aomarl_set_option ("force_f32_dft", "gemm_split_f16") override it for one family.
 * aomarl_arith_*: launches per arithmetic family since aomarl_arith_reset, e.g.
 * "gemm:split_f16_mfma" -- bench.py builds its `dtype` from what was actually launched. */
enum { AOMARL_PRECISION_F32 = 0, AOMARL_PRECISION_SPLIT_F16 = 1 };
int aomarl_set_precision(int mode);
int aomarl_get_precision(void);
/* Fast mode only: number of kernel threads of split-fp16 GEMM launches on the current device since the
 * last query that staged an operand whose scaled value left the fp16 range (|v| > 65504: it is clipped
 * there, the product is wrong).  Synchronises `stream`, clears the counter.  The internal call sites keep
 * margins of 10^2 .. 10^4 over what a closed loop produces (stencil differences up to 255 um, Btt
 * coordinates up to 4094, slopes up to 65504 arcsec); a diverging policy can leave them.  ao_marl_amd.env
 * checks it at every episode boundary, like aomarl_denoiser_overflow.  ONE counter per device for the whole
 * process, cleared by whoever reads it first: with several contexts alive (training + evaluation) a clipping
 * event is reported to the first reader, whichever context it happened in. */
int aomarl_gemm_saturated(unsigned *count, void *stream);
int aomarl_arith_families(void);
const char *aomarl_arith_family_name(int family);
unsigned long long aomarl_arith_launches(int family);
void aomarl_arith_reset(void);

/* Build the static device-side description (replaces the Telescope/Atmos/Sensors/Dms/Target/
 * Rtc constructors + load_arrays calls of shesha/init/xxx_init.py). Host pointers are copied. */
int aomarl_create(const aomarl_desc *desc, aomarl_ctx **out);
int aomarl_destroy(aomarl_ctx *ctx);
/* d_control[0].set_cmat (basis.py:254): cmat is [nactu][nslope] row-major, host memory */
int aomarl_set_cmat(aomarl_ctx *ctx, const float *cmat);
int aomarl_set_gain(aomarl_ctx *ctx, float gain); /* d_control[0].set_gain (ao_env.py:957) */
/* One integrator gain per environment (host [nenv], nenv = st->nenv of the states stepped with this
 * context) instead of the scalar: the gain scan of obtain_best_gain_and_modes_filtered.py:101-150 as
 * ONE batch, environment e running the reference's loop with gain gains[e].  NULL: scalar again.
 * Applies to aomarl_do_control only; the RL control entry points take their gain as an argument. */
int aomarl_set_env_gains(aomarl_ctx *ctx, const float *gains, int nenv);
/* volts2modes [nmodes][nactu], modes2volts [nactu][nmodes] (rlSupervisor.py:170-172),
 * freedom vector [nmodes] (rlSupervisor.py:277-278), action_modes[nact]: the modes an action
 * component drives (rlSupervisor.py:677-691); host memory */
int aomarl_set_modal(aomarl_ctx *ctx, int nmodes, const float *v2m, const float *m2v,
                     const float *freedom, int nact, const int32_t *action_modes);
size_t aomarl_workspace_floats(const aomarl_ctx *ctx, int nenv);
size_t aomarl_screen_stride(const aomarl_ctx *ctx);   /* floats per env in st->screens  */
size_t aomarl_dmshape_stride(const aomarl_ctx *ctx);  /* floats per env in st->dm_shape */

/* RlSupervisor.reset (rlSupervisor.py:236-246): Atmos.set_seed/refresh_screen per layer
 * (atmosCompass.py:137-145), integrator + DM shapes + Strehl meter zeroed.
 * seeds: host [env_count]; accumx/accumy: host [nenv][nlayers], zeroed for the reset envs. */
int aomarl_reset(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                 const uint32_t *seeds, float *accumx, float *accumy, void *stream);

/* The next episode's reset, hidden behind the running one.  The trainer knows the next seeds while an episode
 * runs (train_rpc.py:486-487: `seed += 1`), and refresh_screen is 2 x dim DEPENDENT extrusions per layer
 * (atmosCompass.py:137-145) -- most of a reset's time.  aomarl_reset_prefetch_begin starts them in a SHADOW state
 * (own screens / origin / seeds / ext_count / frame / com, com1, com2, err, voltage / work; the other buffers may
 * alias the live state's: they are not touched), aomarl_reset_prefetch_advance runs `nrounds` more rounds (< 0: all
 * that are left) -- both on a stream of the caller's choice, beside the live episode --, and aomarl_reset_adopt is
 * aomarl_reset with the screens COPIED from the shadow (it first runs whatever rounds are left, on
 * prefetch_stream).  Same kernels on the same columns with the k split aomarl_reset's partition of the batch gives
 * its products ("reset_prefetch_whole" below): the screens are the plain reset's, bit for bit (tests/test_gpu_glue.py).  The seeds must be those of the begin call.
 * `*remaining` = rounds still to run.  aomarl_reset_prefetch_cancel forgets a begun prefetch. */
int aomarl_reset_prefetch_begin(aomarl_ctx *ctx, const aomarl_state *shadow, int env_begin, int env_count,
                                const uint32_t *seeds, void *stream);
int aomarl_reset_prefetch_advance(aomarl_ctx *ctx, int nrounds, void *stream, int *remaining);
int aomarl_reset_prefetch_cancel(aomarl_ctx *ctx);
int aomarl_reset_adopt(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count, const uint32_t *seeds,
                       float *accumx, float *accumy, void *prefetch_stream, void *stream);
/* Atmos.move_atmos (atmosCompass.py:161). accumx/accumy: host [nenv][nlayers], updated. */
int aomarl_move_atmos(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                      float *accumx, float *accumy, void *stream);
/* The same move, issued EARLY: call it after the last kernel of this frame that reads the screens
 * has been enqueued on `stream`; the extrusions run on a stream of the library behind that point,
 * beside whatever `stream` does next (do_control, the agents, next_part_two).  The next
 * aomarl_move_atmos of the same state / range only waits for it (the move is not repeated); every
 * other entry point that touches the screens waits for it too, aomarl_reset drops it.  Between the
 * two calls the screens are one frame ahead of the slopes.  The composite aomarl_next_part_one does
 * this by itself under aomarl_set_option(ctx, "prefetch_atmos", 1); not for states that share their
 * screens (the GEO twin).  Results are those of the plain call order, bit for bit. */
int aomarl_prefetch_atmos(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                          float *accumx, float *accumy, void *stream);
/* Run-time changes of the atmosphere (AtmosCompass.set_wind / set_r0, atmosCompass.py:79-135; the trainer's
 * non-stationary experiments, train_rpc.py:429-450).  Host values of the context: every move PLANNED after the call
 * uses them; a move already issued by aomarl_prefetch_atmos keeps the wind it was planned with (the host side
 * refuses to step across that: ao_marl_amd/env.py VecAtmos), a prefetched reset in flight must be cancelled by the
 * caller when a sign of deltax changes (its rounds run along the old sign).  Both calls synchronise the device.
 *
 * aomarl_set_wind = Tscreen.set_deltax + set_deltay of layer `layer` (pixels per frame), and -- the rule of
 * atmosCompass.py:124-135 -- where old * new < 0 along an axis that axis' stencil is mirrored
 * (istencil -> dim * dim - 1 - istencil: set_istencilx / set_istencily).  mirror_stencils = 0: the deltas alone
 * (the facade's set_deltax / set_deltay; it mirrors through aomarl_set_stencil).
 * aomarl_set_stencil = Tscreen.set_istencilx (axis 0) / set_istencily (axis 1): n = the layer's stencil size, flat
 * logical indices, host memory.
 * aomarl_set_r0 = Atmos.set_r0: the amplitude of the noise term of every layer's new lines, amplitude[nlayers] =
 * r0_layer^(-5/6) * 0.5 / (2 pi) in um (atmos_init.py:115, iterkolmo.py:278); the screens as they stand are kept. */
int aomarl_set_wind(aomarl_ctx *ctx, int layer, float deltax, float deltay, int mirror_stencils);
int aomarl_set_stencil(aomarl_ctx *ctx, int layer, int axis, const uint32_t *istencil, int n);
int aomarl_set_r0(aomarl_ctx *ctx, const float *amplitude, int nlayers);
/* the layer's current host values (deltax, deltay, amplitude); any pointer may be null */
int aomarl_get_layer(const aomarl_ctx *ctx, int layer, float *deltax, float *deltay, float *amplitude);
/* one parallel round of extrusions: op i extrudes layer[i] in direction dir[i] (+-1 x, +-2 y) */
int aomarl_extrude(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count, int nops,
                   const int32_t *layer, const int32_t *dir, void *stream);
/* overwrite one layer with logical screens src [env_count][dim*dim] (device); ring origin -> 0 */
int aomarl_set_screen(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count, int layer,
                      const float *src, void *stream);
/* copy the logical (un-rotated) screen of one layer to dst [env_count][dim*dim] (device) */
int aomarl_get_screen(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count, int layer,
                      float *dst, void *stream);

/* Source.raytrace for the WFS guide star / the target (sourceCompass.py:76-85), materialising
 * st->wfs_phase / st->tar_phase. The fast path (aomarl_comp_image without FROM_PHASE_BUFFER,
 * aomarl_target_psf) fuses this and never writes the phase. */
int aomarl_raytrace_wfs(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                        int flags, void *stream);
int aomarl_raytrace_target(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                           int flags, void *stream);
/* Wfs.comp_image (wfsCompass.py:343) [+ Rtc.do_centroids when AOMARL_IMG_COG] */
int aomarl_comp_image(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count, int flags,
                      void *stream);
/* Rtc.do_centroids (rtcCompass.py:563) from st->bincube */
int aomarl_do_centroids(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                        void *stream);
/* Wfs.slopes_geom(0) from st->wfs_phase (imats.py:103) */
int aomarl_slopes_geom(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                       void *stream);
/* Rtc.do_control (rtcCompass.py:547): err = -cmat.s ; com += gain*err */
int aomarl_do_control(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                      void *stream);
/* d_control[0].set_com (rtcCompass.py:473): com_dev [env_count][nactu] device memory */
int aomarl_set_com(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                   const float *com_dev, void *stream);
/* RlSupervisor.rl_control -> correction_modal_basis (rlSupervisor.py:713-733, 784-818):
 * m = v2m.com ; m[action_modes] += action*freedom[action_modes] ; com = m2v.m
 * action_dev [env_count][nact] device memory */
int aomarl_rl_control(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                      const float *action_dev, void *stream);
/* aomarl_rl_control when the Btt coordinates of the current command are already known:
 *     modes = m0 + g * m1 (+ action * freedom on the action modes) ; com = m2v . modes
 * (m0, m1, modes_out: device [env_count][nmodes], modes_out may be NULL; action may be NULL).
 * The commands never leave span(Btt) (cmat = Btt . D+, rl_control projects on it) and v2m . m2v = I,
 * so v2m . com of the command the integrator left, com_before + g * err, is v2m . com_before +
 * g * v2m . err by linearity -- two vectors the environment has just computed for its state
 * (AoEnv.linear_step, ao_env.py:871-909): the v2m GEMM of rl_control and the one of the next
 * state are saved.  ao_marl_amd/env.py uses it inside step(); results differ from aomarl_rl_control
 * by fp32 round-off only. */
int aomarl_rl_control_modes(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                            const float *m0_dev, const float *m1_dev, float g,
                            const float *action_dev, float *modes_out_dev, void *stream);
/* Rtc.apply_control (rtcCompass.py:582): delay line -> voltage -> Dm.comp_shape per DM;
 * comp_voltage: AOMARL_APPLY_* bits (1 = the reference's compVoltage=True) */
int aomarl_apply_control(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                         int comp_voltage, void *stream);
/* 1 when the one-pass frame kernel can evaluate the stack-array DM from the command lattice
 * (separable influence functions on a regular lattice whose pitch divides the 16-pixel tile) */
int aomarl_dm_from_voltage_available(aomarl_ctx *ctx);
/* Dm.comp_shape of every DM from st->voltage, also after AOMARL_APPLY_DEFER_STACK_SHAPE */
int aomarl_materialize_dm_shape(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                                void *stream);
/* Dm.set_com + comp_shape (dmCompass.py:64-146): volts_dev [env_count][nactu] or NULL=voltage */
int aomarl_comp_dm_shape(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                         const float *volts_dev, void *stream);
/* materialise the shape of DM k into dst [env_count][dim*dim] (device memory). Stack-array shapes
 * live in st->dm_shape; tip-tilt shapes are never stored (consumers evaluate c0*f0 + c1*f1) */
int aomarl_get_dm_shape(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count, int k,
                        float *dst, void *stream);
/* implementation switches for tests / A-B measurements: "force_generic_dm" (per-pixel gather
 * tables instead of the separable-lattice kernel), "force_valu_target" (VALU PSF rows kernel),
 * "force_generic_spot" / "force_generic_target" (layout-agnostic kernels), "force_unfused_frame"
 * (separate target and WFS passes in aomarl_next_part_one), "force_f32_dft" (one-pass frame
 * kernel: 1 = fp32 MFMAs through LDS tiles, 0 = split-fp16 MFMAs from registers, -1 = follow
 * aomarl_set_precision, the default), "precision" (= aomarl_set_precision; process-wide, ctx may be NULL),
 * "gemm_target_blocks" (split-K target of the fast mode's split-fp16 GEMM; the fp32 products run on k_gemm_p,
 * csrc/aomarl_gemm_p.h, which picks tile and k split so that every CU gets the same share),
 * "gemm_xcd_map" (default 1: the split-f16 GEMM's blocks are renumbered so that one XCD works on one k-chunk and
 * its L2 holds that slice of both operands; 0: plain grid order; same values; process-wide, ctx may be NULL),
 * "time_frame_kernel" (see aomarl_frame_kernel_time),
 * "gemm_split_f16" (1: the internal products on split-fp16 operands, see aomarl_gemm_nt_split; 0: on fp32
 * matrix instructions; follows aomarl_set_precision, i.e. 0, until set; process-wide),
 * "prefetch_atmos" (aomarl_next_part_one moves the next frame's atmosphere on a side stream, see
 * aomarl_prefetch_atmos),
 * "gemm_kgroups" (k-groups per tile of aomarl_gemm_batched: 0 = heuristic, 1 / 2 / 4; process-wide,
 * ctx may be NULL),
 * "reset_untransposed" (1: aomarl_reset runs its 2 n x-extrusions on the row-major screen itself; default 0:
 * on the transposed screen -- every new line a row instead of 648 scattered 4-byte writes -- followed by
 * one in-place transposition: same screens, bit for bit, 46 instead of 60 ms per 256 environments),
 * "extrude_unfused" (1: scatter and gather of two consecutive extrusion rounds with the same operations
 * as separate launches; default 0: one launch, k_extrude_sg -- same values),
 * "reset_streams" (default 2: a reset of >= 32 environments runs its 1296 dependent extrusion rounds in two
 * halves of the batch side by side, the second on the library's extrusion stream, each half's kernels filling the
 * other's latency -- 59 -> 51 ms per 256 environments; 1: one chain on the caller's stream; same kernels on the same
 * columns, the split-K rule of a product sees its half's columns; only with "prefetch_atmos" on),
 * "reset_prefetch_whole" (default 1: aomarl_reset_prefetch_* walks the rounds as ONE range whose products take the
 * tile and the k split a half's product gets -- a sum's order depends on the k split alone, so the screens are the
 * plain reset's bit for bit, with half the launches beside the running episode: 47 instead of 60 ms of rounds per 256
 * environments, the step beside them 2.7 % shorter; 0: the plain reset's halves one after the other),
 * "small_move" (default 1: screens of <= 256 pixels with stencil + dim <= 4096 -- the 10x10 files -- move in ONE
 * launch per frame, k_move_small, instead of gather / GEMM / scatter rounds; fp32 vector FMAs in both precision
 * modes; 0: the rounds),
 * "renew_frame_stream" (any value): the frame pipeline's own stream is destroyed and created anew -- the runtime deals
 *   its hardware queues out as streams come, and a frame stream that shares one with another stream of the step makes
 *   the pipelined order nearly twice as slow; nothing may be in flight (behind a full-range reset).  VecAoEnv's probe
 *   of the two call orders asks for it before it settles for the plain order.
 * "small_chain" (default 1: systems with <= 512 actuators / modes, <= 1024 slopes and nactu x nslope <= 65536 run the
 * control / agent chain of aomarl_env_step as two workgroup-per-environment kernels, k_small_head / k_small_tail,
 * instead of three GEMMs and five elementwise kernels; fp32 round-off apart, the same numbers; 0: the general chain),
 * "frame_pipeline" (default 1; 0 = plain call order although aomarl_set_frame_pipeline gave a twin; refused while
 * a frame is in flight),
 * "defer_dm_shape" (the composites
 * aomarl_next_part_two / aomarl_next_part_one use AOMARL_APPLY_DEFER_STACK_SHAPE /
 * AOMARL_IMG_DM_FROM_VOLTAGE when available: st->voltage is the DM state and the stack-array
 * planes of st->dm_shape stay stale until aomarl_materialize_dm_shape) */
int aomarl_set_option(aomarl_ctx *ctx, const char *name, int value);
/* fused target raytrace + PSF window + phase variance into a pending slot
 * (RlSupervisor.raytrace_target, rlSupervisor.py:845-855) */
int aomarl_target_psf(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                      void *stream);
/* aomarl_target_psf + aomarl_comp_image in ONE pass over the phase (the two halves of
 * RlSupervisor.next_part_one between move_atmos and do_control, rlSupervisor.py:829-843): the
 * sub-aperture tiles of the WFS are the 16 x 16 tiles of the pupil grid the target sees, so every
 * screen / DM pixel is read once and feeds both paths.  Needs the geometry to line up (WFS grid =
 * pupil grid + symmetric guard band, same integer offsets, binary pupil, DMs = [stack array,
 * tip-tilt]): aomarl_frame_fused_available() says whether it does
 * (and "force_unfused_frame" was not set).  flags as aomarl_comp_image minus FROM_PHASE_BUFFER /
 * NO_ATMOS / NO_DMS.  aomarl_next_part_one uses it automatically when available. */
int aomarl_frame_fused_available(aomarl_ctx *ctx);
/* Name of the kernel instantiation the last aomarl_frame_fused of this context launched, spelled
 * as rocprofv3 prints it ("k_frame_wave<3, 1, true, false, false, true>": layers, lattice blocks,
 * DM from voltage, noise, bincube, split-fp16 DFT); "" before the first launch.  bench.py matches
 * it against the kernel name recorded in the profiles/ file it takes the HBM traffic from. */
const char *aomarl_frame_kernel_name(aomarl_ctx *ctx);
/* Under aomarl_set_option(ctx, "time_frame_kernel", nlaunches) every k_frame_wave launch carries a HIP
 * event pair on its own stream, attached to the dispatch as its start / stop events
 * (hipExtLaunchKernelGGL: no marker packets of their own on the queue; the first `nlaunches` launches
 * after the option is set or the times were last read).  This reads them: sum of the launch durations and their count; waits for the
 * recorded launches, then starts over.  bench.py's roofline.achieved comes from here. */
int aomarl_frame_kernel_time(aomarl_ctx *ctx, double *total_ms, int *launches);
int aomarl_frame_fused(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count, int flags,
                       void *stream);
/* General batched fp32 GEMM (forward and backward passes of the stacked SAC networks):
 * C[b] = act(opA(A[b]) . opB(B[b]) + bias[b]) (+ C[b] if accumulate); opA(A) is M x K (stored [M][K],
 * or [K][M] when transA), opB(B) is K x N (stored [N][K], or [K][N] when transB).
 * y = x W, dx = dy W^T, dW = x^T dy of a layer with W stored [in][out] are (0,1), (0,0), (1,1). */
int aomarl_gemm_batched(int batch, int transA, int transB, int M, int N, int K, const float *A, int lda,
                        long long strideA, const float *B, int ldb, long long strideB, const float *bias,
                        long long strideBias, float *C, int ldc, long long strideC, int relu,
                        int accumulate, void *stream);
/* ---- multi-agent soft actor-critic update (the learner side of the path) --------
 * One call = update_critic -> update_actor -> update_alpha -> soft_update of EVERY agent on one
 * replay batch per agent (reference: SAC.update_critic / update_actor / update_alpha / soft_update,
 * src/reinforcement_learning/rpc_training/train_rpc.py:985-1038, 1044-1064, 1070-1084, 1128-1129,
 * networks src/reinforcement_learning/rpc_training/model_rpc.py:60-160), forward AND hand-derived
 * backward passes on the batched MFMA GEMM, Adam (torch.optim.Adam defaults) and the target update
 * fused into one kernel per parameter set.
 *
 * Parameters live in caller-owned flat device buffers (zero-padded per agent, fp32):
 *   policy : W1 [A][in_max][H], b1 [A][H], (Wh [A][H][H], bh [A][H]) x (n_hidden - 1),
 *            Whead [A][H][2 act_max] (mean | log_std columns), bhead [A][2 act_max]
 *   critic : Win [A][in_max + act_max][2 Hc] (Q1 | Q2 columns; rows: state then action),
 *            bin [A][2 Hc], Wout [A][2][Hc], bout [A][2]
 * each tensor starting on a multiple of 4 floats (aomarl_sac_layout returns the offsets). */
typedef struct {
  int32_t n_agents, batch, in_max, act_max, hidden, hidden_critic, n_hidden;
  int32_t state_dim, action_dim;         /* row lengths of the replay ring */
  const int32_t *state_gather;           /* host [A][in_max]: column of the state row, < 0 = zero pad */
  const int32_t *action_gather;          /* host [A][act_max]: column of the action row, < 0 = pad */
  const int32_t *n_act;                  /* host [A]: live action columns of each agent */
  const float *target_entropy;           /* host [A] */
  float gamma, tau, lr, beta1, beta2, adam_eps;
  float log_sig_min, log_sig_max, action_scale, action_bias;
  /* device buffers, caller-owned: parameters, Adam moments (m, v: zero before the first update) and
   * the gradients of the last update (written by every call, laid out like the parameters) */
  float *policy, *policy_m, *policy_v, *policy_grad;                  /* policy_len floats each */
  float *critic, *critic_m, *critic_v, *critic_grad, *critic_target;  /* critic_len floats each */
  float *log_alpha, *log_alpha_m, *log_alpha_v, *log_alpha_grad, *alpha;   /* [A] each */
} aomarl_sac_desc;
typedef struct aomarl_sac aomarl_sac;
#define AOMARL_SAC_SOFT_UPDATE 1      /* target <- (1 - tau) target + tau critic after the critic step */
#define AOMARL_SAC_TUNE_ALPHA 2       /* automatic entropy tuning */
/* offsets (floats) of the tensors inside the flat buffers; policy_off [2 n_hidden + 2] in the order
 * above, critic_off [4]; returns the two lengths. */
int aomarl_sac_layout(const aomarl_sac_desc *d, long long *policy_off, long long *critic_off,
                      long long *policy_len, long long *critic_len);
int aomarl_sac_create(const aomarl_sac_desc *d, aomarl_sac **out);
int aomarl_sac_destroy(aomarl_sac *s);
/* replay ring: state / next_state [rows][state_dim], action [rows][action_dim], reward [rows][A],
 * mask [rows].  idx: device [A][batch] row numbers (int64), or NULL: drawn uniformly in
 * [0, replay_rows) from Philox (seed, counter).  eps_next / eps_pi: device [A][batch][act_max]
 * standard-normal draws, or NULL: Philox (seed, counter).  adam_step: 1-based update count.
 * losses: device [5][A] (q1, q2, policy, alpha losses, alpha value) or NULL. */
int aomarl_sac_update(aomarl_sac *s, const float *state, const float *next_state, const float *action,
                      const float *reward, const float *mask, long long replay_rows, const int64_t *idx,
                      const float *eps_next, const float *eps_pi, uint32_t seed, uint32_t counter,
                      int adam_step, int flags, float *losses, void *stream);
/* ---- agent-side glue, all device pointers, no context (fused chains of the tiny host operations
 * the reference does per agent per step):
 * aomarl_split_states   TrainerRPC.divide_states_for_agents (train_rpc.py:418-427):
 *                       out[a][e][k] = state[e][gather[a][k]], gather == state_dim -> 0 (padding)
 * aomarl_policy_sample  GaussianPolicy.sample(only_choosing_action) on the head outputs
 *                       (model_rpc.py:137-144): head [A][nenv][2*act_max] = mean | log_std ->
 *                       action, mean [nenv][action_dim]; eps: standard normals [nenv][action_dim]
 *                       or NULL (Philox4x32-10 keyed by seed, counter = step, env, action index)
 * aomarl_assemble_state AoEnv.linear_step's concatenation + standardise (ao_env.py:470-480,
 *                       871-909): out[e] = concat_k (src_k[e] - mean_k) / std_k  (mean/std NULL:
 *                       raw); src / mean / std_ are HOST arrays of device pointers
 * aomarl_agent_rewards  helper_rewards.get_separated_rewards (helper_rewards.py:14-22):
 *                       out[e][a] = -factor * mean(res[e][lo_a:hi_a]^2), lohi [A][2] on the device */
int aomarl_split_states(int nenv, int state_dim, int n_agents, int in_max, const int32_t *gather,
                        const float *state, float *out, void *stream);
int aomarl_policy_sample(int nenv, int act_max, int action_dim, const float *head, float log_sig_min,
                         float log_sig_max, float scale, float bias, const int32_t *sc_agent,
                         const int32_t *sc_local, const float *eps, uint32_t seed, uint32_t counter,
                         float *action, float *mean, void *stream);
int aomarl_assemble_state(int nenv, int nblocks, const float *const *src, const int32_t *ld,
                          const int32_t *dim, const float *const *mean, const float *const *std_,
                          float *out, void *stream);
/* The same with a column selection: out block k = standardise(src_k[:, sel[0 .. dim_k)]) (the
 * action-range sub-selection of transform_state_to_zernike, ao_env.py:482-505). sel NULL = identity. */
int aomarl_assemble_state_cols(int nenv, int nblocks, const float *const *src, const int32_t *ld,
                          const int32_t *dim, const float *const *mean, const float *const *std_,
                          const int32_t *sel, float *out, void *stream);
int aomarl_agent_rewards(int nenv, int nmodes, int n_agents, const float *res_modes, int ld,
                         const int32_t *lohi, float factor, float *out, void *stream);
/* ---- one native call per half of a training step.  Same launches, same order as the entry points
 * above; what they save is the host: ~25 calls per step at ~10 us each from the Python side.
 *
 * aomarl_actor_forward   TrainerRPC.choose_action for every agent (train_rpc.py:650-675):
 *                        divide_states_for_agents -> GaussianPolicy.forward (Linear + ReLU stack, merged
 *                        mean | log_std head) -> sample(only_choosing_action) -> the global action vector.
 *                        Weights in nn.Linear layout, stacked over agents, zero-padded to in_max / act_max. */
typedef struct {
  int32_t n_agents, nenv, state_dim, in_max, act_max, hidden, n_hidden, action_dim;
  const int32_t *gather;               /* device [A][in_max]: state column, state_dim = zero pad      */
  const float *W1, *b1;                /* device [A][H][in_max], [A][H]                               */
  const float *const *Wh, *const *bh;  /* HOST arrays of n_hidden - 1 device pointers [A][H][H], [A][H] */
  const float *Whead, *bhead;          /* device [A][2 act_max][H], [A][2 act_max]                    */
  const int32_t *sc_agent, *sc_local;  /* device [action_dim]                                         */
  float log_sig_min, log_sig_max, scale, bias;
  float *x, *h0, *h1, *head;           /* device scratch [A][nenv][in_max], [A][nenv][H] x 2, [A][nenv][2 act_max]
                                          (only the layer-by-layer path uses them)                    */
  int32_t flags;                       /* AOMARL_ACTOR_*                                              */
  const float *W1_tiled;               /* device copies of W1 / Wh / Whead in the tile order of      */
  const float *const *Wh_tiled;        /* aomarl_actor_tile_weights (HOST array for Wh), or NULL      */
  const float *Whead_tiled;
} aomarl_actor_desc;
/* With the tiled copies at hand (and hidden % 16 == 0): ONE launch -- one workgroup per agent x 16
 * environments, activations in LDS, fp32 matrix instructions.  Without them, or with this flag:
 * split_states + one batched GEMM per layer + policy_sample.  Same arithmetic up to the order of the
 * fp32 sums. */
#define AOMARL_ACTOR_LAYER_BY_LAYER 1
/* dst[a] = the [N][K] matrix src[a] (nn.Linear layout, stacked over agents) cut into 16 x 16 tiles in
 * the operand order of the 16 x 16 x 4 matrix instruction, zero-padded: tile (n, s) holds 64 x float4,
 * entry l = row 16 n + (l & 15), columns 16 s + 4 (l >> 4) .. + 3.  dst: aomarl_actor_tiled_floats(). */
long long aomarl_actor_tiled_floats(int n_agents, int N, int K);
int aomarl_actor_tile_weights(int n_agents, int N, int K, const float *src, float *dst, void *stream);
int aomarl_actor_forward(const aomarl_actor_desc *d, const float *state, const float *eps, uint32_t seed,
                         uint32_t counter, float *action, float *mean, void *stream);
/* aomarl_env_step        TrainerRPC.env_step (train_rpc.py:633-648) for the default state layout
 *                        (dm_history_n .. 1, dm_before_linear, dm_residual; parameters.cfg:31-37), all
 *                        environments of the state:  rl_step (Btt correction through
 *                        aomarl_rl_control_modes, apply_control, Strehl)  ->  per-agent rewards  ->
 *                        linear_step (aomarl_next_part_one, v2m . err, standardise + concatenate).
 *                        The Btt coordinates of the last nhist + 1 commands live in a ring the caller
 *                        owns; ring_pos (slot of the newest) is advanced by the call. */
typedef struct {
  int32_t nmodes, dm_dim, n_agents, nhist;
  int32_t ring_pos;
  const int32_t *sel;                  /* device [dm_dim]: modes that enter the state, or NULL = all  */
  const float *mean_dm, *std_dm, *mean_res, *std_res;   /* device [dm_dim], or all NULL: raw states  */
  const int32_t *lohi;                 /* device [n_agents][2] mode ranges of the agents' rewards     */
  float reward_factor;
  float *modes_ring;                   /* device [nhist + 1][nenv][nmodes]                            */
  float *res_modes;                    /* device [nenv][nmodes]: v2m . err of the last frame (in/out) */
  void *denoiser;                      /* aomarl_denoiser* or NULL: the autoencoder branch of
                                          next_part_one_integrator (rlSupervisor.py:975-984); needs st->bincube */
  int32_t denoiser_f32;                /* 1: aomarl_denoiser_apply_f32                                 */
  int32_t flags;                       /* AOMARL_ENV_STEP_*                                            */
} aomarl_env_glue;
/* Default: the chain with every split-K reduction folded into the kernel that consumes the product and
 * independent small kernels sharing a launch (10 launches per step on the main stream).  With this
 * flag: the entry points above called one after the other (14).  Bit-identical results. */
#define AOMARL_ENV_STEP_UNFUSED 1
int aomarl_env_step(aomarl_ctx *ctx, aomarl_state *st, aomarl_env_glue *glue, const float *action_dev,
                    float gain, float *accumx, float *accumy, float *state_out, float *reward_out,
                    void *stream);
/* TrainerRPC.choose_action + TrainerRPC.env_step (train_rpc.py:650-675, 633-648) in ONE call: action = the actors on
 * `state` (aomarl_actor_forward's arguments), then aomarl_env_step with that action -> state_out (the next state),
 * reward_out: the two entry points one behind the other, from C -- one host round trip per step instead of two (a
 * host-bound step of a small system notices: BASELINE configs[1]).  action / mean: [nenv][action_dim] outputs. */
int aomarl_policy_env_step(aomarl_ctx *ctx, aomarl_state *st, aomarl_env_glue *glue, const aomarl_actor_desc *d,
                           const float *state_dev, const float *eps_dev, uint32_t seed, uint32_t counter, float gain,
                           float *accumx, float *accumy, float *action_dev, float *mean_dev, float *state_out,
                           float *reward_out, void *stream);
/* aomarl_set_option(ctx, "residual_shortcut", 1) (needs aomarl_set_slopes2modes): aomarl_env_step takes the residual
 * modes v2m . err of a frame from ONE product of its slopes with -(v2m . cmat) instead of aomarl_do_control (cmat . s,
 * integrate) + v2m . err: the integrator then lives in the Btt coordinates alone (the next call's head rebuilds the
 * command from them, as the reference's rl_control does every step: rlSupervisor.py:784-818), st->err is not formed and
 * st->com is not integrated in actuator space -- aomarl_do_control on the same slopes gives both afterwards.  Same
 * mathematics, another order of the fp32 sums (states within 3e-3 relative of the default's over 10 steps of the 40x40
 * system, tests/test_gpu_glue.py).  Returns 1 when a call with this glue would take the shortcut (the option is on, the
 * matrix matches the glue's modes, the chain is fusable and the system is not a small one, whose tail kernel does
 * do_control itself), else 0: a host that defers do_control must know (VecAoEnv._step_native). */
int aomarl_env_step_shortcut(aomarl_ctx *ctx, const aomarl_env_glue *glue);
/* Rtc.do_control (rtcCompass.py:547) for the whole batch on the slopes of the frame the last aomarl_env_step REDUCED:
 * with a frame in flight (aomarl_set_frame_pipeline) those live in the state or in its twin by parity, and the call-by-
 * call aomarl_do_control is refused; this one picks the view itself (err and com are not parity buffers, the next
 * call's head rebuilds com from the Btt coordinates before the delay line takes it).  Without a frame in flight:
 * aomarl_do_control of the whole batch.  What a host that runs with "residual_shortcut" calls when somebody asks for
 * err / the integrated command in actuator space (rtc.get_err / rtc.get_command, rtcCompass.py:114-142). */
int aomarl_do_control_reduced(aomarl_ctx *ctx, aomarl_state *st, void *stream);
/* aomarl_set_option(ctx, "graph_step", 1): aomarl_env_step replays a HIP graph captured from its own launch
 * sequence (one per distinct extrusion plan x ring position x buffer addresses; captured the first time a
 * combination occurs): one hipGraphLaunch instead of ~25 launches + ~8 event operations per step, for the
 * launch-bound regime (small batches).  Same kernels, same arguments, same results.  Inside a graph the side
 * streams join the caller's stream at the end of the step, so at large batches the plain path (whose
 * extrusion chain runs on beside the next step's head) is the faster one: off by default.
 * aomarl_graph_stats: graphs captured / replayed so far on this context. */
int aomarl_graph_stats(aomarl_ctx *ctx, unsigned long long *captures, unsigned long long *replays);

/* Frames one step ahead of the control chain ("frame pipeline").  With a loop delay of exactly one frame
 * (p_controller.delay == 1, the production files) the voltages frame t+1 is formed with are the commands
 * of step t (rlSupervisor.py next_part_two: the delay line of shesha's generic controller), known BEFORE
 * frame t's slopes have been reduced: the frame kernel of step t+1 does not depend on the control / agent
 * chain of step t.  aomarl_set_frame_pipeline hands the library a TWIN of `st` -- same aomarl_state, own
 * slopes / voltage / dm_shape / work buffers, every other pointer equal to st's -- and aomarl_env_step then
 * keeps one frame in flight: the call of step t launches frame t+1 on a stream of its own (even frames in
 * st's buffers, odd ones in the twin's; ring origins from per-parity snapshots), moves the atmosphere for
 * frame t+2 beside it when the lines that move rewrites lie outside the windows the frame kernel reads
 * (checked per move; otherwise behind it), and only then reduces frame t's slopes.  Same kernels, same
 * arguments, same values as the plain call order, bit for bit; the frame kernels run back to back with the
 * two latency chains beside them instead of between them.
 * Used only when the step is eligible (delay == 1, noise-free sensor, one-pass frame kernel with the DM
 * evaluated from the voltages, prefetch_atmos on, no denoiser, no graph_step, the whole batch); otherwise
 * aomarl_env_step takes the plain path.  While a frame is in flight the only calls accepted on this state
 * are aomarl_env_step and a full-range aomarl_reset (which drops it: the episode is over) -- everything
 * else fails loudly, because slopes / voltage of odd frames live in the twin and the screens are a frame
 * ahead.  twin == NULL switches the pipeline off (refused while a frame is in flight).
 * aomarl_frame_pipeline_state: in_flight = a frame is in flight; consumed_in_twin = the slopes / voltage of
 * the LAST REDUCED frame are in the twin's buffers; steps = pipelined steps so far; overlapped = moves that
 * ran beside the frame kernel. */
int aomarl_set_frame_pipeline(aomarl_ctx *ctx, const aomarl_state *st, const aomarl_state *twin);
int aomarl_frame_pipeline_state(aomarl_ctx *ctx, int *in_flight, int *consumed_in_twin,
                                unsigned long long *steps, unsigned long long *overlapped);

/* WFS-image denoiser in the loop (RlSupervisor.autoencoder_denoising, rlSupervisor.py:876-891;
 * DenoisingAutoencoderCNN2DSingleSubapeture.forward, src/autoencoder/autoencoder_models.py:130-197):
 * every 16 x 16 spot image of a bincube goes through the conv autoencoder, in place, in one fused
 * kernel.  weights / biases: 6 HOST arrays each in the reference checkpoint's layouts (encoder1..3
 * Conv2d [Cout][Cin][3][3]; decoder1, 2 ConvTranspose2d [Cin][Cout][4][4]; decoder3
 * ConvTranspose2d [16][1][3][3]).  cube: device [nimg][256], tiles [y][x] as aomarl_comp_image
 * writes them (the network sees them transposed, like the reference feeds them). */
typedef struct aomarl_denoiser aomarl_denoiser;
int aomarl_denoiser_create(const float *const *weights, const float *const *biases,
                           aomarl_denoiser **out);
/* In the library's precision mode (aomarl_set_precision): fp32 by default. */
int aomarl_denoiser_apply(aomarl_denoiser *dn, float *cube, long long nimg, void *stream);
/* Every product on fp32 matrix instructions (the reference's arithmetic; 4x the matrix-pipe time of the
 * split form). */
int aomarl_denoiser_apply_f32(aomarl_denoiser *dn, float *cube, long long nimg, void *stream);
/* Fast mode: each operand carried as an fp16 pair (hi + lo, 22 mantissa bits, fp32 accumulation): same
 * results to fp32 rounding as long as inputs and activations stay inside the fp16 range (|v| < 65504);
 * what leaves it is counted (aomarl_denoiser_overflow). */
int aomarl_denoiser_apply_split_f16(aomarl_denoiser *dn, float *cube, long long nimg, void *stream);
/* Number of kernel threads of split-fp16 denoiser launches since the last query that saw an activation
 * outside the fp16 range (the split-fp16 operands saturate there instead of overflowing loudly);
 * synchronises `stream`, clears the counter.  Non-zero: those results are wrong -- rerun on
 * aomarl_denoiser_apply_f32.  ao_marl_amd.denoiser checks it at every episode boundary. */
int aomarl_denoiser_overflow(aomarl_denoiser *dn, unsigned *count, void *stream);
int aomarl_denoiser_destroy(aomarl_denoiser *dn);
/* PSF window + phase variance of st->tar_phase as it stands (pending, like aomarl_target_psf) */
int aomarl_target_psf_buffer(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                             void *stream);
/* Target.get_tar_image(tar_index, expo_type = "se") (shesha/supervisor/components/targetCompass.py:71-92): the whole
 * short-exposure PSF of every environment of the range, out [env_count][npsf][npsf] ([ky][kx], zero frequency at
 * (npsf/2, npsf/2): what fftshift(np.array(d_image_se)) holds, transposed to this library's [y][x] order), raw
 * |FFT2|^2 like d_image_se.  On demand only -- the hot path forms the central window (aomarl_target_psf); three
 * reward branches of the reference read the full frame (ao_env.py:621-623, 654-656).  Two DFT passes on the
 * library's fp32 GEMM, one environment at a time; the phase is ray-traced from the state as it stands (stack-array
 * shapes are materialised first if they are deferred). */
int aomarl_target_image(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count, float *out, void *stream);
/* Geometric ("GEO") reference controller: COMPASS's sutra_controller_geo as the reference sets it
 * up (rtc_init.py:418-448 init_proj_sparse over the pupil pixels) and drives it
 * (rlSupervisor.py:989-1013: target.raytrace(atmosphere) -> rtc.do_control(sources=target) ->
 * apply_control -> target.raytrace(dms)).  The command is the least-squares projection of the
 * piston-removed pupil phase on the influence functions, tip-tilt first, stack array on the rest:
 *     com = W . [ IF_stack^T (m phi) | TT^T (m phi) | sum(m phi) ]
 * with W [nactu][nactu+1] (row-major, host) from ao_marl_amd.modal.geo_projector.  The products
 * are evaluated as two small batched GEMMs over the separable lattice + one GEMM against the
 * tip-tilt planes.  st->tar_phase must hold the MASKED atmosphere phase of the target
 * (aomarl_raytrace_target with ATMOS | RESET | MASK); the result goes to st->com.  A GEO twin is a
 * second aomarl_state that shares screens / origin / seeds / ext_count with the main one and owns
 * com / voltage / dm_shape / tar_phase / strehl / le_img / work.  `work`: device scratch of
 * aomarl_geo_workspace_floats(ctx, nenv) floats. */
int aomarl_set_geo(aomarl_ctx *ctx, const float *W);
size_t aomarl_geo_workspace_floats(aomarl_ctx *ctx, int nenv);
int aomarl_geo_control(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                       float *work, void *stream);
/* Target.comp_image + comp_strehl (targetCompass.py:193,205): publish the pending PSF */
int aomarl_comp_strehl(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                       void *stream);
/* Target.comp_strehl(do_fit = True), the default of the reference's get_strehl (targetCompass.py:139-159): the Strehl
 * record of an environment is 8 floats -- [0] SR SE, [1] SR LE, [2] phase variance, [3] its sum, [4] frames, [5] peak on
 * the window edge, [6] SR SE with the PSF peak fitted by two 1-D sincs (filled by every commit: the fit comes with
 * the window), [7] SR LE fitted -- which THIS call brings up to date from the accumulated window (it is not on the
 * control chain; between two calls slot 7 holds the un-fitted value of the last commit).  Reads only the state's
 * own record and long-exposure window: allowed while a pipelined frame is in flight. */
int aomarl_strehl_fit(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count, void *stream);
int aomarl_reset_strehl(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                        void *stream);
/* modes[row][nmodes] = v2m . vec[row][0..nactu)  (AoEnv.transform_state_to_zernike,
 * ao_env.py:482-505); vec_dev row stride ldvec >= nactu, modes_dev row stride nmodes; st may be
 * NULL (then no split-K workspace is used) */
int aomarl_volts2modes(aomarl_ctx *ctx, aomarl_state *st, int nrows, const float *vec_dev,
                       int ldvec, float *modes_dev, void *stream);
/* Residual in Btt coordinates straight from the slopes:  v2m . err = -(v2m . cmat) . slopes.  Where the
 * next control step takes its command from modal coordinates (aomarl_rl_control_modes) the frame
 * needs neither err nor the integrated com in actuator space, so ONE product with the pre-multiplied
 * matrix s2m = v2m . cmat (host, [nmodes][nslope]; NULL drops it) replaces aomarl_do_control +
 * aomarl_volts2modes; aomarl_do_control run later on the same slopes gives err / com exactly as the
 * plain order would.  AoEnv.linear_step's get_err -> transform_state_to_zernike (ao_env.py:507-533). */
int aomarl_set_slopes2modes(aomarl_ctx *ctx, int nmodes, const float *s2m);
int aomarl_slopes2modes(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count, float *modes,
                        void *stream);

/* composites: one call per half frame, same order as the reference
 * next_part_one: move_atmos, target trace(+PSF), WFS trace+image+COG, do_control
 *                (rlSupervisor.py:1015-1051, 954-987)
 * next_part_two: [rl_control], apply_control, comp_strehl (rlSupervisor.py:900-947);
 *                action_dev may be NULL (integrator only, linear_control=True) */
int aomarl_next_part_one(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                         float *accumx, float *accumy, int image_flags, void *stream);
int aomarl_next_part_two(aomarl_ctx *ctx, aomarl_state *st, int env_begin, int env_count,
                         const float *action_dev, void *stream);

/* generic fp32 MFMA GEMM used by the calls above, exported for tests:
 * C[M][N] = alpha * A[M][K] . B[N][K]^T + beta * C ; device pointers, row-major, ld in floats */
int aomarl_gemm_nt(int M, int N, int K, float alpha, const float *A, int lda, const float *B,
                   int ldb, float beta, float *C, int ldc, void *stream);

/* The same product on the f16 matrix pipe with split operands (hi + lo fp16 pairs, 22 mantissa bits,
 * fp32 accumulation; the kernel behind the library's own extrusion / command-matrix / Btt-projection
 * products unless aomarl_set_option(ctx, "gemm_split_f16", 0)): A and B are scaled by the powers of two
 * scale_a / scale_b as they are staged (undone in the result), the scaled magnitudes must stay below
 * 65504.  work (device, may be NULL): split-K workspace of work_floats floats.  Exported for tests. */
int aomarl_gemm_nt_split(int M, int N, int K, float alpha, const float *A, int lda, const float *B, int ldb,
                         float beta, float *C, int ldc, float scale_a, float scale_b, float *work,
                         long long work_floats, void *stream);

/* batched fp32 MFMA GEMM with fused bias + ReLU for the stacked SAC MLPs (one batch entry per
 * agent): C[b][M][N] = act(A[b][M][K] . B[b][N][K]^T + bias[b][N]); B is in nn.Linear layout
 * ([out][in]); bias may be NULL; strides in floats; A, B 16-byte aligned */
int aomarl_gemm_nt_batched(int batch, int M, int N, int K, const float *A, int lda, long long strideA,
                           const float *B, int ldb, long long strideB, const float *bias,
                           long long strideBias, float *C, int ldc, long long strideC, int relu,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif
