/*
 * aoref -- CPU ORACLE for the AO environment hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (ao_marl_amd/) never does and fails loudly without its HIP extension.
 *
 * What it restates.  The reference (Tomeu7/AO-MARL) keeps all per-frame arithmetic in a
 * third-party binary that is NOT in its tree: COMPASS 5.1.0 (`sutraWrap`/`carmaWrap`,
 * README.md:15, shesha/sutra_wrap.py:46-72).  This file is a plain-C restatement of the stages
 * the reference drives each frame through that boundary, following the in-tree statements of
 * the algorithms where they exist and the published COMPASS semantics elsewhere:
 *   extrusion          shesha/util/iterkolmo.py:255-288 (x = A(z - zref) + B eps + zref)
 *   move_atmos         shesha/supervisor/components/atmosCompass.py:158-161
 *   raytrace           shesha/supervisor/components/sourceCompass.py:54-85, offsets wfs_init.py:170-204
 *   SH image           shesha/supervisor/components/wfsCompass.py:334-343; maps geom_init.py:622-810
 *   COG                shesha/supervisor/components/rtcCompass.py:557-563; offset/scale rtc_init.py:208,217
 *   LS control         rtcCompass.py:527-547; "rtc.get_err returns -CMAT.slopes" guardians/roket.py:169
 *   delay + DM shape   rtcCompass.py:573-582; gather tables dm_init.py:750-815
 *   PSF / Strehl       shesha/supervisor/components/targetCompass.py:139-205
 * PARITY PIN: the reference holds no golden vectors for this boundary, so
 * exact-value parity with COMPASS is UNPINNED; what is pinned is (a) every geometry array this
 * oracle consumes, bit-exact against the reference's own init code (tests/golden/geom_*.npz),
 * (b) the reference's own Python host logic executed over this oracle (tests/golden/trace_*),
 * (c) the statistical fixtures the reference recorded from real COMPASS (slope / mode std).
 *
 * Conventions: fp32 everywhere (Rtc_FFF); flat pixel index p = x + n*y, x fast.
 */
#ifndef AOREF_H
#define AOREF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- counter-based RNG: Philox4x32-10 (Salmon et al., SC'11; Random123 v1.09 KAT in tests) */
void aoref_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* n standard normals for (seed, stream, counter): Box-Muller on consecutive Philox outputs */
void aoref_normals(uint32_t seed, uint32_t stream, uint64_t counter, int n, float *out);
/* n uniforms in (0,1) */
void aoref_uniforms(uint32_t seed, uint32_t stream, uint64_t counter, int n, float *out);

/* ---- atmosphere */
/* one extrusion of a logical n x n screen; dir = +1/-1 (x) or +2/-2 (y); ist = stencil for that
 * axis (already mirrored for negative deltas, iterkolmo.py:246-249); eps = n normals */
void aoref_extrude(float *screen, int n, const float *A, int ns, const float *B,
                   const uint32_t *ist, int dir, float amplitude, const float *eps, float *tmp);

/* ---- raytrace: out[x + nx*y] (+)= bilinear(in, x + xoff, y + yoff) */
void aoref_raytrace(float *out, int nx, int ny, const float *in, int Nin, float xoff, float yoff,
                    int accumulate);

/* ---- DMs */
void aoref_pzt_shape(float *shape, int dim, const float *influ_flatF, const int32_t *influpos,
                     const int32_t *ninflu, const int32_t *influstart, int ss, const float *com);
void aoref_tt_shape(float *shape, int dim, const float *influ_xy2, const float *com2);

/* ---- Shack-Hartmann: phase (n x n, microns) -> bincube [nvalid][npix*npix] */
void aoref_sh_image(const float *phase, const float *mpupil, int nvalid, int pdiam, int nfft,
                    int npix, int nrebin, const int32_t *phasemap /*[pdiam^2][nvalid]*/,
                    const float *halfxy, const int32_t *binmap /*[nrebin^2][npix^2]*/,
                    const float *flux_valid, float nphot, float lambda_um, float *bincube);
/* photon + read-out noise on a bincube (noise < 0: none; 0: Poisson; > 0: Poisson + N(0,noise)) */
void aoref_sh_noise(float *bincube, int nvalid, int npix2, float noise, uint32_t seed,
                    uint64_t frame);
void aoref_cog(const float *bincube, int nvalid, int npix, float offset, float scale,
               float *slopes /*[2*nvalid]: all x then all y*/);
void aoref_fill_binimg(const float *bincube, int nvalid, int npix, const int32_t *validx,
                       const int32_t *validy, int imgdim, float *binimg);
/* geometric slopes (imat_geom / correct_dm only; scale-invariant use, dm_init.py:857-859) */
void aoref_slopes_geom(const float *phase, const float *mpupil, int n, int nvalid, int pdiam,
                       const int32_t *phasemap, const float *flux_valid, float subapd,
                       float *slopes);

/* ---- controller */
void aoref_gemv(const float *M, int rows, int cols, const float *x, float *y); /* row-major */
void aoref_ls_control(const float *cmat, int nactu, int nslope, const float *slopes, float gain,
                      float *err, float *com);

/* ---- target: PSF of pupil * exp(i 2 pi phase / lambda) on an nfft^2 grid.
 * psf_full (may be NULL): unshifted |FFT|^2 ; returns max over the full grid in *peak_full and the
 * max over the centred window of half-width hw (frequencies -hw .. hw-1) in *peak_win. */
void aoref_psf(const float *phase, const float *pupil, int n, int nfft, float lambda_um, int hw,
               float *psf_full, float *psf_win /*[(2hw)^2], may be NULL*/, float *peak_full,
               float *peak_win);
float aoref_phase_var(const float *phase, const float *pupil, int n); /* um^2 over pupil>0 */
/* Target.comp_strehl(do_fit = True) (targetCompass.py:139-159, the default of get_strehl): sub-pixel peak of a PSF
 * by two 1-D sinc fits through the maximum and its neighbours along x and along y; COMPASS's kernel is absent from
 * the reference tree -- restated from its name and docstring, UNPINNED (see aoref.c).  aoref_sinc_gain: fitted peak
 * over the sampled maximum for one axis; aoref_fit_max_2x1d_sinc: the fitted maximum of an image. */
float aoref_sinc_gain(float ym, float y0, float yp);
float aoref_fit_max_2x1d_sinc(const float *img, int nx, int ny);

/* ---- threading of the OpenMP loops above (bench.py's cpu_baseline times 1 and N threads):
 * set: threads of the following calls, returns the count in effect; max: what the host offers */
int aoref_set_threads(int n);
int aoref_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
