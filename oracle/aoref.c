/* aoref.c -- CPU ORACLE (test infrastructure only; see aoref.h for scope, citations, parity pin).
 * Plain C99 + optional OpenMP.  Straightforward algorithms on purpose: full zero-padded radix-2
 * FFTs, explicit shifts, gather loops -- the HIP product uses different algorithms (pruned DFT on
 * MFMA, ring-buffer screens, fused raytrace) and must agree with this to fp32 round-off. */
#include "aoref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ Philox4x32-10 */
static inline void mulhilo(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo) {
  uint64_t p = (uint64_t)a * (uint64_t)b;
  *hi = (uint32_t)(p >> 32);
  *lo = (uint32_t)p;
}

void aoref_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint32_t hi0, lo0, hi1, lo1;
    mulhilo(0xD2511F53u, c0, &hi0, &lo0);
    mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

#define AOREF_KEY1 0x414F4D52u /* "AOMR" */

void aoref_uniforms(uint32_t seed, uint32_t stream, uint64_t counter, int n, float *out) {
  uint32_t key[2] = {seed, AOREF_KEY1};
  for (int b = 0; 4 * b < n; b++) {
    uint32_t ctr[4] = {(uint32_t)b, (uint32_t)counter, (uint32_t)(counter >> 32), stream}, x[4];
    aoref_philox4x32_10(ctr, key, x);
    for (int j = 0; j < 4 && 4 * b + j < n; j++) out[4 * b + j] = u01(x[j]);
  }
}

void aoref_normals(uint32_t seed, uint32_t stream, uint64_t counter, int n, float *out) {
  uint32_t key[2] = {seed, AOREF_KEY1};
  for (int b = 0; 4 * b < n; b++) {
    uint32_t ctr[4] = {(uint32_t)b, (uint32_t)counter, (uint32_t)(counter >> 32), stream}, x[4];
    aoref_philox4x32_10(ctr, key, x);
    float z[4];
    for (int h = 0; h < 2; h++) {
      float u0 = u01(x[2 * h]), u1 = u01(x[2 * h + 1]);
      float r = sqrtf(-2.0f * logf(u0));
      float a = 6.28318530717958647692f * u1;
      z[2 * h] = r * cosf(a);
      z[2 * h + 1] = r * sinf(a);
    }
    for (int j = 0; j < 4 && 4 * b + j < n; j++) out[4 * b + j] = z[j];
  }
}

/* ------------------------------------------------------------------ atmosphere */
void aoref_extrude(float *p, int n, const float *A, int ns, const float *B, const uint32_t *ist,
                   int dir, float amplitude, const float *eps, float *tmp) {
  /* tmp: ns + n floats */
  float *z = tmp, *nw = tmp + ns;
  size_t iref = (dir == 1 || dir == -2) ? (size_t)(n - 1) : (size_t)n * (n - 1);
  float zref = p[iref];
  for (int k = 0; k < ns; k++) z[k] = p[ist[k]] - zref;
#pragma omp parallel for schedule(static)
  for (int r = 0; r < n; r++) {
    float acc = 0.f;
    const float *a = A + (size_t)r * ns;
    for (int k = 0; k < ns; k++) acc += a[k] * z[k];
    float accb = 0.f;
    const float *b = B + (size_t)r * n;
    for (int k = 0; k < n; k++) accb += b[k] * eps[k];
    nw[r] = acc + amplitude * accb + zref;
  }
  if (dir == 1) {
    for (int y = 0; y < n; y++) {
      memmove(p + (size_t)y * n, p + (size_t)y * n + 1, (size_t)(n - 1) * sizeof(float));
      p[(size_t)y * n + n - 1] = nw[y];
    }
  } else if (dir == -1) {
    for (int y = 0; y < n; y++) {
      memmove(p + (size_t)y * n + 1, p + (size_t)y * n, (size_t)(n - 1) * sizeof(float));
      p[(size_t)y * n] = nw[n - 1 - y];
    }
  } else if (dir == 2) {
    memmove(p, p + n, (size_t)n * (n - 1) * sizeof(float));
    for (int x = 0; x < n; x++) p[(size_t)n * (n - 1) + x] = nw[x];
  } else { /* -2 */
    memmove(p + n, p, (size_t)n * (n - 1) * sizeof(float));
    for (int x = 0; x < n; x++) p[x] = nw[n - 1 - x];
  }
}

/* ------------------------------------------------------------------ raytrace */
void aoref_raytrace(float *out, int nx, int ny, const float *in, int N, float xoff, float yoff,
                    int accumulate) {
#pragma omp parallel for schedule(static)
  for (int y = 0; y < ny; y++) {
    float fy = (float)y + yoff;
    int iy = (int)floorf(fy);
    float wy = fy - (float)iy;
    for (int x = 0; x < nx; x++) {
      float fx = (float)x + xoff;
      int ix = (int)floorf(fx);
      float wx = fx - (float)ix;
      float v = 0.f;
      if (ix >= 0 && iy >= 0 && ix < N && iy < N) {
        int ix1 = ix + 1 < N ? ix + 1 : ix, iy1 = iy + 1 < N ? iy + 1 : iy;
        float v00 = in[(size_t)iy * N + ix], v01 = in[(size_t)iy * N + ix1];
        float v10 = in[(size_t)iy1 * N + ix], v11 = in[(size_t)iy1 * N + ix1];
        v = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
      }
      if (accumulate)
        out[(size_t)y * nx + x] += v;
      else
        out[(size_t)y * nx + x] = v;
    }
  }
}

/* ------------------------------------------------------------------ DMs */
void aoref_pzt_shape(float *shape, int dim, const float *influ, const int32_t *influpos,
                     const int32_t *ninflu, const int32_t *influstart, int ss, const float *com) {
  int ss2 = ss * ss;
#pragma omp parallel for schedule(static)
  for (int p = 0; p < dim * dim; p++) {
    float acc = 0.f;
    int s = influstart[p], c = ninflu[p];
    for (int k = 0; k < c; k++) {
      int pos = influpos[s + k];
      acc += influ[pos] * com[pos / ss2];
    }
    shape[p] = acc;
  }
}

void aoref_tt_shape(float *shape, int dim, const float *influ, const float *com) {
  /* influ: [y][x][2] (C order of the (dim, dim, 2) cube) */
  for (size_t p = 0; p < (size_t)dim * dim; p++)
    shape[p] = com[0] * influ[2 * p] + com[1] * influ[2 * p + 1];
}

/* ------------------------------------------------------------------ FFT (radix-2, in place) */
typedef struct { float re, im; } cpx;

static void fft1d(cpx *a, int n, int stride, const cpx *tw /* n/2 twiddles e^{-2 pi i k/n} */) {
  /* bit reversal */
  for (int i = 1, j = 0; i < n; i++) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { cpx t = a[(size_t)i * stride]; a[(size_t)i * stride] = a[(size_t)j * stride]; a[(size_t)j * stride] = t; }
  }
  for (int len = 2; len <= n; len <<= 1) {
    int half = len >> 1, step = n / len;
    for (int i = 0; i < n; i += len)
      for (int k = 0; k < half; k++) {
        cpx w = tw[k * step];
        cpx *u = &a[(size_t)(i + k) * stride], *v = &a[(size_t)(i + k + half) * stride];
        float tr = v->re * w.re - v->im * w.im, ti = v->re * w.im + v->im * w.re;
        v->re = u->re - tr; v->im = u->im - ti;
        u->re += tr; u->im += ti;
      }
  }
}

static cpx *make_twiddles(int n) {
  cpx *tw = (cpx *)malloc(sizeof(cpx) * (size_t)(n / 2 > 0 ? n / 2 : 1));
  for (int k = 0; k < n / 2; k++) {
    double a = -2.0 * M_PI * (double)k / (double)n;
    tw[k].re = (float)cos(a);
    tw[k].im = (float)sin(a);
  }
  return tw;
}

static void fft2d(cpx *a, int n, const cpx *tw) {
  for (int y = 0; y < n; y++) fft1d(a + (size_t)y * n, n, 1, tw);
  for (int x = 0; x < n; x++) fft1d(a + x, n, n, tw);
}

/* ------------------------------------------------------------------ Shack-Hartmann */
void aoref_sh_image(const float *phase, const float *mpupil, int nvalid, int pdiam, int nfft,
                    int npix, int nrebin, const int32_t *phasemap, const float *halfxy,
                    const int32_t *binmap, const float *flux, float nphot, float lambda_um,
                    float *bincube) {
  const float sc = (float)(2.0 * M_PI) / lambda_um;
  const int npix2 = npix * npix, nr2 = nrebin * nrebin, pd2 = pdiam * pdiam;
  cpx *tw = make_twiddles(nfft);
#pragma omp parallel
  {
    cpx *buf = (cpx *)malloc(sizeof(cpx) * (size_t)nfft * nfft);
    float *hr = (float *)malloc(sizeof(float) * (size_t)nfft * nfft);
#pragma omp for schedule(static)
    for (int i = 0; i < nvalid; i++) {
      memset(buf, 0, sizeof(cpx) * (size_t)nfft * nfft);
      for (int k = 0; k < pd2; k++) {
        int p = phasemap[(size_t)k * nvalid + i];
        float a = mpupil[p];
        float ang = phase[p] * sc - halfxy[k];
        int y = k / pdiam, x = k % pdiam;
        buf[(size_t)y * nfft + x].re = a * cosf(ang);
        buf[(size_t)y * nfft + x].im = a * sinf(ang);
      }
      fft2d(buf, nfft, tw);
      for (int q = 0; q < nfft * nfft; q++) hr[q] = buf[q].re * buf[q].re + buf[q].im * buf[q].im;
      float *out = bincube + (size_t)i * npix2;
      float tot = 0.f;
      for (int px = 0; px < npix2; px++) {
        float s = 0.f;
        for (int r = 0; r < nr2; r++) s += hr[binmap[(size_t)r * npix2 + px]];
        out[px] = s;
        tot += s;
      }
      float g = tot > 0.f ? nphot * flux[i] / tot : 0.f;
      for (int px = 0; px < npix2; px++) out[px] *= g;
    }
    free(buf);
    free(hr);
  }
  free(tw);
}

/* Poisson sample with mean lam from uniform/normal draws: inversion for lam < 30, rounded normal
 * approximation above (the product uses the identical rule). */
static float poisson_draw(float lam, float u, float zn) {
  if (!(lam > 0.f)) return 0.f;
  if (lam < 30.f) {
    float p = expf(-lam), c = p;
    int k = 0;
    while (u > c && k < 200) {
      k++;
      p *= lam / (float)k;
      c += p;
    }
    return (float)k;
  }
  float v = floorf(lam + sqrtf(lam) * zn + 0.5f);
  return v < 0.f ? 0.f : v;
}

/* Photon + read noise of one bincube.  Pixel idx of frame `frame` owns ONE Philox block
 * (counter {idx, frame, stream 4}): word 0 -> the uniform of the Poisson inversion, words 1, 2 -> a
 * Box-Muller pair: its cosine branch drives the rounded-normal Poisson for lam >= 30, its sine
 * branch is the read noise.  (COMPASS draws from cuRAND: parity with it is distributional.) */
void aoref_sh_noise(float *bincube, int nvalid, int npix2, float noise, uint32_t seed,
                    uint64_t frame) {
  if (noise < 0.f) return;
  int n = nvalid * npix2;
  uint32_t key[2] = {seed, AOREF_KEY1};
  for (int i = 0; i < n; i++) {
    uint32_t ctr[4] = {(uint32_t)i, (uint32_t)frame, (uint32_t)(frame >> 32), 4u}, x[4];
    aoref_philox4x32_10(ctr, key, x);
    float u = u01(x[0]);
    float r = sqrtf(-2.0f * logf(u01(x[1])));
    float a = 6.28318530717958647692f * u01(x[2]);
    float v = poisson_draw(bincube[i], u, r * cosf(a));
    if (noise > 0.f) v += noise * (r * sinf(a));
    bincube[i] = v;
  }
}

void aoref_cog(const float *bincube, int nvalid, int npix, float offset, float scale,
               float *slopes) {
  for (int i = 0; i < nvalid; i++) {
    const float *im = bincube + (size_t)i * npix * npix;
    float s = 0.f, sx = 0.f, sy = 0.f;
    for (int y = 0; y < npix; y++)
      for (int x = 0; x < npix; x++) {
        float v = im[y * npix + x];
        s += v;
        sx += v * (float)x;
        sy += v * (float)y;
      }
    if (s != 0.f) {
      slopes[i] = (sx / s - offset) * scale;
      slopes[nvalid + i] = (sy / s - offset) * scale;
    } else {
      slopes[i] = 0.f;
      slopes[nvalid + i] = 0.f;
    }
  }
}

void aoref_fill_binimg(const float *bincube, int nvalid, int npix, const int32_t *validx,
                       const int32_t *validy, int imgdim, float *binimg) {
  for (int i = 0; i < nvalid; i++)
    for (int y = 0; y < npix; y++)
      for (int x = 0; x < npix; x++)
        binimg[(size_t)(validy[i] + y) * imgdim + validx[i] + x] =
            bincube[(size_t)i * npix * npix + y * npix + x];
}

void aoref_slopes_geom(const float *phase, const float *mpupil, int n, int nvalid, int pdiam,
                       const int32_t *phasemap, const float *flux, float subapd, float *slopes) {
  (void)n;
  const float alpha = 0.206265f / subapd;
  for (int i = 0; i < nvalid; i++) {
    float gx = 0.f, gy = 0.f;
    for (int y = 0; y < pdiam; y++)
      for (int x = 0; x < pdiam; x++) {
        int k = y * pdiam + x;
        int p = phasemap[(size_t)k * nvalid + i];
        float m = mpupil[p];
        int xm = x > 0 ? x - 1 : x, xp = x < pdiam - 1 ? x + 1 : x;
        int ym = y > 0 ? y - 1 : y, yp = y < pdiam - 1 ? y + 1 : y;
        float dx = (phase[phasemap[(size_t)(y * pdiam + xp) * nvalid + i]] -
                    phase[phasemap[(size_t)(y * pdiam + xm) * nvalid + i]]) / (float)(xp - xm);
        float dy = (phase[phasemap[(size_t)(yp * pdiam + x) * nvalid + i]] -
                    phase[phasemap[(size_t)(ym * pdiam + x) * nvalid + i]]) / (float)(yp - ym);
        gx += m * dx;
        gy += m * dy;
      }
    float den = (float)pdiam * flux[i];
    slopes[i] = alpha * gx / den;
    slopes[nvalid + i] = alpha * gy / den;
  }
}

/* ------------------------------------------------------------------ controller */
void aoref_gemv(const float *M, int rows, int cols, const float *x, float *y) {
#pragma omp parallel for schedule(static)
  for (int r = 0; r < rows; r++) {
    const float *m = M + (size_t)r * cols;
    float acc = 0.f;
    for (int c = 0; c < cols; c++) acc += m[c] * x[c];
    y[r] = acc;
  }
}

void aoref_ls_control(const float *cmat, int nactu, int nslope, const float *slopes, float gain,
                      float *err, float *com) {
  aoref_gemv(cmat, nactu, nslope, slopes, err);
  for (int a = 0; a < nactu; a++) {
    err[a] = -err[a];           /* err = -cmat . s  (guardians/roket.py:169) */
    com[a] += gain * err[a];    /* integrator, modal gains = 1 (rtc_init.py:507-513) */
  }
}

/* ------------------------------------------------------------------ target */
void aoref_psf(const float *phase, const float *pupil, int n, int nfft, float lambda_um, int hw,
               float *psf_full, float *psf_win, float *peak_full, float *peak_win) {
  const float sc = (float)(2.0 * M_PI) / lambda_um;
  cpx *buf = (cpx *)calloc((size_t)nfft * nfft, sizeof(cpx));
  cpx *tw = make_twiddles(nfft);
  for (int y = 0; y < n; y++)
    for (int x = 0; x < n; x++) {
      float a = pupil[(size_t)y * n + x];
      if (a != 0.f) {
        float ang = phase[(size_t)y * n + x] * sc;
        buf[(size_t)y * nfft + x].re = a * cosf(ang);
        buf[(size_t)y * nfft + x].im = a * sinf(ang);
      }
    }
  /* rows (only the first n are non-zero), then columns */
#pragma omp parallel for schedule(static)
  for (int y = 0; y < n; y++) fft1d(buf + (size_t)y * nfft, nfft, 1, tw);
#pragma omp parallel for schedule(static)
  for (int x = 0; x < nfft; x++) fft1d(buf + x, nfft, nfft, tw);
  float pf = 0.f, pw = 0.f;
  for (size_t q = 0; q < (size_t)nfft * nfft; q++) {
    float v = buf[q].re * buf[q].re + buf[q].im * buf[q].im;
    if (psf_full) psf_full[q] = v;
    if (v > pf) pf = v;
  }
  for (int j = 0; j < 2 * hw; j++)
    for (int i = 0; i < 2 * hw; i++) {
      int ky = (j - hw + nfft) % nfft, kx = (i - hw + nfft) % nfft;
      cpx c = buf[(size_t)ky * nfft + kx];
      float v = c.re * c.re + c.im * c.im;
      if (psf_win) psf_win[j * 2 * hw + i] = v;
      if (v > pw) pw = v;
    }
  *peak_full = pf;
  *peak_win = pw;
  free(buf);
  free(tw);
}

/* Target.comp_strehl(do_fit = True) -- the default of every get_strehl call in the reference
 * (shesha/supervisor/components/targetCompass.py:139-159; ao_env.py:592, train_rpc.py:463-610): "fit the PSF with a
 * sinc before computing SR".  COMPASS's kernel (sutra: fit_max_2x1dSinc) is not in the reference tree: UNPINNED.
 * Restated as what the name and the docstring say: two 1-D fits, along x and along y through the maximum, of
 *      y(x) = A sinc(w (x - x0)),     sinc(t) = sin(t) / t,
 * each through the three samples (max - 1, max, max + 1); the fitted peak is  max * gx * gy,  g = A / y(0) =
 * 1 / sinc(w x0) >= 1.  (w, x0) by Newton from the parabola through the three points (the sinc's second-order
 * expansion); a fit that leaves 0 < w < 3, |x0| <= 0.6, 1 <= g < 1.5 falls back to the parabola's peak.  A maximum
 * on the border of the image has no neighbours: no fit. */
static double sincd(double t) { return fabs(t) < 1e-4 ? 1.0 - t * t / 6.0 : sin(t) / t; }
static double dsincd(double t) { return fabs(t) < 1e-4 ? -t / 3.0 : (cos(t) - sin(t) / t) / t; }
float aoref_sinc_gain(float ym, float y0, float yp) {
  if (!(y0 > 0.f)) return 1.f;
  const double rm = ym / y0, rp = yp / y0;
  const double a = 0.5 * (rm + rp) - 1.0, b = 0.5 * (rp - rm);
  if (!(a < -1e-6)) return 1.f;
  double x0 = -b / (2.0 * a);
  if (x0 > 0.5) x0 = 0.5;
  if (x0 < -0.5) x0 = -0.5;
  const double gpar = 1.0 - b * b / (4.0 * a);
  double w = sqrt(-6.0 * a / gpar);
  if (!(w > 1e-3)) w = 1e-3;
  if (w > 3.0) w = 3.0;
  for (int it = 0; it < 8; it++) {
    const double f0 = sincd(w * x0), d0 = dsincd(w * x0);
    const double fm = sincd(w * (1.0 + x0)), dm = dsincd(w * (1.0 + x0));
    const double fp = sincd(w * (1.0 - x0)), dp = dsincd(w * (1.0 - x0));
    const double F1 = fm - rm * f0, F2 = fp - rp * f0;
    const double J11 = (1.0 + x0) * dm - rm * x0 * d0, J12 = w * dm - rm * w * d0;
    const double J21 = (1.0 - x0) * dp - rp * x0 * d0, J22 = -w * dp - rp * w * d0;
    const double det = J11 * J22 - J12 * J21;
    if (!(fabs(det) > 1e-12)) break;
    w -= (F1 * J22 - F2 * J12) / det;
    x0 -= (J11 * F2 - J21 * F1) / det;
    if (!(w > 1e-3)) w = 1e-3;
    if (w > 3.0) w = 3.0;
    if (x0 > 0.6) x0 = 0.6;
    if (x0 < -0.6) x0 = -0.6;
  }
  const double g = 1.0 / sincd(w * x0);
  if (g >= 1.0 && g < 1.5) return (float)g;
  return (float)((gpar >= 1.0 && gpar < 1.5) ? gpar : 1.0);
}
float aoref_fit_max_2x1d_sinc(const float *img, int nx, int ny) {
  int ax = 0, ay = 0;
  float m = img[0];
  for (int y = 0; y < ny; y++)
    for (int x = 0; x < nx; x++)
      if (img[(size_t)y * nx + x] > m) { m = img[(size_t)y * nx + x]; ax = x; ay = y; }
  if (ax == 0 || ay == 0 || ax == nx - 1 || ay == ny - 1) return m;
  const float gx = aoref_sinc_gain(img[(size_t)ay * nx + ax - 1], m, img[(size_t)ay * nx + ax + 1]);
  const float gy = aoref_sinc_gain(img[(size_t)(ay - 1) * nx + ax], m, img[(size_t)(ay + 1) * nx + ax]);
  return m * gx * gy;
}

float aoref_phase_var(const float *phase, const float *pupil, int n) {
  double s = 0., c = 0.;
  for (size_t p = 0; p < (size_t)n * n; p++)
    if (pupil[p] > 0.f) { s += phase[p]; c += 1.; }
  if (c == 0.) return 0.f;
  double m = s / c, v = 0.;
  for (size_t p = 0; p < (size_t)n * n; p++)
    if (pupil[p] > 0.f) { double d = phase[p] - m; v += d * d; }
  return (float)(v / c);
}

/* ------------------------------------------------------------------ threading */
#ifdef _OPENMP
#include <omp.h>
int aoref_set_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  int got = 1;
#pragma omp parallel
  {
#pragma omp single
    got = omp_get_num_threads();
  }
  return got;
}
int aoref_max_threads(void) { return omp_get_num_procs(); }
#else
int aoref_set_threads(int n) { (void)n; return 1; }
int aoref_max_threads(void) { return 1; }
#endif
