// aomarl_capi_extras.hip -- part of the C ABI implementation (included by aomarl_capi.hip, one translation unit):
// denoiser, geometric controller, full-frame PSF on demand.
// (the WFS-image denoiser, A17, is a translation unit of its own: aomarl_denoise.hip)

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
  rc = aomarl_gemm_nt_batched(n, gw, pd, pd, c->geoUx, pd, 0, phi, pd, (long long)pd * pd, nullptr, 0, T, pd,
                              (long long)gw * pd, 0, stream);
  if (rc) return rc;
  // LAT[e][j][i] = sum_y T[e][j][y] UyT[i][y]
  rc = aomarl_gemm_nt_batched(n, gw, gh, pd, T, pd, (long long)gw * pd, c->geoUy, pd, 0, nullptr, 0, LAT, gh,
                              (long long)gw * gh, 0, stream);
  if (rc) return rc;
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
