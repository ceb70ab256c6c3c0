"""A `sutraWrap` / `carmaWrap`-shaped module over the CPU ORACLE (oracle/aoref.c), batch size 1.

BUILD-CONTAINER TOOL (test infrastructure): lets the reference's *unmodified* Python
(`shesha.supervisor.rlSupervisor.RlSupervisor`, `src...ao_env.AoEnv`) run end to end so that
golden (state, reward, slopes, command) traces can be recorded (tools/gen_golden_trace.py).  It
implements the surface SURVEY.md Appendix B lists, each method forwarding to the oracle stage that
restates it.  2-D arrays are handed back in COMPASS's orientation (first index = x), i.e. the
transpose of this repo's [y][x] arrays.
"""
import ctypes as C
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import aoref  # noqa: E402

L = aoref.lib()
f32 = np.float32


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


class DevArray(object):
    """Stands for a carma device array: np.array(obj) copies it out; .reset() zeroes it."""

    def __init__(self, arr, transpose=False):
        self.a = arr
        self.t = transpose

    def __array__(self, dtype=None, copy=None):
        out = self.a.T.copy() if (self.t and self.a.ndim == 2) else self.a.copy()
        return out.astype(dtype) if dtype is not None else out

    def reset(self):
        self.a[...] = 0

    @property
    def shape(self):
        return self.a.T.shape if self.t else self.a.shape


class context(object):
    active_device = 0
    ndevice = 1

    @staticmethod
    def get_instance_1gpu(d):
        return context()

    @staticmethod
    def get_instance_ngpu(n, d):
        return context()

    def set_active_device(self, d):
        pass

    def set_active_device_force(self, d):
        pass


class Telescope(object):
    def __init__(self, ctx, n_pup, npos, pupil, n_mpup, mpupil):
        self.spupil, self.mpupil = _c(pupil), _c(mpupil)
        self.d_pupil, self.d_pupil_m = DevArray(self.spupil, True), DevArray(self.mpupil, True)


# ------------------------------------------------------------------------------ atmosphere
class _Screen(object):
    pass


class Atmos(object):
    def __init__(self, ctx, nscreens, r0, r0_layers, dim_screens, stencil_size, alt, windspeed,
                 winddir, deltax, deltay, dev):
        self.nscreens = int(nscreens)
        self.r0 = r0
        self.d_screens = []
        for i in range(self.nscreens):
            s = _Screen()
            s.dim = int(dim_screens[i])
            s.screen = np.zeros((s.dim, s.dim), dtype=f32)
            s.d_screen = DevArray(s.screen, True)
            s.deltax, s.deltay = f32(deltax[i]), f32(deltay[i])
            s.amplitude = f32(float(r0_layers[i])**(-5. / 6.) * 0.5 / (2 * np.pi))
            s.accumx, s.accumy, s.count, s.seed = f32(0), f32(0), 0, 1234 + i
            self.d_screens.append(s)

    def init_screen(self, i, A, B, istx, isty, seed):
        s = self.d_screens[i]
        s.A, s.B = _c(np.asarray(A)), _c(np.asarray(B))
        s.istx, s.isty = _c(istx, np.uint32), _c(isty, np.uint32)
        s.seed = int(seed)

    def _extrude(self, i, d):
        s = self.d_screens[i]
        ist = s.istx if abs(d) == 1 else s.isty
        eps = aoref.normals(s.seed, 0, s.count, s.dim)
        s.count += 1
        tmp = np.empty(ist.size + s.dim, dtype=f32)
        L.aoref_extrude(s.screen.reshape(-1), s.dim, s.A, ist.size, s.B, ist, d,
                        float(s.amplitude), eps, tmp)

    def set_seed(self, k, seed):
        self.d_screens[k].seed = int(seed)

    def refresh_screen(self, k):
        s = self.d_screens[k]
        s.screen[:] = 0
        s.accumx, s.accumy, s.count = f32(0), f32(0), 0
        d = 1 if s.deltax > 0 else -1
        for _ in range(2 * s.dim):
            self._extrude(k, d)

    def move_atmos(self):
        for i, s in enumerate(self.d_screens):
            s.accumx = f32(s.accumx + s.deltax)
            s.accumy = f32(s.accumy + s.deltay)
            kx, ky = int(s.accumx), int(s.accumy)
            for _ in range(abs(kx)):
                self._extrude(i, 1 if kx > 0 else -1)
            s.accumx = f32(s.accumx - f32(kx))
            for _ in range(abs(ky)):
                self._extrude(i, 2 if ky > 0 else -2)
            s.accumy = f32(s.accumy - f32(ky))


# ------------------------------------------------------------------------------ sources
class Source(object):
    def __init__(self, size, lam):
        self.size, self.Lambda = int(size), float(lam)
        self.phase = np.zeros((self.size, self.size), dtype=f32)
        self.d_phase = DevArray(self.phase, True)
        self.layers = []

    def add_layer(self, typ, idx, xoff, yoff):
        self.layers.append((str(typ), int(idx), float(xoff), float(yoff)))

    def remove_layer(self, typ, idx):
        self.layers = [l for l in self.layers if not (l[0] == str(typ) and l[1] == int(idx))]

    def raytrace(self, obj=None, rst=0, **kw):
        if rst:
            self.phase[:] = 0
        if obj is None or isinstance(obj, Telescope):
            return                          # NCPA / telescope aberrations: zero in every config
        n = self.size
        if isinstance(obj, Atmos):
            for (t, i, xo, yo) in self.layers:
                if t == "atmos":
                    s = obj.d_screens[i]
                    L.aoref_raytrace(self.phase.reshape(-1), n, n, s.screen.reshape(-1), s.dim,
                                     xo, yo, 1)
        elif isinstance(obj, Dms):
            for (t, i, xo, yo) in self.layers:
                if t in ("pzt", "tt"):
                    d = obj.d_dms[i]
                    L.aoref_raytrace(self.phase.reshape(-1), n, n, d.shape.reshape(-1), d.dim,
                                     xo, yo, 1)
        else:
            raise TypeError("raytrace through %r" % (obj,))


# ------------------------------------------------------------------------------ DMs
class Dm(object):
    def __init__(self, typ, alt, dim, ntotact, influsize, push4imat):
        self.type, self.alt, self.dim = str(typ), float(alt), int(dim)
        self.nactu, self.influsize, self.push4imat = int(ntotact), int(influsize), float(push4imat)
        self.shape = np.zeros((self.dim, self.dim), dtype=f32)
        self.d_shape = DevArray(self.shape, True)
        self.com = np.zeros(self.nactu, dtype=f32)
        self.d_com = DevArray(self.com)

    def pzt_loadarrays(self, influ, influpos, ninflu, influstart, i1, j1):
        self.influ = _c(np.asarray(influ).flatten("F"))
        self.influpos, self.ninflu = _c(influpos, np.int32), _c(ninflu, np.int32)
        self.influstart = _c(influstart, np.int32)

    def tt_loadarrays(self, influ):
        self.influ = _c(influ)

    def set_com(self, com, shape_dm=True):
        self.com[:] = np.asarray(com, dtype=f32).reshape(-1)
        if shape_dm and shape_dm is not None and not isinstance(shape_dm, (int, np.integer)) \
                or shape_dm is True:
            self.comp_shape()

    def comp_shape(self, com=None):
        c = self.com if com is None else _c(com)
        if self.type == "pzt":
            L.aoref_pzt_shape(self.shape.reshape(-1), self.dim, self.influ, self.influpos,
                              self.ninflu, self.influstart, self.influsize, c)
        else:
            L.aoref_tt_shape(self.shape.reshape(-1), self.dim, self.influ.reshape(-1), c)

    def comp_oneactu(self, i, ampli):
        c = np.zeros(self.nactu, dtype=f32)
        c[i] = ampli
        self.comp_shape(c)

    def reset_shape(self):
        self.shape[:] = 0
        self.com[:] = 0


class Dms(object):
    def __init__(self):
        self.d_dms = []

    def add_dm(self, ctx, typ, alt, dim, ntotact, influsize, ninflupos, n_npts, push4imat, nord,
               dev):
        self.d_dms.append(Dm(typ, alt, dim, ntotact, influsize, push4imat))

    def remove_dm(self, i):
        self._removed = i
        self.d_dms.pop(i)

    def insert_dm(self, ctx, typ, alt, dim, ntotact, influsize, ninflupos, n_npts, push4imat, nord,
                  dx, dy, theta, G, dev, idx):
        self.d_dms.insert(idx, Dm(typ, alt, dim, ntotact, influsize, push4imat))

    def set_full_com(self, com, shape_dm=True):
        o = 0
        for d in self.d_dms:
            d.set_com(com[o:o + d.nactu], shape_dm)
            o += d.nactu


# ------------------------------------------------------------------------------ WFS
class Wfs(object):
    def __init__(self, tel, nxsub, nvalid, npix, nphase, nrebin, nfft, ntot, pdiam, nphot):
        self.tel = tel
        self.nxsub, self.nvalid, self.npix, self.nphase = int(nxsub), int(nvalid), int(npix), \
            int(nphase)
        self.nrebin, self.nfft, self.ntot, self.subapd = int(nrebin), int(nfft), int(ntot), \
            float(pdiam)
        self.nphot = f32(nphot)
        self.noise, self.seed, self.frame = -1.0, 1234, 0
        self.bincube = np.zeros((self.nvalid, self.npix * self.npix), dtype=f32)
        self.slopes = np.zeros(2 * self.nvalid, dtype=f32)
        self.d_slopes = DevArray(self.slopes)
        dim = self.npix * self.nxsub
        self.binimg = np.zeros((dim, dim), dtype=f32)
        self.d_binimg = DevArray(self.binimg, True)
        self.d_camimg = self.d_binimg
        self.d_gs = None

    @property
    def d_bincube(self):
        # COMPASS: (npix, npix, nvalid), first index fastest = x
        cube = self.bincube.reshape(self.nvalid, self.npix, self.npix)      # [i][y][x]
        return DevArray(np.ascontiguousarray(cube.transpose(2, 1, 0)))      # [x][y][i]

    def load_arrays(self, phasemap, hrmap, binmap, halfxy, fluxPerSub, validsubsx, validsubsy,
                    validpuppixx, validpuppixy, ftkernel):
        self.phasemap, self.binmap = _c(phasemap, np.int32), _c(binmap, np.int32)
        self.halfxy, self.flux = _c(halfxy), _c(fluxPerSub)
        self.validsubsx, self.validsubsy = _c(validsubsx, np.int32), _c(validsubsy, np.int32)
        self.d_validsubsx, self.d_validsubsy = DevArray(self.validsubsx), DevArray(self.validsubsy)

    def set_noise(self, noise, seed):
        self.noise, self.seed, self.frame = float(noise), int(seed), 0

    def comp_image(self, noise=True):
        L.aoref_sh_image(self.d_gs.phase.reshape(-1), self.tel.mpupil.reshape(-1), self.nvalid,
                         self.nphase, self.nfft, self.npix, self.nrebin, self.phasemap,
                         self.halfxy.reshape(-1), self.binmap, self.flux, float(self.nphot),
                         self.d_gs.Lambda, self.bincube)
        if noise and self.noise >= 0:
            L.aoref_sh_noise(self.bincube, self.nvalid, self.npix * self.npix, self.noise,
                             self.seed & 0xFFFFFFFF, self.frame)
        self.frame += 1
        L.aoref_fill_binimg(self.bincube, self.nvalid, self.npix, self.validsubsx,
                            self.validsubsy, self.binimg.shape[0], self.binimg)

    def set_binimg(self, img, size):
        self.binimg[:] = np.asarray(img, dtype=f32).T
        for i in range(self.nvalid):
            x0, y0 = self.validsubsx[i], self.validsubsy[i]
            self.bincube[i] = self.binimg[y0:y0 + self.npix, x0:x0 + self.npix].reshape(-1)

    def slopes_geom(self, meth=0):
        L.aoref_slopes_geom(self.d_gs.phase.reshape(-1), self.tel.mpupil.reshape(-1),
                            self.d_gs.size, self.nvalid, self.nphase, self.phasemap, self.flux,
                            self.subapd, self.slopes)


class Sensors(object):
    def __init__(self, ctx, tel, t_wfs, nsensors, nxsub, nvalid, nPupils, npix, nphase, nrebin,
                 nfft, ntota, npup, pdiam, nphot, nphot4imat, lgs, fakecam, maxFlux, maxPix, dev,
                 roket):
        self.tel = tel
        self.d_wfs = [Wfs(tel, nxsub[i], nvalid[i], npix[i], nphase[i], nrebin[i], nfft[i],
                          ntota[i], pdiam[i], nphot[i]) for i in range(nsensors)]

    def initgs(self, xpos, ypos, Lambda, mag, zerop, size, noise, seed, G, thetaML, dx, dy):
        for i, w in enumerate(self.d_wfs):
            w.d_gs = Source(size[i], Lambda[i])
            w.noise, w.seed = float(noise[i]), int(seed[i])


# ------------------------------------------------------------------------------ target
class TargetSource(Source):
    HW = 8

    def __init__(self, tel, size, lam):
        Source.__init__(self, size, lam)
        self.tel = tel
        self.npsf = int(2**(int(np.floor(np.log2(2 * size))) + 1))
        self.ref = float(np.sum(tel.spupil, dtype=np.float64))**2
        self.reset_strehlmeter()

    def init_strehlmeter(self):
        self.reset_strehlmeter()

    def reset_strehlmeter(self):
        W = 2 * self.HW
        self.win = np.zeros((W, W), dtype=f32)
        self.le = np.zeros((W, W), dtype=np.float64)
        self.strehl_counter = 0
        self.strehl_se = self.strehl_le = 0.0
        self.phase_var = self.phase_var_avg = 0.0
        self.phase_var_count = 0
        self._pending = False

    def comp_image(self, puponly=0, compLE=True):
        pf, pw = C.c_float(0), C.c_float(0)
        L.aoref_psf(self.phase.reshape(-1), self.tel.spupil.reshape(-1), self.size, self.npsf,
                    self.Lambda, self.HW, None, self.win.ctypes.data_as(C.c_void_p), C.byref(pf),
                    C.byref(pw))
        self.peak_full, self.peak_win = pf.value, pw.value
        if compLE:
            self.le += self.win
            self.strehl_counter += 1
        self._new_image = True

    def comp_strehl(self, do_fit=True):
        # (do_fit: the PSF peak fitted by two 1-D sincs -- COMPASS's default, aoref_fit_max_2x1d_sinc)
        W = 2 * self.HW
        peak = lambda img: float(L.aoref_fit_max_2x1d_sinc(np.ascontiguousarray(img, dtype=f32).reshape(-1), W, W))  # noqa: E731
        self.strehl_se = (peak(self.win) if do_fit else self.peak_win) / self.ref
        if self.strehl_counter > 0:
            self.strehl_le = (peak(self.le) if do_fit else float(self.le.max())) / self.strehl_counter / self.ref
        if getattr(self, "_new_image", False):     # variance bookkeeping once per image
            self.phase_var = float(L.aoref_phase_var(self.phase.reshape(-1),
                                                     self.tel.spupil.reshape(-1), self.size))
            self.phase_var_avg += self.phase_var
            self.phase_var_count += 1
            self._new_image = False

    @property
    def d_image_se(self):
        return DevArray(self.win)

    @property
    def d_image_le(self):
        return DevArray(self.le.astype(f32))


class Target(object):
    def __init__(self, ctx, tel, n, xpos, ypos, Lambda, mag, zerop, sizes, Npts, dev):
        self.d_targets = [TargetSource(tel, sizes[i], Lambda[i]) for i in range(n)]


# ------------------------------------------------------------------------------ RTC
class Centroider(object):
    def __init__(self, nvalid, offset, scale, wfs):
        self.nvalid, self.offset, self.scale, self.wfs = int(nvalid), float(offset), float(scale), wfs
        self.nslopes = 2 * self.nvalid

    def load_validpos(self, x, y, n):
        pass

    def set_npix(self, n):
        self.npix = int(n)


class Controller(object):
    def __init__(self, rtc, nvalid, nslope, nactu, delay, typ, dms, ndm, nwfs):
        self.rtc, self.type = rtc, str(typ)
        self.nslope, self.nactu, self.delay = int(nslope), int(nactu), float(delay)
        self.dms, self.ndm, self.nwfs = dms, [int(k) for k in ndm], [int(k) for k in nwfs]
        self.gain = 0.0
        z = lambda n: np.zeros(n, dtype=f32)  # noqa: E731
        self.com, self.com1, self.com2, self.err, self.voltage = z(nactu), z(nactu), z(nactu), \
            z(nactu), z(nactu)
        self.centroids = z(nslope)
        self.imat = np.zeros((nslope, nactu), dtype=f32)
        self.cmat = np.zeros((nactu, nslope), dtype=f32)
        self.open_loop = 0
        for n in ("com", "err", "voltage", "centroids", "imat", "cmat"):
            setattr(self, "d_" + n, DevArray(getattr(self, n)))

    def set_gain(self, g):
        self.gain = float(g)

    def set_modal_gains(self, m):
        self.mgain = _c(m)

    def set_cmat(self, cmat):
        self.cmat[:] = np.asarray(cmat, dtype=f32)

    def set_imat(self, imat):
        self.imat[:] = np.asarray(imat, dtype=f32)

    def set_com(self, com, size=None):
        self.com[:] = np.asarray(com, dtype=f32)

    def set_open_loop(self, flag, reset=True):
        self.open_loop = int(flag)
        if flag and reset:
            for a in (self.com, self.com1, self.com2, self.err, self.voltage):
                a[:] = 0

    def init_proj_sparse(self, dms, indx_dm, unitpervolt, indx_pup, indx_mpup, roket=False):
        """rtc_init.py:418-448: influence functions of the controller's DMs on the pupil pixels.

        indx_pup: the lit pixels of the pupil grid (flat, first-index-fastest = this module's
        C-order [y][x]).  indx_dm is meant to hold, per DM, those pixels' indices inside the DM's
        own support -- but the reference builds block j from p_dms[j], the j-th mirror of the FILE,
        not from p_dms[ndm[j]], the j-th mirror of the CONTROLLER (rtc_init.py:431-436).  With the
        stock files (ndm = [1, 3]) the tip-tilt block is therefore computed for a stack array's
        258-pixel support and would address a 288-pixel tip-tilt array: meaningless pixels.  What
        COMPASS's native code does with it is not in the tree (unpinned).  Both facades take the
        geometry the indices are meant to encode instead: every DM support is centred on the pupil,
        so pupil pixel (x, y) is DM pixel (x + o, y + o), o = (dim - pupdiam) / 2 -- the offsets the
        reference itself gives the target (target_init.py:119-141).  Where the handed-in block is
        usable (the stack array's), it is checked to say the same."""
        import scipy.sparse as sp
        indx_dm = np.asarray(indx_dm).reshape(len(self.ndm), -1)
        pup = np.asarray(indx_pup, dtype=np.int64)
        cols = []
        for j, k in enumerate(self.ndm):
            d = dms.d_dms[k]
            npup = self.rtc_pupdiam(dms)
            o = (d.dim - npup) // 2
            py, px = pup // npup, pup % npup
            own = (py + o) * d.dim + (px + o)                 # this DM's pixels under the lit pupil
            given = indx_dm[j]
            if given.max() < d.dim * d.dim and j == 0:
                assert np.array_equal(np.sort(given), np.sort(own)), "indx_dm disagrees with the DM geometry"
            keep_com, keep_shape = d.com.copy(), d.shape.copy()
            for i in range(d.nactu):
                d.comp_oneactu(i, 1.0)
                cols.append(d.shape.reshape(-1)[own].astype(np.float64))
            d.com[:], d.shape[:] = keep_com, keep_shape
        self.geo_IF = sp.csc_matrix(np.stack(cols, axis=1))
        self.geo_pup = pup

    def rtc_pupdiam(self, dms):
        """Side of the pupil grid the GEO projection works on (the science target's grid)."""
        return int(self._pupdiam)

    def comp_dphi(self, source, is_wfs=False):
        """rtcCompass.py:545-546: the phase of `source` on the pupil pixels, kept for do_control."""
        self._dphi = source.phase.reshape(-1)[self.geo_pup].astype(np.float64)

    def svdec_imat(self):
        w = np.linalg.eigvalsh(self.imat.astype(np.float64).T @ self.imat.astype(np.float64))
        self.eigenvals = w[::-1].astype(f32)           # descending
        self.d_eigenvals = DevArray(self.eigenvals)

    def build_cmat(self, nfilt):
        D = self.imat.astype(np.float64)
        w, V = np.linalg.eigh(D.T @ D)
        inv = np.zeros_like(w)
        keep = np.argsort(w)[int(nfilt):]
        inv[keep] = 1.0 / w[keep]
        self.cmat[:] = ((V * inv[None, :]) @ V.T @ D.T).astype(f32)


class Rtc_FFF(object):
    def __init__(self):
        self.d_centro, self.d_control = [], []

    def add_centroider(self, ctx, nvalid, offset, scale, filter_TT, dev, typ, wfs=None):
        self.d_centro.append(Centroider(nvalid, offset, scale, wfs))

    def add_controller(self, ctx, nvalid, nslope, nactu, delay, dev, typ, dms=None, ndm=(), ndm_size=0,
                       nwfs=(), nwfs_size=0, Nphi=0, roket=False, nstates=0):
        self.d_control.append(Controller(self, nvalid, nslope, nactu, delay, typ, dms, ndm, nwfs))
        if self.d_centro:                     # pupil grid of the science path = the sensors' telescope
            self.d_control[-1]._pupdiam = self.d_centro[0].wfs.tel.spupil.shape[0]

    def _wfs_of(self, n):
        c = self.d_control[n]
        return [self.d_centro[k] for k in range(len(self.d_centro)) if k in c.nwfs]

    def do_centroids(self, n):
        c = self.d_control[n]
        o = 0
        for cen in self._wfs_of(n):
            w = cen.wfs
            # Rtc.do_centroids reads the camera image (d_binimg), e.g. after set_binimg
            out = np.zeros(cen.nslopes, dtype=f32)
            L.aoref_cog(w.bincube, w.nvalid, w.npix, cen.offset, cen.scale, out)
            c.centroids[o:o + cen.nslopes] = out
            o += cen.nslopes

    def do_control(self, n, *a, **k):
        c = self.d_control[n]
        if c.type == "geo":
            # sutra_controller_geo restated (ao_marl_amd.modal.geo_command, unpinned like every
            # native stage): least-squares projection of the pupil phase comp_dphi captured
            from ao_marl_amd import modal
            c.com[:] = modal.geo_command(c.geo_IF, c._dphi).astype(f32)
            return
        if c.open_loop:
            L.aoref_gemv(c.cmat, c.nactu, c.nslope, c.centroids, c.err)
            c.err[:] = -c.err
            return
        L.aoref_ls_control(c.cmat, c.nactu, c.nslope, c.centroids, c.gain, c.err, c.com)

    def apply_control(self, n, comp_voltage=True):
        c = self.d_control[n]
        if comp_voltage:
            a, b, cc = aoref.delay_weights(c.delay)
            c.voltage[:] = f32(a) * c.com + f32(b) * c.com1 + f32(cc) * c.com2
            c.com2[:] = c.com1
            c.com1[:] = c.com
        else:
            c.voltage[:] = c.com
        o = 0
        for k in c.ndm:
            d = c.dms.d_dms[k]
            d.com[:] = c.voltage[o:o + d.nactu]
            d.comp_shape()
            o += d.nactu

    def do_clipping(self, n):
        pass

    def do_imat(self, n, dms):
        c = self.d_control[n]
        col = 0
        for k in c.ndm:
            d = dms.d_dms[k]
            for i in range(d.nactu):
                res = []
                for sgn in (1.0, -1.0):
                    d.comp_oneactu(i, sgn * d.push4imat)
                    sl = []
                    for cen in self._wfs_of(n):
                        cen.wfs.d_gs.raytrace(dms, rst=1)
                        cen.wfs.comp_image(noise=False)
                        out = np.zeros(cen.nslopes, dtype=f32)
                        L.aoref_cog(cen.wfs.bincube, cen.wfs.nvalid, cen.wfs.npix, cen.offset,
                                    cen.scale, out)
                        sl.append(out)
                    res.append(np.concatenate(sl))
                c.imat[:, col] = (res[0] - res[1]) / f32(2 * d.push4imat)
                d.reset_shape()
                col += 1
        # the imat pass must not advance the WFS noise stream
        for cen in self._wfs_of(n):
            cen.wfs.frame = 0


def install():
    sw = types.ModuleType("sutraWrap")
    for name, cls in (("Dms", Dms), ("Rtc_FFF", Rtc_FFF), ("Sensors", Sensors), ("Atmos", Atmos),
                      ("Telescope", Telescope), ("Target", Target)):
        setattr(sw, name, cls)

    def _missing(n):
        def ctor(*a, **k):
            raise RuntimeError("%s is not provided by the oracle facade" % n)
        return ctor

    for n in ("Rtc_FHF", "Rtc_UFF", "Rtc_UHF", "Rtc_FFU", "Rtc_FHU", "Rtc_UFU", "Rtc_UHU",
              "Target_brahma", "Gamora", "Groot", "Rtc_brahma", "Rtc_cacao_FFF", "Rtc_cacao_UFF",
              "Rtc_cacao_FHF", "Rtc_cacao_UHF"):
        setattr(sw, n, type(n, (object,), {"__init__": _missing(n)}))
    cw = types.ModuleType("carmaWrap")
    cw.context = context
    sys.modules["sutraWrap"], sys.modules["carmaWrap"] = sw, cw
    return sw, cw
