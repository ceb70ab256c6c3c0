"""`sutraWrap` / `carmaWrap`-shaped facade over libaomarl_hip.so: the HIP library behind the
reference's OWN native API (shesha/sutra_wrap.py:46-72; surface and call sites: SURVEY.md
Appendix B), batch size 1, NumPy in / NumPy out.

    from ao_marl_amd import sutra_facade
    sutra_facade.install()           # sys.modules["sutraWrap"], ["carmaWrap"] -> this module's classes

after which the reference's unmodified `shesha.init.*` / `shesha.supervisor.*` construct and drive
these objects exactly as they drive COMPASS's.  Every per-frame method forwards to one entry point
of include/aomarl.h through ao_marl_amd.sim.HipSim (env_count = 1); nothing is computed on the
host except what the reference's native library also does once at init on small matrices
(eigen-decomposition of the interaction matrix, the GEO projector).  There is no CPU fallback.

How the object model maps.  COMPASS's objects are built one by one and refer to each other; the C
ABI wants one static description per control path (aomarl_create).  A control path here is an
"engine": WFS i, the DMs its guide star sees (Source.add_layer, in that order: stack array, then
tip-tilt), target i -- the indices the reference's supervisor pairs up (rlSupervisor.py:954-1013:
controller n uses WFS n and target n).  Engines are (re)built lazily from whatever the facade
objects hold when a call first needs the device, and again when the configuration changes (DM
re-inserted by correct_dm, controller added).  All engines share ONE atmosphere: engine k > 0
aliases engine 0's screen buffers (like the library's GEO twin).  Arrays come back in COMPASS's
orientation (first index = x), the transpose of this repository's [y][x] layout.

Not provided (raise): sensors whose sampling the spot kernel is not specialised for can be
ray-traced and give geometric slopes but no images (the reference never images its second,
128-point WFS); LGS, pyramid, KL DMs, ROKET, the cacao / brahma variants.  `comp_strehl(do_fit)`: the
two-1-D-sinc fit of the PSF peak on the Strehl window (k_strehl_commit; COMPASS's kernel is not in the reference
tree: restated).
"""
import ctypes as C
import sys
import types

import numpy as np

from . import libaomarl as la
from . import modal, system

f32 = np.float32
HW = 8                                   # PSF window half-width of the strehl meter


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


class DevArray(object):
    """Stands for a carma device array: np.array(obj) copies it out through `get`; .reset() zeroes
    it through `zero`."""

    def __init__(self, get, zero=None):
        self._get, self._zero = get, zero

    def __array__(self, dtype=None, copy=None):
        out = np.array(self._get())
        return out.astype(dtype) if dtype is not None else out

    def reset(self):
        if self._zero is None:
            raise NotImplementedError("reset of this device array")
        self._zero()

    @property
    def shape(self):
        return np.array(self._get()).shape


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


# one simulation per process, like carmaWrap's context singleton
_HUB = {"atmos": None, "sensors": None, "target": None, "dms": [], "rtc": None, "engines": None,
        "device": "cuda:0", "sealed": False}


def _engines():
    if _HUB["engines"] is None:
        _build_engines()
    return _HUB["engines"]


def _invalidate(reason):
    if _HUB["engines"] is not None:
        if _HUB["sealed"]:
            raise NotImplementedError("the configuration changed after the first reset (%s): the "
                                      "device state would be lost" % reason)
        import torch
        torch.cuda.synchronize()
        _HUB["engines"] = None


class _DmArrays(object):
    """What libaomarl.make_desc reads from a DM."""
    pass


class _Engine(object):
    """WFS i + the DMs it sees + target i as one aomarl_ctx / HipSim(nenv = 1)."""

    def __init__(self, i, first):
        from .sim import HipSim
        self.i = i
        sens, tgt, atm = _HUB["sensors"], _HUB["target"], _HUB["atmos"]
        w = sens.d_wfs[i]
        t = tgt.d_targets[i] if tgt is not None and i < len(tgt.d_targets) else None
        tel = sens.tel
        s = system.SimArrays()
        s.name = "sutra_facade.engine%d" % i
        s.n, s.pupdiam = int(tel.mpupil.shape[0]), int(tel.spupil.shape[0])
        s.mpupil, s.spupil = tel.mpupil, tel.spupil
        # sampling the spot kernel is specialised for; otherwise borrow a sibling's image-formation
        # arrays (same sub-aperture geometry) and refuse to form images
        self.image_ok = (w.nphase, w.nfft, w.nrebin, w.npix) == (16, 64, 2, 16)
        wi = w
        if not self.image_ok:
            sib = [x for x in sens.d_wfs if (x.nphase, x.nfft, x.nrebin, x.npix) == (16, 64, 2, 16)
                   and x.nvalid == w.nvalid]
            if not sib:
                raise la.AomarlError("WFS %d: sampling %r is not supported by the spot kernel" %
                                     (i, (w.nphase, w.nfft, w.nrebin, w.npix)))
            wi = sib[0]
        s.nvalid, s.pdiam, s.nfft, s.npix, s.nrebin, s.nxsub = (w.nvalid, wi.nphase, wi.nfft, wi.npix,
                                                               wi.nrebin, w.nxsub)
        s.phasemap, s.halfxy, s.binmap = w.phasemap, wi.halfxy, wi.binmap
        s.flux, s.nphot = w.flux, f32(w.nphot)
        s.wfs_lambda, s.noise = float(w.d_gs.Lambda), float(w.noise)
        s.validsubsx, s.validsubsy = w.validsubsx, w.validsubsy
        s.subapd = float(w.subapd)
        cen = [c for c in (_HUB["rtc"].d_centro if _HUB["rtc"] is not None else []) if c.wfs is w]
        s.cog_offset = float(cen[0].offset) if cen else float(w.npix // 2. - 0.5)
        s.cog_scale = float(cen[0].scale) if cen else 1.0
        s.nslope = 2 * w.nvalid
        # atmosphere
        s.nscreens = atm.nscreens
        s.screen_dim = [sc.dim for sc in atm.d_screens]
        s.deltax = np.asarray([sc.deltax for sc in atm.d_screens], dtype=np.float32)
        s.deltay = np.asarray([sc.deltay for sc in atm.d_screens], dtype=np.float32)
        s.amplitude = np.asarray([sc.amplitude for sc in atm.d_screens], dtype=np.float32)
        s.A = [sc.A for sc in atm.d_screens]
        s.B = [sc.B for sc in atm.d_screens]
        s.istx = [sc.istx for sc in atm.d_screens]
        s.isty = [sc.isty for sc in atm.d_screens]

        def layer_offsets(src, kinds):
            return {(k, idx): (xo, yo) for (k, idx, xo, yo) in src.layers if k in kinds}
        wat, tat = layer_offsets(w.d_gs, ("atmos",)), layer_offsets(t, ("atmos",)) if t is not None else {}
        s.wfs_atm_off = [wat.get(("atmos", l), (0.0, 0.0)) for l in range(s.nscreens)]
        s.tar_atm_off = [tat.get(("atmos", l), s.wfs_atm_off[l]) for l in range(s.nscreens)]
        # DMs: the layers of the guide star, stack arrays first, tip-tilt last
        seen = [(k, idx, xo, yo) for (k, idx, xo, yo) in w.d_gs.layers if k in ("pzt", "tt")]
        seen.sort(key=lambda l: 0 if l[0] == "pzt" else 1)
        tdm = layer_offsets(t, ("pzt", "tt")) if t is not None else {}
        self.dm_index = [idx for (_, idx, _, _) in seen]
        s.dms, s.wfs_dm_off, s.tar_dm_off = [], [], []
        for (k, idx, xo, yo) in seen:
            d = _HUB["dms"].d_dms[idx]
            m = _DmArrays()
            m.type, m.dim, m.ntotact, m.influsize = d.type, d.dim, d.nactu, d.influsize
            m.influ = d.influ
            if d.type == "pzt":
                m.influpos, m.ninflu, m.influstart = d.influpos, d.ninflu, d.influstart
            s.dms.append(m)
            s.wfs_dm_off.append((xo, yo))
            s.tar_dm_off.append(tdm.get((k, idx), (xo, yo)))
        s.nactu = int(sum(m.ntotact for m in s.dms))
        s.tar_lambda = float(t.Lambda) if t is not None else 1.65
        s.npsf = system.psf_fft_size(s.pupdiam)
        s.strehl_halfwin = HW
        # controller attached to this WFS (delay / gain / command matrix)
        self.ctl = None
        if _HUB["rtc"] is not None:
            for c in _HUB["rtc"].d_control:
                if c.nwfs and c.nwfs[0] == i:
                    self.ctl = c
        s.delay = float(self.ctl.delay) if self.ctl is not None else 0.0
        s.gain = float(self.ctl.gain) if self.ctl is not None else 0.0
        s.cmat = None
        self.s = s
        self.sim = HipSim(s, nenv=1, device=_HUB["device"], keep_bincube=True, keep_phase=True)
        self.sim.defer_shape = False              # every stage is called on its own: shapes in memory
        if first is not None:                     # one atmosphere: alias engine 0's screens
            for k in ("screens", "origin", "seeds", "ext_count"):
                self.sim.t[k] = first.sim.t[k]
                setattr(self.sim.st, k, first.sim.t[k].data_ptr())
            self.sim.accumx, self.sim.accumy = first.sim.accumx, first.sim.accumy
        if self.ctl is not None and self.ctl.type == "ls":
            self.sim.set_cmat(self.ctl.cmat)
        self.geo_ready = False
        self.volts = np.zeros(s.nactu, dtype=np.float32)     # what the DM shapes were last built from

    # ---- helpers
    def dm_slot(self, idx):
        k = self.dm_index.index(idx)
        a = sum(m.ntotact for m in self.s.dms[:k])
        return k, a, a + self.s.dms[k].ntotact

    def shape_from(self, volts):
        self.volts[:] = volts
        self.sim.comp_dm_shape(self.volts[None, :])

    def lib_call(self, fn, *extra):
        sim = self.sim
        la.check(fn(sim.ctx, C.byref(sim.st), 0, 1, *extra, sim._stream()))


def _build_engines():
    sens = _HUB["sensors"]
    if sens is None or _HUB["atmos"] is None or _HUB["dms"] is None:
        raise la.AomarlError("sutra_facade: the simulation is not assembled yet (need Atmos, Dms, "
                             "Sensors before the first device call)")
    engines = []
    for i in range(len(sens.d_wfs)):
        engines.append(_Engine(i, engines[0] if engines else None))
    _HUB["engines"] = engines
    # the DMs' current commands survive a rebuild (correct_dm re-inserts mirrors at rest anyway)
    for e in engines:
        v = np.zeros(e.s.nactu, dtype=np.float32)
        for idx in e.dm_index:
            k, a, b = e.dm_slot(idx)
            v[a:b] = _HUB["dms"].d_dms[idx].com
        if np.any(v):
            e.shape_from(v)


def _engine_of_dm(idx):
    for e in _engines():
        if idx in e.dm_index:
            return e
    raise la.AomarlError("DM %d is seen by no WFS: it belongs to no control path" % idx)


# ------------------------------------------------------------------------------ telescope
class Telescope(object):
    def __init__(self, ctx, n_pup, npos, pupil, n_mpup, mpupil):
        self.spupil, self.mpupil = _c(pupil), _c(mpupil)
        self.d_pupil = DevArray(lambda: self.spupil.T.copy())
        self.d_pupil_m = DevArray(lambda: self.mpupil.T.copy())


# ------------------------------------------------------------------------------ atmosphere
class _Screen(object):
    """sutra's Tscreen as atmosCompass.py uses it at run time (:103-135): set_deltax / set_deltay, d_istencilx /
    d_istencily (np.array(...) of them) and set_istencilx / set_istencily -- over aomarl_set_wind (the deltas alone)
    and aomarl_set_stencil of the engine that moves the atmosphere."""

    def _push(self, fn):
        if _HUB["engines"]:                       # built: the library holds these values (else they are read at the build)
            fn(_HUB["engines"][0].sim)

    def set_deltax(self, v):
        self.deltax = f32(v)
        self._push(lambda sim: sim.set_wind(self.index, self.deltax, self.deltay, mirror_stencils=False))

    def set_deltay(self, v):
        self.deltay = f32(v)
        self._push(lambda sim: sim.set_wind(self.index, self.deltax, self.deltay, mirror_stencils=False))

    def set_istencilx(self, ist):
        self.istx = _c(ist, np.uint32)
        self._push(lambda sim: sim.set_stencil(self.index, 0, self.istx))

    def set_istencily(self, ist):
        self.isty = _c(ist, np.uint32)
        self._push(lambda sim: sim.set_stencil(self.index, 1, self.isty))


class Atmos(object):
    def __init__(self, ctx, nscreens, r0, r0_layers, dim_screens, stencil_size, alt, windspeed,
                 winddir, deltax, deltay, dev):
        self.nscreens = int(nscreens)
        self.r0 = r0
        self.d_screens = []
        for i in range(self.nscreens):
            s = _Screen()
            s.index = i
            s.d_istencilx = DevArray(lambda s=s: s.istx.astype(np.int64))
            s.d_istencily = DevArray(lambda s=s: s.isty.astype(np.int64))
            s.dim = int(dim_screens[i])
            s.deltax, s.deltay = f32(deltax[i]), f32(deltay[i])
            s.amplitude = f32(float(r0_layers[i])**(-5. / 6.) * 0.5 / (2 * np.pi))
            s.seed = 1234 + i
            s.d_screen = DevArray(lambda i=i: _engines()[0].sim.screen(i)[0].cpu().numpy().T.copy())
            self.d_screens.append(s)
        _HUB["atmos"] = self
        _invalidate("new Atmos")

    def init_screen(self, i, A, B, istx, isty, seed):
        s = self.d_screens[i]
        s.A, s.B = _c(np.asarray(A)), _c(np.asarray(B))
        s.istx, s.isty = _c(istx, np.uint32), _c(isty, np.uint32)
        s.seed = int(seed)
        _invalidate("init_screen")

    def set_seed(self, k, seed):
        self.d_screens[k].seed = int(seed)

    def set_r0(self, r0):
        """sutra's Atmos.set_r0 (atmosCompass.py:93): every layer's noise amplitude follows r0^(-5/6); the screens as
        they stand are kept."""
        scale = (float(r0) / float(self.r0))**(-5. / 6.)
        for sc in self.d_screens:
            sc.amplitude = f32(float(sc.amplitude) * scale)
        self.r0 = float(r0)
        if _HUB["engines"]:
            _HUB["engines"][0].sim.set_amplitudes([sc.amplitude for sc in self.d_screens])

    def _base_seed(self):
        base = {sc.seed - k for k, sc in enumerate(self.d_screens)}
        if len(base) != 1:
            raise NotImplementedError("layer seeds must be base + layer index (Atmos.set_seed(k, seed + k), "
                                      "atmosCompass.py:141-145): the library derives them from one seed")
        return base.pop() & 0xFFFFFFFF

    def refresh_screen(self, k):
        """Zero layer k, reseed it, 2 * dim extrusions along x (atmosCompass.py:139-145)."""
        import torch
        sim = _engines()[0].sim
        s = self.d_screens[k]
        base = self._base_seed()
        sim.t["seeds"].fill_(base if base < 2**31 else base - 2**32)      # uint32 bits in an int32 tensor
        sim.set_screen(k, torch.zeros(1, s.dim, s.dim, dtype=torch.float32, device=sim.device))
        sim.t["ext_count"][:, k] = 0
        sim.accumx[:, k] = 0
        sim.accumy[:, k] = 0
        d = 1 if s.deltax > 0 else -1
        for _ in range(2 * s.dim):
            sim.extrude([k], [d])
        _HUB["sealed"] = True

    def move_atmos(self):
        _engines()[0].sim.move_atmos()


# ------------------------------------------------------------------------------ sources
class Source(object):
    """A guide star / science source: the phase buffer of its engine (wfs_phase / tar_phase)."""

    def __init__(self, owner_index, size, lam, target):
        self.i, self.size, self.Lambda, self.is_target = owner_index, int(size), float(lam), target
        self.layers = []
        self.d_phase = DevArray(self._read, self._zero)

    def _sim(self):
        return _engines()[self.i].sim

    def _buf(self):
        sim = self._sim()
        sim._need_phase()
        return sim.t["tar_phase" if self.is_target else "wfs_phase"]

    def _read(self):
        return self._buf()[0].cpu().numpy().T.copy()

    def _zero(self):
        self._trace(False, False, True)

    def _trace(self, atm, dms, reset):
        sim = self._sim()
        (sim.raytrace_target if self.is_target else sim.raytrace_wfs)(atm=atm, dms=dms, reset=reset)

    def add_layer(self, typ, idx, xoff, yoff):
        self.layers.append((str(typ), int(idx), float(xoff), float(yoff)))
        _invalidate("add_layer")

    def remove_layer(self, typ, idx):
        self.layers = [l for l in self.layers if not (l[0] == str(typ) and l[1] == int(idx))]
        _invalidate("remove_layer")

    def raytrace(self, obj=None, rst=0, **kw):
        """Source.raytrace(atmos | tel | dms | nothing) (sourceCompass.py:76-85)."""
        if obj is None or isinstance(obj, Telescope):
            if rst:
                self._zero()
            return                          # NCPA / telescope aberrations: zero in every config
        if isinstance(obj, Atmos):
            self._trace(True, False, bool(rst))
        elif isinstance(obj, Dms):
            self._trace(False, True, bool(rst))
        else:
            raise TypeError("raytrace through %r" % (obj,))


# ------------------------------------------------------------------------------ DMs
class Dm(object):
    def __init__(self, typ, alt, dim, ntotact, influsize, push4imat):
        self.type, self.alt, self.dim = str(typ), float(alt), int(dim)
        self.nactu, self.influsize, self.push4imat = int(ntotact), int(influsize), float(push4imat)
        self.com = np.zeros(self.nactu, dtype=f32)
        self.d_com = DevArray(lambda: self._pull_com())
        self.d_shape = DevArray(lambda: self._shape())
        self.influ = None

    def _index(self):
        return _HUB["dms"].d_dms.index(self)

    def _pull_com(self):
        return self.com.copy()

    def _shape(self):
        e = _engine_of_dm(self._index())
        k, _, _ = e.dm_slot(self._index())
        return e.sim.dm_shape(k)[0].cpu().numpy().T.copy()

    def pzt_loadarrays(self, influ, influpos, ninflu, influstart, i1, j1):
        self.influ = _c(np.asarray(influ))                       # (ss, ss, nact), first index = x
        self.influpos, self.ninflu = _c(influpos, np.int32), _c(ninflu, np.int32)
        self.influstart = _c(influstart, np.int32)
        _invalidate("pzt_loadarrays")

    def tt_loadarrays(self, influ):
        self.influ = _c(influ)                                   # (dim, dim, 2)
        _invalidate("tt_loadarrays")

    def _apply(self):
        idx = self._index()
        e = _engine_of_dm(idx)
        k, a, b = e.dm_slot(idx)
        v = e.volts.copy()
        v[a:b] = self.com
        e.shape_from(v)

    def set_com(self, com, shape_dm=True):
        self.com[:] = np.asarray(com, dtype=f32).reshape(-1)
        if shape_dm:
            self._apply()

    def comp_shape(self, com=None):
        if com is not None:
            self.com[:] = np.asarray(com, dtype=f32).reshape(-1)
        self._apply()

    def comp_oneactu(self, i, ampli):
        """Shape of actuator i pushed by `ampli`; the stored command is not touched."""
        idx = self._index()
        e = _engine_of_dm(idx)
        k, a, b = e.dm_slot(idx)
        v = e.volts.copy()
        v[a:b] = 0
        v[a + int(i)] = ampli
        keep = e.volts.copy()
        e.shape_from(v)
        e.volts[:] = keep
        e.volts[a:b] = 0
        e.volts[a + int(i)] = ampli

    def reset_shape(self):
        self.com[:] = 0
        if _HUB["sensors"] is None or self.influ is None:
            return                            # nothing on the device yet
        self._apply()


class Dms(object):
    def __init__(self):
        self.d_dms = []
        _HUB["dms"] = self
        _invalidate("new Dms")

    def add_dm(self, ctx, typ, alt, dim, ntotact, influsize, ninflupos, n_npts, push4imat, nord,
               dev):
        self.d_dms.append(Dm(typ, alt, dim, ntotact, influsize, push4imat))
        _invalidate("add_dm")

    def remove_dm(self, i):
        self.d_dms.pop(i)
        _invalidate("remove_dm")

    def insert_dm(self, ctx, typ, alt, dim, ntotact, influsize, ninflupos, n_npts, push4imat, nord,
                  dx, dy, theta, G, dev, idx):
        self.d_dms.insert(idx, Dm(typ, alt, dim, ntotact, influsize, push4imat))
        _invalidate("insert_dm")

    def set_full_com(self, com, shape_dm=True):
        o = 0
        for d in self.d_dms:
            d.set_com(com[o:o + d.nactu], shape_dm)
            o += d.nactu


# ------------------------------------------------------------------------------ WFS
class Wfs(object):
    def __init__(self, i, tel, nxsub, nvalid, npix, nphase, nrebin, nfft, ntot, pdiam, nphot):
        self.i, self.tel = i, tel
        self.nxsub, self.nvalid, self.npix, self.nphase = int(nxsub), int(nvalid), int(npix), int(nphase)
        self.nrebin, self.nfft, self.ntot, self.subapd = int(nrebin), int(nfft), int(ntot), float(pdiam)
        self.nphot = f32(nphot)
        self.noise, self.seed = -1.0, 1234
        self.d_gs = None
        self.d_slopes = DevArray(lambda: self._sim().slopes[0].cpu().numpy())
        self.d_binimg = DevArray(self._binimg)
        self.d_camimg = self.d_binimg
        self.d_bincube = DevArray(self._bincube)

    def _engine(self):
        return _engines()[self.i]

    def _sim(self):
        return self._engine().sim

    def _cube(self):
        sim = self._sim()
        sim._need_bincube()
        return sim.t["bincube"][0].cpu().numpy().reshape(self.nvalid, self.npix, self.npix)   # [i][y][x]

    def _bincube(self):
        return np.ascontiguousarray(self._cube().transpose(2, 1, 0))                        # [x][y][i]

    def _binimg(self):
        dim = self.npix * self.nxsub
        img, cube = np.zeros((dim, dim), dtype=f32), self._cube()
        for k in range(self.nvalid):
            x0, y0 = self.validsubsx[k], self.validsubsy[k]
            img[y0:y0 + self.npix, x0:x0 + self.npix] = cube[k]
        return img.T.copy()

    def load_arrays(self, phasemap, hrmap, binmap, halfxy, fluxPerSub, validsubsx, validsubsy,
                    validpuppixx, validpuppixy, ftkernel):
        self.phasemap, self.binmap = _c(phasemap, np.int32), _c(binmap, np.int32)
        self.halfxy, self.flux = _c(halfxy), _c(fluxPerSub)
        self.validsubsx, self.validsubsy = _c(validsubsx, np.int32), _c(validsubsy, np.int32)
        self.d_validsubsx = DevArray(lambda: self.validsubsx.copy())
        self.d_validsubsy = DevArray(lambda: self.validsubsy.copy())
        _invalidate("load_arrays")

    def set_noise(self, noise, seed):
        """Wfs.set_noise (wfsCompass.py:345-350): restart this sensor's noise stream."""
        if float(noise) != float(self.noise):
            self.noise = float(noise)
            _invalidate("set_noise: new noise level")
        self.seed = int(seed)
        if self.noise >= 0:
            base = _HUB["atmos"]._base_seed()
            if (int(seed) & 0xFFFFFFFF) != base:
                raise NotImplementedError("the sensor's noise seed must be the atmosphere's base seed "
                                          "(the library keeps one seed per environment)")
        if _HUB["engines"] is not None:
            self._sim().t["frame"].zero_()

    def comp_image(self, noise=True):
        e = self._engine()
        if not e.image_ok:
            raise NotImplementedError("WFS %d: no image formation for this sampling (the spot kernel is "
                                      "specialised for 16 / 64 / 2 / 16)" % self.i)
        e.sim.comp_image(from_phase_buffer=True, noise=bool(noise), write_bincube=True, cog=False)

    def set_binimg(self, img, size):
        """Replace the camera image (rlSupervisor.py:884-889: the denoised image goes back)."""
        import torch
        img = np.asarray(img, dtype=f32).T
        cube = np.zeros((self.nvalid, self.npix * self.npix), dtype=f32)
        for k in range(self.nvalid):
            x0, y0 = self.validsubsx[k], self.validsubsy[k]
            cube[k] = img[y0:y0 + self.npix, x0:x0 + self.npix].reshape(-1)
        sim = self._sim()
        sim._need_bincube()
        sim.t["bincube"][0].copy_(torch.from_numpy(cube))

    def slopes_geom(self, meth=0):
        self._sim().slopes_geom()


class Sensors(object):
    def __init__(self, ctx, tel, t_wfs, nsensors, nxsub, nvalid, nPupils, npix, nphase, nrebin,
                 nfft, ntota, npup, pdiam, nphot, nphot4imat, lgs, fakecam, maxFlux, maxPix, dev,
                 roket):
        for t in t_wfs:
            if str(t) != "sh":
                raise NotImplementedError("only Shack-Hartmann sensors are on the hot path")
        self.tel = tel
        self.d_wfs = [Wfs(i, tel, nxsub[i], nvalid[i], npix[i], nphase[i], nrebin[i], nfft[i],
                          ntota[i], pdiam[i], nphot[i]) for i in range(nsensors)]
        _HUB["sensors"] = self
        _invalidate("new Sensors")

    def initgs(self, xpos, ypos, Lambda, mag, zerop, size, noise, seed, G, thetaML, dx, dy):
        for i, w in enumerate(self.d_wfs):
            w.d_gs = Source(i, size[i], Lambda[i], target=False)
            w.noise, w.seed = float(noise[i]), int(seed[i])
        _invalidate("initgs")


# ------------------------------------------------------------------------------ target
class TargetSource(Source):
    def __init__(self, i, tel, size, lam):
        Source.__init__(self, i, size, lam, target=True)
        self.tel = tel
        self.strehl_counter = 0
        self.d_image_se = DevArray(lambda: self._window(False))
        self.d_image_le = DevArray(lambda: self._window(True))

    def init_strehlmeter(self):
        self.reset_strehlmeter()

    def reset_strehlmeter(self):
        self.strehl_counter = 0
        if _HUB["engines"] is not None:
            self._sim().reset_strehl()

    def comp_image(self, puponly=0, compLE=True):
        """PSF of the phase as it stands; committed at once (short exposure, long-exposure sum,
        phase variance: aomarl_target_psf_buffer + aomarl_comp_strehl)."""
        e = _engines()[self.i]
        e.lib_call(e.sim.lib.aomarl_target_psf_buffer)
        e.sim.comp_strehl()
        self.strehl_counter += 1

    _fit = True

    def comp_strehl(self, do_fit=True):
        self._fit = bool(do_fit)              # numbers are read from the device on access: fitted peaks in slots 6 / 7

    def _st(self):
        sim = self._sim()
        if self._fit:
            la.check(sim.lib.aomarl_strehl_fit(sim.ctx, C.byref(sim.st), 0, sim.nenv, sim._stream()))
        return sim.t["strehl"][0].cpu().numpy()

    strehl_se = property(lambda self: float(self._st()[6 if self._fit else 0]))
    strehl_le = property(lambda self: float(self._st()[7 if self._fit else 1]))
    phase_var = property(lambda self: float(self._st()[2]))
    phase_var_avg = property(lambda self: float(self._st()[3]))
    phase_var_count = property(lambda self: int(self._st()[4]))

    def _window(self, le):
        sim = self._sim()
        W = 2 * HW
        if le:
            return sim.t["le_img"][0].cpu().numpy().reshape(W, W).copy()
        raise NotImplementedError("the short-exposure PSF window is not kept after the commit")


class Target(object):
    def __init__(self, ctx, tel, n, xpos, ypos, Lambda, mag, zerop, sizes, Npts, dev):
        self.d_targets = [TargetSource(i, tel, sizes[i], Lambda[i]) for i in range(n)]
        _HUB["target"] = self
        _invalidate("new Target")


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
        self._gain = 0.0
        self.open_loop = 0
        self.imat = np.zeros((self.nslope, self.nactu), dtype=f32)
        self.cmat = np.zeros((self.nactu, self.nslope), dtype=f32)
        self.d_imat = DevArray(lambda: self.imat.copy())
        self.d_cmat = DevArray(lambda: self.cmat.copy())
        for n in ("com", "err", "voltage"):
            setattr(self, "d_" + n, DevArray(lambda n=n: getattr(self._sim(), n)[0].cpu().numpy()))
        self.d_centroids = DevArray(lambda: self._sim().slopes[0].cpu().numpy())

    def _engine(self):
        return _engines()[self.nwfs[0]]

    def _sim(self):
        return self._engine().sim

    # gain: an attribute the reference reads and a setter it calls (ao_env.py:950-958)
    gain = property(lambda self: self._gain)

    def set_gain(self, g):
        self._gain = float(g)
        if _HUB["engines"] is not None:
            self._sim().set_gain(self._gain)

    def set_modal_gains(self, m):
        self.mgain = _c(m)
        if not np.all(self.mgain == 1.0):
            raise NotImplementedError("modal gains other than 1 (rtc_init.py:507-513 sets ones)")

    def set_cmat(self, cmat):
        self.cmat[:] = np.asarray(cmat, dtype=f32)
        if _HUB["engines"] is not None:
            self._sim().set_cmat(self.cmat)

    def set_imat(self, imat):
        self.imat[:] = np.asarray(imat, dtype=f32)

    def set_com(self, com, size=None):
        import torch
        com = np.asarray(com, dtype=f32).reshape(1, -1)
        if com.shape[1] != self.nactu:
            raise ValueError("Dimension mismatch")
        self._sim().set_com(torch.from_numpy(com))

    def set_open_loop(self, flag, reset=True):
        self.open_loop = int(flag)
        if flag and reset:
            sim = self._sim()
            for k in ("com", "com1", "com2", "err", "voltage"):
                sim.t[k].zero_()

    def svdec_imat(self):
        w = np.linalg.eigvalsh(self.imat.astype(np.float64).T @ self.imat.astype(np.float64))
        self.eigenvals = w[::-1].astype(f32)           # descending
        self.d_eigenvals = DevArray(lambda: self.eigenvals.copy())

    def build_cmat(self, nfilt):
        D = self.imat.astype(np.float64)
        w, V = np.linalg.eigh(D.T @ D)
        inv = np.zeros_like(w)
        keep = np.argsort(w)[int(nfilt):]
        inv[keep] = 1.0 / w[keep]
        self.set_cmat(((V * inv[None, :]) @ V.T @ D.T).astype(f32))

    # ---- geometric controller (rtc_init.py:418-448, rtcCompass.py:545-547)
    def init_proj_sparse(self, dms, indx_dm, unitpervolt, indx_pup, indx_mpup, roket=False):
        """Influence functions of this controller's DMs on the lit pupil pixels, as the science
        target sees them (unit pokes through the library's own DM-shape and ray-trace kernels), and
        the projector of ao_marl_amd.modal.geo_projector -> aomarl_set_geo."""
        import scipy.sparse as sp
        e = self._engine()
        sim, s = e.sim, e.s
        lit = (s.spupil.reshape(-1) > 0)
        cols = []
        keep = e.volts.copy()
        for a in range(s.nactu):
            v = np.zeros(s.nactu, dtype=np.float32)
            v[a] = 1.0
            e.shape_from(v)
            sim.raytrace_target(atm=False, dms=True, reset=True)
            cols.append(sim.t["tar_phase"][0].cpu().numpy().reshape(-1)[lit].astype(np.float64))
        e.shape_from(keep)
        IF = sp.csc_matrix(np.stack(cols, axis=1))
        W = np.ascontiguousarray(modal.geo_projector(IF), dtype=np.float32)
        la.check(sim.lib.aomarl_set_geo(sim.ctx, la.fptr(W)))
        import torch
        self._gwork = torch.zeros(int(sim.lib.aomarl_geo_workspace_floats(sim.ctx, 1)),
                                  dtype=torch.float32, device=sim.device)
        e.geo_ready = True

    def comp_dphi(self, source, is_wfs=False):
        """The pupil phase of `source` (a target of this engine) enters the projection at
        do_control; the library wants it masked by the pupil (aomarl_geo_control)."""
        if is_wfs or source.i != self.nwfs[0]:
            raise NotImplementedError("comp_dphi from another control path's source")
        e = self._engine()
        e.lib_call(e.sim.lib.aomarl_raytrace_target, la.TRACE_MASK)       # multiply by the pupil in place


class Rtc_FFF(object):
    def __init__(self):
        self.d_centro, self.d_control = [], []
        _HUB["rtc"] = self
        _invalidate("new Rtc")

    def add_centroider(self, ctx, nvalid, offset, scale, filter_TT, dev, typ, wfs=None):
        if str(typ) != "cog":
            raise NotImplementedError("only the centre-of-gravity centroider is on the hot path")
        self.d_centro.append(Centroider(nvalid, offset, scale, wfs))
        _invalidate("add_centroider")

    def add_controller(self, ctx, nvalid, nslope, nactu, delay, dev, typ, dms=None, ndm=(), ndm_size=0,
                       nwfs=(), nwfs_size=0, Nphi=0, roket=False, nstates=0):
        if str(typ) not in ("ls", "geo"):
            raise NotImplementedError("controller type %r" % (typ,))
        self.d_control.append(Controller(self, nvalid, nslope, nactu, delay, typ, dms, ndm, nwfs))
        _invalidate("add_controller")

    def do_centroids(self, n):
        self.d_control[n]._sim().do_centroids()

    def do_control(self, n, *a, **k):
        c = self.d_control[n]
        e, sim = c._engine(), c._sim()
        if c.type == "geo":
            if not e.geo_ready:
                raise la.AomarlError("GEO controller used before init_proj_sparse")
            e.lib_call(sim.lib.aomarl_geo_control, c._gwork.data_ptr())
            return
        if c.open_loop:                       # err only: the integrator is frozen
            sim.set_gain(0.0)
            sim.do_control()
            sim.set_gain(c._gain)
            return
        sim.do_control()

    def apply_control(self, n, comp_voltage=True):
        c = self.d_control[n]
        e = c._engine()
        e.sim.apply_control(comp_voltage=bool(comp_voltage))
        e.volts[:] = e.sim.voltage[0].cpu().numpy()
        for idx in e.dm_index:                # Dm.com follows the controller's voltages
            k, a, b = e.dm_slot(idx)
            _HUB["dms"].d_dms[idx].com[:] = e.volts[a:b]

    def do_clipping(self, n):
        pass

    def do_imat(self, n, dms):
        """Push-pull interaction matrix through the full image-formation + centroiding chain
        (imats.py:115-167); one frame per poke, noise off, the noise stream is not advanced."""
        c = self.d_control[n]
        e = c._engine()
        sim, s = e.sim, e.s
        col = 0
        for k, m in enumerate(s.dms):
            push = float(_HUB["dms"].d_dms[e.dm_index[k]].push4imat)
            p = np.zeros((m.ntotact, s.nactu), dtype=np.float32)
            a0 = sum(x.ntotact for x in s.dms[:k])
            p[np.arange(m.ntotact), a0 + np.arange(m.ntotact)] = push
            plus = sim.dm_response(p, geometric=False)
            minus = sim.dm_response(-p, geometric=False)
            c.imat[:, col:col + m.ntotact] = ((plus - minus) / f32(2 * push)).T
            col += m.ntotact
        e.shape_from(np.zeros(s.nactu, dtype=np.float32))
        sim.t["frame"].zero_()


def install():
    """Make `import sutraWrap` / `import carmaWrap` (shesha/sutra_wrap.py:4-38) resolve to this
    facade."""
    sw = types.ModuleType("sutraWrap")
    for name, cls in (("Dms", Dms), ("Rtc_FFF", Rtc_FFF), ("Sensors", Sensors), ("Atmos", Atmos),
                      ("Telescope", Telescope), ("Target", Target)):
        setattr(sw, name, cls)

    def _missing(n):
        def ctor(*a, **k):
            raise RuntimeError("%s is not provided by ao_marl_amd.sutra_facade" % n)
        return ctor

    for n in ("Rtc_FHF", "Rtc_UFF", "Rtc_UHF", "Rtc_FFU", "Rtc_FHU", "Rtc_UFU", "Rtc_UHU",
              "Target_brahma", "Gamora", "Groot", "Rtc_brahma", "Rtc_cacao_FFF", "Rtc_cacao_UFF",
              "Rtc_cacao_FHF", "Rtc_cacao_UHF"):
        setattr(sw, n, type(n, (object,), {"__init__": _missing(n)}))
    cw = types.ModuleType("carmaWrap")
    cw.context = context
    sys.modules["sutraWrap"], sys.modules["carmaWrap"] = sw, cw
    return sw, cw


def reset_hub(device="cuda:0"):
    """Forget the current simulation (tests build several in one process)."""
    import torch
    if _HUB["engines"] is not None:
        torch.cuda.synchronize()
    _HUB.update(atmos=None, sensors=None, target=None, dms=[], rtc=None, engines=None, device=device,
                sealed=False)
