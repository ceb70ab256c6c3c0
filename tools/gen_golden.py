#!/usr/bin/env python3
"""Generate golden geometry fixtures by IMPORTING the reference's pure-Python init code.

Runs ONLY in the build container (needs /root/reference); the fixtures it writes under
tests/golden/ are plain data (.npz) and are the only reference-derived artefacts committed.

How: a *recording* fake of the `sutraWrap`/`carmaWrap` native modules is installed before
`shesha` is imported, so the reference's own `tel_init / atmos_init / dm_init / target_init /
wfs_init` (shesha/init/*.py) run unmodified and every array / scalar they would hand to COMPASS
is captured at the drop-in boundary (shesha/sutra_wrap.py:46-72).

Usage: python tools/gen_golden.py [10x10|40x40|all]
"""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

import types  # noqa: E402
import numpy as np  # noqa: E402

CALLS = []


class Rec(object):
    """Records constructor + method calls; grows d_* lists like the native objects do."""

    def __init__(self, *a, **k):
        self._ctor = (a, k)
        self._calls = []
        self.d_dms, self.d_wfs, self.d_targets, self.d_screens = [], [], [], []
        kind = type(self).__name__
        CALLS.append((kind, "__init__", a, k))
        if kind == "Sensors":
            n = a[3]
            for i in range(n):
                w = Rec_sub("wfs%d" % i)
                w.d_gs = Rec_sub("wfs%d.gs" % i)
                self.d_wfs.append(w)
        if kind == "Target":
            n = a[2]
            for i in range(n):
                self.d_targets.append(Rec_sub("target%d" % i))

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def f(*a, **k):
            CALLS.append((type(self).__name__, name, a, k))
            if name in ("add_dm",):
                self.d_dms.append(Rec_sub("dm%d" % len(self.d_dms)))
            return None

        return f


class Rec_sub(object):
    def __init__(self, tag):
        self._tag = tag

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def f(*a, **k):
            CALLS.append((self._tag, name, a, k))

        return f


def install_fake_native():
    sw = types.ModuleType("sutraWrap")
    for n in ("Dms", "Rtc_FFF", "Rtc_FHF", "Rtc_UFF", "Rtc_UHF", "Rtc_FFU", "Rtc_FHU", "Rtc_UFU",
              "Rtc_UHU", "Sensors", "Atmos", "Telescope", "Target", "Target_brahma", "Gamora",
              "Groot", "Rtc_brahma", "Rtc_cacao_FFF", "Rtc_cacao_UFF", "Rtc_cacao_FHF",
              "Rtc_cacao_UHF"):
        setattr(sw, n, type(n, (Rec,), {}))
    cw = types.ModuleType("carmaWrap")

    class context(object):
        active_device = 0

        @staticmethod
        def get_instance_1gpu(d):
            return context()

        @staticmethod
        def get_instance_ngpu(n, d):
            return context()

        def set_active_device(self, d):
            pass

    cw.context = context
    sys.modules["sutraWrap"] = sw
    sys.modules["carmaWrap"] = cw


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def run(param_name, out_path, full):
    """full=True stores whole arrays (10x10); else hashes + small slices (40x40)."""
    del CALLS[:]
    from shesha.util.utilities import load_config_from_file
    from shesha.init.geom_init import tel_init
    from shesha.init.atmos_init import atmos_init
    from shesha.init.dm_init import dm_init
    from shesha.init.target_init import target_init
    from shesha.init.wfs_init import wfs_init
    import carmaWrap
    cfg = load_config_from_file(
            os.path.join(_ref_shims.REF, "data/par/par4rl/production", param_name + ".py"))
    ctx = carmaWrap.context.get_instance_1gpu(0)
    # same order as GenericSupervisor._init_components (genericSupervisor.py:116-142)
    tel = tel_init(ctx, cfg.p_geom, cfg.p_tel, cfg.p_atmos.r0, cfg.p_loop.ittime, cfg.p_wfss)
    atm = atmos_init(ctx, cfg.p_atmos, cfg.p_tel, cfg.p_geom, cfg.p_loop.ittime, cfg.p_wfss,
                     cfg.p_targets)
    dms = dm_init(ctx, cfg.p_dms, cfg.p_tel, cfg.p_geom, cfg.p_wfss)
    tar = target_init(ctx, tel, cfg.p_targets, cfg.p_atmos, cfg.p_tel, cfg.p_geom, cfg.p_dms)
    wfs = wfs_init(ctx, tel, cfg.p_wfss, cfg.p_tel, cfg.p_geom, cfg.p_dms, cfg.p_atmos)

    g, w, a = cfg.p_geom, cfg.p_wfss[0], cfg.p_atmos
    out = {}

    def put(name, arr, always=False):
        arr = np.asarray(arr)
        out["sha_" + name] = np.array(sha(arr))
        out["shape_" + name] = np.array(arr.shape, dtype=np.int64)
        if full or always or arr.size <= 4096:
            out[name] = arr

    scal = dict(pupdiam=g.pupdiam, ssize=g.ssize, n=g._n, n1=g._n1, n2=g._n2, p1=g._p1, p2=g._p2,
                cent=g.cent, pdiam=w._pdiam, Nfft=w._Nfft, Ntot=w._Ntot, nrebin=w._nrebin,
                qpixsize=w._qpixsize, pixsize=w.pixsize, nvalid=w._nvalid, nphotons=w._nphotons,
                subapd=w._subapd, npix=w.npix, pupixsize=a.pupixsize, nscreens=a.nscreens)
    for k, v in scal.items():
        out["s_" + k] = np.array(v)
    put("spupil", g._spupil)
    put("mpupil", g._mpupil)
    put("isvalid", w._isvalid)
    put("fluxPerSub", w._fluxPerSub)
    put("validsubsx", w._validsubsx, True)
    put("validsubsy", w._validsubsy, True)
    put("validpuppixx", w._validpuppixx, True)
    put("validpuppixy", w._validpuppixy, True)
    put("phasemap", w._phasemap)
    put("binmap", w._binmap, True)
    put("halfxy", w._halfxy, True)
    put("ftkernel", w._ftkernel)
    out["ftkernel_is_dirac"] = np.array(
            bool(np.allclose(w._ftkernel, w._ftkernel.flat[0], atol=1e-9)))
    # atmosphere
    put("dim_screens", a.dim_screens, True)
    put("deltax", a._deltax, True)
    put("deltay", a._deltay, True)
    put("frac", a.frac, True)
    put("alt", a.alt, True)
    for c in CALLS:
        if c[0] == "Atmos" and c[1] == "__init__":
            args = c[2]
            put("atm_r0_layers", args[3], True)
            put("atm_stencil_size", args[5], True)
        if c[0] == "Atmos" and c[1] == "init_screen":
            i, A, B, istx, isty, seed = c[2]
            put("A%d" % i, A)
            put("B%d" % i, B)
            put("istx%d" % i, istx, True)
            put("isty%d" % i, isty, True)
            out["seed%d" % i] = np.array(seed)
            if not full:  # a few rows so that float content is pinned, not only the hash
                out["A%d_rows" % i] = np.asarray(A)[:4].copy()
                out["B%d_rows" % i] = np.asarray(B)[:4].copy()
    # layer offsets handed to the sources
    offs = [(c[0], c[2]) for c in CALLS if c[1] == "add_layer"]
    out["add_layer_tags"] = np.array(["%s|%s|%d" % (t, a_[0], a_[1]) for t, a_ in offs])
    out["add_layer_xy"] = np.array([[a_[2], a_[3]] for t, a_ in offs], dtype=np.float64)
    # DMs
    for i, d in enumerate(cfg.p_dms):
        pre = "dm%d_" % i
        out[pre + "type"] = np.array(str(d.type))
        out[pre + "n1"] = np.array(d._n1)
        out[pre + "n2"] = np.array(d._n2)
        out[pre + "ntotact"] = np.array(d._ntotact)
        out[pre + "influsize"] = np.array(d._influsize)
        if str(d.type) == "pzt":
            out[pre + "pitch"] = np.array(d._pitch)
            put(pre + "xpos", d._xpos, True)
            put(pre + "ypos", d._ypos, True)
            put(pre + "i1", d._i1, True)
            put(pre + "j1", d._j1, True)
            put(pre + "influ0", d._influ[:, :, 0], True)
            out[pre + "influ_maxdev"] = np.array(
                    float(np.max(np.abs(d._influ - d._influ[:, :, :1]))))
            put(pre + "influpos", d._influpos)
            put(pre + "ninflu", d._ninflu)
            put(pre + "influstart", d._influstart)
        else:
            put(pre + "influ", d._influ)
            if not full:
                out[pre + "influ_corner"] = d._influ[:8, :8, :].copy()
                c0 = d._influ.shape[0] // 2
                out[pre + "influ_center"] = d._influ[c0 - 4:c0 + 4, c0 - 4:c0 + 4, :].copy()
    # Sensors / Target ctor scalars
    for c in CALLS:
        if c[0] == "Sensors" and c[1] == "__init__":
            out["sensors_nphot"] = np.asarray(c[2][14])
        if c[0] == "Target" and c[1] == "__init__":
            out["target_Npts"] = np.array(c[2][9])
            out["target_lambda"] = np.asarray(c[2][5])
        if c[0] == "wfs0" and c[1] == "load_arrays":
            put("fluxPerSub_valid", c[2][4], True)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, "keys:", len(out), "size: %.1f KB" % (os.path.getsize(out_path) / 1e3))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    install_fake_native()
    _ref_shims.install()
    outdir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    if which in ("10x10", "all"):
        run("production_sh_10x10_2m", os.path.join(outdir, "geom_10x10.npz"), True)
    if which in ("40x40", "all"):
        run("production_sh_40x40_8m_3layers", os.path.join(outdir, "geom_40x40.npz"), False)
        run("production_sh_40x40_8m_3layers_d0_noise",
            os.path.join(outdir, "geom_40x40_d0_noise.npz"), False)
