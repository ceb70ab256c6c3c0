"""Simulation parameters for the AO environment hot path.

Two ways in:
  * `builtin(name)`  -- the four production configurations the reference ships
    (data/par/par4rl/production/production_sh_10x10_2m.py, ..._40x40_8m_3layers[_d0_noise|
    _d1_noise].py), restated here as plain data so nothing from the reference tree is needed at
    run time (the GPU box has no /root/reference).
  * `load_param_file(path)` -- executes a COMPASS/shesha-style parameter module (a Python file of
    `conf.Param_xxx()` objects and `set_yyy(...)` calls, loaded in the reference by
    shesha/util/utilities.py:159-231) against a recording stand-in for `shesha.config`, so a user
    of the reference can keep their parameter files.

Defaults follow shesha/config/P*.py (PDMS.py:50-68, PWFS.py:48-130, PTEL.py:48-72,
PTARGET.py:49-61, PCONTROLLER.py:50-97).
"""
import copy
import importlib.util
import os
import sys
import types

import numpy as np

__all__ = ["ParamSet", "builtin", "load_param_file", "BUILTIN_NAMES"]


class _P(object):
    """Attribute bag with defaults + `set_xxx` / `get_xxx` methods, like shesha.config.Param_*."""
    _defaults = {}

    def __init__(self, **kw):
        for k, v in self._defaults.items():
            object.__setattr__(self, k, copy.deepcopy(v))
        for k, v in kw.items():
            setattr(self, k, v)

    def __getattr__(self, name):
        if name.startswith("set_"):
            key = name[4:]

            def setter(v):
                setattr(self, key, v)

            return setter
        if name.startswith("get_"):
            key = name[4:]
            return lambda: getattr(self, key)
        raise AttributeError(name)

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join(
                "%s=%r" % kv for kv in sorted(self.__dict__.items()) if not kv[0].startswith("_")))


class Param_loop(_P):
    _defaults = dict(niter=0, ittime=0.0, devices=[0])


class Param_geom(_P):
    _defaults = dict(zenithangle=0.0, pupdiam=0)


class Param_tel(_P):
    _defaults = dict(diam=0.0, cobs=0.0, type_ap="Generic", t_spiders=-1.0, spiders_type=None,
                     pupangle=0.0)


class Param_atmos(_P):
    _defaults = dict(nscreens=0, r0=None, L0=None, alt=None, winddir=None, windspeed=None,
                     frac=None, seeds=None)


class Param_target(_P):
    _defaults = dict(apod=False, Lambda=None, xpos=0.0, ypos=0.0, mag=None, zerop=1.0,
                     dms_seen=None)


class Param_wfs(_P):
    _defaults = dict(type=None, nxsub=0, npix=0, pixsize=0.0, Lambda=0.0, optthroughput=0.0,
                     fracsub=0.0, open_loop=False, atmos_seen=0, dms_seen=None, xpos=0.0, ypos=0.0,
                     gsalt=0.0, gsmag=0.0, zerop=0.0, noise=0.0, kernel=0.0, G=1.0, thetaML=0.0,
                     dx=0.0, dy=0.0)

    def __init__(self, roket=False, **kw):
        _P.__init__(self, **kw)
        self.roket = roket


class Param_dm(_P):
    _defaults = dict(type=None, nact=0, alt=0.0, thresh=0.0, coupling=0.2, gain=1.0,
                     unitpervolt=0.01, push4imat=1.0, margin_out=None, margin_in=0.0,
                     pzt_extent=5.0, influ_type="default", type_pattern=None)


class Param_centroider(_P):
    _defaults = dict(nwfs=None, type=None, thresh=1.0e-4, filter_TT=False)


class Param_controller(_P):
    _defaults = dict(type=None, nwfs=None, ndm=None, maxcond=None, delay=None, gain=None,
                     nmodes=None, modopti=False, do_kl_imat=False, nstates=0)


class ParamSet(object):
    """What the reference calls a `config` module: p_loop, p_geom, p_tel, p_atmos, p_targets,
    p_wfss, p_dms, p_centroiders, p_controllers, simul_name."""
    _names = ("p_loop", "p_geom", "p_tel", "p_atmos", "p_targets", "p_wfss", "p_dms",
              "p_centroiders", "p_controllers", "simul_name")

    def __init__(self, **kw):
        for n in self._names:
            setattr(self, n, kw.get(n))

    def validate(self):
        for n in self._names[:-1]:
            if getattr(self, n) is None:
                raise ValueError("parameter set is missing %s" % n)
        types_dm = [d.type for d in self.p_dms]
        if "tt" in types_dm:
            first = types_dm.index("tt")
            if any(t != "tt" for t in types_dm[first:]):
                # same check as shesha/init/dm_init.py:83-86
                raise RuntimeError("TT must be defined at the end of the dms parameters")
        a = self.p_atmos
        for k in ("frac", "alt", "windspeed", "winddir", "L0"):
            v = getattr(a, k)
            if v is None or len(v) != a.nscreens:
                raise ValueError("p_atmos.%s must have nscreens=%d entries" % (k, a.nscreens))
        return self


# ------------------------------------------------------------------------------------------
# reference-style parameter files


def _conf_module():
    m = types.ModuleType("shesha.config")
    for c in (Param_loop, Param_geom, Param_tel, Param_atmos, Param_target, Param_wfs, Param_dm,
              Param_centroider, Param_controller):
        setattr(m, c.__name__, c)
    return m


def load_param_file(path):
    """Execute a shesha-style parameter module without shesha; returns a ParamSet."""
    path = os.path.abspath(path)
    if not path.endswith(".py"):
        raise ValueError("Config file must be .py or a module")  # utilities.py:183-184
    saved = {k: sys.modules.get(k) for k in ("shesha", "shesha.config")}
    pkg = types.ModuleType("shesha")
    conf = _conf_module()
    pkg.config = conf
    sys.modules["shesha"], sys.modules["shesha.config"] = pkg, conf
    try:
        spec = importlib.util.spec_from_file_location("_aomarl_param_" + os.path.basename(path)[:-3],
                                                      path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    ps = ParamSet(**{n: getattr(mod, n, None) for n in ParamSet._names})
    return _normalise(ps).validate()


def _normalise(ps):
    a = ps.p_atmos
    for k in ("frac", "alt", "windspeed", "winddir", "L0"):
        v = getattr(a, k)
        if v is not None:
            setattr(a, k, np.asarray(v, dtype=np.float32).reshape(-1))  # PATMOS.py: float32 arrays
    for w in ps.p_wfss:
        if w.dms_seen is not None:
            w.dms_seen = np.asarray(w.dms_seen, dtype=np.int32)
    for t in ps.p_targets:
        if t.dms_seen is not None:
            t.dms_seen = np.asarray(t.dms_seen, dtype=np.int32)
    for c in ps.p_controllers:
        c.nwfs = np.asarray(c.nwfs, dtype=np.int32)
        c.ndm = np.asarray(c.ndm, dtype=np.int32)
    return ps


# ------------------------------------------------------------------------------------------
# built-in production configurations (values: the reference's parameter files)

BUILTIN_NAMES = ("production_sh_10x10_2m", "production_sh_40x40_8m_3layers",
                 "production_sh_40x40_8m_3layers_d0_noise",
                 "production_sh_40x40_8m_3layers_d1_noise")
# The wind variants of the 40x40 file the reference ships (data/par/par4rl/production/
# production_sh_40x40_8m_3layers_<variant>.py), each with recorded COMPASS statistics
# (state_normalization/*.pickle).  They differ from the base file in exactly three numbers:
#   _dir_0_15_30 / _same_dir      winddir [0, 15, 30] / [0, 0, 0]        (base: [0, 45, 90])
#   _v_10_5_15 / _v_20_15_25      windspeed [10, 5, 15] / [20, 15, 25]   (base: [15, 10, 20])
#   gain                          0.6 in the three _v_10_5_15 files, 0.7 otherwise
# plus two statistics files recorded on _same_dir with the integrator gain changed on the command line
# (--gain_change, train_rpc.py:80-82); their gains are the `g` of the geo/ parameter files of the same
# name: _same_dir_gain_change_high = 0.9, _same_dir_gain_change_low = 0.2.  _same_dir_roket is
# _same_dir without the geometric twin (identical statistics).
_BASE_L = "production_sh_40x40_8m_3layers"
WIND_VARIANTS = tuple(_BASE_L + sfx for sfx in (
        "_dir_0_15_30", "_dir_0_15_30_v_10_5_15", "_dir_0_15_30_v_20_15_25", "_same_dir",
        "_same_dir_v_10_5_15", "_same_dir_v_20_15_25", "_v_10_5_15", "_v_20_15_25", "_same_dir_roket",
        "_same_dir_gain_change_high", "_same_dir_gain_change_low"))


def _wind_variant(name):
    """(winddir, windspeed, gain) of a WIND_VARIANTS name, or None."""
    if name not in WIND_VARIANTS:
        return None
    sfx = name[len(_BASE_L):]
    wdir = [0, 15, 30] if "_dir_0_15_30" in sfx else ([0, 0, 0] if "_same_dir" in sfx else [0, 45, 90])
    speed, gain = [15, 10, 20], 0.7
    if "_v_10_5_15" in sfx:
        speed, gain = [10, 5, 15], 0.6
    elif "_v_20_15_25" in sfx:
        speed = [20, 15, 25]
    if sfx.endswith("_gain_change_high"):
        gain = 0.9
    elif sfx.endswith("_gain_change_low"):
        gain = 0.2
    return wdir, speed, gain


def _wfs(nxsub, gsmag, noise, dms_seen):
    return Param_wfs(type="sh", nxsub=nxsub, npix=16, dms_seen=dms_seen, pixsize=0.25,
                     fracsub=0.8, xpos=0., ypos=0., Lambda=0.5, gsmag=gsmag, optthroughput=0.12,
                     zerop=1.e11, noise=noise, atmos_seen=1)


def _pzt(nact):
    return Param_dm(type="pzt", nact=nact, alt=0., thresh=0.3, coupling=0.2, unitpervolt=0.01,
                    push4imat=100.)


def _tt():
    return Param_dm(type="tt", alt=0., unitpervolt=0.0005, push4imat=10.)


def _target(dms_seen):
    return Param_target(dms_seen=dms_seen, xpos=0., ypos=0., Lambda=1.65, mag=10.)


def builtin(name):
    """Return a fresh ParamSet for one of BUILTIN_NAMES (a trailing '.py' is accepted, as the
    reference's `parameters_telescope` option carries one, ao_env.py:290)."""
    if name.endswith(".py"):
        name = name[:-3]
    variant = _wind_variant(name)
    if variant is not None:
        ps = builtin(_BASE_L)
        ps.simul_name = name
        ps.p_atmos.winddir = np.asarray(variant[0], dtype=np.float32)
        ps.p_atmos.windspeed = np.asarray(variant[1], dtype=np.float32)
        for c in ps.p_controllers:
            c.gain = variant[2]
        return ps.validate()
    if name not in BUILTIN_NAMES:
        raise NotImplementedError("unknown built-in parameter set %r" % name)
    small = name == "production_sh_10x10_2m"
    noise = name.endswith("_noise")
    nxsub = 10 if small else 40
    diam = 2.0 if small else 8.0
    if noise:
        d0 = name.endswith("_d0_noise")
        delay, ron, gain = (0., 3., 0.3) if d0 else (1., 3., 0.65)
    else:
        delay, ron, gain = 1., -1., 0.7
    p_loop = Param_loop(niter=2000, ittime=0.002)
    p_geom = Param_geom(zenithangle=0.)
    p_tel = Param_tel(diam=diam, cobs=0.12)
    if small:
        p_atmos = Param_atmos(r0=0.16, nscreens=1, frac=[1.0], alt=[0.0], windspeed=[20.0],
                              winddir=[45.], L0=[1.e5])
    else:
        p_atmos = Param_atmos(r0=0.16, nscreens=3, frac=[0.6, 0.25, 0.15],
                              alt=[0.0, 4500.0, 14000.0], windspeed=[15, 10, 20],
                              winddir=[0, 45, 90], L0=[1.e5, 1.e5, 1.e5])
    if noise:
        # single controller/target, DMs = [pzt, tt]; second WFS is a noise-free twin (unused by
        # controller 0)
        p_targets = [_target([0, 1])]
        p_wfss = [_wfs(nxsub, 9., ron, [0, 1]), _wfs(nxsub, 9., -1., [0, 1])]
        p_dms = [_pzt(nxsub + 1), _tt()]
        p_centroiders = [Param_centroider(nwfs=0, type="cog")]
        p_controllers = [Param_controller(type="ls", nwfs=[0], ndm=[0, 1], maxcond=1500.,
                                          delay=delay, gain=gain)]
    else:
        # DM list order is [pzt, pzt_geo, tt, tt_geo] (production_sh_10x10_2m.py:~105)
        p_targets = [_target([0, 2]), _target([1, 3])]
        p_wfss = [_wfs(nxsub, 4., ron, [0, 2]), _wfs(nxsub, 4., -1., [1, 3])]
        p_dms = [_pzt(nxsub + 1), _pzt(nxsub + 1), _tt(), _tt()]
        p_centroiders = [Param_centroider(nwfs=0, type="cog"),
                         Param_centroider(nwfs=1, type="cog")]
        p_controllers = [Param_controller(type="ls", nwfs=[0], ndm=[0, 2], maxcond=1500.,
                                          delay=delay, gain=gain),
                         Param_controller(type="geo", nwfs=[1], ndm=[1, 3], maxcond=1500.,
                                          delay=0., gain=gain)]
    ps = ParamSet(p_loop=p_loop, p_geom=p_geom, p_tel=p_tel, p_atmos=p_atmos,
                  p_targets=p_targets, p_wfss=p_wfss, p_dms=p_dms, p_centroiders=p_centroiders,
                  p_controllers=p_controllers, simul_name=name)
    return _normalise(ps).validate()
