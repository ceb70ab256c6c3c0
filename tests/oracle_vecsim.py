"""OracleVecSim: the HipSim interface over the CPU oracle (torch CPU tensors, a Python loop over
environments).  TEST INFRASTRUCTURE: lets tests/ drive ao_marl_amd.env (host logic) without a GPU
and check it against traces of the reference's own Python; never used by the product."""
import numpy as np
import torch

from oracle import aoref


class _Sim(aoref.OracleSim):
    def __init__(self, s, lazy):
        self._lazy = lazy
        aoref.OracleSim.__init__(self, s, seed=0)

    def reset(self, seed, grown=None):
        if self._lazy:              # construction: no screen generation yet
            self.seed, self.frame = int(seed), 0
            self.accumx = np.zeros(self.s.nscreens, dtype=np.float32)
            self.accumy = np.zeros(self.s.nscreens, dtype=np.float32)
            self.ext_count = [0] * self.s.nscreens
            self._alloc_ctrl()
            self.reset_strehl()
            self._lazy = False
            return
        aoref.OracleSim.reset(self, seed, grown)


class OracleVecSim(object):
    def __init__(self, s, nenv, device="cpu", keep_bincube=False, keep_phase=False):
        self.s, self.nenv, self.device = s, int(nenv), torch.device("cpu")
        self._build()

    def _build(self):
        self.sims = [_Sim(self.s, True) for _ in range(self.nenv)]
        self.v2m = self.m2v = self.freedom = self.amodes = None

    # configuration
    def set_cmat(self, cmat):
        self.s.cmat = np.ascontiguousarray(cmat, dtype=np.float32)

    def set_gain(self, g):
        self.s.gain = float(g)

    def set_env_gains(self, gains):
        self.env_gains = None if gains is None else np.asarray(gains, dtype=np.float32).reshape(-1)

    def _with_gain(self, e, fn):
        if getattr(self, "env_gains", None) is None:
            return fn()
        g0, self.s.gain = self.s.gain, float(self.env_gains[e])
        try:
            return fn()
        finally:
            self.s.gain = g0

    def set_modal(self, v2m, m2v, freedom=None, action_modes=None):
        self.v2m, self.m2v = np.asarray(v2m, np.float32), np.asarray(m2v, np.float32)
        self.nmodes = self.v2m.shape[0]
        self.freedom = None if freedom is None else np.asarray(freedom, np.float32)
        self.amodes = None if action_modes is None else np.asarray(action_modes) % self.nmodes
        self.nact = 0 if self.amodes is None else int(self.amodes.size)

    def reload_dms(self):
        self._build()

    # state views
    def _stack(self, name):
        return torch.from_numpy(np.stack([getattr(o, name) for o in self.sims]).astype(np.float32))

    com = property(lambda self: self._stack("com"))
    err = property(lambda self: self._stack("err"))
    voltage = property(lambda self: self._stack("voltage"))
    slopes = property(lambda self: self._stack("slopes"))

    @property
    def strehl(self):
        return torch.tensor([o.get_strehl() for o in self.sims], dtype=torch.float32)

    def target_image(self, retrace=True):
        """Target.get_tar_image("se") of every environment from the oracle: the full |FFT2|^2 of the science phase,
        centred.  retrace: of the state as it stands (atmosphere + mirrors, like aomarl_target_image); False: of the
        phase the oracle's last raytrace_target left (COMPASS's d_phase, what comp_tar_image sees)."""
        import ctypes as C
        out = []
        for o in self.sims:
            if retrace:
                o.raytrace_target()
            s = o.s
            full = np.zeros((s.npsf, s.npsf), dtype=np.float32)
            pf, pw = C.c_float(0), C.c_float(0)
            o.L.aoref_psf(o.tar_phase.reshape(-1), s.spupil.reshape(-1), s.pupdiam, s.npsf, s.tar_lambda, s.strehl_halfwin,
                          full.ctypes.data_as(C.c_void_p), None, C.byref(pf), C.byref(pw))
            out.append(np.fft.fftshift(full))
        return torch.from_numpy(np.stack(out))

    @property
    def strehl_fit(self):
        return torch.tensor([o.get_strehl(do_fit=True) for o in self.sims], dtype=torch.float32)

    # per-frame API
    def reset(self, seeds):
        seeds = np.broadcast_to(np.asarray(seeds), (self.nenv,))
        for o, sd in zip(self.sims, seeds):
            o.reset(int(sd))

    # run-time wind / r0 (the HipSim surface of the same names)
    def layer_values(self, layer):
        o = self.sims[0]
        return float(o.deltax[layer]), float(o.deltay[layer]), float(o.amplitude[layer])

    def atmos_change_blocked(self):
        return None                                                  # call-by-call order: nothing runs ahead

    def set_wind(self, layer, deltax, deltay, mirror_stencils=True):
        for o in self.sims:
            o.set_wind(layer, deltax, deltay, mirror_stencils)
        return False

    def set_stencil(self, layer, axis, istencil):
        for o in self.sims:
            o.set_stencil(layer, axis, istencil)

    def set_amplitudes(self, amplitude):
        for o in self.sims:
            o.set_amplitudes(amplitude)
        return False

    def rl_control(self, action):
        a = np.asarray(action, dtype=np.float32)
        for e, o in enumerate(self.sims):
            m = self.v2m.dot(o.com)                                  # rlSupervisor.py:800
            m[self.amodes] += a[e] * self.freedom[self.amodes]       # :813
            o.set_com(self.m2v.dot(m))                               # :816, :733

    def apply_control(self, defer_shape=False):
        for o in self.sims:
            o.apply_control()

    def comp_strehl(self):
        for o in self.sims:
            o.comp_strehl()

    def next_part_one(self):
        for e, o in enumerate(self.sims):
            self._with_gain(e, o.next_part_one)

    def move_atmos(self):
        for o in self.sims:
            o.move_atmos()

    def target_psf(self):
        for o in self.sims:
            o.raytrace_target()

    def comp_image(self, noise=True, cog=True, **kw):
        for o in self.sims:
            o.raytrace_wfs(atm=True, dms=False, reset=True)
            o.raytrace_wfs(atm=False, dms=True, reset=False)
            o.comp_image(noise=noise)
            if cog:
                o.do_centroids()

    def do_control(self):
        for e, o in enumerate(self.sims):
            self._with_gain(e, o.do_control)

    def volts2modes(self, vec):
        return torch.from_numpy(np.asarray(vec, dtype=np.float32) @ self.v2m.T)

    # the sensor's phase, as HipSim exposes it: raytrace_wfs fills t["wfs_phase"]
    def raytrace_wfs(self, atm=True, dms=True, reset=True, env_begin=0, env_count=None):
        n = self.nenv if env_count is None else env_count
        for o in self.sims[env_begin:env_begin + n]:
            o.raytrace_wfs(atm=atm, dms=dms, reset=reset)

    @property
    def t(self):
        return {"wfs_phase": self._stack("wfs_phase")}

    def comp_dm_shape(self, volts, env_begin=0, env_count=None):
        v = np.asarray(volts, dtype=np.float32)
        n = self.nenv if env_count is None else env_count
        for i, o in enumerate(self.sims[env_begin:env_begin + n]):
            o.comp_shapes(v[i])

    def dm_response(self, commands, geometric):
        return self.sims[0].dm_response(np.ascontiguousarray(commands, dtype=np.float32), geometric)
