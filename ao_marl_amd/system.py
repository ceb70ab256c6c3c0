"""SimArrays: the flat, backend-neutral description of ONE control path of the AO system
(controller 0 of the reference's parameter file: its WFS, its DMs, its science target) -- exactly
the arrays and scalars the reference hands to the native library at init
(shesha/init/*_init.py -> sutraWrap constructors / load_arrays).

The HIP library (ao_marl_amd/csrc) uploads these once; the per-frame kernels read nothing else.
"""
import math

import numpy as np

from . import geometry as G


class SimArrays(object):
    """Plain attribute container; see `from_system` for the fields."""

    def __repr__(self):
        return "SimArrays(%s)" % ", ".join(sorted(self.__dict__))


def psf_fft_size(pupdiam):
    """Support of the science PSF, 2^(floor(log2(2*pupdiam))+1) (512 / 2048 for the two
    production pupils; SURVEY.md Appendix A, last row -- upstream semantics)."""
    return int(2**(int(math.floor(math.log2(2 * pupdiam))) + 1))


def from_system(sysm, ncontrol=0, strehl_halfwin=16):
    """Flatten `geometry.build_system()` output for controller `ncontrol`."""
    ps = sysm.params
    ctl = ps.p_controllers[ncontrol]
    if ctl.type != "ls":
        raise NotImplementedError("only the LS (integrator) controller path is on the hot path; "
                                  "the GEO reference controller is SURVEY.md section 8(f) item 1")
    if len(ctl.nwfs) != 1:
        raise NotImplementedError("one WFS per controller")
    iw = int(ctl.nwfs[0])
    w, g, a = sysm.wfss[iw], sysm.geom, sysm.atm
    s = SimArrays()
    s.name = ps.simul_name
    # pupil
    s.n, s.pupdiam = g.n, g.pupdiam
    s.mpupil = np.ascontiguousarray(g.mpupil, dtype=np.float32)
    s.spupil = np.ascontiguousarray(g.spupil, dtype=np.float32)
    # WFS
    s.nvalid, s.pdiam, s.nfft, s.npix, s.nrebin = w.nvalid, w.pdiam, w.Nfft, w.npix, w.nrebin
    s.nxsub = w.nxsub
    s.phasemap = np.ascontiguousarray(w.phasemap, dtype=np.int32)          # [pdiam^2][nvalid]
    s.halfxy = np.ascontiguousarray(w.halfxy, dtype=np.float32)
    s.binmap = np.ascontiguousarray(w.binmap, dtype=np.int32)              # [nrebin^2][npix^2]
    s.flux = np.ascontiguousarray(w.fluxPerSub_valid, dtype=np.float32)
    s.nphot = np.float32(w.nphotons)
    s.wfs_lambda = float(w.Lambda)
    s.noise = float(w.noise)
    s.validsubsx = np.ascontiguousarray(w.validsubsx, dtype=np.int32)
    s.validsubsy = np.ascontiguousarray(w.validsubsy, dtype=np.int32)
    s.subapd = float(w.subapd)
    s.cog_offset = float(w.npix // 2. - 0.5)      # rtc_init.py:208
    s.cog_scale = float(w.pixsize)                # rtc_init.py:217
    s.nslope = 2 * w.nvalid
    # atmosphere
    s.nscreens = len(a.dim_screens)
    s.screen_dim = [int(d) for d in a.dim_screens]
    s.deltax = np.asarray(a.deltax, dtype=np.float32)
    s.deltay = np.asarray(a.deltay, dtype=np.float32)
    # screens are generated in microns: r0_layers^(-5/6) [rad @ 0.5 um] * 0.5 / (2 pi)
    s.amplitude = (a.r0_layers.astype(np.float64)**(-5. / 6.) * 0.5 / (2 * np.pi)).astype(
            np.float32)
    s.A, s.B, s.istx, s.isty = [], [], [], []
    for l in range(s.nscreens):
        A, B, ix, iy = G.extrusion_for_layer(a, l)
        s.A.append(A)
        s.B.append(B)
        s.istx.append(ix)
        s.isty.append(iy)
    s.wfs_atm_off = [tuple(o) for o in w.atm_off]
    s.tar_atm_off = [tuple(o) for o in sysm.targets[ncontrol].atm_off]
    # DMs of this controller, in controller order (pzt first, TT last)
    s.dm_index = [int(k) for k in ctl.ndm]
    s.dms = [sysm.dms[k] for k in s.dm_index]
    s.wfs_dm_off = [tuple(w.dm_off[k]) for k in s.dm_index]
    s.tar_dm_off = [tuple(sysm.targets[ncontrol].dm_off[k]) for k in s.dm_index]
    s.nactu = int(sum(d.ntotact for d in s.dms))
    # target
    t = sysm.targets[ncontrol]
    s.tar_lambda = float(t.Lambda)
    s.npsf = psf_fft_size(g.pupdiam)
    s.strehl_halfwin = int(strehl_halfwin)
    # controller
    s.delay = float(ctl.delay)
    s.gain = float(ctl.gain)
    s.cmat = None   # filled by modal.calibrate()
    return s


def refresh_dms(s):
    """Recount actuators after `correct_dm` filtered the stack-array mirrors."""
    s.nactu = int(sum(d.ntotact for d in s.dms))
    return s


def dm_command_slices(s):
    out, c = [], 0
    for d in s.dms:
        out.append((c, c + d.ntotact))
        c += d.ntotact
    return out
