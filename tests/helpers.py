"""Shared test helpers: systems calibrated with the CPU ORACLE as the calibration backend (the
oracle is test infrastructure; the product calibrates through the HIP backend)."""
import copy
import functools

import numpy as np

from ao_marl_amd import geometry as G
from ao_marl_amd import modal, params, system
from oracle import aoref


class CalSim(aoref.OracleSim):
    """OracleSim without the (slow) screen generation at construction: calibration only."""

    def reset(self, seed):
        self.seed = seed
        self.frame = 0
        self._alloc_ctrl()
        self.reset_strehl()


class OracleBackend(object):
    def __init__(self, s):
        self.s = s
        self.sim = CalSim(s)

    def dm_response(self, commands, geometric):
        return self.sim.dm_response(np.ascontiguousarray(commands, dtype=np.float32), geometric)

    def reload_dms(self):
        self.sim = CalSim(self.s)


@functools.lru_cache(maxsize=4)
def _calibrated(name, nfilt, hw):
    sysm = G.build_system(params.builtin(name))
    s = system.from_system(sysm, strehl_halfwin=hw)
    cal = modal.calibrate(s, sysm, OracleBackend(s), nfilt=nfilt)
    return sysm, s, cal


def calibrated(name="production_sh_10x10_2m", nfilt=5, hw=8):
    """(sysm, SimArrays, Calibration), deep-copied so tests may mutate them."""
    return copy.deepcopy(_calibrated(name, nfilt, hw))


def uncalibrated(name="production_sh_10x10_2m", hw=8):
    sysm = G.build_system(params.builtin(name))
    return sysm, system.from_system(sysm, strehl_halfwin=hw)


def calibrated_hip(name, nfilt=5, hw=8):
    """(sysm, SimArrays, Calibration) calibrated through the HIP backend (GPU tests of the large system: the oracle
    backend needs minutes for its 1286 actuators; memoised by modal.calibrate)."""
    from ao_marl_amd.sim import HipSim
    sysm = G.build_system(params.builtin(name))
    s = system.from_system(sysm, strehl_halfwin=hw)
    cal = modal.calibrate(s, sysm, lambda: HipSim(s, nenv=512, keep_phase=True), nfilt=nfilt,
                          backend_id=HipSim.calibration_id())
    return sysm, s, cal
