"""The GPU's full 40x40 reset against the oracle's, per GEMM kernel and precision (development aid; what bounds
tests/test_gpu_env_step_large.py's reset check): max and rms difference of the screens, and how it grows with
the number of extrusion rounds (a reset of a SMALLER screen is not available, so the growth is read off rows:
row r of the final transposed screen was written by round ~r).   python tools/reset_parity_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ao_marl_amd import geometry as G, libaomarl as la, params, system
from ao_marl_amd.sim import HipSim
from oracle import aoref

sysm = G.build_system(params.builtin("production_sh_40x40_8m_3layers"))
s = system.from_system(sysm, strehl_halfwin=8)
seeds = [1234, 1250]
o = [aoref.OracleSim(s, seed=sd) for sd in seeds]
want = [np.stack([x.screens[l] for x in o]) for l in range(s.nscreens)]
for mode in ("f32", "split_f16"):
    la.set_precision(mode)
    for bal in (1,):            # (profiles/r04_reset_parity.txt also holds round 3's k_gemm_nt2, removed since: gemm_balanced=0)
        sim = HipSim(s, nenv=len(seeds))
        sim.reset(seeds)
        line = "%-9s gemm_balanced=%d:" % (mode, bal)
        for l in range(s.nscreens):
            d = sim.screen(l).cpu().numpy() - want[l]
            line += "  layer %d max %.2e rms %.2e (screen rms %.2f)" % (l, np.abs(d).max(), d.std(), want[l].std())
        print(line, flush=True)
        del sim
la.set_precision("f32")
