"""Run the hot stages a few times on a warm 256-env 40x40 system (target of rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ao_marl_amd import params, geometry as G, system, modal
from ao_marl_amd.sim import HipSim
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sysm = G.build_system(params.builtin("production_sh_40x40_8m_3layers")); s = system.from_system(sysm, strehl_halfwin=8)
cal = modal.calibrate(s, sysm, HipSim(s, nenv=512, keep_phase=True), nfilt=5)
sim = HipSim(s, nenv=nenv)
nm = cal.volts2modes.shape[0]
sim.set_modal(cal.volts2modes, cal.modes2volts, np.full(nm, 0.01, np.float32), np.arange(nm))
sim.reset(1234 + 16 * np.arange(nenv))
a = torch.zeros(nenv, nm, device="cuda")
for _ in range(5):
    sim.next_part_two(a); sim.next_part_one()
torch.cuda.synchronize()
for _ in range(reps):
    sim.next_part_two(a)          # rl_control GEMMs, delay line, (deferred) DM shapes, Strehl commit
    sim.next_part_one()           # extrusion, one-pass frame kernel, command GEMM
    sim.volts2modes(sim.com)
torch.cuda.synchronize()
print("done")
