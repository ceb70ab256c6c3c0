"""A/B timing of the one-pass frame kernel (development aid): AOMARL_LIB=<variant .so> python tools/fw_ab.py [nenv]
Times aomarl_frame_fused (no noise, COG, stack-array DM from the voltages) through the library's own
event pairs in both arithmetics, plus the kernel's development switches ("fused_debug": 1 no WFS path, 2 no PSF rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ao_marl_amd import params, geometry as G, system, libaomarl as la
from ao_marl_amd.sim import HipSim

name = "production_sh_40x40_8m_3layers"
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sysm = G.build_system(params.builtin(name))
s = system.from_system(sysm, strehl_halfwin=8)
s.cmat = np.zeros((s.nactu, s.nslope), dtype=np.float32)
sim = HipSim(s, nenv=nenv)
sim.reset(1234 + 16 * np.arange(nenv))
for _ in range(37):                      # ring origins off their reset value (window alignment as in a running loop)
    sim.move_atmos()
v = torch.randn(nenv, s.nactu, device="cuda") * 0.5
sim.t["voltage"][:, :s.nactu] = v


def t_frame(reps=30, **kw):
    for _ in range(5):
        sim.frame_fused(noise=False, cog=True, dm_from_voltage=True, **kw)
    torch.cuda.synchronize()
    sim.set_option("time_frame_kernel", reps)
    for _ in range(reps):
        sim.frame_fused(noise=False, cog=True, dm_from_voltage=True, **kw)
    tot, n = sim.frame_kernel_time()
    sim.set_option("time_frame_kernel", 0)
    return tot / n


for prec in ("f32", "split_f16"):
    la.set_precision(prec)
    out = ["%-10s" % prec]
    for d, what in ((0, "full"), (0, "full"), (1, "no WFS path"), (2, "no PSF rows")):
        sim.set_option("fused_debug", d)
        out.append("%s %.4f ms" % (what, t_frame()))
    sim.set_option("fused_debug", 0)
    print(" | ".join(out), flush=True)
la.set_precision("f32")
