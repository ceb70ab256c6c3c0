"""A/B timing of the one-pass frame kernel (development aid): AOMARL_LIB=<variant .so> python tools/fw_ab.py [nenv]
Times aomarl_frame_fused (no noise, COG, stack-array DM from the voltages) through the library's own
event pairs, plus the kernel's development switches (no loads / loads + amplitudes only / ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ao_marl_amd import params, geometry as G, system
from ao_marl_amd.sim import HipSim

name = "production_sh_40x40_8m_3layers"
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sysm = G.build_system(params.builtin(name))
s = system.from_system(sysm, strehl_halfwin=8)
s.cmat = np.zeros((s.nactu, s.nslope), dtype=np.float32)
sim = HipSim(s, nenv=nenv)
sim.reset(1234 + 16 * np.arange(nenv))
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


lib = os.environ.get("AOMARL_LIB", "default")
out = ["%-28s" % os.path.basename(lib)]
out.append("full %.4f ms" % t_frame())
for d, what in ((1, "no spot"), (2, "no psf"), (3, "loads+ampl"), (4, "no loads")):
    sim.set_option("fused_debug", d)
    out.append("%s %.4f" % (what, t_frame()))
sim.set_option("fused_debug", 0)
out.append("| noise+cube %.4f" % t_frame(write_bincube=True) if False else "")
print("  ".join(out), flush=True)
