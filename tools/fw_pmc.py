"""Frame kernel alone, for rocprofv3 --pmc passes (development aid): python3 tools/fw_pmc.py [nenv] [dbg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ao_marl_amd import params, geometry as G, system
from ao_marl_amd.sim import HipSim
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dbg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sysm = G.build_system(params.builtin("production_sh_40x40_8m_3layers"))
s = system.from_system(sysm, strehl_halfwin=8)
s.cmat = np.zeros((s.nactu, s.nslope), dtype=np.float32)
sim = HipSim(s, nenv=nenv)
sim.reset(1234 + 16 * np.arange(nenv))
for _ in range(37):                      # ring origins off their reset value (a reset starts every ring on a 128-byte line;
    sim.move_atmos()                     # in a running loop only the layers without wind along x stay there)
sim.t["voltage"][:, :s.nactu] = torch.randn(nenv, s.nactu, device="cuda") * 0.5
sim.set_option("fused_debug", dbg)
for _ in range(6):
    sim.frame_fused(noise=False, cog=True, dm_from_voltage=True)
torch.cuda.synchronize()
open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fw_kernel_name.txt"), "w").write(sim.frame_kernel_name())
