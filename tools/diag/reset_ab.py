"""Reset: fused scatter+gather rounds against separate launches -- same screens, time (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ao_marl_amd import params, geometry as G, system
from ao_marl_amd.sim import HipSim
nenv = 256
s = system.from_system(G.build_system(params.builtin("production_sh_40x40_8m_3layers")), strehl_halfwin=8)
s.cmat = np.zeros((s.nactu, s.nslope), dtype=np.float32)
out = {}
for unf in (1, 0):
    sim = HipSim(s, nenv=nenv)
    sim.set_option("reset_untransposed", unf)
    sim.reset(1234 + 16 * np.arange(nenv))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sim.reset(1234 + 16 * np.arange(nenv))
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    for _ in range(7): sim.move_atmos()
    torch.cuda.synchronize()
    out[unf] = (sim.t["screens"].clone(), sim.t["origin"].clone(), sim.t["ext_count"].clone())
    print("row-major reset  " if unf else "transposed reset ", "reset %.1f ms (host %.1f ms)" % (t * 1e3, th * 1e3), flush=True)
    del sim
print("screens equal:", torch.equal(out[0][0], out[1][0]), " origins:", torch.equal(out[0][1], out[1][1]), " counters:", torch.equal(out[0][2], out[1][2]))
