"""Reset and one-step extrusion with and without the XCD block map of the split-f16 GEMM -- same screens, time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ao_marl_amd import params, geometry as G, system, libaomarl as la
from ao_marl_amd.sim import HipSim
lib = la.load()
nenv = 256
s = system.from_system(G.build_system(params.builtin("production_sh_40x40_8m_3layers")), strehl_halfwin=8)
s.cmat = np.zeros((s.nactu, s.nslope), dtype=np.float32)
out = {}
sim = HipSim(s, nenv=nenv)
for x in (0, 1, 0, 1):
    la.check(lib.aomarl_set_option(None, b"gemm_xcd_map", x))
    sim.reset(1234 + 16 * np.arange(nenv))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sim.reset(1234 + 16 * np.arange(nenv))
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): sim.move_atmos()
    torch.cuda.synchronize(); tm = (time.perf_counter() - t0) / 50
    out[x] = sim.t["screens"].clone()
    print("xcd map %d: reset %.1f ms, move_atmos %.1f us" % (x, t * 1e3, tm * 1e6), flush=True)
print("screens equal:", torch.equal(out[0], out[1]))
