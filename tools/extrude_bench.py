"""One extrusion round (gather, product, scatter: three launches) alone, by direction of its operations (development aid):
column-type operations (x moves) gather and scatter one element per row of the screen, row-type ones (y moves) whole lines.
   python tools/extrude_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ao_marl_amd import params, geometry as G, system
from ao_marl_amd.sim import HipSim

nenv = 256
sysm = G.build_system(params.builtin("production_sh_40x40_8m_3layers"))
s = system.from_system(sysm, strehl_halfwin=8)
s.cmat = np.zeros((s.nactu, s.nslope), dtype=np.float32)
sim = HipSim(s, nenv=nenv)
sim.reset(1234 + 16 * np.arange(nenv))


def timeit(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, layers, dirs in (("3 layers, all row-type (y moves)", [0, 1, 2], [-2, -2, -2]),
                           ("3 layers, all column-type (x moves)", [0, 1, 2], [-1, -1, -1]),
                           ("3 layers as a frame's first round (y, x, x)", [0, 1, 2], [-2, -1, -1]),
                           ("1 layer, row-type", [2], [-2]), ("1 layer, column-type", [2], [-1])):
    print("%-48s %6.1f us per round" % (name, timeit(lambda: sim.extrude(layers, dirs))), flush=True)
