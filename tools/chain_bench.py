#!/usr/bin/env python3
"""Stand-alone timings of the small kernels around the frame kernel, per precision mode (development aid):
do_control (cmat GEMM + integrate), volts2modes (v2m GEMM), one extrusion round over all layers
(gather + [A|B] GEMM + scatter), the full reset.  production 40x40, 256 environments.
    python tools/chain_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ao_marl_amd import geometry as G, libaomarl as la, params, system  # noqa: E402
from ao_marl_amd.sim import HipSim  # noqa: E402

nenv = 256
sysm = G.build_system(params.builtin("production_sh_40x40_8m_3layers"))
s = system.from_system(sysm, strehl_halfwin=8)
rng = np.random.default_rng(0)
s.cmat = (rng.standard_normal((s.nactu, s.nslope)) * 0.1).astype(np.float32)
sim = HipSim(s, nenv=nenv)
nm = s.nactu - 3
sim.set_modal((rng.standard_normal((nm, s.nactu)) * 1e-2).astype(np.float32),
              (rng.standard_normal((s.nactu, nm)) * 1e-2).astype(np.float32))
sim.t["slopes"].normal_(0, 0.1)


def timeit(fn, n=50):
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


vec = torch.randn(nenv, sim.ld_actu, device="cuda")[:, :s.nactu]
out = torch.empty(nenv, nm, device="cuda")
for mode, tb in (("f32", 0), ("f32", 512), ("split_f16", 0), ("split_f16", 512)):
    la.set_precision(mode)
    sim.set_option("gemm_target_blocks", tb)
    print("---- %s, split-K rule: %s" % (mode, "cost model" if tb == 0 else "about %d blocks" % tb))
    sim.reset(1234 + 16 * np.arange(nenv))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.reset(1234 + 16 * np.arange(nenv))
    torch.cuda.synchronize()
    t_reset = (time.perf_counter() - t0) * 1e3
    gf_ctrl = 2e-9 * nenv * s.nactu * s.nslope
    gf_v2m = 2e-9 * nenv * nm * s.nactu
    K = s.screen_dim[0] + int(s.istx[0].size)
    gf_ext = 2e-9 * nenv * 3 * s.screen_dim[0] * K
    us = timeit(lambda: sim.do_control())
    print("%-9s do_control (256 x %d x %d + integrate)   %6.1f us  %5.1f TFLOP/s" % (mode, s.nactu, s.nslope, us, gf_ctrl / us * 1e3))
    us = timeit(lambda: sim.volts2modes(vec, out=out))
    print("%-9s volts2modes (256 x %d x %d)               %6.1f us  %5.1f TFLOP/s" % (mode, nm, s.nactu, us, gf_v2m / us * 1e3))
    us1 = timeit(lambda: sim.extrude([2], [-1]))
    print("%-9s extrusion round, 1 layer  (256 x %d x %d) %6.1f us" % (mode, s.screen_dim[0], K, us1))
    us = timeit(lambda: sim.extrude([0, 1, 2], [-2, -1, -1]))
    print("%-9s extrusion round, 3 layers (768 x %d x %d) %6.1f us  %5.1f TFLOP/s (GEMM flops / whole round)" %
          (mode, s.screen_dim[0], K, us, gf_ext / us * 1e3))
    print("%-9s reset (2 x %d rounds)                      %6.1f ms  %5.1f TFLOP/s" %
          (mode, s.screen_dim[0], t_reset, gf_ext * 2 * s.screen_dim[0] / t_reset))
la.set_precision("f32")
sim.set_option("gemm_target_blocks", 0)
