"""What would line-aligned layer fetches be worth?  (development aid)  With a row pitch that is a multiple of 128 bytes
(a library build with -DRING_PAD=56: 648 + 56 = 704 floats) and ring origins chosen so that every pair of tiles starts on a
128-byte line, the pair walk's pieces ARE whole lines: the frame kernel alone and the HBM-side bytes in that state against
the same build at arbitrary origins.   AOMARL_LIB=<pad56 build> python tools/aligned_probe.py"""
import os, sys
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
sim.t["voltage"][:, :s.nactu] = torch.randn(nenv, s.nactu, device="cuda") * 0.5
tox = [int(round(o[0])) for o in s.tar_atm_off]
dims = list(s.screen_dim)


def t_frame(reps=30):
    for _ in range(5):
        sim.frame_fused(noise=False, cog=True, dm_from_voltage=True)
    torch.cuda.synchronize()
    sim.set_option("time_frame_kernel", reps)
    for _ in range(reps):
        sim.frame_fused(noise=False, cog=True, dm_from_voltage=True)
    tot, n = sim.frame_kernel_time()
    sim.set_option("time_frame_kernel", 0)
    return tot / n


org = sim.t["origin"]
rng = np.random.default_rng(0)
for name in ("random origins", "aligned origins", "random origins", "aligned origins"):
    o = org.cpu().numpy().copy()
    for l in range(len(dims)):
        if name.startswith("aligned"):
            o[:, l, 0] = (-tox[l]) % 32                       # (tox + origin_x) % 32 == 0: pairs start on a line
        else:
            o[:, l, 0] = rng.integers(0, dims[l], size=nenv)
        o[:, l, 1] = rng.integers(0, dims[l], size=nenv)
    org.copy_(torch.as_tensor(o))
    print("%-16s frame kernel %.4f ms  (layer offsets %s)" % (name, t_frame(), tox), flush=True)
