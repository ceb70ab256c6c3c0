"""Stage-level timing with option sweeps on the GPU (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ao_marl_amd import params, geometry as G, system, modal
from ao_marl_amd.sim import HipSim
name = "production_sh_40x40_8m_3layers"
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sysm = G.build_system(params.builtin(name)); s = system.from_system(sysm, strehl_halfwin=8)
cal = modal.calibrate(s, sysm, HipSim(s, nenv=512, keep_phase=True), nfilt=5)
sim = HipSim(s, nenv=nenv)
nm = cal.volts2modes.shape[0]
sim.set_modal(cal.volts2modes, cal.modes2volts, np.full(nm, 0.01, np.float32), np.arange(nm))
sim.reset(1234 + 16 * np.arange(nenv))
for _ in range(5):
    sim.next_part_two(None); sim.next_part_one()
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("spot default     %.3f ms" % timeit(lambda: sim.comp_image(noise=False, cog=True)))
for g in (4, 8, 16, 32, 64, 150, 300):
    sim.set_option("spot_blocks_per_env", g)
    print("spot gx=%-4d      %.3f ms" % (g, timeit(lambda: sim.comp_image(noise=False, cog=True))))
sim.set_option("spot_blocks_per_env", 0)
for pad in (0, 30000, 45000, 70000, 140000):
    sim.set_option("spot_lds_pad", pad)
    print("spot lds_pad=%-6d %.3f ms" % (pad, timeit(lambda: sim.comp_image(noise=False, cog=True))))
sim.set_option("spot_lds_pad", 0)
print("spot no_atmos     %.3f ms" % timeit(lambda: sim.comp_image(noise=False, cog=True, atm=False)))
print("spot no_dms       %.3f ms" % timeit(lambda: sim.comp_image(noise=False, cog=True, dms=False)))
print("spot none         %.3f ms" % timeit(lambda: sim.comp_image(noise=False, cog=True, atm=False, dms=False)))
print("target_psf        %.3f ms" % timeit(lambda: sim.target_psf()))
print("dm_shape          %.3f ms" % timeit(lambda: sim.comp_dm_shape()))
print("move_atmos        %.3f ms" % timeit(lambda: sim.move_atmos()))
print("do_control        %.3f ms" % timeit(lambda: sim.do_control()))
a = torch.zeros(nenv, nm, device="cuda")
print("rl_control        %.3f ms" % timeit(lambda: sim.rl_control(a)))
print("volts2modes       %.3f ms" % timeit(lambda: sim.volts2modes(sim.com)))

print("target+wfs serial  %.3f ms" % timeit(lambda: (sim.target_psf(), sim.comp_image(noise=False, cog=True))))
print("target||wfs        %.3f ms" % timeit(lambda: sim.target_and_wfs(noise=False)))
