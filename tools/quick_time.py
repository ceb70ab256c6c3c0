"""Ad-hoc timing of the half-frame composites on the GPU (development aid, not the bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ao_marl_amd import params, geometry as G, system, modal
from ao_marl_amd.sim import HipSim
name = sys.argv[1] if len(sys.argv) > 1 else "production_sh_40x40_8m_3layers"
nenv = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
t = time.time(); sysm = G.build_system(params.builtin(name)); s = system.from_system(sysm, strehl_halfwin=8); print("geometry %.1fs" % (time.time() - t), flush=True)
t = time.time(); cal_sim = HipSim(s, nenv=min(512, 1500), keep_phase=True)
cal = modal.calibrate(s, sysm, cal_sim, nfilt=5, verbose=True); del cal_sim
print("calibration %.1fs nactu %d" % (time.time() - t, s.nactu), flush=True)
sim = HipSim(s, nenv=nenv)
nm = cal.volts2modes.shape[0]
modes = np.r_[np.arange(0, nm - 2 - 5 if nm > 200 else 80), nm - 2, nm - 1]
sim.set_modal(cal.volts2modes, cal.modes2volts, np.full(nm, 0.01, np.float32), modes)
t = time.time(); sim.reset(1234 + 16 * np.arange(nenv)); torch.cuda.synchronize(); print("reset %.2fs" % (time.time() - t), flush=True)
act = torch.zeros(nenv, modes.size, device="cuda")
def ev():
    return torch.cuda.Event(enable_timing=True)
names = ["part_two(rl+apply+strehl)", "move_atmos", "target_psf", "wfs image+cog", "do_control"]
acc = np.zeros(len(names))
for it in range(steps + 3):
    e = [ev() for _ in range(len(names) + 1)]
    e[0].record(); sim.next_part_two(act)
    e[1].record(); sim.move_atmos()
    e[2].record(); sim.target_psf()
    e[3].record(); sim.comp_image(noise=True, cog=True)
    e[4].record(); sim.do_control()
    e[5].record(); torch.cuda.synchronize()
    if it >= 3:
        acc += np.array([e[i].elapsed_time(e[i + 1]) for i in range(len(names))])
acc /= steps
for n_, a in zip(names, acc): print("  %-28s %8.3f ms" % (n_, a))
print("total %.3f ms / batch step -> %.0f env-steps/s ; SR %s" % (acc.sum(), nenv / acc.sum() * 1e3, sim.strehl[:3, 0].cpu().numpy()), flush=True)
t = time.time()
for it in range(steps):
    sim.next_part_two(act); sim.next_part_one()
torch.cuda.synchronize(); dt = (time.time() - t) / steps
print("composite wall: %.3f ms/step -> %.0f env-steps/s" % (dt * 1e3, nenv / dt))
