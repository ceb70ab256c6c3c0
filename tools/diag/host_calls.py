"""Host time of the individual library calls of a 10x10 step (development aid)."""
import sys, os, time, gc, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
from ao_marl_amd import libaomarl as la
w = B.Workload(B.SMALL, 64, 0, 1, "cuda:0")
w.reset()
for _ in range(10): w.one_step()
sim = w.sim
torch.cuda.synchronize()
gc.collect(); gc.disable()
def t(name, fn, n=300):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%-22s host %.1f us" % (name, th / n * 1e6), flush=True)
a, _ = w.policy.select_action(w.state)
t("move_atmos", lambda: sim.move_atmos())
sim.set_option("prefetch_atmos", 0)
t("move_atmos (no prefetch pending)", lambda: sim.move_atmos())
t("frame_fused", lambda: sim.frame_fused(noise=True, cog=True))
t("do_control", lambda: sim.do_control())
t("apply_control", lambda: sim.apply_control(defer_shape=True))
t("comp_strehl", lambda: sim.comp_strehl())
t("volts2modes", lambda: sim.volts2modes(sim.err))
t("rl_control", lambda: sim.rl_control(a))
t("select_action", lambda: w.policy.select_action(w.state))
sim.set_option("prefetch_atmos", 1)
t("env.step", lambda: w.env.step(a))
