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
print("spot              %.3f ms" % timeit(lambda: sim.comp_image(noise=False, cog=True)))
print("spot noise        %.3f ms" % timeit(lambda: sim.comp_image(noise=True, cog=True)))
print("target_psf        %.3f ms" % timeit(lambda: sim.target_psf()))
print("frame_fused       %.3f ms" % timeit(lambda: sim.frame_fused(noise=False, cog=True, dm_from_voltage=False)))
print("frame_fused otf   %.3f ms" % timeit(lambda: sim.frame_fused(noise=False, cog=True, dm_from_voltage=True)))
for d, what in ((1, "no spot"), (2, "no psf mfma"), (3, "loads+amplitudes only"), (4, "no loads"), (7, "amplitudes only")):
    sim.set_option("fused_debug", d)
    print("frame_fused otf dbg=%d (%s) %.3f ms" % (d, what, timeit(lambda: sim.frame_fused(noise=False, cog=True, dm_from_voltage=True))))
sim.set_option("fused_debug", 0)
print("frame_fused noise %.3f ms" % timeit(lambda: sim.frame_fused(noise=True, cog=True)))
print("dm_shape          %.3f ms" % timeit(lambda: sim.comp_dm_shape()))
print("move_atmos        %.3f ms" % timeit(lambda: sim.move_atmos()))
print("do_control        %.3f ms" % timeit(lambda: sim.do_control()))
a = torch.zeros(nenv, nm, device="cuda")
print("rl_control        %.3f ms" % timeit(lambda: sim.rl_control(a)))
print("volts2modes       %.3f ms" % timeit(lambda: sim.volts2modes(sim.com)))
print("part_two deferred %.3f ms" % timeit(lambda: sim.next_part_two(None)))
print("part_one fused    %.3f ms" % timeit(lambda: sim.next_part_one()))
sim.defer_shape = False
sim.set_option("force_unfused_frame", 1)
print("part_one unfused  %.3f ms" % timeit(lambda: sim.next_part_one()))
sim.set_option("force_unfused_frame", 0)
print("part_two          %.3f ms" % timeit(lambda: sim.next_part_two(None)))

for (M, N, K) in ((256, 1296, 2624), (256, 1286, 2400), (256, 1276, 1288)):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); Cc = torch.zeros(M, N, device="cuda")
    t = timeit(lambda: sim.gemm_nt(A, B, Cout=Cc))
    print("gemm no-split %dx%dx%d  %.1f us  %.1f TFLOP/s" % (M, N, K, t * 1e3, 2e-9 * M * N * K / t))
for tb in (256, 512, 768, 1024):
    sim.set_option("gemm_target_blocks", tb)
    print("target_blocks=%d: do_control %.1f us  rl_control %.1f us  v2m %.1f us  move_atmos %.1f us" % (
        tb, 1e3 * timeit(lambda: sim.do_control()), 1e3 * timeit(lambda: sim.rl_control(a)),
        1e3 * timeit(lambda: sim.volts2modes(sim.com)), 1e3 * timeit(lambda: sim.move_atmos())))
sim.set_option("gemm_target_blocks", 512)
