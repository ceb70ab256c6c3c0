"""cProfile of the host side of one step of the small configuration (development aid)."""
import sys, os, time, gc, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
w = B.Workload(B.SMALL, 64, 0, 1, "cuda:0")
w.reset()
gc.collect(); gc.disable()
for _ in range(50): w.one_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): w.one_step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print("host %.1f us/step, wall %.1f us/step" % (th / 500 * 1e6, tt / 500 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(500): w.one_step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
