import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
w = B.Workload(B.SMALL, 64, 0, 1, "cuda:0")
w.reset()
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); w.reset(); torch.cuda.synchronize()
    print("reset %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
for _ in range(20): w.one_step()
torch.cuda.synchronize(); t0 = time.perf_counter(); w.reset(); torch.cuda.synchronize()
print("reset after steps %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
for _ in range(20): w.one_step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); w.reset(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
