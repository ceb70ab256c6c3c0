import sys, os, time, cProfile, pstats, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
w = B.Workload(B.WORKLOAD, 256, 0, 1, "cuda:0")
w.reset()
for _ in range(30): w.one_step()
torch.cuda.synchronize()
if len(sys.argv) > 1:
    from ao_marl_amd.sac import BatchedSAC
    r = B.sac_update_rate(w.layout, "cuda:0"); print("sac", r, flush=True)
del w; gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
w = B.Workload(B.SMALL, 64, 0, 1, "cuda:0")
w.reset()
for i in range(3):
    pr = cProfile.Profile(); pr.enable()
    torch.cuda.synchronize(); t0 = time.perf_counter(); w.reset(); torch.cuda.synchronize()
    pr.disable()
    print("reset %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    if i == 0: pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
