"""Host time per step of the 40x40 configuration: a tiny batch makes the loop host-bound (development aid)."""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
for nenv in (8, 256):
    w = B.Workload(B.WORKLOAD, nenv, 0, 1, "cuda:0")
    w.reset()
    gc.collect(); gc.disable()
    for _ in range(30): w.one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): w.one_step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print("nenv %d: host %.1f us/step, wall %.1f us/step" % (nenv, th / 300 * 1e6, tt / 300 * 1e6), flush=True)
    gc.enable(); del w; gc.collect(); torch.cuda.synchronize()
