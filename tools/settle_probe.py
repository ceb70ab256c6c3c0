import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
S = torch.cuda.Stream(); torch.cuda.set_stream(S)
w = bench.Workload(bench.WORKLOAD, 256, 0, 1, "cuda:0")
w.reset()
print("probe", w.env.order_probe)
for rep in range(3):
    rs = w.time_reset()
    torch.cuda.synchronize()
    ts = []
    for k in range(12):
        t0 = time.perf_counter(); w.one_step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    for k in range(40): w.one_step()
    torch.cuda.synchronize(); t40 = (time.perf_counter() - t0) * 1e3
    print("reset %.1f ms; first steps (synchronised each): %s ; next 40 unsynchronised: %.2f ms" % (rs * 1e3, " ".join("%.2f" % t for t in ts), t40))
