"""Per-step period behind a reset (development aid): the time between consecutive states becoming ready on the
caller's stream, in bins of 10 steps, pipelined and plain order.   python tools/transient_probe.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S = torch.cuda.Stream()
for pipe in (1, 0):
    w = bench.Workload(bench.WORKLOAD, 256, 0, 1, "cuda:0", pipeline=bool(pipe))
    with torch.cuda.stream(S):
        for rep in range(2):
            w.reset()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
            ev[0].record()
            for k in range(n):
                w.one_step()
                ev[k + 1].record()
            torch.cuda.synchronize()
            dt = [ev[k].elapsed_time(ev[k + 1]) for k in range(n)]
            print("pipeline %d rep %d: first 10 steps " % (pipe, rep) + " ".join("%.3f" % d for d in dt[:10]))
            print("   bins of 10: " + " ".join("%.3f" % (sum(dt[i:i + 10]) / 10) for i in range(0, n, 10)), flush=True)
    del w
    torch.cuda.synchronize()
