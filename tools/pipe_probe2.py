"""Frame pipeline stability probe (development aid): one Workload timed repeatedly, then fresh Workloads on ONE torch stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
S = torch.cuda.Stream()
w = bench.Workload(bench.WORKLOAD, 256, 0, 1, "cuda:0")
with torch.cuda.stream(S):
    for rep in range(5):
        w.reset()
        n = 100
        e, enq, fk = w.timed(n, 20)
        print("same workload, rep %d: %.4f ms/step  host enqueue %.4f ms  frame kernel %.4f ms" % (rep, e / n * 1e3, enq / n * 1e3, fk), flush=True)
del w
torch.cuda.synchronize()
for rep in range(4):
    w = bench.Workload(bench.WORKLOAD, 256, 0, 1, "cuda:0")
    with torch.cuda.stream(S):
        w.reset()
        n = 100
        e, enq, fk = w.timed(n, 20)
        print("fresh workload %d, same torch stream: %.4f ms/step  host enqueue %.4f ms  frame kernel %.4f ms" % (rep, e / n * 1e3, enq / n * 1e3, fk), flush=True)
    del w
    torch.cuda.synchronize()
