"""Pipelined step against the split-K rule of the chain GEMMs (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
S = torch.cuda.Stream()
w = bench.Workload(bench.WORKLOAD, 256, 0, 1, "cuda:0")
with torch.cuda.stream(S):
    for tb in (0, 1, 0, 1, 264, 132, 0):
        w.reset()
        w.sim.set_option("gemm_target_blocks", tb)
        n = 100
        e, enq, fk = w.timed(n, 20)
        print("gemm_target_blocks %4d: %.4f ms/step  %8.0f steps/s  frame kernel %.4f ms" % (tb, e / n * 1e3, 256 * n / e, fk), flush=True)
        w.sim.set_option("gemm_target_blocks", 0)
