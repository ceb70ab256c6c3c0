"""Pipelined / plain step against an option of the library, same process, same workload (development aid):
python tools/pipe_probe3.py <option> <value> [<value> ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
opt, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
S = torch.cuda.Stream()
for pipe in (1, 0):
    w = bench.Workload(bench.WORKLOAD, 256, 0, 1, "cuda:0", pipeline=bool(pipe))
    with torch.cuda.stream(S):
        for v in vals * 2:
            w.reset()
            w.sim.set_option(opt, v)
            n = 100
            e, enq, fk = w.timed(n, 20)
            print("pipeline %d  %s %4d: %.4f ms/step  %8.0f steps/s  frame kernel %.4f ms" % (pipe, opt, v, e / n * 1e3, 256 * n / e, fk), flush=True)
    del w
    torch.cuda.synchronize()
