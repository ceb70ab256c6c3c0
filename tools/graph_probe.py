"""Step time with aomarl_env_step as HIP graphs vs plain launches (development aid): python tools/graph_probe.py
graph 1 + prefetch 1: the three-stream graph (fork / join inside); graph 1 + prefetch 0: ONE stream, a linear graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
S = torch.cuda.Stream()
for cfg, envs in ((bench.SMALL, 64), (bench.SMALL, 256), (bench.WORKLOAD, 256)):
    for prefetch in (True, False):
        w = bench.Workload(cfg, envs, 0, 1, "cuda:0", prefetch=prefetch, pipeline=False)
        with torch.cuda.stream(S):
            for graph in (0, 1, 0, 1):
                w.sim.set_option("graph_step", graph)
                w.policy.out_ring = 6 if graph else 0
                w.reset()
                n = 300 if "10x10" in cfg else 100
                e, enq, fk = w.timed(n, 60, time_frame=False)
                print("%-34s envs %4d prefetch %d graph %d: %.4f ms/step  %8.0f steps/s  host enqueue %.4f ms  graphs %s" %
                      (cfg, envs, prefetch, graph, e / n * 1e3, envs * n / e, enq / n * 1e3, w.sim.graph_stats()), flush=True)
        del w
        torch.cuda.synchronize()
