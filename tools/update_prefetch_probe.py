"""Does the next episode's reset hide beside the learner's updates?  1000 aomarl_sac_update calls of the bench layout
alone, then the same with the 2 x 648 rounds of a prefetched reset dealt over them on the library's side stream
(aomarl_reset_prefetch_*), against updates + reset in the open (development aid): python tools/update_prefetch_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ao_marl_amd.sac import BatchedSAC

w = bench.Workload(bench.WORKLOAD, 256, 0, 1, "cuda:0")
layout, dev = w.env.layout, "cuda:0"
S = torch.cuda.Stream()
with torch.cuda.stream(S):
    w.reset()
    sac = BatchedSAC(layout, dict(memory_size=20000), device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    sac.memory.push(torch.randn(20000, layout.state_dim, generator=g, device=dev),
                    torch.rand(20000, layout.action_dim, generator=g, device=dev) * 2 - 1,
                    -torch.rand(20000, layout.n_agents, generator=g, device=dev),
                    torch.randn(20000, layout.state_dim, generator=g, device=dev), 1.0)
    for _ in range(5):
        sac.update_from_memory(256)
    sim = w.sim
    seeds = 4321 + 16 * np.arange(256)
    N = 1000

    def updates(prefetch):
        if prefetch:
            sim.prefetch_reset_begin(seeds)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        acc, left = 0.0, 1
        for i in range(N):
            sac.update_from_memory(256)
            if prefetch and left:
                acc += 1296.0 / (0.95 * N)
                k = int(acc)
                if k:
                    acc -= k
                    left = sim.prefetch_reset_advance(k)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, left

    def reset_time(adopt):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sim.reset(seeds)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for rep in range(2):
        tu, _ = updates(False)
        tr = reset_time(False)
        tup, left = updates(True)
        tra = reset_time(True)
        print("updates alone %.1f ms + reset in the open %.1f ms = %.1f | updates with the prefetch beside them %.1f ms (rounds left %d) "
              "+ adopting reset %.1f ms = %.1f" % (tu, tr, tu + tr, tup, left, tra, tup + tra), flush=True)
