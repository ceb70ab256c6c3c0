"""Host cost of one step on the host-bound configuration (development aid): python tools/host_probe.py
Step time and host enqueue time with the one-launch move (small_move) on / off, then cProfile of the
Python side of 300 steps."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
for envs in (64, 256):
    w = bench.Workload(bench.SMALL, envs, 0, 1, "cuda:0", pipeline=True)
    with torch.cuda.stream(torch.cuda.Stream()):
        for small in (1, 0, 1, 0):
            w.sim.set_option("small_move", small)
            w.reset()
            n = 300
            e, enq, fk = w.timed(n, 60, time_frame=False)
            print("%-26s envs %4d small_move %d: %.4f ms/step  %8.0f steps/s  host enqueue %.4f ms" %
                  (bench.SMALL, envs, small, e / n * 1e3, envs * n / e, enq / n * 1e3), flush=True)
        if envs == 64:
            w.sim.set_option("small_move", 1)
            w.reset()
            for _ in range(50):
                w.one_step()
            torch.cuda.synchronize()
            pr = cProfile.Profile()
            pr.enable()
            t0 = time.perf_counter()
            for _ in range(300):
                w.one_step()
            t1 = time.perf_counter()
            pr.disable()
            torch.cuda.synchronize()
            print("profiled loop: %.4f ms/step of host time" % ((t1 - t0) / 300 * 1e3))
            pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
    del w
    torch.cuda.synchronize()
