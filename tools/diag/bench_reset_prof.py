import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
orig = B.Workload.time_reset
def prof(self, *a, **k):
    pr = cProfile.Profile(); pr.enable()
    r = orig(self, *a, **k)
    pr.disable()
    sys.stderr.write("time_reset %.2f ms\n" % (r * 1e3))
    if r > 0.03:
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(14)
    return r
B.Workload.time_reset = prof
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "40"]
B.main()
