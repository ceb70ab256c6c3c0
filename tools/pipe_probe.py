"""Frame pipeline A/B (development aid): python tools/pipe_probe.py  -- step time of the headline workload, pipelined
against the plain call order, same process, one caller stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
S = torch.cuda.Stream()          # ONE caller stream: a new one per pass changes which hardware queues alias
for pipe in (1, 0, 1, 0):
    w = bench.Workload(bench.WORKLOAD, 256, 0, 1, "cuda:0", pipeline=bool(pipe))
    with torch.cuda.stream(S):
        w.reset()
        n = 100
        e, enq, fk = w.timed(n, 20)
        print("frame pipeline %d: %.4f ms/step  %8.0f steps/s  host enqueue %.4f ms  frame kernel %.4f ms  pipe %s" %
              (pipe, e / n * 1e3, 256 * n / e, enq / n * 1e3, fk, w.sim.frame_pipeline_state()), flush=True)
    del w
    torch.cuda.synchronize()
