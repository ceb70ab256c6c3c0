"""Frame pipeline A/B (development aid), same process, ONE caller stream (a new torch stream per pass changes which
hardware queues alias):   python tools/pipe_probe.py                      pipelined against the plain call order
                          python tools/pipe_probe.py <option> <v> [<v>..]  a library option varied under both orders"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
opt = sys.argv[1] if len(sys.argv) > 2 else None
vals = [int(v) for v in sys.argv[2:]] if opt else [0]
S = torch.cuda.Stream()
for pipe in (1, 0, 1, 0) if opt is None else (1, 0):
    w = bench.Workload(bench.WORKLOAD, 256, 0, 1, "cuda:0", pipeline=bool(pipe))
    with torch.cuda.stream(S):
        for v in (vals * 2 if opt else vals):
            w.reset()
            if opt:
                w.sim.set_option(opt, v)
            n = 100
            e, enq, fk = w.timed(n, 20)
            print("frame pipeline %d %s: %.4f ms/step  %8.0f steps/s  host enqueue %.4f ms  frame kernel %.4f ms  pipe %s" %
                  (pipe, "%s=%d" % (opt, v) if opt else "", e / n * 1e3, 256 * n / e, enq / n * 1e3, fk,
                   w.sim.frame_pipeline_state()), flush=True)
    del w
    torch.cuda.synchronize()
