"""Reset time against library options (development aid): python tools/reset_probe.py [envs]
round 3: split-K rule of the extrusion GEMM (the blocks-per-CU cost model picks 396 workgroups for 768 x 648 x 1957:
58.7 ms; 264: 71.5, 528-1056: 62.3-64.3); the batch in parts side by side on several streams (reset_streams)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
envs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
once = len(sys.argv) > 2 and sys.argv[2] == "once"          # one pass with the default (tools/reset_trace.sh)
w = bench.Workload(bench.WORKLOAD, envs, 0, 1, "cuda:0")
S = torch.cuda.Stream()
with torch.cuda.stream(S):
    for two in ((2,) if once else (1, 2, 3, 4, 1, 2, 3, 4)):
        w.sim.set_option("reset_streams", two)
        w.reset()
        t = min(w.time_reset() for _ in range(2))
        print("envs %d reset_streams %d: reset %.2f ms" % (envs, two, t * 1e3), flush=True)
