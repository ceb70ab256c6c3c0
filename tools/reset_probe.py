"""Reset time against the split-K rule of the extrusion GEMM (development aid): python tools/reset_probe.py
(round 3: the blocks-per-CU cost model picks 396 workgroups for 768 x 648 x 1957 -- 58.7 ms; 264: 71.5, 528-1056: 62.3-64.3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
w = bench.Workload(bench.WORKLOAD, 256, 0, 1, "cuda:0")
S = torch.cuda.Stream()
with torch.cuda.stream(S):
    for xcd in (1, 0):
        w.sim.set_option("gemm_xcd_map", xcd)
        for tb in (0, 264, 396, 528, 660, 792, 1056, 0):
            w.sim.set_option("gemm_target_blocks", tb)
            w.reset()
            t = min(w.time_reset() for _ in range(2))
            print("gemm_xcd_map %d gemm_target_blocks %4d: reset %.2f ms" % (xcd, tb, t * 1e3), flush=True)
w.sim.set_option("gemm_target_blocks", 0)
