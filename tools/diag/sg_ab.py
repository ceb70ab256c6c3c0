"""Step time with the scatter + gather of consecutive extrusion rounds fused / as separate launches (development aid)."""
import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
w = B.Workload(B.WORKLOAD, 256, 0, 1, "cuda:0")
w.reset()
for rep in range(3):
    for unf in (1, 0):
        w.sim.set_option("extrude_unfused", unf)
        el, tq, fk = w.timed(300, 20, None, "nccl", time_frame=True)
        print("extrude_unfused=%d: %.1f us/step frame kernel %.4f" % (unf, el / 300 * 1e6, fk), flush=True)
