"""Cost of the event pair around the frame kernel in the timed loop (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
w = B.Workload(B.WORKLOAD, 256, 0, 1, "cuda:0")
w.reset()
for rep in range(3):
    for tf in (True, False):
        el, tq, fk = w.timed(300, 20, None, "nccl", time_frame=tf)
        print("time_frame=%s: %.1f us/step (enqueue %.1f us/step) frame kernel %s" % (tf, el / 300 * 1e6, tq / 300 * 1e6, fk), flush=True)
