"""Step time of the headline loop under a few switches (development aid): frame-kernel event pair on / off,
atmosphere prefetch on / off.   python tools/gap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
w = bench.Workload(bench.WORKLOAD, 256, 0, 1, "cuda:0")
w.reset()
def run(steps=60, label="", time_frame=True):
    e, enq, fk = w.timed(steps, 5, time_frame=time_frame)
    print("%-40s %.4f ms/step  host enqueue %.4f  frame kernel %s" % (label, e / steps * 1e3, enq / steps * 1e3, fk))
for rep in range(2):
    run(label="events on the frame kernel dispatch", time_frame=True)
    run(label="no events", time_frame=False)
