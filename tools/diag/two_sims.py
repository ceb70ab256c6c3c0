"""Two environments alive in one process (a training and an evaluation one, say): step time of each (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
ws = []
for i in range(2):
    w = B.Workload(B.WORKLOAD, 256, 0, 1, "cuda:0")
    w.reset()
    ws.append(w)
    for j, v in enumerate(ws):
        el, tq, fk = v.timed(200, 20, None, "nccl", time_frame=True)
        print("%d alive, environment %d: %.1f us/step, frame kernel %.4f ms" % (len(ws), j, el / 200 * 1e6, fk), flush=True)
