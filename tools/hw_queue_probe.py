"""How many streams of one process run side by side?  N streams, one spinning kernel each (torch.cuda._sleep), wall time
against one stream's: with Q hardware queues the N kernels take ceil(N / Q) rounds.  python tools/hw_queue_probe.py
(GPU_MAX_HW_QUEUES=<n> in the environment before the first HIP call changes Q on ROCm.)"""
import os
import time
import torch

cycles = 200_000_000          # about 0.1 s
torch.cuda._sleep(1000)
torch.cuda.synchronize()
res = []
for n in (1, 2, 3, 4, 5, 6, 8):
    streams = [torch.cuda.Stream() for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    res.append((n, time.perf_counter() - t0))
one = res[0][1]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for n, t in res:
    print("%d streams: %.3f s = %.2f x one stream" % (n, t, t / one))
