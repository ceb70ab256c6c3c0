"""Host and GPU time of VecAoEnv.step in its variants (development aid)."""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
cfg, nenv = (B.SMALL, 64) if len(sys.argv) < 2 else (B.WORKLOAD, 256)
for label, native, fused in (("call by call", False, False), ("one call, 14 launches", True, False), ("one call, fused tail", True, True)):
    w = B.Workload(cfg, nenv, 0, 1, "cuda:0")
    w.env.native_step, w.env.fused_tail = native, fused
    w.reset()
    gc.collect(); gc.disable()
    for _ in range(20): w.one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): w.one_step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    a, _ = w.policy.select_action(w.state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): w.env.step(a)
    te = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%-24s host %.1f us/step, wall %.1f us/step; env.step alone: host %.1f us" % (label, th / 300 * 1e6, tt / 300 * 1e6, te / 300 * 1e6), flush=True)
    del w
    gc.enable(); gc.collect(); torch.cuda.synchronize()
