"""aomarl_gemm_batched on the shapes of the production SAC update, per k-group count (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ao_marl_amd import libaomarl as L
lib = L.load()
A, B, I, Na, H = 14, 256, 552, 98, 256
shapes = [("P1 fwd", 0, 1, B, H, I), ("P2 fwd", 0, 1, B, H, H), ("head fwd", 0, 1, B, 2 * Na, H),
          ("critic fwd", 0, 1, B, 2 * H, I + Na), ("dWin", 1, 1, I + Na, 2 * H, B), ("dW1", 1, 1, I, H, B),
          ("dWh", 1, 1, H, H, B), ("dWhead", 1, 1, H, 2 * Na, B), ("dpi", 0, 0, B, Na, 2 * H),
          ("dact", 0, 0, B, H, 2 * Na), ("dA", 0, 0, B, H, H)]
for name, ta, tb, M, N, K in shapes:
    a = torch.randn((A, K, M) if ta else (A, M, K), device="cuda")
    b = torch.randn((A, K, N) if tb else (A, N, K), device="cuda")
    out = torch.empty(A, M, N, device="cuda")
    ref = torch.bmm(a.transpose(1, 2) if ta else a, b if tb else b.transpose(1, 2))
    line = "%-11s M%4d N%4d K%4d  %5.2f GF " % (name, M, N, K, 2e-9 * A * M * N * K)
    for G in (1, 2, 4):
        L.check(lib.aomarl_set_option(None, b"gemm_kgroups", G))
        for _ in range(3):
            L.gemm_batched(a, b, bool(ta), bool(tb), out=out)
        err = (out - ref).abs().max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            L.gemm_batched(a, b, bool(ta), bool(tb), out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        line += "| G%d %6.1f us %5.1f TF err %.1e " % (G, us, 2e-6 * A * M * N * K / us, err)
    print(line)
