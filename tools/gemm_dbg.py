"""Phase ablation of k_gemm_batched_gen (debug bits in the relu argument); run under rocprofv3
--kernel-trace and read the per-call durations (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ao_marl_amd import libaomarl as L
lib = L.load()
A, B, I, Na, H = 14, 256, 552, 98, 256
for name, ta, tb, M, N, K in [("P2 fwd", 0, 1, B, H, H), ("dA", 0, 0, B, H, H), ("critic", 0, 1, B, 2 * H, I + Na)]:
    a = torch.randn((A, K, M) if ta else (A, M, K), device="cuda")
    b = torch.randn((A, K, N) if tb else (A, N, K), device="cuda")
    out = torch.empty(A, M, N, device="cuda")
    for G in (1, 2, 4):
        L.check(lib.aomarl_set_option(None, b"gemm_kgroups", G))
        for dbg in (0, 256, 512, 512 + 1024, 256 + 512, 256 + 512 + 1024):
            for _ in range(4):
                L.check(lib.aomarl_gemm_batched(A, ta, tb, M, N, K, a.data_ptr(), a.stride(1), a.stride(0), b.data_ptr(), b.stride(1), b.stride(0), None, 0, out.data_ptr(), N, M * N, dbg, 0, None))
            torch.cuda.synchronize()
