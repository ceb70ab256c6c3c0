"""Kernel-only durations of the split-f16 GEMM variants (run under rocprofv3 --kernel-trace; development aid)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ao_marl_amd import libaomarl as la
lib = la.load()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in ((768, 648, 1960), (256, 1286, 2400), (256, 648, 1960)):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); Cd = torch.zeros(M, N, device="cuda")
    ws = torch.zeros(8 * M * N + 4096, device="cuda")
    for t128 in (0,):
        for xm in (0, 1):
            la.check(lib.aomarl_set_option(None, b"gemm_xcd_map", xm))
            for _ in range(10):
                la.check(lib.aomarl_gemm_nt_split(M, N, K, 1.0, A.data_ptr(), K, B.data_ptr(), K, 0.0, Cd.data_ptr(), N, 1.0, 1.0, ws.data_ptr(), ws.numel(), stream))
            torch.cuda.synchronize()
