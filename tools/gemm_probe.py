"""k_gemm_nt2 on large and loop-sized shapes (development aid): python tools/gemm_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from ao_marl_amd import libaomarl as la
L = la.load()
def run(M, N, K, n=20):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); Cc = torch.zeros(M, N, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: la.check(L.aomarl_gemm_nt(M, N, K, 1.0, A.data_ptr(), K, B.data_ptr(), K, 0.0, Cc.data_ptr(), N, st))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    ref = A @ B.T
    print("%5d x %5d x %5d  %8.1f us  %6.1f TFLOP/s  blocks %d  err %.1e" % (M, N, K, us, 2e-6 * M * N * K / us, ((M+63)//64)*((N+63)//64), (Cc-ref).abs().max().item()/ref.abs().max().item()))
for shp in ((4096, 4096, 4096), (2048, 2048, 2048), (1024, 1024, 4096), (768, 648, 1957+3), (256, 1288, 2400), (256, 1284, 1288), (1024, 1024, 512)):
    run(*shp)
