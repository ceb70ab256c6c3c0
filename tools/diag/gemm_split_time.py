"""Timing of the split-f16 GEMM against the fp32 one on the loop's shapes (development aid)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ao_marl_amd import libaomarl as la
lib = la.load()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K) in ((256, 1286, 2400), (256, 1283, 1288), (768, 648, 1960), (256, 648, 1960)):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); Cd = torch.zeros(M, N, device="cuda")
    ws = torch.zeros(8 * M * N + 4096, device="cuda")
    t32 = timeit(lambda: la.check(lib.aomarl_gemm_nt(M, N, K, 1.0, A.data_ptr(), K, B.data_ptr(), K, 0.0, Cd.data_ptr(), N, stream)))
    th0 = timeit(lambda: la.check(lib.aomarl_gemm_nt_split(M, N, K, 1.0, A.data_ptr(), K, B.data_ptr(), K, 0.0, Cd.data_ptr(), N, 1.0, 1.0, None, 0, stream)))
    th1 = timeit(lambda: la.check(lib.aomarl_gemm_nt_split(M, N, K, 1.0, A.data_ptr(), K, B.data_ptr(), K, 0.0, Cd.data_ptr(), N, 1.0, 1.0, ws.data_ptr(), ws.numel(), stream)))
    fl = 2.0 * M * N * K
    for tb in (256, 512, 1024):
        la.check(lib.aomarl_set_option(None, b"gemm_target_blocks", tb))
        t = timeit(lambda: la.check(lib.aomarl_gemm_nt_split(M, N, K, 1.0, A.data_ptr(), K, B.data_ptr(), K, 0.0, Cd.data_ptr(), N, 1.0, 1.0, ws.data_ptr(), ws.numel(), stream)))
        print("      target blocks %4d: %.1f us" % (tb, t))
    la.check(lib.aomarl_set_option(None, b"gemm_target_blocks", 512))
    for x in (0, 1):
        la.check(lib.aomarl_set_option(None, b"gemm_xcd_map", x))
        t = timeit(lambda: la.check(lib.aomarl_gemm_nt_split(M, N, K, 1.0, A.data_ptr(), K, B.data_ptr(), K, 0.0, Cd.data_ptr(), N, 1.0, 1.0, ws.data_ptr(), ws.numel(), stream)))
        print("      xcd map %d: %.1f us" % (x, t))
    print("%4d x %4d x %4d: fp32 no-split %.1f us | split-f16 no-split-K %.1f us (%.0f TFLOP/s eq.) | split-f16 + split-K + reduce %.1f us" % (M, N, K, t32, th0, fl / th0 * 1e-6, th1))
