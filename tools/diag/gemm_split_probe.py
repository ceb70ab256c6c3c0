"""Probe (development aid): accuracy of aomarl_gemm_nt_split on the loop's real operands and on small
operands (does the f16 matrix pipe flush subnormal inputs?)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ao_marl_amd import libaomarl as la
lib = la.load()
t = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "trace_10x10_stock.npz"))
cm, sl = t["cmat"].astype(np.float32), t["slopes"].astype(np.float32)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

def run(A, B, sa, sb):
    M, K = A.shape; N = B.shape[0]
    Kp = (K + 3) // 4 * 4
    Ad = torch.zeros(M, Kp, device="cuda"); Ad[:, :K] = torch.from_numpy(A).cuda()
    Bd = torch.zeros(N, Kp, device="cuda"); Bd[:, :K] = torch.from_numpy(B).cuda()
    Cd = torch.zeros(M, N, device="cuda")
    la.check(lib.aomarl_gemm_nt_split(M, N, K, 1.0, Ad.data_ptr(), Kp, Bd.data_ptr(), Kp, 0.0, Cd.data_ptr(), N,
                                      float(sa), float(sb), None, 0, stream))
    want = A.astype(np.float64) @ B.astype(np.float64).T
    got = Cd.cpu().numpy().astype(np.float64)
    return np.abs(got - want).max(), np.abs(want).max()

for sa, sb in ((64, 32), (1, 1), (1024, 32), (4096, 32)):
    print("slopes x cmat, scales", sa, sb, "max err %.3e of %.3e" % run(sl, cm, sa, sb))
for row in (0, 10, 30):
    print(" one row (M=1) frame", row, "max err %.3e of %.3e" % run(sl[row:row + 1], cm, 64, 32))
g = np.random.default_rng(0)
for mag in (1.0, 1e-2, 1e-4, 1e-5, 1e-6):
    A = (g.normal(size=(64, 256)) * mag).astype(np.float32); B = g.normal(size=(64, 256)).astype(np.float32)
    e, s = run(A, B, 1, 1)
    print("A magnitude %.0e unscaled: relative err %.2e" % (mag, e / s))
