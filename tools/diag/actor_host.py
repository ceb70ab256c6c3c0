"""Where does the host time of a fused select_action go (development aid)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ao_marl_amd.agents import AgentLayout, BatchedGaussianPolicy
from ao_marl_amd import libaomarl as la

small = AgentLayout(85, [0, 80], 2, include_tip_tilt=True, n_filtered=5)
big = AgentLayout(1283, [0, 1274], 13, include_tip_tilt=True, window_n_zernike=20,
                  include_tip_tilt_windowed=True, n_filtered=5)
for lay, nenv in ((small, 64), (big, 256), (small, 256), (small, 1024)):
    p = BatchedGaussianPolicy(lay, last_layer_zero=False, seed=5, device="cuda:0")
    st = torch.randn(nenv, lay.state_dim, device="cuda:0")
    for _ in range(20):
        p.select_action(st)
    torch.cuda.synchronize()
    d = p._actor_desc(nenv)
    a = torch.empty(nenv, lay.action_dim, device="cuda:0"); m = torch.empty_like(a)
    fn = la.load().aomarl_actor_forward
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ts = []
    for i in range(200):
        t0 = time.perf_counter()
        fn(C.byref(d), st.data_ptr(), None, 1, i, a.data_ptr(), m.data_ptr(), stream)
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    ts = sorted(ts)
    print("A %d nenv %d: C call median %.1f us  p90 %.1f  max %.1f" % (lay.n_agents, nenv, ts[100] * 1e6, ts[180] * 1e6, ts[-1] * 1e6), flush=True)
    # spaced calls: the GPU is idle at each launch
    ts = []
    for i in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(C.byref(d), st.data_ptr(), None, 1, i, a.data_ptr(), m.data_ptr(), stream)
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts)
    print("   after a sync: median %.1f us" % (ts[25] * 1e6), flush=True)

import cProfile, pstats
p = BatchedGaussianPolicy(small, last_layer_zero=False, seed=5, device="cuda:0")
st = torch.randn(64, small.state_dim, device="cuda:0")
for _ in range(20):
    p.select_action(st)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    p.select_action(st)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
