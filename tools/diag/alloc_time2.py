import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ao_marl_amd.agents import AgentLayout, BatchedGaussianPolicy
from ao_marl_amd import libaomarl as la
small = AgentLayout(85, [0, 80], 2, include_tip_tilt=True, n_filtered=5)
p = BatchedGaussianPolicy(small, last_layer_zero=False, seed=5, device="cuda:0")
nenv = 64
st = torch.randn(nenv, small.state_dim, device="cuda:0")
for _ in range(20): p.select_action(st)
torch.cuda.synchronize()
d = p._actor_desc(nenv)
fn = la.load().aomarl_actor_forward
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda:0")
def loop(mode, n=300):
    te = tc = 0.0
    keep = []
    for i in range(n):
        t0 = time.perf_counter()
        a = torch.empty(nenv, small.action_dim, dtype=torch.float32, device=dev); m = torch.empty_like(a)
        t1 = time.perf_counter()
        if mode != "noc":
            fn(C.byref(d), st.data_ptr(), None, 1, i, a.data_ptr(), m.data_ptr(), stream)
        else:
            a.zero_()
        t2 = time.perf_counter()
        te += t1 - t0; tc += t2 - t1
        if mode == "keep": keep.append((a, m))
    torch.cuda.synchronize()
    print(mode, "alloc %.1f us, launch %.1f us" % (te / n * 1e6, tc / n * 1e6), flush=True)
loop("plain"); loop("keep"); loop("noc"); loop("plain")
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["segment.all.current"], os.environ.get("PYTORCH_HIP_ALLOC_CONF"), os.environ.get("PYTORCH_CUDA_ALLOC_CONF"))
for lay, nenv in ((small, 64),):
    for rep in range(3):
        p = BatchedGaussianPolicy(lay, last_layer_zero=False, seed=5, device="cuda:0")
        st = torch.randn(nenv, lay.state_dim, device="cuda:0")
        for _ in range(20): p.select_action(st)
        torch.cuda.synchronize()
        ts = []
        for i in range(300):
            t0 = time.perf_counter(); p.select_action(st); ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        order = sorted(range(300), key=lambda i: -ts[i])[:5]
        print("rep", rep, "mean %.1f us median %.1f us; slowest:" % (sum(ts) / 300 * 1e6, sorted(ts)[150] * 1e6), [(i, round(ts[i] * 1e6)) for i in order], flush=True)
