"""SAC update time at production size: the library's update vs the torch-autograd statement of it
(development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ao_marl_amd.agents import AgentLayout
from ao_marl_amd.sac import BatchedSAC
lay = AgentLayout(1283, [0, 1274], 13, include_tip_tilt=True, window_n_zernike=20,
                  include_tip_tilt_windowed=True, n_filtered=5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
if os.environ.get("AOMARL_KG"):
    from ao_marl_amd import libaomarl as L
    L.check(L.load().aomarl_set_option(None, b"gemm_kgroups", int(os.environ["AOMARL_KG"])))
for native in (True,):
    sac = BatchedSAC(lay, dict(memory_size=20000), native=native)
    sac.memory.push(torch.randn(20000, lay.state_dim, device="cuda"), torch.rand(20000, lay.action_dim, device="cuda") * 2 - 1,
                    -torch.rand(20000, lay.n_agents, device="cuda"), torch.randn(20000, lay.state_dim, device="cuda"), 1.0)
    for _ in range(5):
        sac.update_from_memory(256)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        sac.update_from_memory(256)
    torch.cuda.synchronize()
    print("native" if native else "torch ", "%.3f ms per update of %d agents" % ((time.time() - t0) / n * 1e3, lay.n_agents))
    if native:
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(n):
            sac.update_from_memory(256)
        t1 = time.time()
        torch.cuda.synchronize()
        print("   host enqueue %.3f ms per update, drained after %.3f ms" % ((t1 - t0) / n * 1e3, (time.time() - t0) / n * 1e3))
