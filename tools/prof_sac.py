"""One production-size SAC update loop (target of rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ao_marl_amd.agents import AgentLayout
from ao_marl_amd.sac import BatchedSAC
lay = AgentLayout(1283, [0, 1274], 13, include_tip_tilt=True, window_n_zernike=20,
                  include_tip_tilt_windowed=True, n_filtered=5)
sac = BatchedSAC(lay, dict(memory_size=20000))
sac.memory.push(torch.randn(20000, lay.state_dim, device="cuda"), torch.rand(20000, lay.action_dim, device="cuda") * 2 - 1,
                -torch.rand(20000, lay.n_agents, device="cuda"), torch.randn(20000, lay.state_dim, device="cuda"), 1.0)
for _ in range(25):
    sac.update_from_memory(256)
torch.cuda.synchronize()
print("done")
