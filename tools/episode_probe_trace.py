"""The bench loop + one torch op per step (r_agents += r), 80 steps: target of rocprofv3 --kernel-trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from ao_marl_amd.env import VecAoEnv
from ao_marl_amd.agents import BatchedGaussianPolicy
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
rl = dict(n_zernike_start_end=[0, 1274], n_reverse_filtered_from_cmat=5, window_n_zernike=20, include_tip_tilt_windowed=True)
env = VecAoEnv("production_sh_40x40_8m_3layers", 256, rl, initial_seed=1234, seed_stride=16, n_agents_modal=13, frame_pipeline=True)
pol = BatchedGaussianPolicy(env.layout, last_layer_zero=True, seed=1234, device="cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "acc"
s = env.reset()
r_agents = torch.zeros(256, env.layout.n_agents, device="cuda:0")
for _ in range(80):
    a, _ = pol.select_action(s)
    s, r, _, _ = env.step(a)
    if mode == "acc":
        r_agents += r
torch.cuda.synchronize()
print("done")
