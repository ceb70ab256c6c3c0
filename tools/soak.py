"""Soak run: 4 episodes x 750 steps of 256 production environments with a sampling policy; every state,
reward and Strehl must stay finite across resets (development aid)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from ao_marl_amd.env import VecAoEnv
from ao_marl_amd.agents import BatchedGaussianPolicy
rl = dict(n_zernike_start_end=[0, 1274], n_reverse_filtered_from_cmat=5, window_n_zernike=20, include_tip_tilt_windowed=True)
env = VecAoEnv("production_sh_40x40_8m_3layers", 256, rl, initial_seed=99, seed_stride=16, n_agents_modal=13, device="cuda:0")
pol = BatchedGaussianPolicy(env.layout, last_layer_zero=True, seed=1, device="cuda:0")
t0 = time.time()
for ep in range(4):
    st = env.reset()
    for it in range(750):
        a, _ = pol.select_action(st)
        st, r, _, _ = env.step(a)
    sr = env.supervisor.get_strehl()
    ok = bool(torch.isfinite(st).all() and torch.isfinite(r).all() and torch.isfinite(sr).all())
    print("episode %d: finite %s  SR_le mean %.4f min %.4f  reward mean %.4f" % (ep, ok, sr[:, 1].mean().item(), sr[:, 1].min().item(), r.mean().item()))
    assert ok
torch.cuda.synchronize()
print("3000 steps x 256 envs in %.1f s" % (time.time() - t0))
