"""Soak run: 4 episodes x 1000 steps of 256 production environments with a sampling policy, with the frame pipeline
and in the plain order; every state, reward and Strehl must stay finite across resets and the two orders must give
the same numbers, bit for bit, at the end of every episode (development aid): python tools/soak.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ao_marl_amd.env import VecAoEnv
from ao_marl_amd.agents import BatchedGaussianPolicy
rl = dict(n_zernike_start_end=[0, 1274], n_reverse_filtered_from_cmat=5, window_n_zernike=20, include_tip_tilt_windowed=True)
ends = {}
with torch.cuda.stream(torch.cuda.Stream()):
    for pipe in (True, False):
        env = VecAoEnv("production_sh_40x40_8m_3layers", 256, rl, initial_seed=99, seed_stride=16, n_agents_modal=13,
                       device="cuda:0", frame_pipeline=pipe)
        pol = BatchedGaussianPolicy(env.layout, last_layer_zero=True, seed=1, device="cuda:0")
        for ep in range(4):
            torch.cuda.synchronize(); t0 = time.time()
            st = env.reset()
            torch.cuda.synchronize(); t1 = time.time()
            for it in range(1000):
                a, _ = pol.select_action(st)
                st, r, _, _ = env.step(a)
            torch.cuda.synchronize(); t2 = time.time()
            sr = env.supervisor.get_strehl()
            ok = bool(torch.isfinite(st).all() and torch.isfinite(r).all() and torch.isfinite(sr).all())
            print("pipeline %d episode %d: reset %.1f ms, 1000 steps %.1f ms -> %.0f env steps/s with the reset | finite %s  SR_le mean "
                  "%.4f min %.4f  reward mean %.4f  pipe %s" %
                  (pipe, ep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, 256e3 / (t2 - t0), ok, sr[:, 1].mean().item(), sr[:, 1].min().item(),
                   r.mean().item(), env.supervisor.sim.frame_pipeline_state()), flush=True)
            assert ok
            ends.setdefault(ep, []).append((st.clone(), r.clone(), sr.clone(), env.supervisor.get_slopes().clone()))
        del env, pol
for ep, (a, b) in ends.items():
    same = all(torch.equal(x, y) for x, y in zip(a, b))
    print("episode %d: pipelined == plain order, bit for bit: %s" % (ep, same))
    assert same
