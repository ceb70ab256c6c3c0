"""aomarl_actor_forward alone (development aid): the fused actor kernel on the layouts of the bench configurations,
back-to-back launches between two events.   AOMARL_LIB=<variant .so> python tools/actor_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ao_marl_amd.agents import AgentLayout, BatchedGaussianPolicy

dev = "cuda:0"
cases = [("40x40, 256 envs, 14 agents (13 x 98 modes, window 20, + tip-tilt)", 1283, 256,
          dict(n_zernike_start_end=[0, 1274], n_agents_modal=13, window_n_zernike=20, include_tip_tilt_windowed=True, n_filtered=5)),
         ("40x40, 256 envs, 43 agents (42 x 30 modes + tip-tilt)", 1283, 256,
          dict(n_zernike_start_end=[0, 1260], n_agents_modal=42, n_filtered=5)),
         ("10x10, 64 envs, 2 agents (80 modes + tip-tilt)", 87, 64,
          dict(n_zernike_start_end=[0, 80], n_agents_modal=1, n_filtered=5))]
for name, nmodes, nenv, kw in cases:
    lay = AgentLayout(nmodes, **kw)
    pol = BatchedGaussianPolicy(lay, last_layer_zero=False, seed=1, device=dev)
    state = torch.randn(nenv, lay.state_dim, device=dev)
    for _ in range(5):
        pol._select_action_one_call(state, False, None)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    a.record()
    for _ in range(reps):
        pol._select_action_one_call(state, False, None)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / reps * 1e3
    ins, acts = lay.state_shapes(), lay.action_shapes()
    H = pol.H
    fl = 2.0 * nenv * sum(i * H + (pol.L - 1) * H * H + H * 2 * ac for i, ac in zip(ins, acts))
    print("%-72s in %d..%d  %.1f us  (%.2f GFLOP useful, %.1f TFLOP/s)" % (name, min(ins), max(ins), us, fl * 1e-9, fl / us * 1e-6), flush=True)
