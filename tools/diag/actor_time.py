"""Time select_action: one kernel vs layer by layer (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ao_marl_amd.agents import AgentLayout, BatchedGaussianPolicy

def run(lay, nenv, tag):
    for lbl in (False, True):
        p = BatchedGaussianPolicy(lay, last_layer_zero=False, seed=5, device="cuda:0")
        p.layer_by_layer = lbl
        st = torch.randn(nenv, lay.state_dim, device="cuda:0")
        for _ in range(20):
            p.select_action(st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(300):
            p.select_action(st)
        e1.record()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("%s nenv %d %s: %.1f us per call on the GPU, host %.1f us" %
              (tag, nenv, "layered" if lbl else "fused", e0.elapsed_time(e1) / 300 * 1e3, th / 300 * 1e6),
              "in_max", p.in_max, "act_max", p.act_max, flush=True)

big = AgentLayout(1283, [0, 1274], 13, include_tip_tilt=True, window_n_zernike=20,
                  include_tip_tilt_windowed=True, n_filtered=5)
small = AgentLayout(85, [0, 80], 2, include_tip_tilt=True, n_filtered=5)
run(big, 256, "40x40")
run(small, 64, "10x10")
run(big, 1024, "40x40") if len(sys.argv) > 1 else None
