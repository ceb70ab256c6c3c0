"""Where a training episode's stepping phase spends its time (development aid): the bench's loop, then + the SAC
object's own policy, + the delayed-MDP replay writes, + the return accumulation."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from ao_marl_amd.env import VecAoEnv, DelayedMDP
from ao_marl_amd.agents import BatchedGaussianPolicy
from ao_marl_amd.sac import BatchedSAC, BatchedReplay
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
rl = dict(n_zernike_start_end=[0, 1274], n_reverse_filtered_from_cmat=5, window_n_zernike=20, include_tip_tilt_windowed=True)
env = VecAoEnv("production_sh_40x40_8m_3layers", 256, rl, initial_seed=1234, seed_stride=16, n_agents_modal=13, frame_pipeline=True)
lay = env.layout
pol = BatchedGaussianPolicy(lay, last_layer_zero=True, seed=1234, device="cuda:0")
sac = BatchedSAC(lay, dict(memory_size=256 * 300), device="cuda:0")
master = BatchedReplay(lay.state_dim, lay.action_dim, lay.n_agents, 256 * 300, "cuda:0")
N = 200
def run(name, policy, push, acc):
    s = env.reset()
    mdp = DelayedMDP(1, False)
    master.reset()
    r_agents = torch.zeros(256, lay.n_agents, device="cuda:0")
    for _ in range(20):
        a, _ = policy.select_action(s); s, r, _, _ = env.step(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        a, mu = policy.select_action(s)
        s2, r, done, _ = env.step(a)
        if push:
            if mdp.check_update_possibility():
                s0, a0, sn = mdp.credit_assignment()
                master.push(s0, a0, r, sn, 1.0)
            mdp.save(s, a, s2)
        if acc:
            r_agents += r
        s = s2
    te = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%-46s %.3f ms per step (host enqueue %.3f)" % (name, (time.perf_counter() - t0) / N * 1e3, te / N * 1e3), flush=True)
run("bench loop (stand-alone policy)", pol, False, False)
run("the SAC object's policy", sac.policy, False, False)
run("+ return accumulation", sac.policy, False, True)
run("+ delayed-MDP replay writes", sac.policy, True, True)
run("bench loop again", pol, False, False)
from ao_marl_amd.sac import run_episode
for _ in range(2):
    tm = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_episode(env, sac, max_steps=N + 20, train=True, n_updates=1, batch_size=256, timing=tm)
    torch.cuda.synchronize()
    print("run_episode (trajectory-resident, one call per step) %.3f ms per step incl. the reset (46 ms / %d steps = %.3f)" %
          (tm["steps_s"] / (N + 20) * 1e3, N + 20, 46.0 / (N + 20)), flush=True)
