"""Experiment: two groups of environments stepped on two HIP streams (frame kernel of one group
beside the control / agent / extrusion chain of the other).  Development aid."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ao_marl_amd.env import VecAoEnv
from ao_marl_amd.agents import BatchedGaussianPolicy
import bench as B

ngroups = int(sys.argv[1]) if len(sys.argv) > 1 else 2
total = int(sys.argv[2]) if len(sys.argv) > 2 else 256
threads = len(sys.argv) > 3 and sys.argv[3] == "threads"
steps = 100
rl = dict(n_zernike_start_end=[0, 1274], n_reverse_filtered_from_cmat=5,
          window_n_zernike=20, include_tip_tilt_windowed=True)
per = total // ngroups
envs, pols, streams, states = [], [], [], []
for g in range(ngroups):
    env = VecAoEnv(B.WORKLOAD, per, rl, initial_seed=1234 + 16 * per * g,
                   seed_stride=16, n_agents_modal=13, device="cuda:0")
    envs.append(env)
    pols.append(BatchedGaussianPolicy(env.layout, last_layer_zero=False, seed=1234, device="cuda:0"))
    streams.append(torch.cuda.Stream())
for g in range(ngroups):
    with torch.cuda.stream(streams[g]):
        states.append(envs[g].reset())

def run(g, n):
    with torch.cuda.stream(streams[g]):
        st = states[g]
        for _ in range(n):
            a, _ = pols[g].select_action(st)
            st, r, _, _ = envs[g].step(a)
        states[g] = st

def run_all(n):
    if threads:
        ts = [threading.Thread(target=run, args=(g, n)) for g in range(ngroups)]
        [t.start() for t in ts]; [t.join() for t in ts]
    else:
        for _ in range(n):
            for g in range(ngroups):
                run(g, 1)

run_all(10)
torch.cuda.synchronize()
t0 = time.perf_counter()
run_all(steps)
t1 = time.perf_counter()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("groups %d x %d envs %s: %.3f ms per step of all groups, %.0f env steps/s (host enqueue %.3f ms)" %
      (ngroups, per, "threads" if threads else "interleaved", el / steps * 1e3, total * steps / el, (t1 - t0) / steps * 1e3))
