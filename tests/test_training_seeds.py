"""Episode-level seed handling of the training loop (TrainerRPC.train_agent, train_rpc.py:452-501):
every training episode and every evaluation pair must see atmosphere seeds no earlier episode saw;
the RL and the integrator evaluation of one test share their seeds.  CPU: VecAoEnv over the
oracle-backed simulator, torch statement of the SAC update."""
import os

import numpy as np
import torch

from ao_marl_amd import params
from ao_marl_amd.env import VecAoEnv, load_norm
from ao_marl_amd.sac import BatchedSAC, run_episode, train_agent
from tests.oracle_vecsim import OracleVecSim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAR = os.path.join(ROOT, "tools", "par", "production_aomarl_sh_10x10_2m_single.py")


def _env(nenv=2, stride=16):
    ps = params.load_param_file(PAR)
    norm, zn = load_norm("production_sh_10x10_2m")
    return VecAoEnv(ps, nenv, dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5,
                                   max_steps_per_episode=3),
                    initial_seed=1234, seed_stride=stride, n_agents_modal=1, device="cpu",
                    norm=norm, zn_norm=zn, sim_factory=OracleVecSim)


def test_consecutive_episodes_use_fresh_seed_blocks():
    env = _env()
    sac = BatchedSAC(env.layout, dict(hidden_size_actor=16, hidden_size_critic=16, batch_size=2,
                                      updates_per_episode_rpc=1),
                     device="cpu", memory_size=64, native=False)
    seen = []
    real_reset = env.supervisor.sim.reset

    def spy(seeds, *a, **k):
        seen.append([int(x) for x in np.asarray(seeds).reshape(-1)])
        return real_reset(seeds, *a, **k)
    env.supervisor.sim.reset = spy
    log = train_agent(env, sac, n_episodes=2, max_steps=3, test_every=1, n_updates=1, batch_size=2)
    # per episode: train, test RL, test integrator
    assert len(seen) == 6 and len(log) == 2
    train0, rl0, lin0, train1, rl1, lin1 = seen
    assert rl0 == lin0 and rl1 == lin1                    # the two evaluations see one atmosphere
    blocks = [train0, rl0, train1, rl1]
    # layer k of environment e uses seed + k: blocks must stay apart by more than the layer count
    flat = sorted(x for b in blocks for x in b)
    assert len(set(flat)) == len(flat)
    assert min(np.diff(flat)) >= 16
    assert [r["seed"] for r in log] == [train0[0], train1[0]]
    assert log[0]["test_seed"] == rl0[0]
    # run_episode on its own does not move the seed (the caller does, like the reference's trainer)
    before = env.supervisor.current_seed
    run_episode(env, sac, max_steps=2, train=False, linear_control=True)
    assert env.supervisor.current_seed == before
    assert env.next_seed_block(world_size=4) == before + 4 * 16 * env.nenv


def test_episode_returns_are_gathered_without_a_process_group():
    env = _env()
    sac = BatchedSAC(env.layout, dict(hidden_size_actor=16, hidden_size_critic=16),
                     device="cpu", memory_size=16, native=False)
    out = run_episode(env, sac, max_steps=2, train=False, linear_control=True)
    assert torch.equal(out["r_total_all"], out["r_total"])
    assert out["sr_le_all"].shape == (env.nenv,)
