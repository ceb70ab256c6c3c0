#!/usr/bin/env python3
"""Learning acceptance: does an agent trained HERE beat the integrator?

The reference's loop (TrainerRPC.train_agent, train_rpc.py:452-501): training episodes of 1000 steps followed
by 1000 SAC updates per agent; every `test_every` episodes one evaluation with the policy's MEAN action and one
with the integrator alone, on the same fresh atmosphere seeds (train_rpc.py:484-490, 555-631), reporting the
summed per-agent reward and the long-exposure Strehl.  Same loop here on BASELINE configs[1]:
production_sh_10x10_2m, 64 environments, 2 agents (80 Btt modes + tip-tilt), everything on the device --
VecAoEnv.step through aomarl_env_step, actors through aomarl_actor_forward, updates through aomarl_sac_update.

    python tools/learning_acceptance.py [--episodes 60] [--test-every 10] [--envs 64] [--precision f32]

--config 40x40: the same loop on BASELINE configs[2] -- production_sh_40x40_8m_3layers, 256 environments, 14 agents
(13 windowed modal agents of 98 modes + the windowed tip-tilt agent, states 552 / 168), the reference's recorded
statistics file for the standardisation (its five filtered modes masked), the loop bench.py times
(VecAoEnv.throughput_mode through sac.train_agent: frame pipeline, residual shortcut, prefetched resets):
profiles/r06_learning_acceptance_40x40.txt.
Prints one line per evaluation and a verdict line; exit code 1 when the last evaluation's RL reward or LE
Strehl is not above the integrator's.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="10x10", choices=("10x10", "40x40"))
    ap.add_argument("--episodes", type=int, default=60)
    ap.add_argument("--test-every", type=int, default=10)
    ap.add_argument("--envs", type=int, default=None, help="default: 64 (10x10) / 256 (40x40)")
    ap.add_argument("--memory", type=int, default=1000000, help="rows of the agents' replay ring (46 KB each at 40x40)")
    ap.add_argument("--lr", type=float, default=None, help="learning rate of actors, critics and temperature (default: the reference's)")
    ap.add_argument("--no-throughput", action="store_true", help="train_agent(throughput=False): the environment as built here")
    ap.add_argument("--agents", type=int, default=14, choices=(14, 43),
                    help="40x40: 14 = BASELINE configs[2]'s 13 x 98 modes + tip-tilt; 43 = the reference's published 42 x 30 "
                         "modes + tip-tilt (README.md:116-119), both with the 20-mode window")
    ap.add_argument("--reward-scale", default=None,
                    help="BatchedSAC(reward_scale=...): 'auto' divides every agent's rewards by their own standard deviation "
                         "before they enter its replay memory; 'integrator' by what the integrator alone earns the agent "
                         "per step (measured in the 200 frames in front of the training); not in the reference, default: off")
    ap.add_argument("--reward-factor", type=float, default=None,
                    help="the factor of the per-agent reward -factor x mean(residual modes^2): reward_type "
                         "avg_squared_modes_<factor> (helper_rewards.py:18 parses any number; the reference's default is 1000)")
    ap.add_argument("--action-scale", type=float, default=None,
                    help="norm_scale_zernike_actions (parameters.cfg:43, default 10): the action range of a mode is its "
                         "recorded maximum / this")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--updates", type=int, default=1000)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--recorded-normalisation", action="store_true",
                    help="standardise the states with the reference's recorded 10x10 file instead of statistics generated "
                         "here.  That file was recorded with TEN filtered modes (modes 75..84 have std 1e-8), so with the "
                         "README's `--n_reverse_filtered_from_cmat 5 --n_zernike_start_end 0 80` the controlled modes "
                         "75..79 enter the state divided by 1e-8: inputs of 1e6..1e8, the critic's loss starts at 1e12 and "
                         "the 80-mode agent never learns (profiles/r03_learning_acceptance.txt)")
    ap.add_argument("--frame-pipeline", action="store_true",
                    help="the RL episodes with a frame in flight (aomarl_set_frame_pipeline): same environment, bit for bit")
    ap.add_argument("--torch-update", action="store_true",
                    help="the torch-autograd statement of the SAC update instead of aomarl_sac_update (slow; A/B)")
    a = ap.parse_args(argv)
    import torch
    from ao_marl_amd import libaomarl as la
    from ao_marl_amd.env import VecAoEnv
    from ao_marl_amd.sac import BatchedSAC, train_agent
    la.set_precision(a.precision)
    large = a.config == "40x40"
    name = "production_sh_40x40_8m_3layers" if large else "production_sh_10x10_2m"
    if a.envs is None:
        a.envs = 256 if large else 64
    if large:
        rl = dict(n_zernike_start_end=[0, 1274 if a.agents == 14 else 1260], n_reverse_filtered_from_cmat=5, window_n_zernike=20,
                  include_tip_tilt_windowed=True, max_steps_per_episode=a.steps)
        n_modal = a.agents - 1
    else:
        rl, n_modal = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5, max_steps_per_episode=a.steps), 1
    if a.reward_factor is not None:
        rl["reward_type"] = "avg_squared_modes_%g" % a.reward_factor
    if a.action_scale is not None:
        rl["norm_scale_zernike_actions"] = a.action_scale
    norm_kw = {}
    if not a.recorded_normalisation and not large:
        # the reference's own workflow for a (parameter file, filtered modes) pair: run the normalisation recipe
        # first (obtain_normalization.py:246-300) -- here on the device, 20 seeds x 1000 integrator frames
        # (the 40x40 file's recorded statistics were made with the README's five filtered modes: used as they are,
        # the five dead columns masked)
        from ao_marl_amd.normalization import obtain_normalization
        norm, zn, _ = obtain_normalization(name, modes_filtered=5)
        norm_kw = dict(norm=norm, zn_norm=zn)
    env = VecAoEnv(name, a.envs, rl, initial_seed=a.seed, seed_stride=16, n_agents_modal=n_modal,
                   frame_pipeline=a.frame_pipeline, **norm_kw)
    cfg = dict(updates_per_episode_rpc=a.updates, memory_size=a.memory)
    if a.lr is not None:
        cfg.update(lr=a.lr)
    if a.reward_scale is not None and a.reward_scale != "integrator":
        cfg.update(reward_scale=a.reward_scale if a.reward_scale == "auto" else float(a.reward_scale))
    sac = BatchedSAC(env.layout, cfg, seed=a.seed, native=not a.torch_update)
    print("reward_type %s  norm_scale_zernike_actions %s  lr %s  reward_scale %s" %
          (env.reward_type, env.config_rl["norm_scale_zernike_actions"], sac.lr, cfg.get("reward_scale")))
    print("config %s  envs %d  agents %d (state dims %s, action dims %s)  %d steps + %d updates per "
          "episode  precision %s" % (name, a.envs, env.layout.n_agents, env.layout.state_shapes(), env.layout.action_shapes(),
                                     a.steps, a.updates, la.get_precision()), flush=True)
    t0 = time.time()
    evals = []
    # before any update the actors' last layer is zero: the mean action is 0 and the RL evaluation must BE the
    # integrator's (same seeds, one through aomarl_env_step, one stage by stage)
    from ao_marl_amd.sac import run_episode
    rl0 = run_episode(env, sac, max_steps=200, train=False, eval_mode=True)
    lin0 = run_episode(env, sac, max_steps=200, train=False, linear_control=True)
    print("untrained (mean action 0), 200 steps: RL reward %.3f SR_LE %.5f SR_SE %.5f | integrator reward %.3f SR_LE %.5f "
          "SR_SE %.5f | per-agent RL %s integrator %s" %
          (rl0["r_total"].mean(), rl0["sr_le"].mean(), rl0["sr_se_mean"].mean(), lin0["r_total"].mean(),
           lin0["sr_le"].mean(), lin0["sr_se_mean"].mean(), rl0["r_per_agent"].mean(dim=0).tolist(),
           lin0["r_per_agent"].mean(dim=0).tolist()), flush=True)
    env.next_seed_block(1)
    if a.reward_scale == "integrator":
        # every agent's rewards in units of what the integrator alone earns it per step (the 200 frames above): all
        # fourteen learners see rewards of about -1 per step where the loop operates, whatever their modes' amplitudes
        per_step = (lin0["r_per_agent"].mean(dim=0).abs() / 200.0).clamp(min=1e-12)
        sac.cfg["reward_scale"] = (1.0 / per_step).tolist()

    def on_episode(rec):
        if "test_r_rl" in rec:
            evals.append(rec)
            print("episode %3d (%.0f s): train reward %9.2f SR_LE %.4f | eval on seed %d: RL reward %9.2f SR_LE %.4f | "
                  "integrator reward %9.2f SR_LE %.4f | RL - integrator: reward %+8.2f  SR_LE %+.4f | SR_SE %.4f vs %.4f | "
                  "per agent RL %s integrator %s | alpha %s q1 loss %s" %
                  (rec["episode"], time.time() - t0, rec["r_total"], rec["sr_le"], rec["test_seed"], rec["test_r_rl"],
                   rec["test_sr_le_rl"], rec["test_r_integrator"], rec["test_sr_le_integrator"],
                   rec["test_r_rl"] - rec["test_r_integrator"], rec["test_sr_le_rl"] - rec["test_sr_le_integrator"],
                   rec["test_sr_se_rl"], rec["test_sr_se_integrator"],
                   ["%.1f" % v for v in rec["test_r_agents_rl"]], ["%.1f" % v for v in rec["test_r_agents_integrator"]],
                   ["%.4f" % v for v in sac.last_losses["alpha_value"].reshape(-1).tolist()] if sac.last_losses else "-",
                   ["%.2e" % v for v in sac.last_losses["q1"].reshape(-1).tolist()] if sac.last_losses else "-"),
                  flush=True)
    train_agent(env, sac, a.episodes, max_steps=a.steps, test_every=a.test_every, n_updates=a.updates, on_episode=on_episode,
                throughput=not a.no_throughput)
    torch.cuda.synchronize()
    if sac._rscale is not None:
        print("reward scales per agent (%s): %s" % ("1 / |what the integrator earns the agent per step|" if a.reward_scale == "integrator" else "BatchedSAC(reward_scale=%r)" % (a.reward_scale,),
              ["%.3g" % v for v in sac._rscale.reshape(-1).tolist()]), flush=True)
    print("environment: frame_pipeline %s (probe %s), residual_shortcut %s, reset_prefetch %s, prefetched resets adopted %d" %
          (env.frame_pipeline, env.order_probe, env.residual_shortcut, env.supervisor.reset_prefetch,
           int(getattr(env.supervisor.sim, "prefetched_resets", 0))), flush=True)
    last = evals[-1]
    ok = last["test_r_rl"] > last["test_r_integrator"] and last["test_sr_le_rl"] > last["test_sr_le_integrator"]
    print("verdict after %d training episodes (%.0f s): RL %s the integrator (reward %.2f vs %.2f, LE Strehl %.4f vs %.4f)" %
          (a.episodes, time.time() - t0, "BEATS" if ok else "does NOT beat", last["test_r_rl"], last["test_r_integrator"],
           last["test_sr_le_rl"], last["test_sr_le_integrator"]))
    main.evals = evals
    main.alphas = sac.last_losses["alpha_value"].reshape(-1).tolist() if sac.last_losses else []   # per-agent temperatures at the end
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
