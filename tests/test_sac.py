"""Batched SAC update + replay memory against the reference's own update code.

tests/golden/host_sac_update.pt was produced by tools/gen_golden_sac.py, which runs
`SAC.update_critic / update_actor / update_alpha` + `soft_update` of the reference
(train_rpc.py:1016-1133) on the reference's GaussianPolicy / QNetwork for two agents of different
sizes and three consecutive updates, recording the normal draws."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from ao_marl_amd.agents import AgentLayout  # noqa: E402
from ao_marl_amd.sac import BatchedReplay, BatchedSAC  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_sac_update.pt")


def _layout():
    # 10x10 layout of BASELINE configs[1]: one 80-mode agent (state 4 x 80) + the tip-tilt agent
    return AgentLayout(87, [0, 80], 1, include_tip_tilt=True, n_filtered=5)


def _pad(t, width):
    out = t.new_zeros(t.shape[0], width)
    out[:, :t.shape[1]] = t
    return out


def test_batched_update_reproduces_the_references_sac_update():
    g = torch.load(GOLD)
    h = g["hyper"]
    lay = _layout()
    assert lay.state_shapes() == [a["nin"] for a in g["agents"]]
    assert lay.action_shapes() == [a["nact"] for a in g["agents"]]
    sac = BatchedSAC(lay, dict(hidden_size_actor=h["hidden"], hidden_size_critic=h["hidden"],
                               lr=h["lr"], gamma=h["gamma"], tau=h["tau"], memory_size=64,
                               initialize_last_layer_0=False), device="cpu")
    for i, ag in enumerate(g["agents"]):
        sac.load_reference_agent(i, ag["policy0"], ag["critic0"])
    for u in range(h["updates"]):
        ups = [ag["updates"][u] for ag in g["agents"]]
        x = torch.stack([_pad(t["s"], sac.in_max) for t in ups])
        x2 = torch.stack([_pad(t["s2"], sac.in_max) for t in ups])
        a = torch.stack([_pad(t["a"], sac.act_max) for t in ups])
        r = torch.stack([t["r"] for t in ups])
        mask = torch.stack([t["mask"] for t in ups])
        e1 = torch.stack([_pad(t["eps_next"], sac.act_max) for t in ups])
        e2 = torch.stack([_pad(t["eps_pi"], sac.act_max) for t in ups])
        losses = sac.update(x, a, r, x2, mask, eps_next=e1, eps_pi=e2)
        for i, t in enumerate(ups):
            assert abs(losses["q1"][i].item() - t["q1_loss"]) < 1e-5 * max(1, abs(t["q1_loss"])), (u, i)
            assert abs(losses["q2"][i].item() - t["q2_loss"]) < 1e-5 * max(1, abs(t["q2_loss"])), (u, i)
            assert abs(losses["policy"][i].item() - t["policy_loss"]) < 1e-5 * max(1, abs(t["policy_loss"]))
            assert abs(losses["alpha"][i].item() - t["alpha_loss"]) < 1e-5 * max(1, abs(t["alpha_loss"]))
            assert abs(sac.alpha[i].item() - t["alpha"]) < 1e-6
            actor, critic = sac.export_agent(i)
            _, target = sac.export_agent(i, target=True)
            for name, ref in t["policy"].items():
                assert torch.allclose(actor[name], ref, atol=2e-6, rtol=1e-5), (u, i, name)
            for name, ref in t["critic"].items():
                assert torch.allclose(critic[name], ref, atol=2e-6, rtol=1e-5), (u, i, name)
            for name, ref in t["critic_target"].items():
                assert torch.allclose(target[name], ref, atol=2e-6, rtol=1e-5), (u, i, name)
    # padding never learns anything
    assert sac.policy.W1[1, 8:].abs().max().item() == 0.0
    assert sac.policy.Wm[1, :, 2:].abs().max().item() == 0.0
    assert sac.critic[0]["Win"][1, 8:sac.in_max].abs().max().item() == 0.0


def test_replay_ring_and_per_agent_batches():
    lay = _layout()
    sac = BatchedSAC(lay, dict(hidden_size_actor=16, hidden_size_critic=16, memory_size=50), device="cpu")
    m = sac.memory
    rng = torch.Generator().manual_seed(0)
    rows = []
    for step in range(9):                       # 9 x 8 = 72 rows into a ring of 50
        s = torch.randn(8, lay.state_dim, generator=rng)
        a = torch.rand(8, lay.action_dim, generator=rng)
        r = -torch.rand(8, lay.n_agents, generator=rng)
        s2 = torch.randn(8, lay.state_dim, generator=rng)
        m.push(s, a, r, s2, 1.0)
        rows += [(s[k], a[k], r[k], s2[k]) for k in range(8)]
    assert len(m) == 50 and m.position == 72 % 50
    # the ring holds the newest 50 rows
    newest = torch.stack([t[0] for t in rows[-50:]])
    held = torch.cat([m.state[m.position:], m.state[:m.position]])
    assert torch.equal(held, newest)
    x, a, r, x2, mask = sac.batch_from_memory(32)
    assert x.shape == (2, 32, sac.in_max) and a.shape == (2, 32, sac.act_max)
    assert r.shape == (2, 32, 1) and mask.shape == (2, 32, 1) and mask.min() == 1.0
    # every sampled per-agent tuple is the split of ONE stored row (same row for s, a, r, s')
    for i in range(2):
        ni, na = lay.state_shapes()[i], lay.action_shapes()[i]
        idx = torch.as_tensor(list(lay.modes_chosen.values())[i])
        lo, hi = list(lay.action_slices.values())[i]
        for b in range(0, 32, 7):
            cand = (m.state[:, idx] == x[i, b, :ni]).all(dim=1).nonzero().reshape(-1)
            assert cand.numel() >= 1
            k = cand[0]
            assert torch.equal(m.next_state[k, idx], x2[i, b, :ni])
            assert torch.equal(m.action[k, lo:hi], a[i, b, :na])
            assert m.reward[k, i] == r[i, b, 0]
        assert x[i, :, ni:].abs().max() == 0 if ni < sac.in_max else True
        assert a[i, :, na:].abs().max() == 0 if na < sac.act_max else True


def test_update_parameters_trickles_the_episode_into_memory():
    lay = _layout()
    sac = BatchedSAC(lay, dict(hidden_size_actor=16, hidden_size_critic=16, memory_size=1000,
                               batch_size=16), device="cpu")
    master = BatchedReplay(lay.state_dim, lay.action_dim, lay.n_agents, 200, "cpu")
    master.push(torch.randn(120, lay.state_dim), torch.rand(120, lay.action_dim),
                -torch.rand(120, lay.n_agents), torch.randn(120, lay.state_dim), 1.0)
    before = [t.detach().clone() for t in sac._policy_params()]
    done = sac.update_parameters(master, n_updates=40)        # 3 rows per update
    # updates start once the memory holds more than one batch: 16 / 3 -> from the 6th push on
    assert done == 40 - 5 and len(sac.memory) == 120 and sac.total_update == done
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, sac._policy_params()))
    assert torch.isfinite(sac.last_losses["q1"]).all()


def test_reward_scale_is_an_opt_in_that_only_touches_what_the_learner_stores():
    """BatchedSAC(reward_scale=...): not in the reference, off by default.  Numbers per agent (or "auto" = 1 / std of every
    agent's rewards over the first episode's transitions, its first tenth left out, measured once) multiply the rewards on
    their way from the episode's master memory into the agents' replay ring -- the master memory (what the episode
    reports) is untouched, states / actions / next states are stored as they are."""
    lay = _layout()
    S, A = torch.randn(120, lay.state_dim), torch.rand(120, lay.action_dim)
    R = -torch.rand(120, lay.n_agents) * torch.tensor([[1e-3, 5.0] + [1.0] * (lay.n_agents - 2)])[:, :lay.n_agents]

    def fill(cfg):
        sac = BatchedSAC(lay, dict(hidden_size_actor=16, hidden_size_critic=16, memory_size=1000, batch_size=16, **cfg),
                         device="cpu")
        master = BatchedReplay(lay.state_dim, lay.action_dim, lay.n_agents, 200, "cpu")
        master.push(S, A, R.clone(), S.flip(0), 1.0)
        sac.update_parameters(master, n_updates=40)
        assert torch.equal(master.reward[:120], R)                       # the episode's own record: as reported
        return sac
    plain = fill({})
    assert plain._reward_scale(None) is None and torch.equal(plain.memory.reward[:120], R)
    fixed = fill(dict(reward_scale=[2.0] * lay.n_agents))
    assert torch.allclose(fixed.memory.reward[:120], 2.0 * R) and torch.equal(fixed.memory.state[:120], S)
    auto = fill(dict(reward_scale="auto"))
    want = 1.0 / R[12:].std(dim=0)
    assert torch.allclose(auto._rscale.reshape(-1), want, rtol=1e-5)
    stored = auto.memory.reward[:120]
    assert torch.allclose(stored, R * want, rtol=1e-5)
    assert torch.allclose(stored[12:].std(dim=0), torch.ones(lay.n_agents), rtol=1e-4)     # every agent: rewards of order one
    # measured once: a second episode with other rewards keeps the scale
    master = BatchedReplay(lay.state_dim, lay.action_dim, lay.n_agents, 200, "cpu")
    master.push(S, A, 100.0 * R, S, 1.0)
    keep = auto._rscale.clone()
    auto.update_parameters(master, n_updates=10)
    assert torch.equal(auto._rscale, keep)
    with pytest.raises(ValueError):
        fill(dict(reward_scale="median"))


def test_training_episode_on_the_oracle_backed_env():
    from tests.oracle_vecsim import OracleVecSim
    from ao_marl_amd.env import VecAoEnv
    from ao_marl_amd.sac import run_episode
    env = VecAoEnv("production_sh_10x10_2m", 2, dict(n_zernike_start_end=[0, 80],
                                                     n_reverse_filtered_from_cmat=5),
                   n_agents_modal=1, device="cpu", sim_factory=OracleVecSim)
    sac = BatchedSAC(env.layout, dict(hidden_size_actor=32, hidden_size_critic=32, batch_size=4,
                                      memory_size=100), device="cpu")
    out = run_episode(env, sac, max_steps=6, train=True, n_updates=5)
    # delay 1, no modification: the first tuple is available at step 3 -> 4 steps x 2 envs stored
    assert len(sac.memory) == 8 and out["updates"] >= 1
    assert out["r_total"].shape == (2,) and out["r_per_agent"].shape == (2, 2)
    assert (out["r_total"] <= 0).all() and torch.isfinite(out["sr_le"]).all()
    ev = run_episode(env, sac, max_steps=3, train=False, eval_mode=True)
    assert "updates" not in ev
    # train_agent takes the package's throughput configuration by default (VecAoEnv.throughput_mode); on a simulator
    # without an atmosphere prefetch (this CPU stand-in) that leaves the call order alone and only asks for the one-product
    # residual, which the stage-by-stage path ignores: the loop runs as before, evaluations included
    from ao_marl_amd.sac import train_agent
    log = train_agent(env, sac, 2, max_steps=5, test_every=1, n_updates=3)
    assert env.residual_shortcut and env.frame_pipeline is False and env.supervisor.reset_prefetch is None
    assert len(log) == 2 and all("test_r_rl" in r and "test_r_integrator" in r for r in log)
    assert log[1]["seed"] > log[0]["test_seed"] > log[0]["seed"]          # a fresh block of seeds per episode and evaluation


def test_checkpoints_use_the_references_container(tmp_path):
    """SAC.save_model layout (train_rpc.py:1140-1161): {'worker_id', 'models_controlled',
    'model_state_dict'} per actor, a bare state_dict per critic; both load back into any slot."""
    lay = _layout()
    a = BatchedSAC(lay, dict(hidden_size_actor=16, hidden_size_critic=16, memory_size=8,
                             initialize_last_layer_0=False), seed=1, device="cpu")
    b = BatchedSAC(lay, dict(hidden_size_actor=16, hidden_size_critic=16, memory_size=8), seed=2,
                   device="cpu")
    paths = a.save_model(str(tmp_path / "exp"), episode=800, experiment_name="demo")
    assert len(paths) == lay.n_agents
    for i, (ap, cp) in enumerate(paths):
        ck = torch.load(ap)
        assert set(ck) == {"worker_id", "models_controlled", "model_state_dict"}
        assert ck["worker_id"] == i + 1 and ck["models_controlled"] == list(list(lay.agents.values())[i])
        assert list(ck["model_state_dict"]) == ["linear1.weight", "linear1.bias", "hidden.0.weight",
                                                "hidden.0.bias", "mean_linear.weight",
                                                "mean_linear.bias", "log_std_linear.weight",
                                                "log_std_linear.bias"]
        b.load_reference_agent(i, ck["model_state_dict"], torch.load(cp))
    s = torch.randn(5, lay.state_dim)
    ma, la_ = a.policy.forward(s)
    mb, lb = b.policy.forward(s)
    assert torch.equal(ma, mb) and torch.equal(la_, lb)
    x = a.policy.split_states(s)
    act = torch.rand(lay.n_agents, 5, a.act_max) * a.act_mask
    qa, qb = a._q(a.critic, x, act), b._q(b.critic, x, act)
    assert torch.equal(qa[0], qb[0]) and torch.equal(qa[1], qb[1])


def test_load_model_accepts_the_three_actor_file_layouts(tmp_path):
    """src/error_budget/sac/sac.py:187-240: {'alpha', 'log_alpha', 'target_entropy',
    'model_state_dict'}, the trainer's {'worker_id', ...}, or the bare policy state_dict."""
    lay = _layout()
    cfg = dict(hidden_size_actor=16, hidden_size_critic=16, memory_size=8)
    src = BatchedSAC(lay, dict(cfg, initialize_last_layer_0=False), seed=1, device="cpu")
    dst = BatchedSAC(lay, cfg, seed=2, device="cpu")
    actor0, critic0 = src.export_agent(0)
    actor1, critic1 = src.export_agent(1)
    p_full, p_bare, p_crit = (str(tmp_path / n) for n in ("full.pt", "bare.pt", "critic.pt"))
    torch.save({"model_state_dict": actor0, "alpha": torch.tensor([0.37]),
                "log_alpha": torch.tensor([-0.99]), "target_entropy": -7.0}, p_full)
    torch.save(actor1, p_bare)
    torch.save(critic0, p_crit)
    dst.load_model(0, p_full, p_crit)
    dst.load_model(1, p_bare)
    assert dst.alpha[0].item() == pytest.approx(0.37) and dst.log_alpha[0].item() == pytest.approx(-0.99)
    assert dst.target_entropy[0].item() == pytest.approx(-7.0)
    assert dst.alpha[1].item() == pytest.approx(0.2)                  # untouched slot keeps its default
    paths = src.save_model(str(tmp_path / "exp"), episode=1)
    dst2 = BatchedSAC(lay, cfg, seed=3, device="cpu")
    for i, (ap, cp) in enumerate(paths):                               # the 'worker_id' layout
        dst2.load_model(i, ap, cp)
    s = torch.randn(4, lay.state_dim)
    m_src, _ = src.policy.forward(s)
    m_dst, _ = dst.policy.forward(s)
    m_dst2, _ = dst2.policy.forward(s)
    assert torch.equal(m_src, m_dst) and torch.equal(m_src, m_dst2)
    x = src.policy.split_states(s)
    act = torch.rand(lay.n_agents, 4, src.act_max) * src.act_mask
    q_src, q_dst = src._q(src.critic, x, act), dst._q(dst.critic, x, act)
    assert torch.equal(q_src[0][0], q_dst[0][0]) and torch.equal(q_src[1][0], q_dst[1][0])
    assert torch.equal(dst.critic_target[0]["Win"][0], dst.critic[0]["Win"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("pipe,online", [(False, False), (True, False), (False, True)])
def test_trajectory_resident_episode_is_the_step_by_step_bookkeeping(pipe, online):
    """run_episode on the GPU writes states, actions and rewards straight into trajectory buffers (TrajectoryReplay:
    no copy per step) -- the transitions the agents then learn from are the ones the reference's delayed-MDP
    bookkeeping (DelayedMDP + manage_memory: one replay push per step) stores, row for row and bit for bit, and so
    are the episode's returns and the learner it leaves behind."""
    from ao_marl_amd.env import VecAoEnv
    from ao_marl_amd.sac import BatchedReplay, TrajectoryReplay, run_episode
    # online: `modification_online` -- the delayed-MDP tuple then spans delay + 0 steps (delayed_mdp.py:11-16) and the
    # environment steps call by call (pure-delay-0 order): the trajectory views must follow both
    rl = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5, modification_online=online)
    T, E = 14, 6
    lag = 1 if online else 2
    got = {}
    for mode in ("steps", "traj"):
        torch.manual_seed(11)
        env = VecAoEnv("production_sh_10x10_2m", E, rl, n_agents_modal=1, initial_seed=77, frame_pipeline=pipe)
        sac = BatchedSAC(env.layout, dict(memory_size=4096, batch_size=32), seed=5)
        lay = env.layout
        seen = []

        def record(m, **kw):                # (the episode's master memory as the learner would receive it)
            assert isinstance(m, TrajectoryReplay if mode == "traj" else BatchedReplay)
            assert len(m) == (T - lag) * E                        # delay 1 (+ 1 without online modification): tuples from step `lag` on
            seen.append([t.clone() for t in m.rows(0, len(m))])
            return 0
        sac.update_parameters = record
        master = BatchedReplay(lay.state_dim, lay.action_dim, lay.n_agents, T * E, "cuda:0") if mode == "steps" else None
        out = run_episode(env, sac, max_steps=T, train=True, master=master, n_updates=4)
        got[mode] = (seen[0], out["r_total"].clone(), out["r_per_agent"].clone(), out["sr_le"].clone())
    for a, b in zip(got["steps"][0], got["traj"][0]):
        assert a.shape == b.shape and torch.equal(a, b.reshape(a.shape))
    assert torch.allclose(got["steps"][1], got["traj"][1], rtol=1e-6) and torch.equal(got["steps"][3], got["traj"][3])
    assert torch.allclose(got["steps"][2], got["traj"][2], rtol=1e-6)  # (sums of the same rewards in another order)
    # rows() in pieces, as update_parameters walks them, and its bounds
    tr = TrajectoryReplay(3, 2, 1, 5, 2, 2, "cuda:0")
    tr.S.copy_(torch.arange(6 * 2 * 3, dtype=torch.float32).reshape(6, 2, 3)); tr.t = 5
    s0, a0, r0, s2, mk = tr.rows(2, 3)
    assert len(tr) == 6 and torch.equal(s0, tr.S.reshape(-1, 3)[2:5]) and torch.equal(s2, tr.S.reshape(-1, 3)[6:9])
    assert torch.equal(r0, tr.R.reshape(-1, 1)[6:9]) and torch.equal(a0, tr.A.reshape(-1, 2)[2:5]) and bool((mk == 1).all())
    with pytest.raises(IndexError):
        tr.rows(4, 3)


@pytest.mark.gpu
def test_training_episode_on_the_gpu_and_update_rate():
    """End-to-end on the HIP path: rollout with the native batched GEMM actors, replay in HBM,
    batched SAC updates through autograd; prints the production-size update rate."""
    import time
    from ao_marl_amd.env import VecAoEnv
    from ao_marl_amd.sac import run_episode
    env = VecAoEnv("production_sh_10x10_2m", 16, dict(n_zernike_start_end=[0, 80],
                                                      n_reverse_filtered_from_cmat=5),
                   n_agents_modal=1)
    sac = BatchedSAC(env.layout, dict(memory_size=4096, batch_size=64))
    w0 = sac.policy.W1.detach().clone()
    out = run_episode(env, sac, max_steps=30, train=True, n_updates=20)
    assert len(sac.memory) == 28 * 16 and out["updates"] == 18     # 23 rows per update, batch 64
    assert not torch.equal(w0, sac.policy.W1.detach())
    assert torch.isfinite(out["r_total"]).all() and (out["sr_le"] > 0).all()
    # production layout: 14 agents (13 x 98 modes + TT, window 20), batch 256
    lay = AgentLayout(1283, [0, 1274], 13, include_tip_tilt=True, window_n_zernike=20,
                      include_tip_tilt_windowed=True, n_filtered=5)
    big = BatchedSAC(lay, dict(memory_size=20000))
    big.memory.push(torch.randn(20000, lay.state_dim, device="cuda"),
                    torch.rand(20000, lay.action_dim, device="cuda") * 2 - 1,
                    -torch.rand(20000, lay.n_agents, device="cuda"),
                    torch.randn(20000, lay.state_dim, device="cuda"), 1.0)
    for _ in range(3):
        big.update(*big.batch_from_memory(256))
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        big.update(*big.batch_from_memory(256))
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 20
    print("SAC update, 14 agents x batch 256: %.2f ms per update (%.0f agent-updates/s)" %
          (dt * 1e3, 14 / dt))
    assert torch.isfinite(big.last_losses["q1"]).all()


def _twin_sacs(lay, cfg, rows, seed=3):
    """A native and a torch-autograd BatchedSAC with identical parameters and replay contents."""
    import torch
    from ao_marl_amd.sac import BatchedSAC
    cfg = dict(dict(memory_size=rows, initialize_last_layer_0=False), **cfg)
    nat = BatchedSAC(lay, cfg, seed=seed, device="cuda:0", native=True)
    ref = BatchedSAC(lay, cfg, seed=seed, device="cuda:0", native=False)
    assert torch.equal(nat._pflat, ref._pflat) and torch.equal(nat._cflat, ref._cflat)
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():               # biases and the twin critics away from their symmetric start
        for s in (nat, ref):
            s._pflat.add_(0.02 * torch.randn(s._pflat.shape, generator=torch.Generator(device="cuda").manual_seed(9), device="cuda") * (s._pflat != 0))
        nat_b = 0.05 * torch.randn(nat._pflat.shape, generator=g, device="cuda")
        for s in (nat, ref):
            for b in [s.policy.b1] + s.policy.bh:
                b.copy_(nat_b[:b.numel()].view(b.shape))
            s.policy.bm.copy_((0.1 * nat_b[:s.policy.bm.numel()].view(s.policy.bm.shape)) * s.act_mask)
            s.policy.bs.copy_((0.1 * nat_b[7:7 + s.policy.bs.numel()].view(s.policy.bs.shape) - 1.0) * s.act_mask)
            for k, q in enumerate(s.critic):
                q["bin"].copy_(nat_b[11 + k:11 + k + q["bin"].numel()].view(q["bin"].shape))
                q["bout"].fill_(0.1 * (k + 1))
            s._ctflat.copy_(s._cflat * 0.9)
    data = (torch.randn(rows, lay.state_dim, generator=g, device="cuda"),
            torch.rand(rows, lay.action_dim, generator=g, device="cuda") * 2 - 1,
            -torch.rand(rows, lay.n_agents, generator=g, device="cuda"),
            torch.randn(rows, lay.state_dim, generator=g, device="cuda"),
            (torch.rand(rows, generator=g, device="cuda") > 0.1).float())
    for s in (nat, ref):
        s.memory.push(*data)
    return nat, ref, g


def _grad_views(sac, u):
    """The updater's flat gradient buffers cut into the tensors of the torch path."""
    p, A = sac.policy, sac.A
    I, Na, H, Hc, L = sac.in_max, sac.act_max, p.H, sac.Hc, p.L
    po, co = sac._poff, sac._coff

    def v(flat, off, *shape):
        n = 1
        for k in shape:
            n *= k
        return flat[off:off + n].view(*shape)
    gp = u.policy_grad
    head, bhead = v(gp, po[2 * L], A, H, 2 * Na), v(gp, po[2 * L + 1], A, 1, 2 * Na)
    pol = [v(gp, po[0], A, I, H), v(gp, po[1], A, 1, H)] + \
          [v(gp, po[2 * l], A, H, H) for l in range(1, L)] + \
          [v(gp, po[2 * l + 1], A, 1, H) for l in range(1, L)] + \
          [head[:, :, :Na], bhead[:, :, :Na], head[:, :, Na:], bhead[:, :, Na:]]
    gc = u.critic_grad
    win, bin_ = v(gc, co[0], A, I + Na, 2 * Hc), v(gc, co[1], A, 1, 2 * Hc)
    wout, bout = v(gc, co[2], A, 2, Hc), v(gc, co[3], A, 2)
    cri = dict(Win=[win[:, :, k * Hc:(k + 1) * Hc] for k in range(2)],
               bin=[bin_[:, :, k * Hc:(k + 1) * Hc] for k in range(2)],
               Wout=[wout[:, k, :].unsqueeze(2) for k in range(2)],
               bout=[bout[:, k].reshape(A, 1, 1) for k in range(2)])
    return pol, cri


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["two_layer", "one_layer_wide_critic", "three_layer_fixed_alpha", "production"])
def test_native_update_matches_the_autograd_update(case):
    """aomarl_sac_update (hand-written forward + backward + Adam + soft update) against the torch
    autograd statement of the update on the same replay rows and the same normal draws: gradients of
    the first update, then parameters, temperature and losses over several updates."""
    import torch
    cfg, B, n_upd = {}, 64, 4
    lay = _layout()
    if case == "one_layer_wide_critic":
        cfg = dict(num_layers_actor=1, hidden_size_actor=64, hidden_size_critic=192, gamma=0.5)
        B = 37
    elif case == "three_layer_fixed_alpha":
        cfg = dict(num_layers_actor=3, hidden_size_actor=128, hidden_size_critic=64,
                   automatic_entropy_tuning=False, target_update_interval=2, alpha=0.3)
    elif case == "production":
        lay = AgentLayout(1283, [0, 1274], 13, include_tip_tilt=True, window_n_zernike=20,
                          include_tip_tilt_windowed=True, n_filtered=5)
        B, n_upd = 256, 2
    nat, ref, g = _twin_sacs(lay, cfg, rows=700)
    ref.keep_grads = True
    A, Na = nat.A, nat.act_max
    for it in range(n_upd):
        idx = torch.randint(0, 700, (A, B), generator=g, device="cuda")
        e2 = torch.randn(A, B, Na, generator=g, device="cuda")
        e1 = torch.randn(A, B, Na, generator=g, device="cuda")
        ln = nat.update_from_memory(B, idx, e2, e1)
        lr = ref.update(*ref.batch_from_memory(B, idx), eps_next=e2, eps_pi=e1)
        if it == 0:
            u = nat.updater(B)
            pol, cri = _grad_views(nat, u)
            for name in cri:
                for k in range(2):
                    want = ref.grad_log["critic"][name][k]
                    err = (cri[name][k] - want).abs().max().item()
                    assert err < 2e-4 * want.abs().max().item() + 1e-7, (name, k, err)
            for i, (got, want) in enumerate(zip(pol, ref.grad_log["policy"])):
                err = (got - want).abs().max().item()
                assert err < 5e-4 * want.abs().max().item() + 1e-8, (i, err)
            if ref.automatic_entropy_tuning:
                assert torch.allclose(u.la_grad, ref.grad_log["log_alpha"].reshape(-1), rtol=1e-4, atol=1e-6)
        for k in ("q1", "q2", "policy", "alpha", "alpha_value"):
            assert torch.allclose(ln[k].reshape(-1), lr[k].reshape(-1), rtol=2e-3, atol=2e-4), (it, k)
    # Adam's first steps are ~ lr * sign(g): a few lr is the resolution of a parameter comparison
    tol = 3 * nat.lr
    assert (nat._pflat - ref._pflat).abs().max().item() < tol
    assert (nat._cflat - ref._cflat).abs().max().item() < tol
    assert (nat._ctflat - ref._ctflat).abs().max().item() < tol
    assert (nat._pflat - ref._pflat).abs().mean().item() < 0.02 * nat.lr
    assert (nat._cflat - ref._cflat).abs().mean().item() < 0.02 * nat.lr
    assert torch.allclose(nat.alpha, ref.alpha, rtol=1e-4)
    assert nat.total_update == ref.total_update == n_upd


@pytest.mark.gpu
def test_native_update_draws_its_own_rows_and_noise():
    """Without idx / eps the kernels draw rows and normals from Philox (seed, update count):
    reproducible bit for bit, different from one update to the next, finite."""
    import torch
    a1, _, _ = _twin_sacs(_layout(), {}, rows=500, seed=5)
    a2, _, _ = _twin_sacs(_layout(), {}, rows=500, seed=5)
    p0 = a1._pflat.clone()
    for _ in range(3):
        l1 = a1.update_from_memory(128)
        l2 = a2.update_from_memory(128)
    assert torch.equal(a1._pflat, a2._pflat) and torch.equal(a1._cflat, a2._cflat)
    assert all(torch.equal(l1[k], l2[k]) for k in l1)
    assert all(torch.isfinite(v).all() for v in l1.values())
    assert (a1._pflat - p0).abs().max().item() > 1e-5
    first = a1.update_from_memory(128)["q1"].clone()
    second = a1.update_from_memory(128)["q1"].clone()
    assert not torch.equal(first, second)
