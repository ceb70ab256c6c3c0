"""Fused agent-side kernels of the library (state split, policy tail, state assembly, per-agent
rewards) against the tensor-operation paths they replace (which the CPU tests pin to the
reference's host code)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from ao_marl_amd.agents import AgentLayout, BatchedGaussianPolicy  # noqa: E402
from tests import helpers  # noqa: E402


def _layout():
    return AgentLayout(1283, [0, 1274], 13, include_tip_tilt=True, window_n_zernike=20,
                       include_tip_tilt_windowed=True, n_filtered=5)


def test_split_states_and_policy_tail_match_tensor_ops():
    lay = _layout()
    pol = BatchedGaussianPolicy(lay, last_layer_zero=False, seed=5, device="cuda:0")
    with torch.no_grad():
        pol.bm.normal_(0, 0.3); pol.bs.normal_(0, 0.3)
    st = torch.randn(33, lay.state_dim, device="cuda:0")
    pol.use_native = True
    x1 = pol.split_states(st)
    pol.use_native = False
    x0 = pol.split_states(st)
    assert torch.equal(x0, x1)
    eps = torch.randn(33, lay.action_dim, device="cuda:0")
    a0, m0 = pol.select_action(st, eps=eps)
    pol.use_native = True
    a1, m1 = pol.select_action(st, eps=eps)
    assert torch.allclose(a0, a1, atol=5e-5, rtol=1e-4) and torch.allclose(m0, m1, atol=5e-5, rtol=1e-4)
    # own counter-based normals: reproducible per (seed, draw), fresh per draw, N(0, 1)
    big = torch.randn(512, lay.state_dim, device="cuda:0")
    pol._draws = 0
    b1, mu = pol.select_action(big)
    b2, _ = pol.select_action(big)
    pol._draws = 0
    b3, _ = pol.select_action(big)
    assert torch.equal(b1, b3) and not torch.equal(b1, b2)
    # with a zero last layer the action is tanh(eps): recover the normals
    flat = BatchedGaussianPolicy(lay, last_layer_zero=True, seed=9, device="cuda:0")
    act, mu0 = flat.select_action(big)
    assert mu0.abs().max().item() == 0.0
    z = torch.atanh(act.clamp(-0.9999999, 0.9999999))
    assert abs(z.mean().item()) < 0.01 and abs(z.std().item() - 1.0) < 0.01
    assert abs((z ** 4).mean().item() - 3.0) < 0.1                  # Gaussian kurtosis
    c = torch.corrcoef(torch.stack([z[:, 0], z[:, 1], z[0, :512], z[1, :512]]))
    assert (c - torch.eye(4, device=c.device)).abs().max().item() < 0.2


def test_state_assembly_and_rewards_match_tensor_ops():
    from ao_marl_amd import libaomarl as la
    g = torch.Generator(device="cuda:0").manual_seed(1)
    nenv, nm = 19, 1283
    blocks = [torch.randn(nenv, nm, device="cuda:0", generator=g) for _ in range(3)]
    padded = torch.randn(nenv, 1288, device="cuda:0", generator=g)
    blocks.append(padded[:, :nm])                                   # a strided view
    norms = [(torch.randn(nm, device="cuda:0", generator=g),
              torch.rand(nm, device="cuda:0", generator=g) + 0.5) for _ in range(2)]
    nl = [norms[0], norms[0], norms[0], norms[1]]
    got = la.assemble_state(blocks, nl)
    want = torch.cat([(b - m) / s for b, (m, s) in zip(blocks, nl)], dim=1)
    assert torch.allclose(got, want, atol=1e-6, rtol=1e-6)
    raw = la.assemble_state(blocks, None)
    assert torch.equal(raw, torch.cat(blocks, dim=1))
    lay = _layout()
    lohi = torch.tensor([list(v) for v in lay.agents.values()], dtype=torch.int32, device="cuda:0")
    r = la.agent_rewards(blocks[3], lohi, 1000.0)
    want = torch.stack([-1000.0 * (blocks[3][:, a:b] ** 2).mean(dim=1) for a, b in lay.agents.values()], dim=1)
    assert torch.allclose(r, want, rtol=1e-5, atol=1e-6)


def test_env_step_with_fused_glue_equals_tensor_op_path():
    from ao_marl_amd.env import VecAoEnv
    rl = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5)
    a = VecAoEnv("production_sh_10x10_2m", 3, rl, n_agents_modal=1, frame_pipeline=False)
    b = VecAoEnv("production_sh_10x10_2m", 3, rl, n_agents_modal=1, frame_pipeline=False)
    assert a._native_glue and a._default_state_layout
    b._native_glue = False
    sa, sb = a.reset(), b.reset()
    assert torch.allclose(sa, sb, rtol=1e-5, atol=1e-5 * sb.abs().max().item())
    g = torch.Generator(device="cuda:0").manual_seed(0)
    for _ in range(4):
        act = torch.rand(3, a.layout.action_dim, device="cuda:0", generator=g) * 2 - 1
        sa, ra, _, _ = a.step(act)
        sb, rb, _, _ = b.step(act)
        assert torch.allclose(sa, sb, rtol=1e-4, atol=1e-5 * sb.abs().max().item())
        assert torch.allclose(ra, rb, rtol=1e-4, atol=1e-6)


def test_modal_shortcut_equals_full_projection_path():
    """Carrying the Btt coordinates of the command by linearity (aomarl_rl_control_modes: no v2m
    GEMM in rl_control, none for the next state) gives the states / rewards of the path that
    projects the command explicitly, to fp32 round-off, over a closed-loop rollout."""
    from ao_marl_amd.env import VecAoEnv
    rl = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5)
    a = VecAoEnv("production_sh_10x10_2m", 3, rl, n_agents_modal=1, frame_pipeline=False)
    b = VecAoEnv("production_sh_10x10_2m", 3, rl, n_agents_modal=1, frame_pipeline=False)
    b.modal_shortcut = False
    a.native_step = False                        # call by call: supervisor.last_modes is observable
    sa, sb = a.reset(), b.reset()
    assert torch.equal(sa, sb)
    g = torch.Generator(device="cuda:0").manual_seed(5)
    used = 0
    for it in range(12):
        act = torch.rand(3, a.layout.action_dim, device="cuda:0", generator=g) * 2 - 1
        lin = it in (4, 5)                       # integrator-only steps in between
        sa, ra, _, _ = a.step(act, linear_control=lin)
        sb, rb, _, _ = b.step(act, linear_control=lin)
        used += a.supervisor.last_modes is not None
        # states are standardised with tiny std for some modes: compare in modal units
        std = torch.cat([a.norm["dm"][1]] * 3 + [a.norm["dm_residual"][1]])
        std = torch.where(torch.isfinite(std), std, torch.zeros_like(std))      # (masked columns: states are 0 on both sides)
        scale = (sb * std).abs().max().item()
        assert ((sa - sb) * std).abs().max().item() < 2e-5 * scale, it
        assert torch.allclose(ra, rb, rtol=2e-4, atol=1e-6), it
        cb = b.supervisor.get_command()
        assert (a.supervisor.get_command() - cb).abs().max().item() < 2e-5 * cb.abs().max().item()
    assert used == 10


@pytest.mark.parametrize("denoise,fused_tail", [(False, True), (False, False), (True, True)])
def test_one_call_step_is_the_call_by_call_step(denoise, fused_tail):
    """aomarl_env_step issues the work of rl_step + rewards + linear_step from C -- by default in fused
    form (10 launches instead of 14, see include/aomarl.h): states, rewards,
    commands, Strehl of a rollout are those of the call-by-call path BIT FOR BIT, also when the two
    are mixed (integrator-only steps and dictionary states go call by call), with per-agent rewards
    for several agents, and through the denoiser branch."""
    from ao_marl_amd.env import VecAoEnv
    from ao_marl_amd.denoiser import SubapDenoiser
    name = "production_sh_40x40_8m_3layers_d0_noise" if denoise else "production_sh_10x10_2m"
    rl = dict(n_zernike_start_end=[0, 1274] if denoise else [0, 80], n_reverse_filtered_from_cmat=5)
    nag = 13 if denoise else 2
    kw = {}
    if denoise:     # no statistics shipped for the d0 file: those of its d1 sibling (same system) do
        from ao_marl_amd.env import load_norm
        kw["norm"], kw["zn_norm"] = load_norm("production_sh_40x40_8m_3layers_d1_noise")
    mk = lambda: VecAoEnv(name, 3, rl, n_agents_modal=nag, frame_pipeline=False,     # noqa: E731  (pieces are mixed in)
                          autoencoder=SubapDenoiser.load(device="cuda:0") if denoise else None, **kw)
    a, b = mk(), mk()
    b.native_step = False
    a.fused_tail = fused_tail       # True: split-K sums folded into their consumers, kernels sharing launches
    a.supervisor.sim.set_option("small_chain", 0)     # the general chain: the one the call-by-call path is, bit for bit
    sa, sb = a.reset(), b.reset()
    assert torch.equal(sa, sb)
    g = torch.Generator(device="cuda:0").manual_seed(11)
    native = 0
    for it in range(10):
        act = torch.rand(3, a.layout.action_dim, device="cuda:0", generator=g) * 2 - 1
        lin = it in (3, 4)
        native += a._native_step_ok(lin) and it != 7
        if it == 7:                              # the pieces, by hand, with a dictionary state
            for e in (a, b):
                e.rl_step(act)
                e._r = e.divide_rewards_for_agents()
                e._s = torch.cat(list(e.linear_step(return_dict=True).values()), dim=1)
            sa, ra, sb, rb = a._s, a._r, b._s, b._r
        else:
            sa, ra, _, _ = a.step(act, linear_control=lin)
            sb, rb, _, _ = b.step(act, linear_control=lin)
        assert torch.equal(sa, sb), it
        assert torch.equal(ra, rb), it
        assert torch.equal(a.supervisor.sim.t["voltage"], b.supervisor.sim.t["voltage"]), it
        assert torch.equal(a.supervisor.sim.t["strehl"], b.supervisor.sim.t["strehl"]), it
    assert native == 7                           # steps 3, 4 (integrator only) and 7 (by hand) go call by call
    assert torch.equal(a.supervisor.get_command(), b.supervisor.get_command())
    assert torch.equal(a.supervisor.get_err(), b.supervisor.get_err())


PUBLISHED = dict(n_zernike=[0, 1260], n_modal=42)     # README.md:116-119: 42 x 30 modes + tip-tilt = 43 agents


@pytest.mark.parametrize("which", ["bench14", "published43", "published43_w20"])
def test_one_call_actor_is_the_layer_by_layer_actor(which):
    """aomarl_actor_forward: with AOMARL_ACTOR_LAYER_BY_LAYER the very kernels of the call-by-call
    path (bit for bit); by default ONE kernel (k_actor_fused) whose fp32 sums run in another order:
    same draws, actions within fp32 round-off of the layered ones, and of a float64 evaluation.  Layouts: the
    bench's 14 windowed agents and the reference's published 43 agents (plain: 120-wide states, 30 actions;
    window 20: 280-wide states)."""
    if which == "bench14":
        lay = _layout()
    else:
        w20 = which.endswith("w20")
        lay = AgentLayout(1283, PUBLISHED["n_zernike"], PUBLISHED["n_modal"], include_tip_tilt=True,
                          window_n_zernike=20 if w20 else -1, include_tip_tilt_windowed=w20, n_filtered=5)
        assert lay.n_agents == 43 and lay.state_shapes()[0] == (280 if w20 else 120)
    mk = lambda: BatchedGaussianPolicy(lay, last_layer_zero=False, seed=5, device="cuda:0")   # noqa: E731
    a, b, f = mk(), mk(), mk()
    a.layer_by_layer = True
    b.native_forward = False
    for p in (a, b, f):
        with torch.no_grad():
            g = torch.Generator(device="cuda:0").manual_seed(3)
            p.bm.copy_(torch.randn(p.bm.shape, device="cuda:0", generator=g) * 0.3)
            p.bs.copy_(torch.randn(p.bs.shape, device="cuda:0", generator=g) * 0.3)
            p.b1.copy_(torch.randn(p.b1.shape, device="cuda:0", generator=g) * 0.1)
        p._native = None
    for nenv in (33, 256, 7):
        st = torch.randn(nenv, lay.state_dim, device="cuda:0") * 3
        for ev in (False, True):
            x, mx = a.select_action(st, eval_mode=ev)
            y, my = b.select_action(st, eval_mode=ev)
            z, mz = f.select_action(st, eval_mode=ev)
            assert torch.equal(x, y) and torch.equal(mx, my)
            assert (z - y).abs().max().item() < 2e-5 and (mz - my).abs().max().item() < 2e-5, nenv
    eps = torch.randn(7, lay.action_dim, device="cuda:0")
    assert torch.equal(a.select_action(st, eps=eps)[0], b.select_action(st, eps=eps)[0])
    zf, mf = f.select_action(st, eps=eps)
    assert (zf - b.select_action(st, eps=eps)[0]).abs().max().item() < 2e-5
    # float64 evaluation of the same network from the stacked weights
    pad = torch.cat([st, st.new_zeros(st.shape[0], 1)], dim=1).double()
    x = pad[:, f.gather].permute(1, 0, 2)
    x = torch.relu(torch.baddbmm(f.b1.double(), x, f.W1.double()))
    for W, bb in zip(f.Wh, f.bh):
        x = torch.relu(torch.baddbmm(bb.double(), x, W.double()))
    mu = torch.tanh(torch.baddbmm(f.bm.double(), x, f.Wm.double()))
    want = mu[f.sc_agent, :, f.sc_local].T
    assert (mf.double() - want).abs().max().item() < 5e-6
    # new weights are picked up (the update invalidates the inference copies)
    for p in (a, b, f):
        with torch.no_grad():
            p.W1.mul_(0.5)
        p._native = None
    a._draws = b._draws = f._draws = 100
    y = b.select_action(st)[0]
    assert torch.equal(a.select_action(st)[0], y)
    assert (f.select_action(st)[0] - y).abs().max().item() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 1377), (256, 91, 256), (37, 256, 250), (1, 1, 3), (130, 66, 64)])
def test_gemm_batched_all_transposes(ta, tb, M, N, K):
    """aomarl_gemm_batched against torch (fp64 accumulate): every operand layout, ragged sizes,
    bias + relu + accumulate."""
    import torch
    from ao_marl_amd import libaomarl as L
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    nb = 5
    A = torch.randn((nb, K, M) if ta else (nb, M, K), generator=g, device="cuda")
    B = torch.randn((nb, K, N) if tb else (nb, N, K), generator=g, device="cuda")
    bias = torch.randn(nb, N, generator=g, device="cuda")
    opA = A.transpose(1, 2) if ta else A
    opB = B if tb else B.transpose(1, 2)
    ref = torch.bmm(opA.double(), opB.double())
    tol = 2e-5 * (K ** 0.5) + 1e-6
    out = L.gemm_batched(A, B, bool(ta), bool(tb))
    assert (out.double() - ref).abs().max().item() < tol
    out2 = L.gemm_batched(A, B, bool(ta), bool(tb), bias=bias, relu=True)
    assert (out2.double() - torch.relu(ref + bias.double().unsqueeze(1))).abs().max().item() < tol
    acc = torch.randn(nb, M, N, generator=g, device="cuda")
    want = acc.double() + ref
    L.gemm_batched(A, B, bool(ta), bool(tb), out=acc, accumulate=True)
    assert (acc.double() - want).abs().max().item() < tol


@pytest.mark.gpu
def test_stacked_linear_gradients_match_torch():
    """The HIP-backed autograd layer gives the gradients torch.baddbmm + relu gives."""
    import torch
    from ao_marl_amd.sac import stacked_linear
    g = torch.Generator(device="cuda").manual_seed(5)
    A, Bn, ni, no = 14, 256, 183, 256
    x = torch.randn(A, Bn, ni, generator=g, device="cuda", requires_grad=True)
    W = (torch.randn(A, ni, no, generator=g, device="cuda") * 0.1).requires_grad_()
    b = torch.randn(A, 1, no, generator=g, device="cuda", requires_grad=True)
    w2 = torch.randn(A, Bn, no, generator=g, device="cuda")
    for relu in (True, False):
        y = stacked_linear(x, W, b, relu)
        gx, gW, gb = torch.autograd.grad((y * w2).sum(), [x, W, b])
        yr = torch.baddbmm(b, x, W)
        yr = torch.relu(yr) if relu else yr
        rx, rW, rb = torch.autograd.grad((yr * w2).sum(), [x, W, b])
        assert (y - yr).abs().max().item() < 2e-4
        for u, v in ((gx, rx), (gW, rW), (gb, rb)):
            assert (u - v).abs().max().item() < 3e-4 * max(1.0, v.abs().max().item())


@pytest.mark.gpu
def test_prefetched_atmosphere_gives_the_same_frames():
    """move_atmos of the next frame on the side stream (aomarl_prefetch_atmos, the default of the
    vectorised supervisor) against the plain call order: slopes, commands, Strehl and the screens
    after an explicit move are identical bit for bit, across a reset."""
    import torch
    from ao_marl_amd.env import VecRlSupervisor
    sups = [VecRlSupervisor("production_sh_10x10_2m", {}, 5, initial_seed=77, prefetch_atmos=p)
            for p in (True, False)]
    assert sups[0].prefetch_atmos and not sups[1].prefetch_atmos
    for ep in range(2):
        for s in sups:
            s.reset()
        for it in range(7):
            outs = []
            for s in sups:
                s.next_part_one()
                s.next_part_two(None, linear_control=True)
                outs.append((s.get_slopes().clone(), s.get_command().clone(), s.get_strehl().clone()))
            assert sups[0].sim.pending_atmos and not sups[1].sim.pending_atmos
            for a, b in zip(*outs):
                assert torch.equal(a, b), (ep, it)
    # imaging the same atmosphere twice is refused while a prefetched frame is pending ...
    with pytest.raises(RuntimeError):
        sups[0].next_part_one(move_atmos=False)
    # ... and the explicit move consumes the prefetched one: same screens as two plain moves
    sups[0].sim.move_atmos()
    sups[1].sim.move_atmos()
    for layer in range(sups[0].s.nscreens):
        assert torch.equal(sups[0].sim.screen(layer), sups[1].sim.screen(layer))


@pytest.mark.gpu
def test_two_live_environments_share_the_side_streams():
    """Every context of the process prefetches on the SAME pair of side streams (one pair per device): two
    supervisors alive and stepped in turn, each one frame ahead on those streams, give what each gives
    alone in plain call order, bit for bit."""
    import torch
    from ao_marl_amd.env import VecRlSupervisor
    mk = lambda seed, p: VecRlSupervisor("production_sh_10x10_2m", {}, 4, initial_seed=seed, prefetch_atmos=p)  # noqa: E731
    a, b = mk(11, True), mk(500, True)
    ra, rb = mk(11, False), mk(500, False)
    for s in (a, b, ra, rb):
        s.reset()
    for it in range(9):
        for s, r in ((a, ra), (b, rb)):
            s.next_part_one(); r.next_part_one()
            s.next_part_two(None, linear_control=True); r.next_part_two(None, linear_control=True)
        assert a.sim.pending_atmos and b.sim.pending_atmos
        for s, r in ((a, ra), (b, rb)):
            assert torch.equal(s.get_slopes(), r.get_slopes()) and torch.equal(s.get_command(), r.get_command()), it
            assert torch.equal(s.get_strehl(), r.get_strehl()), it
    assert not torch.equal(a.get_slopes(), b.get_slopes())


@pytest.mark.gpu
def test_prefetch_with_partial_ranges_equals_batch_stepping():
    """prefetch_atmos on, the batch stepped as two halves through the composite: the first half runs
    one frame ahead, the second in plain order -- same bits as stepping the whole batch."""
    import torch
    from ao_marl_amd import geometry as G, params, system
    from ao_marl_amd.sim import HipSim
    s = system.from_system(G.build_system(params.builtin("production_sh_10x10_2m")))
    s.cmat = (np.random.default_rng(0).standard_normal((s.nactu, s.nslope)) * 1e-3).astype(np.float32)
    a, b = HipSim(s, nenv=4), HipSim(s, nenv=4)
    b.set_option("prefetch_atmos", 1)
    for x in (a, b):
        x.reset([5, 6, 7, 8])
    for it in range(4):
        a.next_part_two(None)
        a.next_part_one()
        for (e0, cnt) in ((0, 2), (2, 2)):
            b.next_part_two(None, env_begin=e0, env_count=cnt)
            b.next_part_one(env_begin=e0, env_count=cnt)
    assert torch.equal(a.slopes, b.slopes) and torch.equal(a.com, b.com)
    assert torch.equal(a.strehl, b.strehl)


@pytest.mark.gpu
def test_residual_modes_from_slopes_equals_do_control_path():
    """v2m . err straight from the slopes (aomarl_slopes2modes, do_control deferred until somebody
    needs err / com in actuator space) against do_control + volts2modes every frame: states and
    rewards over a rollout with integrator-only steps in between, then err, command and voltages."""
    from ao_marl_amd.env import VecAoEnv
    rl = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5)
    a = VecAoEnv("production_sh_10x10_2m", 3, rl, n_agents_modal=1, frame_pipeline=False)
    b = VecAoEnv("production_sh_10x10_2m", 3, rl, n_agents_modal=1, frame_pipeline=False)
    assert not a.residual_shortcut and not b.residual_shortcut          # opt-in
    a.residual_shortcut = True
    sa, sb = a.reset(), b.reset()
    g = torch.Generator(device="cuda:0").manual_seed(11)
    std = torch.cat([a.norm["dm"][1]] * 3 + [a.norm["dm_residual"][1]])
    std = torch.where(torch.isfinite(std), std, torch.zeros_like(std))          # (masked columns: states are 0 on both sides)
    deferred = 0
    for it in range(10):
        act = torch.rand(3, a.layout.action_dim, device="cuda:0", generator=g) * 2 - 1
        lin = it in (3, 7)
        sa, ra, _, _ = a.step(act, linear_control=lin)
        sb, rb, _, _ = b.step(act, linear_control=lin)
        deferred += a.supervisor._control_pending
        assert not b.supervisor._control_pending
        scale = (sb * std).abs().max().item()
        assert ((sa - sb) * std).abs().max().item() < 2e-5 * scale, it
        assert torch.allclose(ra, rb, rtol=2e-4, atol=1e-6), it
    # (the one-call step of this SMALL system runs do_control in its tail kernel -- aomarl_env_step_shortcut is 0 there --;
    # the two call-by-call steps defer it.  The 40x40 system's one-call step defers it too: see
    # test_residual_shortcut_inside_the_one_call_step)
    assert deferred == 2
    # actuator-space quantities on demand: err of the last frame, the integrated command, voltages
    eb = b.supervisor.get_err()
    assert (a.supervisor.get_err() - eb).abs().max().item() < 2e-5 * eb.abs().max().item()
    assert not a.supervisor._control_pending
    cb = b.supervisor.get_command()
    assert (a.supervisor.get_command() - cb).abs().max().item() < 2e-5 * cb.abs().max().item()
    vb = b.supervisor.get_voltages()
    assert (a.supervisor.get_voltages() - vb).abs().max().item() < 2e-5 * vb.abs().max().item()
    # err after a step whose command came from modal coordinates: recomputed without integrating
    act = torch.rand(3, a.layout.action_dim, device="cuda:0", generator=g) * 2 - 1
    a.rl_step(act)
    b.rl_step(act)
    ca = a.supervisor.get_command().clone()
    eb = b.supervisor.get_err()
    assert (a.supervisor.get_err() - eb).abs().max().item() < 2e-5 * eb.abs().max().item()
    assert torch.equal(a.supervisor.get_command(), ca)


@pytest.mark.gpu
@pytest.mark.parametrize("config,nenv,rl,n_modal,prefetch", [
    ("production_sh_10x10_2m", 8, dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5), 1, True),
    ("production_sh_10x10_2m", 8, dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5), 1, False),
    ("production_sh_40x40_8m_3layers", 4, dict(n_zernike_start_end=[0, 1274], n_reverse_filtered_from_cmat=5,
                                               window_n_zernike=20, include_tip_tilt_windowed=True), 13, True)])
def test_graph_step_replays_the_same_step(config, nenv, rl, n_modal, prefetch):
    """aomarl_set_option("graph_step", 1): aomarl_env_step captured as HIP graphs (one per extrusion plan x ring
    position x buffer addresses) and replayed -- bit for bit the plain call's states, rewards, slopes, commands and
    Strehl over 40 steps, with graphs both captured and replayed along the way.  prefetch False: the whole step on
    ONE stream, a linear graph (no fork / join), the form meant for the host-bound small systems."""
    from ao_marl_amd.env import VecAoEnv
    out = {}
    rng = np.random.default_rng(3)
    actions = None
    for mode in ("plain", "graph"):
        env = VecAoEnv(config, nenv, rl, initial_seed=77, seed_stride=16, n_agents_modal=n_modal, prefetch_atmos=prefetch, frame_pipeline=False)
        sim = env.supervisor.sim
        sim.set_option("graph_step", 1 if mode == "graph" else 0)
        if actions is None:
            actions = torch.from_numpy(rng.uniform(-1, 1, size=(40, nenv, env.action_dim)).astype(np.float32)).cuda()
        torch.cuda.synchronize()
        with torch.cuda.stream(torch.cuda.Stream()):        # the null stream cannot be captured
            s = env.reset()
            rec = [s.clone()]
            abuf = [torch.empty_like(actions[0]) for _ in range(3)]   # stable addresses, as a policy's output ring gives
            for t in range(40):
                assert env._native_step_ok(False)
                abuf[t % 3].copy_(actions[t])
                s, r, _, _ = env.step(abuf[t % 3])
                rec += [s.clone(), r.clone()]
            rec += [env.supervisor.get_slopes().clone(), env.supervisor.get_command().clone(),
                    env.supervisor.get_strehl().clone()]
            torch.cuda.synchronize()
            out[mode] = (rec, sim.graph_stats())
            # a second episode on the same context: the reset's plain calls and the graphs mix
            if mode == "graph":
                s2 = env.reset()
                for t in range(5):
                    abuf[t % 3].copy_(actions[t])
                    s2, _, _, _ = env.step(abuf[t % 3])
                torch.cuda.synchronize()
                assert torch.isfinite(s2).all()
        del env
    cap, rep = out["graph"][1]
    # 40 steps: at most (plans x 3 ring positions x 2 output-ring phases) distinct graphs, the rest replays
    assert out["plain"][1] == (0, 0) and 2 <= cap <= 30 and rep >= 10 and 38 <= cap + rep <= 40, (cap, rep)   # the first step validates the glue on the plain path
    for a, b in zip(out["plain"][0], out["graph"][0]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_split_gemm_counts_operands_that_leave_the_fp16_range():
    """gh_split clamps scaled operands to +-65504; what is clipped is now counted
    (aomarl_gemm_saturated) and the supervisor raises at the episode boundary, like the denoiser's counter."""
    import ctypes as C
    from ao_marl_amd import libaomarl as la
    L = la.load()
    keep = la.get_precision()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    try:
        la.set_precision("split_f16")
        assert la.gemm_saturated(st) >= 0                  # clears whatever earlier tests left
        M, N, K = 64, 64, 256
        A = torch.randn(M, K, device="cuda")
        B = torch.randn(N, K, device="cuda")
        out = torch.zeros(M, N, device="cuda")
        work = torch.zeros(8 * M * N, device="cuda")

        def gemm(a):
            la.check(L.aomarl_gemm_nt_split(M, N, K, 1.0, a.data_ptr(), K, B.data_ptr(), K, 0.0, out.data_ptr(), N,
                                            16.0, 1.0, work.data_ptr(), work.numel(), st))
        gemm(A)
        assert la.gemm_saturated(st) == 0
        assert (out - A @ B.T).abs().max().item() < 1e-4 * K ** 0.5 * 4
        A2 = A.clone()
        A2[3, 17] = 5000.0                                 # x 16 = 80000 > 65504: clipped
        gemm(A2)
        n = la.gemm_saturated(st)
        assert n >= 1
        assert la.gemm_saturated(st) == 0                  # reading clears
        # the environment turns it into an error at the next reset
        from ao_marl_amd.env import VecRlSupervisor
        sup = VecRlSupervisor("production_sh_10x10_2m", dict(n_reverse_filtered_from_cmat=5), 2)
        sup.reset()
        gemm(A2)
        with pytest.raises(FloatingPointError):
            sup.reset()
        sup.reset()                                        # cleared: the next episode starts normally
    finally:
        la.gemm_saturated(st)
        la.set_precision(keep)


@pytest.mark.gpu
def test_env_step_refuses_inconsistent_arguments():
    """aomarl_env_step validates its glue before the first launch (a C caller got a device fault or a
    silently wrong gain): null rings, rewards without agent ranges, a column selection outside the modes,
    per-environment gains left on the context."""
    import ctypes as C
    from ao_marl_amd import libaomarl as la
    from ao_marl_amd.env import VecAoEnv
    env = VecAoEnv("production_sh_10x10_2m", 2, dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5),
                   n_agents_modal=1, frame_pipeline=False)
    s = env.reset()
    a = torch.zeros(2, env.action_dim, device="cuda")
    s, r, _, _ = env.step(a)                               # builds the glue, validates the selection
    sim, g = env.supervisor.sim, env._glue
    state = torch.empty_like(s)
    rew = torch.empty_like(r)

    def call(glue, reward=rew):
        return sim.lib.aomarl_env_step(sim.ctx, C.byref(sim.st), C.byref(glue), a.data_ptr(), 0.7, la.fptr(sim.accumx),
                                       la.fptr(sim.accumy), state.data_ptr(), reward.data_ptr() if reward is not None else None,
                                       sim._stream())

    def variant(**kw):
        v = la.EnvGlue()
        C.memmove(C.byref(v), C.byref(g), C.sizeof(v))
        for k, val in kw.items():
            setattr(v, k, val)
        return v
    assert call(variant(modes_ring=None)) != 0 and b"modes_ring" in sim.lib.aomarl_last_error()
    assert call(variant(lohi=None)) != 0 and b"lohi" in sim.lib.aomarl_last_error()
    assert call(variant(dm_dim=env.nmodes + 1)) != 0
    assert call(variant(std_res=None)) != 0
    bad = torch.full((g.dm_dim,), env.nmodes + 5, dtype=torch.int32, device="cuda")
    assert call(variant(sel=bad.data_ptr())) != 0 and b"outside" in sim.lib.aomarl_last_error()
    sim.set_env_gains([0.5, 0.6])
    assert call(variant()) != 0 and b"per-environment" in sim.lib.aomarl_last_error()
    sim.set_env_gains(None)
    torch.cuda.synchronize()
    s2, _, _, _ = env.step(a)                              # and the loop goes on unharmed
    assert torch.isfinite(s2).all()


@pytest.mark.gpu
@pytest.mark.parametrize("layers", [1, 3])
def test_small_screen_move_in_one_launch_equals_the_generic_rounds(layers):
    """Screens of at most 256 pixels: aomarl_move_atmos is ONE k_move_small launch (every extrusion of the
    frame, one block per environment and layer) instead of gather / GEMM / scatter rounds.  Same Philox
    draws, same ring writes, fp32 sums in another order: the logical screens agree to rounding after the
    ring has wrapped, with winds of all four sign combinations and different extrusion counts per layer."""
    import torch
    from ao_marl_amd import geometry as G, params, system
    from ao_marl_amd.sim import HipSim
    ps = params.builtin("production_sh_10x10_2m")
    if layers == 3:
        a = ps.p_atmos
        a.nscreens = 3
        a.frac = np.asarray([0.5, 0.3, 0.2], dtype=np.float32)
        a.alt = np.asarray([0.0, 0.0, 0.0], dtype=np.float32)
        a.windspeed = np.asarray([20.0, 33.0, 9.0], dtype=np.float32)
        a.winddir = np.asarray([45.0, 200.0, -60.0], dtype=np.float32)
        a.L0 = np.asarray([1.e5, 25.0, 1.e5], dtype=np.float32)
        ps = ps.validate()
    s = system.from_system(G.build_system(ps))
    s.cmat = np.zeros((s.nactu, s.nslope), dtype=np.float32)
    one, gen = HipSim(s, nenv=5), HipSim(s, nenv=5)
    gen.set_option("small_move", 0)
    seeds = [3, 1000, 77, 12345, 9]
    one.reset(seeds); gen.reset(seeds)
    nmoves = 3 * max(s.screen_dim) // 2
    for it in range(nmoves):
        one.move_atmos(); gen.move_atmos()
        if it in (0, 1, 7, nmoves - 1):
            for layer in range(s.nscreens):
                x, y = one.screen(layer), gen.screen(layer)
                scale = y.abs().max().item()
                assert (x - y).abs().max().item() < (2e-5 if it < 8 else 2e-4) * scale, (it, layer)   # rounding drifts on the undamped low orders
    # and through the step: frames formed from either atmosphere agree
    one.next_part_one(); gen.next_part_one()
    assert (one.slopes - gen.slopes).abs().max().item() < 1e-3 * gen.slopes.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("config,nenv,rl,n_modal", [
    ("production_sh_10x10_2m", 8, dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5), 1),
    ("production_sh_40x40_8m_3layers", 4, dict(n_zernike_start_end=[0, 1274], n_reverse_filtered_from_cmat=5,
                                               window_n_zernike=20, include_tip_tilt_windowed=True), 13)])
def test_frame_pipeline_gives_the_plain_steps(config, nenv, rl, n_modal):
    """aomarl_set_frame_pipeline (VecAoEnv.frame_pipeline, the default where the loop delay is one frame): frame
    t+1 launched by the call of step t, before frame t is reduced, the atmosphere moved beside the frame in
    flight -- bit for bit the plain call order's states, rewards, slopes, voltages, commands and Strehl over two
    episodes; while a frame is in flight anything but env_step / reset is refused."""
    from ao_marl_amd.env import VecAoEnv
    from ao_marl_amd.libaomarl import AomarlError
    rng = np.random.default_rng(5)
    out, actions = {}, None
    for mode in ("plain", "pipe"):
        env = VecAoEnv(config, nenv, rl, initial_seed=31, seed_stride=16, n_agents_modal=n_modal, frame_pipeline=False)
        env.frame_pipeline = mode == "pipe"
        sim = env.supervisor.sim
        if actions is None:
            actions = torch.from_numpy(rng.uniform(-1, 1, size=(24, nenv, env.action_dim)).astype(np.float32)).cuda()
        rec = []
        with torch.cuda.stream(torch.cuda.Stream()):
            for ep in range(2):
                s = env.reset()
                rec.append(s.clone())
                for t in range(24 if ep == 0 else 7):
                    assert env._native_step_ok(False)
                    s, r, _, _ = env.step(actions[t])
                    rec += [s.clone(), r.clone()]
                    if t in (0, 1, 2, 11, 23):
                        rec += [env.supervisor.get_slopes().clone(), sim.voltage.clone(), env.supervisor.get_command().clone(),
                                env.supervisor.get_strehl().clone()]
                if mode == "pipe":
                    flying, _, steps, beside = sim.frame_pipeline_state()
                    assert flying and steps >= 6 and beside >= 6, (steps, beside)
                    with pytest.raises(AomarlError):
                        sim.move_atmos()                    # screens are a frame ahead: refused, loudly
                    with pytest.raises(AomarlError):
                        sim.comp_strehl()
            torch.cuda.synchronize()
        out[mode] = rec
        if mode == "pipe":
            env.reset()
            assert not sim.frame_pipeline_state()[0]        # the reset dropped the frame in flight
            sim.move_atmos()
            torch.cuda.synchronize()
        else:
            assert sim.frame_pipeline_state() == (False, False, 0, 0)
        del env
    assert len(out["plain"]) == len(out["pipe"])
    for k, (a, b) in enumerate(zip(out["plain"], out["pipe"])):
        assert torch.equal(a, b), k


@pytest.mark.gpu
def test_frame_pipeline_is_not_used_where_the_voltages_depend_on_the_new_command():
    """Loop delay below one frame: the voltages of a frame contain the command computed from the previous frame's
    slopes, nothing can run ahead -- an environment that asks for the pipeline steps in the plain order (same
    values as one that does not ask), and a partial reset while a frame IS in flight is refused."""
    from ao_marl_amd import params
    from ao_marl_amd.env import VecAoEnv
    from ao_marl_amd.libaomarl import AomarlError
    rl = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5)
    outs = []
    for pipe in (True, False):
        ps = params.builtin("production_sh_10x10_2m")
        for c in ps.p_controllers:
            c.delay = 0.5
        env = VecAoEnv(ps.validate(), 4, rl, initial_seed=9, n_agents_modal=1, frame_pipeline=pipe)
        s = env.reset()
        g = torch.Generator(device="cuda:0").manual_seed(3)
        for it in range(6):
            assert env._native_step_ok(False)
            s, r, _, _ = env.step(torch.rand(4, env.action_dim, device="cuda:0", generator=g) * 2 - 1)
        assert env.supervisor.sim.frame_pipeline_state()[:3] == (False, False, 0)
        outs.append((s.clone(), r.clone(), env.supervisor.get_slopes().clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    env = VecAoEnv("production_sh_10x10_2m", 4, rl, initial_seed=9, n_agents_modal=1, frame_pipeline=True)
    env.reset()
    for it in range(3):
        env.step(torch.zeros(4, env.action_dim, device="cuda:0"))
    sim = env.supervisor.sim
    assert sim.frame_pipeline_state()[0]
    with pytest.raises(AomarlError):
        sim.reset([1, 2], env_begin=0, env_count=2)
    env.reset()
    assert not sim.frame_pipeline_state()[0]


@pytest.mark.gpu
def test_frame_pipeline_over_a_long_episode_with_ring_wraps():
    """700 pipelined steps of the production system (every ring wraps at least once, the fastest layer three
    times; every extrusion plan of the file occurs) against the plain order: states and rewards of every 50th
    step, slopes, voltages, commands and Strehl at the end, bit for bit; every move of the pipelined run went beside
    the frame in flight."""
    from ao_marl_amd.env import VecAoEnv
    rl = dict(n_zernike_start_end=[0, 1274], n_reverse_filtered_from_cmat=5, window_n_zernike=20,
              include_tip_tilt_windowed=True)
    nenv, nstep = 4, 700
    rec = {}
    for mode in ("plain", "pipe"):
        env = VecAoEnv("production_sh_40x40_8m_3layers", nenv, rl, initial_seed=5, seed_stride=16, n_agents_modal=13,
                       frame_pipeline=mode == "pipe")
        sim = env.supervisor.sim
        g = torch.Generator(device="cuda:0").manual_seed(17)
        out = []
        with torch.cuda.stream(torch.cuda.Stream()):
            s = env.reset()
            for t in range(nstep):
                a = (torch.rand(nenv, env.action_dim, device="cuda:0", generator=g) * 2 - 1) * 0.05
                s, r, _, _ = env.step(a)
                if t % 50 == 49:
                    out += [s.clone(), r.clone()]
            out += [env.supervisor.get_slopes().clone(), sim.voltage.clone(), env.supervisor.get_command().clone(),
                    env.supervisor.get_strehl().clone()]
            if mode == "pipe":
                flying, _, steps, beside = sim.frame_pipeline_state()
                assert flying and steps == nstep - 1 and beside == nstep, (steps, beside)
            torch.cuda.synchronize()
        rec[mode] = out
        assert torch.isfinite(out[-1]).all() and float(out[-1][:, 1].min()) > 0.0
        del env
    for k, (a, b) in enumerate(zip(rec["plain"], rec["pipe"])):
        assert torch.equal(a, b), k


@pytest.mark.gpu
def test_small_chain_equals_the_general_chain_to_rounding():
    """Small systems: the control / agent chain of aomarl_env_step as two workgroup-per-environment kernels
    (k_small_head / k_small_tail, default) against the general chain of GEMMs and elementwise kernels ("small_chain" =
    0): same formulas, other summation order -- states, rewards, commands, voltages and Strehl of a 40-step
    closed loop agree to fp32 round-off, in the plain order and with the frame pipeline."""
    from ao_marl_amd.env import VecAoEnv
    rl = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5)
    for pipe in (False, True):
        envs = [VecAoEnv("production_sh_10x10_2m", 6, rl, initial_seed=41, n_agents_modal=1, frame_pipeline=pipe) for _ in range(2)]
        envs[1].supervisor.sim.set_option("small_chain", 0)
        s0, s1 = envs[0].reset(), envs[1].reset()
        assert torch.equal(s0, s1)
        g = torch.Generator(device="cuda:0").manual_seed(23)
        for it in range(40):
            act = torch.rand(6, envs[0].action_dim, device="cuda:0", generator=g) * 2 - 1
            (s0, r0, _, _), (s1, r1, _, _) = envs[0].step(act), envs[1].step(act)
            scale = max(1.0, s1.abs().max().item())
            assert (s0 - s1).abs().max().item() < 2e-4 * scale, (pipe, it)
            assert (r0 - r1).abs().max().item() < 2e-4 * max(1e-6, r1.abs().max().item()), (pipe, it)
        c0, c1 = envs[0].supervisor.get_command(), envs[1].supervisor.get_command()
        assert (c0 - c1).abs().max().item() < 2e-4 * c1.abs().max().item()
        v0, v1 = envs[0].supervisor.sim.voltage, envs[1].supervisor.sim.voltage
        assert (v0 - v1).abs().max().item() < 2e-4 * v1.abs().max().item()
        assert (envs[0].supervisor.get_strehl() - envs[1].supervisor.get_strehl()).abs().max().item() < 1e-4
        assert (envs[0].supervisor.get_err() - envs[1].supervisor.get_err()).abs().max().item() < 2e-4 * envs[1].supervisor.get_err().abs().max().item()
        del envs


def test_default_call_order_is_probed_and_falls_back_loudly():
    """VecAoEnv(frame_pipeline='auto') (opt-in; the constructor's default is the plain order): behind the first reset of an eligible environment both call orders
    are timed on the caller's stream and the pipelined one is kept unless it is the slower one; the fallback (forced
    here through the probe's margin) warns, drops the twin and runs the plain order.  Either way the episode is the
    plain order's, bit for bit; an environment that is not eligible (noisy sensor) never probes."""
    from ao_marl_amd.env import VecAoEnv
    rl = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5)
    VecAoEnv._ORDER_CACHE.clear()
    ref = VecAoEnv("production_sh_10x10_2m", 8, rl, initial_seed=3, n_agents_modal=1, frame_pipeline=False)
    assert ref.frame_pipeline is False
    assert VecAoEnv("production_sh_10x10_2m", 8, rl, initial_seed=3, n_agents_modal=1).frame_pipeline is False   # the default
    auto = VecAoEnv("production_sh_10x10_2m", 8, rl, initial_seed=3, n_agents_modal=1, frame_pipeline="auto")
    assert auto.frame_pipeline == "auto" and auto.order_probe is None
    s0, s1 = ref.reset(), auto.reset()
    assert ref.order_probe is None
    p = auto.order_probe
    assert p is not None and p["chosen"] in ("pipelined", "plain") and p["pipelined"] > 0 and p["plain"] > 0
    assert auto.frame_pipeline is (p["chosen"] == "pipelined")
    assert torch.equal(s0, s1)
    g = torch.Generator(device="cuda:0").manual_seed(4)
    acts = [torch.rand(8, ref.action_dim, device="cuda:0", generator=g) * 2 - 1 for _ in range(12)]
    for a in acts:
        (sa, ra, _, _), (sb, rb, _, _) = ref.step(a), auto.step(a)
        assert torch.equal(sa, sb) and torch.equal(ra, rb)
    if auto.frame_pipeline:
        assert auto.supervisor.sim.frame_pipeline_state()[2] >= 10         # pipelined steps were taken
    # a second environment on the same stream reuses the decision
    again = VecAoEnv("production_sh_10x10_2m", 8, rl, initial_seed=3, n_agents_modal=1, frame_pipeline="auto")
    again.reset()
    assert again.order_probe.get("cached") and again.frame_pipeline is auto.frame_pipeline
    # the frame stream created anew behind a reset (what the probe does when the pipelined order looks aliased): same episode
    s0b, s1b = ref.reset(), auto.reset()
    if auto.frame_pipeline:
        auto.supervisor.sim.renew_frame_stream()
    assert torch.equal(s0b, s1b)
    for a in acts[:8]:
        (sa, ra, _, _), (sb, rb, _, _) = ref.step(a), auto.step(a)
        assert torch.equal(sa, sb) and torch.equal(ra, rb)
    # the fallback, forced: nothing is 'not more than 0 x slower'
    VecAoEnv._ORDER_CACHE.clear()
    fb = VecAoEnv("production_sh_10x10_2m", 8, rl, initial_seed=3, n_agents_modal=1, frame_pipeline="auto")
    fb.frame_pipeline = False
    s2 = fb.reset()
    fb.frame_pipeline = "auto"
    with pytest.warns(UserWarning, match="plain order"):
        s2 = fb._probe_order(s2, margin=0.0, alias_ratio=0.0)
    assert fb.frame_pipeline is False and fb.order_probe["chosen"] == "plain"
    assert fb.order_probe["frame_stream_renewed"] == 2         # (it tried the frame stream anew, twice, first)
    assert getattr(fb.supervisor.sim, "_twin", None) is None
    assert torch.equal(s2, s0)
    ref.reset()
    for a in acts[:6]:
        (sa, ra, _, _), (sb, rb, _, _) = ref.step(a), fb.step(a)
        assert torch.equal(sa, sb) and torch.equal(ra, rb)
    assert fb.supervisor.sim.frame_pipeline_state()[0] is False
    VecAoEnv._ORDER_CACHE.clear()
    # not eligible: no probe, plain order
    noisy = VecAoEnv("production_sh_40x40_8m_3layers_d1_noise", 2,
                     dict(n_zernike_start_end=[0, 1274], n_reverse_filtered_from_cmat=5), n_agents_modal=13,
                     frame_pipeline="auto")
    noisy.reset()
    assert noisy.frame_pipeline is False and noisy.order_probe is None


@pytest.mark.parametrize("name,nenv,whole", [("production_sh_10x10_2m", 6, 1), ("production_sh_40x40_8m_3layers", 32, 1),
                                             ("production_sh_40x40_8m_3layers", 32, 0)])
def test_prefetched_reset_is_the_plain_reset_bit_for_bit(name, nenv, whole):
    """aomarl_reset_prefetch_begin / _advance / aomarl_reset_adopt: the next episode's screens grown in a shadow state
    beside the running episode (a few rounds per step, on a stream of their own) are the plain reset's -- screens,
    ring origins, extrusion counters, and the frames that follow -- bit for bit; the running episode is not
    disturbed; a reset that comes before the rounds are through runs what is left; other seeds drop the prefetch.
    32 environments of the 40x40 system: the reset walks its rounds in two parts of the batch (reset_streams); the
    prefetched one as one range with the parts' k split ("reset_prefetch_whole", the default) or in the same parts."""
    from ao_marl_amd.sim import HipSim
    _, s, cal = helpers.calibrated(name) if "10x10" in name else helpers.calibrated_hip(name)
    sims = []
    for _ in range(2):
        sim = HipSim(s, nenv=nenv)
        sim.set_modal(cal.volts2modes, cal.modes2volts)
        sim.set_option("prefetch_atmos", 1)
        sim.set_option("reset_prefetch_whole", whole)
        sims.append(sim)
    a, b = sims
    s1, s2, s3 = 100 + 16 * np.arange(nenv), 9000 + 16 * np.arange(nenv), 555 + 16 * np.arange(nenv)
    total = 2 * max(s.screen_dim)

    def frames(k):
        for _ in range(k):
            for x in (a, b):
                x.next_part_two(None)
                x.next_part_one()

    def same_state(what):
        for l in range(s.nscreens):
            assert torch.equal(a.screen(l), b.screen(l)), (what, "screen", l)
        assert torch.equal(a.t["origin"], b.t["origin"]) and torch.equal(a.t["ext_count"], b.t["ext_count"]), what
        assert torch.equal(a.t["seeds"], b.t["seeds"]), what
        assert torch.equal(a.slopes, b.slopes) and torch.equal(a.com, b.com) and torch.equal(a.t["strehl"], b.t["strehl"]), what

    a.reset(s1); b.reset(s1)
    b.prefetch_reset_begin(s2)
    assert b.prefetch_reset_pending(s2) and not b.prefetch_reset_pending(s3)
    left = total
    while left:                                  # the rounds a few at a time, frames of the live episode in between
        left = b.prefetch_reset_advance(97)
        frames(1)
    same_state("the running episode beside the prefetch")
    a.reset(s2); b.reset(s2)                     # b adopts
    assert getattr(b, "prefetched_resets", 0) == 1 and not b.prefetch_reset_pending()
    same_state("adopted reset")
    frames(3)
    same_state("frames behind the adopted reset")
    # a reset that comes early: what is left of the rounds runs inside it
    b.prefetch_reset_begin(s3)
    assert b.prefetch_reset_advance(11) == total - 11
    frames(2)
    a.reset(s3); b.reset(s3)
    assert b.prefetched_resets == 2
    same_state("early adoption")
    frames(2)
    same_state("frames behind the early adoption")
    # other seeds: the prefetch is dropped, the reset runs in the open
    b.prefetch_reset_begin(s1)
    b.prefetch_reset_advance(5)
    a.reset(s2); b.reset(s2)
    assert b.prefetched_resets == 2 and not b.prefetch_reset_pending()
    same_state("dropped prefetch")


@pytest.mark.parametrize("pipe", [False, True])
def test_environment_with_prefetched_resets_gives_the_same_episodes(pipe):
    """VecAoEnv(reset_prefetch=...): the next episode's screens grow beside the running one (a share of the rounds per
    step, the rest inside reset()); states, rewards and Strehl of three episodes -- same seeds ("same") and a
    trainer's seed schedule (next_seed_block) -- equal the plain environment's bit for bit, with and without a frame
    in flight; an episode on unexpected seeds drops the prefetch and resets in the open."""
    from ao_marl_amd.env import VecAoEnv
    rl = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5, max_steps_per_episode=30)
    mk = lambda rp: VecAoEnv("production_sh_10x10_2m", 5, rl, initial_seed=11, n_agents_modal=1, frame_pipeline=pipe,  # noqa: E731
                             reset_prefetch=rp)
    g = torch.Generator(device="cuda:0").manual_seed(8)
    acts = [torch.rand(5, 82, device="cuda:0", generator=g) * 2 - 1 for _ in range(30)]

    def episodes(env, schedule):
        rec = []
        for ep, nsteps in enumerate((30, 12, 30)):
            s = env.reset()
            rec.append(s.clone())
            for t in range(nsteps):
                s, r, _, _ = env.step(acts[t])
                rec += [s.clone(), r.clone()]
            rec.append(env.supervisor.get_strehl().clone())
            if schedule == "blocks":
                env.next_seed_block(1)
            elif schedule == "surprise" and ep == 0:
                env.set_sim_seed(4242)
        return rec

    for rp, schedule in (("same", "same"), (1, "blocks"), ("same", "surprise")):
        a, b = mk(None), mk(rp)
        ra, rb = episodes(a, schedule), episodes(b, schedule)
        assert len(ra) == len(rb) and all(torch.equal(x, y) for x, y in zip(ra, rb)), (rp, schedule)
        want = 1 if schedule == "surprise" else 2          # the surprise episode's reset ran in the open
        assert getattr(b.supervisor.sim, "prefetched_resets", 0) == want and getattr(a.supervisor.sim, "prefetched_resets", 0) == 0


def test_residual_shortcut_inside_the_one_call_step():
    """VecAoEnv.residual_shortcut in aomarl_env_step ("residual_shortcut"): the residual modes from ONE product of the
    slopes with v2m . cmat instead of do_control (cmat . s, integrate) + v2m . err -- the integrator lives in the Btt
    coordinates, the head of the next step rebuilds the command from them.  Same mathematics, another order of the
    fp32 sums: states, rewards and commands agree with the reference order to round-off of 2400-term dot products;
    with a frame in flight the shortcut step is the plain shortcut step bit for bit; err / com in actuator space
    appear on demand (do_control on the frame's slopes)."""
    from ao_marl_amd.env import VecAoEnv
    name = "production_sh_40x40_8m_3layers"
    rl = dict(n_zernike_start_end=[0, 1274], n_reverse_filtered_from_cmat=5, window_n_zernike=20, include_tip_tilt_windowed=True)
    mk = lambda pipe: VecAoEnv(name, 4, rl, initial_seed=5, seed_stride=16, n_agents_modal=13, frame_pipeline=pipe)   # noqa: E731
    ref, cut, cutp = mk(False), mk(False), mk(True)
    cut.residual_shortcut = cutp.residual_shortcut = True
    s0, s1, s2 = ref.reset(), cut.reset(), cutp.reset()
    assert torch.allclose(s0, s1, rtol=0, atol=2e-3) and torch.equal(s1, s2)
    g = torch.Generator(device="cuda:0").manual_seed(12)
    worst = 0.0
    for it in range(10):
        a = torch.rand(4, ref.action_dim, device="cuda:0", generator=g) * 2 - 1
        (sa, ra, _, _), (sb, rb, _, _), (sc, rc, _, _) = ref.step(a), cut.step(a), cutp.step(a)
        assert cut._native_step_ok(False) and cut._native_shortcut and cutp._native_shortcut and not ref._native_shortcut
        assert torch.equal(sb, sc) and torch.equal(rb, rc), it            # a frame in flight changes nothing
        live = torch.isfinite(sa) & (sa.abs() < 1e3)
        d = ((sa - sb).abs() / (1.0 + sa.abs()))[live].max().item()
        worst = max(worst, d)
        assert d < 3e-3, (it, d)
        assert torch.allclose(ra, rb, rtol=5e-3, atol=1e-3)
    assert cutp.supervisor.sim.frame_pipeline_state()[2] >= 8
    # actuator space on demand: the integrated command and err of the last frame
    ca, cb = ref.supervisor.get_command().clone(), cut.supervisor.get_command().clone()
    assert (ca - cb).abs().max().item() < 2e-3 * ca.abs().max().item()
    ea, eb = ref.supervisor.get_err().clone(), cut.supervisor.get_err().clone()
    assert (ea - eb).abs().max().item() < 5e-3 * ea.abs().max().item() + 1e-5
    # ... and the loop goes on from there
    a = torch.zeros(4, ref.action_dim, device="cuda:0")
    (sa, _, _, _), (sb, _, _, _) = ref.step(a), cut.step(a)
    assert ((sa - sb).abs() / (1.0 + sa.abs()))[torch.isfinite(sa) & (sa.abs() < 1e3)].max().item() < 3e-3
    print("residual shortcut: worst relative state difference over 10 steps %.2e" % worst)
