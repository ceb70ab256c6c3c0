"""The composition bench.py times, with an assert: VecAoEnv("production_sh_40x40_8m_3layers", 13 windowed
modal agents + the windowed tip-tilt agent) stepping through ONE library call per step
(aomarl_env_step: rl_step -> per-agent rewards -> linear_step -> state assembly, Btt coordinates carried by
linearity, fused tail) against the same VecAoEnv host logic over the CPU oracle (tests/oracle_vecsim.py:
every native stage restated in C, explicit v2m / m2v projections, call-by-call order).  State layout =
helper_states.py:202-283 (windowed), rewards = helper_rewards.py:14-22."""
import copy

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from ao_marl_amd import geometry as G, modal, system  # noqa: E402
from tests.oracle_vecsim import OracleVecSim  # noqa: E402

NAME = "production_sh_40x40_8m_3layers"
RL = dict(n_zernike_start_end=[0, 1274], n_reverse_filtered_from_cmat=5, window_n_zernike=20,
          include_tip_tilt_windowed=True)
NENV, NSTEP = 4, 6


def _pushed_sim_class():
    from ao_marl_amd.sim import HipSim

    class PushedSim(HipSim):
        """HipSim whose full reset ends on the ORACLE's screens: the two sides generate theirs with
        differently ordered fp32 sums over a 1296-step recursion; the step is what is compared here.  The GPU's own
        reset is compared with the oracle's screens (all layers, every pixel) right before they are pushed."""
        source = None
        reset_diff, reset_limit = 0.0, 6e-3

        def reset(self, seeds, env_begin=0, env_count=None):
            HipSim.reset(self, seeds, env_begin, env_count)
            src = type(self).source
            if src is not None and env_begin == 0 and env_count in (None, self.nenv):
                for l in range(self.s.nscreens):
                    # the GPU's OWN full reset first (2 x 648 dependent extrusion rounds per layer, reset kernels on
                    # the transposed screen, two streams) against the oracle's, all environments: <= 1e-3 um
                    want = np.stack([scr[l] for scr, _ in src.snap])
                    got = self.screen(l).cpu().numpy()
                    assert got.shape == want.shape
                    d, rms = float(np.abs(got - want).max()), float((got - want).std())
                    type(self).reset_diff = max(type(self).reset_diff, d)
                    # fp32 round-off of a 1296-round recursion whose two sides sum 1957 products per pixel in
                    # different orders (the oracle sequentially): measured 1.6e-3 .. 2.9e-3 um max, 3e-4 .. 1.2e-3 rms
                    # on screens of 1 .. 4 um rms -- the same with round 3's GEMM (tools/reset_parity_probe.py,
                    # profiles/r04_reset_parity.txt).  Split-fp16 operands: up to 8e-3.
                    lim = type(self).reset_limit
                    assert d < lim and rms < 0.4 * lim and float(want.std()) > 0.05, ("reset screens, layer %d" % l, d, rms)
                    cnt = self.t["ext_count"][:, l].cpu().numpy()
                    assert (cnt == np.array([c[l] for _, c in src.snap])).all()
                    self.set_screen(l, np.stack([scr[l] for scr, _ in src.snap]))
                    self.t["ext_count"][:, l] = torch.tensor([cnt[l] for _, cnt in src.snap], dtype=torch.int32)
                self.target_psf()           # the pending PSF of the pushed screens
    return PushedSim


_GROWN = {}          # (configuration, seed) -> the oracle's screens behind its reset: grown once per session


class SnapOracleVecSim(OracleVecSim):
    """Keeps the screens as the oracle's reset left them (VecAoEnv.reset goes straight on to the first
    linear_step, which moves them).  The four parametrisations reset the same seeds of the same system: the oracle
    grows each environment's screens once (3 x 1296 extrusions, half a minute for four environments) and restores
    them afterwards."""

    def reset(self, seeds):
        seeds = np.broadcast_to(np.asarray(seeds), (self.nenv,))
        for o, sd in zip(self.sims, seeds):
            key = (getattr(self.s, "name", ""), tuple(self.s.screen_dim), int(sd))
            if key in _GROWN:
                o.reset(int(sd), grown=_GROWN[key])
            else:
                o.reset(int(sd))
                _GROWN[key] = ([scr.copy() for scr in o.screens], list(o.ext_count))
        self.snap = [([scr.copy() for scr in o.screens], list(o.ext_count)) for o in self.sims]


# the reference's published layout (README.md:116-119, src/error_budget/helper_experiments.py:19-36): 42 agents of 30
# modes + the tip-tilt agent, here with the `_w20` experiments' window
RL43 = dict(n_zernike_start_end=[0, 1260], n_reverse_filtered_from_cmat=5, window_n_zernike=20,
            include_tip_tilt_windowed=True)


@pytest.mark.parametrize("precision,pipeline,layout,bench", [("f32", False, 14, False), ("f32", True, 14, False),
                                                             ("split_f16", True, 14, False), ("f32", True, 43, False),
                                                             ("f32", True, 14, True), ("f32", False, 14, True)])
def test_env_step_40x40_windowed_agents_match_the_oracle_env(monkeypatch, precision, pipeline, layout, bench):
    """pipeline: the frame pipeline bench.py runs with (frame t+1 in flight while frame t is reduced).
    layout: the bench's 14 agents, or the reference's published 43.
    bench: the EXACT composition bench.py times and `run_episode` steps with -- `VecAoEnv.policy_step`
    (aomarl_policy_env_step: the actors inside the call, sampled actions) with `residual_shortcut=True` (v2m . cmat in
    one product) -- against the oracle environment in the reference's order (do_control, then v2m . err), which gets
    the actions the call returned."""
    from ao_marl_amd import libaomarl as la
    from ao_marl_amd.env import VecAoEnv
    keep = la.get_precision()
    la.set_precision(precision)
    try:
        _run(monkeypatch, precision, pipeline, layout, bench)
    finally:
        la.set_precision(keep)


def _run(monkeypatch, precision, pipeline, layout=14, bench=False):
    from ao_marl_amd import libaomarl as la
    from ao_marl_amd.env import VecAoEnv
    PushedSim = _pushed_sim_class()
    PushedSim.reset_limit = 6e-3 if precision == "f32" else 2e-2
    RL = globals()["RL"] if layout == 14 else RL43
    nmod = layout - 1
    env = VecAoEnv(NAME, NENV, RL, initial_seed=1234, seed_stride=16, n_agents_modal=nmod, device="cuda:0",
                   sim_factory=PushedSim, frame_pipeline=pipeline)
    lay = env.layout
    assert lay.n_agents == layout and lay.state_shapes()[0] == (552 if layout == 14 else 280) and lay.state_shapes()[-1] == 168
    assert env._native_glue and env._default_state_layout
    pol = None
    if bench:
        from ao_marl_amd.agents import BatchedGaussianPolicy
        env.residual_shortcut = True
        # a policy that acts (random last layer, sampled actions: the whole action range is exercised)
        pol = BatchedGaussianPolicy(lay, last_layer_zero=False, seed=7, device="cuda:0")
    cal = env.supervisor.cal

    # the oracle-backed twin takes the GPU side's calibration (an interaction matrix of 1286 actuators
    # through the CPU oracle would take minutes; the calibration itself is compared in test_gpu_parity)
    def calibrate(s, sysm, backend, nfilt=0, verbose=False, **kw):
        # (`backend` is a factory of the calibration simulator: never built here)
        for k, d in enumerate(s.dms):
            if d.type == "pzt":
                G.pzt_select(d, sysm.geom, cal.kept[k])
        system.refresh_dms(s)
        s.cmat = np.ascontiguousarray(cal.cmat)
        return copy.copy(cal)
    monkeypatch.setattr(modal, "calibrate", calibrate)
    oenv = VecAoEnv(NAME, NENV, RL, initial_seed=1234, seed_stride=16, n_agents_modal=nmod, device="cpu",
                    sim_factory=SnapOracleVecSim)
    monkeypatch.undo()
    assert oenv.supervisor.s.nactu == env.supervisor.s.nactu == 1286
    assert [tuple(v) for v in oenv.layout.agents.values()] == [tuple(v) for v in lay.agents.values()]

    so = oenv.reset().numpy()                       # full 40x40 reset in the oracle (3 x 1296 extrusions per env)
    PushedSim.source = oenv.supervisor.sim
    sg = env.reset()
    PushedSim.source = None
    assert sg.shape == (NENV, env.state_dim) == so.shape
    rng = np.random.default_rng(5)
    worst = dict(state=0.0, reward=0.0, slopes=0.0, com=0.0)
    # Columns that carry signal.  The 5 Btt modes filtered out of the command matrix (and never commanded)
    # have recorded standard deviations of ~1e-10 in the reference's statistics: their standardised values
    # are round-off divided by 1e-10 on both sides (the reference feeds its agents the same noise) -- not
    # comparable, and excluded here.
    sd_dm, sd_res = env.norm["dm"][1].cpu().numpy(), env.norm["dm_residual"][1].cpu().numpy()
    # (VecAoEnv masks them by default: dead_columns="mask" turns their recorded std into inf, i.e. a state of 0)
    assert sorted(env.dead_columns) == ["dm", "dm_residual"] and all(len(v) == 5 for v in env.dead_columns.values())
    live = np.concatenate([np.isfinite(sd_dm)] * 3 + [np.isfinite(sd_res)])
    assert live.shape == (env.state_dim,) and (~live).sum() == 4 * 5
    assert float(np.abs(sg.cpu().numpy()[:, ~live]).max()) == 0.0 and float(np.abs(so[:, ~live]).max()) == 0.0
    la.arith_launches(reset=True)
    used_native = 0
    for it in range(NSTEP):
        scale = np.maximum(1.0, np.abs(so[:, live]).max())
        d = np.abs(sg.cpu().numpy() - so)[:, live].max() / scale
        worst["state"] = max(worst["state"], d)
        # standardised Btt coordinates: O(1) columns; fp32 round-off of two differently ordered chains
        assert d < 2e-3, ("state", it, d)
        ok = env._native_step_ok(False)
        if bench:
            calls = la.arith_launches().get("actor:f32_mfma", 0)
            a_t, sg, rg, done, _ = env.policy_step(pol, sg)
            a = a_t.cpu().numpy()
            assert la.arith_launches().get("actor:f32_mfma", 0) == calls + 1          # the actors ran inside the call
            assert np.isfinite(a).all() and np.abs(a).max() <= 1.0 and np.abs(a).max() > 0.5
            # the shortcut ran inside the one-call step: the integrator has not run in actuator space (the getters
            # below run do_control on demand)
            assert env._native_shortcut and env.supervisor._control_pending
        else:
            a = rng.uniform(-1, 1, size=(NENV, env.action_dim)).astype(np.float32)
            sg, rg, done, _ = env.step(torch.from_numpy(a).cuda())
        used_native += int(ok)
        so_t, ro, _, _ = oenv.step(torch.from_numpy(a))
        so, ro = so_t.numpy(), ro.numpy()
        rg = rg.cpu().numpy()
        assert rg.shape == ro.shape == (NENV, layout) and done is False
        dr = np.abs(rg - ro).max() / np.maximum(np.abs(ro).max(), 1e-12)
        worst["reward"] = max(worst["reward"], dr)
        assert dr < 2e-3, ("reward", it, dr)
        sl = env.supervisor.get_slopes().cpu().numpy()
        slo = oenv.supervisor.get_slopes().numpy()
        worst["slopes"] = max(worst["slopes"], np.abs(sl - slo).max())
        assert np.abs(sl - slo).max() < 1e-4, ("slopes", it)            # arcsec, the north-star tolerance
        cm = env.supervisor.get_command().cpu().numpy()
        cmo = oenv.supervisor.get_command().numpy()
        dc = np.abs(cm - cmo).max() / np.abs(cmo).max()
        worst["com"] = max(worst["com"], dc)
        assert dc < 5e-4, ("com", it, dc)
    assert used_native == NSTEP                      # every step went through aomarl_env_step
    flying, _, piped, beside = env.supervisor.sim.frame_pipeline_state()
    assert (flying, piped, beside) == ((True, NSTEP - 1, NSTEP) if pipeline else (False, 0, 0))
    launched = {k: v for k, v in la.arith_launches().items() if v}
    if precision == "f32":
        assert not any("split" in k for k in launched), launched
        assert env.supervisor.sim.frame_kernel_name() == "k_frame_wave<3, 1, true, false, false, false>"
    else:
        assert env.supervisor.sim.frame_kernel_name() == "k_frame_wave<3, 1, true, false, false, true>"
        assert "gemm:f32_mfma" not in launched, launched
    st = env.supervisor.get_strehl().cpu().numpy()
    sto = oenv.supervisor.get_strehl().numpy()
    assert np.abs(st[:, :2] - sto[:, :2]).max() < 2e-4
    print("worst deviations over %d steps (%s): %s; full reset against the oracle's screens: %.2e um" %
          (NSTEP, precision, worst, PushedSim.reset_diff))


def test_bench_batch_through_env_step_with_the_frame_pipeline():
    """What bench.py times, at ITS batch, with asserts: 256 environments x 14 windowed agents through
    `policy_step` (aomarl_policy_env_step = aomarl_actor_forward + aomarl_env_step in one call) with the residual
    shortcut and a frame in flight, 32 steps.  Size-independent properties: equal
    seeds with equal actions stay equal bit for bit, different seeds decorrelate, everything is finite, the loop
    closes, and the first environments agree with the same seeds stepped as a batch of 8 (another GEMM tiling:
    fp32 round-off of differently ordered sums, not bits)."""
    from ao_marl_amd.agents import BatchedGaussianPolicy
    from ao_marl_amd.env import VecAoEnv
    n, small_n, steps = 256, 8, 32

    def make(nenv):
        env = VecAoEnv(NAME, nenv, RL, initial_seed=1234, seed_stride=16, n_agents_modal=13, device="cuda:0",
                       frame_pipeline=True)
        env.residual_shortcut = True
        seeds = env.supervisor.env_seeds().copy()
        seeds[1] = seeds[0]                                  # twin environments
        env.supervisor.env_seeds = lambda: seeds
        return env
    big, small = make(n), make(small_n)
    pol = BatchedGaussianPolicy(big.layout, last_layer_zero=False, seed=7, device="cuda:0")
    sb, ss = big.reset(), small.reset()
    assert torch.equal(sb[0], sb[1]) and (sb[:small_n] - ss).abs().max().item() < 5e-3 * max(1.0, ss.abs().max().item())
    g = torch.Generator(device="cuda:0").manual_seed(9)
    worst = 0.0
    for it in range(steps):
        eps = torch.randn(n, big.layout.action_dim, device="cuda:0", generator=g)
        eps[1] = eps[0]
        assert big._native_step_ok(False)
        a_want, _ = pol.select_action(sb, eps=eps)           # k_actor_fused on the batch's states
        a, sb, rb, _, _ = big.policy_step(pol, sb, eps=eps)  # ... and inside the one-call step: the same launch, the same bits
        assert torch.equal(a, a_want) and torch.equal(a[0], a[1])
        ss, rs, _, _ = small.step(a[:small_n].contiguous())
        assert torch.isfinite(sb).all() and torch.isfinite(rb).all() and rb.shape == (n, 14)
        assert torch.equal(sb[0], sb[1]) and torch.equal(rb[0], rb[1])          # same seed, same actions: same bits
        livec = torch.cat([torch.isfinite(big.norm["dm"][1])] * 3 + [torch.isfinite(big.norm["dm_residual"][1])])
        d = (sb[:small_n] - ss)[:, livec].abs().max().item() / max(1.0, ss[:, livec].abs().max().item())
        worst = max(worst, d)
        assert d < 5e-3, (it, d)                             # batch of 256 against batch of 8, closed loop, 32 steps
    flying, _, piped, beside = big.supervisor.sim.frame_pipeline_state()
    assert flying and piped >= steps - 1 and beside >= steps - 2, (piped, beside)
    assert big._native_shortcut and small._native_shortcut
    sl = big.supervisor.get_slopes()
    assert torch.isfinite(sl).all() and torch.equal(sl[0], sl[1])
    c = np.corrcoef(sl[2].cpu().numpy(), sl[3].cpu().numpy())[0, 1]
    assert abs(c) < 0.3, c                                   # different seeds decorrelate
    st = big.supervisor.get_strehl().cpu().numpy()
    assert np.isfinite(st).all() and st[:, 0].min() > 0.0
    print("256 x 14 agents, %d pipelined steps: batch against batch-of-8 worst %.2e (standardised states)" % (steps, worst))


@pytest.mark.parametrize("pipe", [True, False])
def test_configs1_batch_through_the_small_system_path(pipe):
    """BASELINE configs[1] at ITS batch -- production_sh_10x10_2m, 64 environments, 2 agents (80 modes + tip-tilt) --
    through what bench.py times there: k_actor_fused at 64 x 2, aomarl_env_step on the small-system kernels
    (k_move_small, k_small_head / k_small_tail), with and without a frame in flight.  Properties that do not depend
    on the size: twin environments (same seed, same actions) stay equal bit for bit, the batch agrees with its own
    first 8 environments stepped as a batch of 8 and with the GENERAL chain ("small_chain" / "small_move" = 0: the
    kernels the 40x40 system runs) to fp32 round-off, the pipelined order is the plain order bit for bit, different
    seeds decorrelate, the loop closes."""
    from ao_marl_amd.agents import BatchedGaussianPolicy
    from ao_marl_amd.env import VecAoEnv
    name, rl = "production_sh_10x10_2m", dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5)
    n, small_n, steps = 64, 8, 40

    def make(nenv, pipeline, general=False):
        env = VecAoEnv(name, nenv, rl, initial_seed=1234, seed_stride=16, n_agents_modal=1, device="cuda:0",
                       frame_pipeline=pipeline)
        seeds = env.supervisor.env_seeds().copy()
        seeds[1] = seeds[0]                                  # twin environments
        env.supervisor.env_seeds = lambda: seeds
        if general:
            env.supervisor.sim.set_option("small_chain", 0)
            env.supervisor.sim.set_option("small_move", 0)
        return env
    big, small, gen = make(n, pipe), make(small_n, pipe), make(n, False, general=True)
    other = make(n, not pipe)
    one, two = make(n, pipe), make(n, pipe)                  # choose_action + env_step as ONE call (aomarl_policy_env_step)
    lay = big.layout
    assert lay.n_agents == 2 and lay.action_dim == 82 and lay.state_shapes() == [320, 8]
    pol = BatchedGaussianPolicy(lay, last_layer_zero=False, seed=7, device="cuda:0")
    sb, ss, sg, so = big.reset(), small.reset(), gen.reset(), other.reset()
    s1, s2 = one.reset(), two.reset()
    assert torch.equal(sb[0], sb[1]) and torch.equal(sb, so) and torch.equal(sb, s1)
    pol1 = BatchedGaussianPolicy(lay, last_layer_zero=False, seed=7, device="cuda:0")     # the same weights, its own draw counter
    pol2 = BatchedGaussianPolicy(lay, last_layer_zero=False, seed=7, device="cuda:0")
    worst1 = 0.0
    g = torch.Generator(device="cuda:0").manual_seed(9)
    live = torch.cat([torch.isfinite(big.norm["dm"][1])] * 3 + [torch.isfinite(big.norm["dm_residual"][1])])
    worst = dict(sub=0.0, general=0.0)
    for it in range(steps):
        eps = torch.randn(n, lay.action_dim, device="cuda:0", generator=g) * 0.3
        eps[1] = eps[0]
        a, _ = pol.select_action(sb, eps=eps)
        assert torch.equal(a[0], a[1]) and big._native_step_ok(False)
        sb, rb, _, _ = big.step(a)
        ss, rs, _, _ = small.step(a[:small_n].contiguous())
        sg, rg, _, _ = gen.step(a)
        so, ro, _, _ = other.step(a)
        assert torch.isfinite(sb).all() and torch.isfinite(rb).all() and rb.shape == (n, 2)
        assert torch.equal(sb[0], sb[1]) and torch.equal(rb[0], rb[1])          # twins: same bits
        assert torch.equal(sb, so) and torch.equal(rb, ro), it                   # frame in flight or not: same bits
        scale = max(1.0, sb[:, live].abs().max().item())
        d1 = (sb[:small_n] - ss)[:, live].abs().max().item() / scale
        d2 = (sb - sg)[:, live].abs().max().item() / scale
        worst["sub"], worst["general"] = max(worst["sub"], d1), max(worst["general"], d2)
        assert d1 < 1e-6 and d2 < 1e-4, (it, d1, d2)    # (one workgroup per environment: the sub-batch is the batch bit for bit; general chain: 1e-5 measured)
        assert torch.allclose(rb, rg, rtol=5e-3, atol=1e-4)
        # choose_action + env_step as one library call: the same launches, the same numbers
        a_same, _ = pol.select_action(s1, eps=eps)
        a1, s1n, r1, _, _ = one.policy_step(pol1, s1, eps=eps)
        assert torch.equal(a1, a_same), it
        assert torch.equal(a1[0], a1[1]) and torch.equal(s1n[0], s1n[1]) and torch.equal(r1[0], r1[1]), it
        s1 = s1n
        worst1 = max(worst1, (s1 - sb)[:, live].abs().max().item() / max(1.0, sb[:, live].abs().max().item()))
        # ... and on the policy's own Philox draws (keyed by seed, number of calls, environment, action index: what
        # select_action draws with the same counter)
        pol2._draws = 100 + it
        a_ref, _ = pol2.select_action(s2)
        pol2._draws = 100 + it
        a2, s2, _, _, _ = two.policy_step(pol2, s2)
        assert torch.equal(a2, a_ref), it
    if pipe:
        flying, _, piped, _ = big.supervisor.sim.frame_pipeline_state()
        assert flying and piped >= steps - 1
    sl = big.supervisor.get_slopes() if not pipe else other.supervisor.get_slopes()
    assert torch.isfinite(sl).all() and torch.equal(sl[0], sl[1])
    assert abs(np.corrcoef(sl[2].cpu().numpy(), sl[3].cpu().numpy())[0, 1]) < 0.4     # different seeds decorrelate
    st = (big if not pipe else other).supervisor.get_strehl().cpu().numpy()
    assert np.isfinite(st).all() and st[:, 0].min() > 0.0
    assert worst1 == 0.0, worst1                          # the same launches: the same bits
    assert torch.isfinite(s1).all() and one.supervisor.sim.frame_pipeline_state()[0] == bool(pipe)
    print("64 x 2 agents, %d steps (%s): batch against batch-of-8 worst %.2e, small-system kernels against the general chain "
          "%.2e, one-call policy step against select_action + step %.2e (standardised states)" %
          (steps, "pipelined" if pipe else "plain order", worst["sub"], worst["general"], worst1))
