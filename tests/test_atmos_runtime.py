"""Run-time changes of the atmosphere: supervisor.atmos.set_wind / set_r0 (atmosCompass.py:79-135), what the
trainer's non-stationary experiments call between episodes (train_rpc.py:429-450; flags GlobalConfig.py:126-130),
over aomarl_set_wind / aomarl_set_stencil / aomarl_set_r0.

CPU: the arithmetic of the change against systems BUILT from the parameter files the change leads to (whose geometry
is pinned bit-exact to the reference's init code, tests/test_geometry.py); the mirroring rule as a property of the
screens it produces.  GPU: HIP against the oracle across a change in mid-episode, the refusal to step across a frame
whose atmosphere was already moved, and the trainer's flow (change, then reset) against an environment built with the
new wind from the start, bit for bit."""
import types

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from ao_marl_amd import geometry as G, params, system  # noqa: E402
from ao_marl_amd.env import VecAtmos  # noqa: E402
from tests import helpers  # noqa: E402
from tests.oracle_vecsim import OracleVecSim  # noqa: E402

BASE = "production_sh_40x40_8m_3layers"


def _bare_supervisor(name):
    """What VecAtmos reads of a supervisor, without a calibration: parameters, derived geometry, arrays, and the
    oracle-backed simulator (lazy: no screens are grown)."""
    ps = params.builtin(name)
    sysm = G.build_system(ps)
    s = system.from_system(sysm)
    sup = types.SimpleNamespace(config=ps, sysm=sysm, s=s, sim=OracleVecSim(s, 1), _rp_left=0,
                                _atmos_changed_behind=None)
    sup.atmos = VecAtmos(sup)
    return sup


@pytest.mark.parametrize("start,target,calls", [
    # change_atmospheric_3_layers_1 (train_rpc.py:430-433): directions 0 0 0 -> 0 15 30 (sin(0 + pi) is -8.7e-8 in
    # the float32 of PATMOS.py: deltax is negative before and after, nothing is mirrored)
    (BASE + "_same_dir", BASE + "_dir_0_15_30", [dict(screen_index=1, winddir=15), dict(screen_index=2, winddir=30)]),
    # a direction that turns layer 1 around: both components change sign, both stencils are mirrored
    (BASE, (BASE, dict(winddir=[0, 200, 90])), [dict(screen_index=1, winddir=200)]),
    # change_atmospheric_3_layers_2 / _4 (train_rpc.py:434-447): speeds
    (BASE, BASE + "_v_20_15_25", [dict(screen_index=0, windspeed=20), dict(screen_index=1, windspeed=15),
                                  dict(screen_index=2, windspeed=25)]),
    (BASE, BASE + "_v_10_5_15", [dict(screen_index=0, windspeed=10), dict(screen_index=1, windspeed=5),
                                 dict(screen_index=2, windspeed=15)]),
    (BASE + "_dir_0_15_30", BASE + "_dir_0_15_30_v_10_5_15", [dict(screen_index=k, windspeed=v) for k, v in enumerate((10, 5, 15))]),
])
def test_set_wind_leads_to_the_system_of_the_parameter_file(start, target, calls):
    sup = _bare_supervisor(start)
    if isinstance(target, tuple):
        tps = params.builtin(target[0])
        for k, v in target[1].items():
            setattr(tps.p_atmos, k, np.asarray(v, dtype=np.float32))
    else:
        tps = params.builtin(target)
    want = system.from_system(G.build_system(tps))
    for kw in calls:
        sup.atmos.set_wind(**kw)
    o = sup.sim.sims[0]
    assert np.array_equal(o.deltax, want.deltax) and np.array_equal(o.deltay, want.deltay)
    for l in range(want.nscreens):
        assert np.array_equal(o.istx[l], want.istx[l]) and np.array_equal(o.isty[l], want.isty[l]), l
    assert np.array_equal(np.asarray(sup.config.p_atmos.winddir), np.asarray(tps.p_atmos.winddir))
    assert sup._atmos_changed_behind is None
    flipped = [[l for l in range(want.nscreens) if not np.array_equal(ist0[l], ist1[l])]
               for ist0, ist1 in ((sup.s.istx, want.istx), (sup.s.isty, want.isty))]
    assert flipped == ([[1], [1]] if isinstance(target, tuple) else [[], []])      # the rule fired exactly where a sign changed


def test_set_r0_is_the_amplitude_of_a_system_built_with_that_r0():
    sup = _bare_supervisor(BASE)
    ps = params.builtin(BASE)
    ps.p_atmos.r0 = 0.08                                     # change_atmospheric_3_layers_3 (train_rpc.py:442-444)
    want = system.from_system(G.build_system(ps))
    before = sup.sim.sims[0].amplitude.copy()
    sup.atmos.set_r0(0.08)
    got = sup.sim.sims[0].amplitude
    assert np.array_equal(got, want.amplitude)
    assert np.allclose(got / before, 0.5**(-5. / 6.), rtol=1e-6)       # amplitude ~ r0^(-5/6)
    assert sup.atmos.r0 == 0.08 and sup.config.p_atmos.r0 == 0.08
    with pytest.raises(NotImplementedError):
        sup.atmos.set_r0(0.1, reset_seed=3)
    with pytest.raises(IndexError):
        sup.atmos.set_wind(5, windspeed=1.0)


def test_mirrored_stencil_keeps_the_screen_continuous_when_the_wind_turns():
    """The rule of atmosCompass.py:124-135 as a property: a screen grown with the wind along +x, then extruded along
    -x with the MIRRORED stencil, stays a von Karman screen -- the new column next to the old edge differs from it by
    about one pixel of structure function; with the stencil left as it was (the deltas alone) the new columns are
    conditioned on the wrong points and jump."""
    from oracle import aoref
    _, s = helpers.uncalibrated("production_sh_10x10_2m")
    n = s.screen_dim[0]
    ps = params.builtin("production_sh_10x10_2m")
    assert s.deltax[0] < 0 and ps.p_atmos.winddir[0] == 45.

    def jump(mirror):
        o = aoref.OracleSim(s, seed=77)                      # reset: 2 n extrusions along -x (the file's wind)
        o.set_wind(0, -s.deltax[0], s.deltay[0], mirror_stencils=mirror)
        assert (o.deltax[0] > 0) and np.array_equal(o.istx[0], s.istx[0]) == (not mirror)
        d = []
        for _ in range(12):
            edge = o.screens[0][:, -1].copy()                # +x: the screen shifts left, the new column is the last
            o._extrude(0, 1)
            d.append(float(np.sqrt(np.mean((o.screens[0][:, -1] - edge)**2))))
        return float(np.mean(d)), float(o.screens[0].std())
    good, rms = jump(True)
    bad, _ = jump(False)
    neighbours = float(np.sqrt(np.mean(np.diff(aoref.OracleSim(s, seed=77).screens[0], axis=1)**2)))
    assert good < 1.5 * neighbours and good < 0.2 * rms, (good, neighbours, rms)
    assert bad > 3.0 * good, (bad, good)


def test_facade_screen_primitives_follow_the_reference_call_sequence():
    """AtmosCompass.set_wind itself (atmosCompass.py:103-135) drives Tscreen.set_deltax / set_deltay, reads
    d_istencilx and writes set_istencilx: the facade's primitives, before any engine exists, keep the values the
    engine is later built from."""
    from ao_marl_amd import sutra_facade as F
    keep = dict(F._HUB)
    try:
        F._HUB.update(engines=None, sealed=False)
        atm = F.Atmos(None, 1, 0.16, np.float32([30.0]), [168], [345], [0.0], [20.0], [45.0], np.float32([-1.5]), np.float32([-1.5]), 0)
        _, s = helpers.uncalibrated("production_sh_10x10_2m")
        atm.init_screen(0, s.A[0], s.B[0], s.istx[0], s.isty[0], 1234)
        sc = atm.d_screens[0]
        a0 = float(sc.amplitude)
        # the reference's own sequence for a sign change along x
        sc.set_deltax(1.25)
        sc.set_deltay(-0.5)
        st = np.array(sc.d_istencilx)
        st = (168 * 168 - 1) - st
        sc.set_istencilx(st)
        assert float(sc.deltax) == 1.25 and float(sc.deltay) == -0.5
        assert np.array_equal(sc.istx, (168 * 168 - 1) - s.istx[0].astype(np.int64))
        atm.set_r0(0.08)
        assert atm.r0 == 0.08 and np.isclose(float(sc.amplitude) / a0, 0.5**(-5. / 6.), rtol=1e-6)
    finally:
        F._HUB.clear()
        F._HUB.update(keep)


# ------------------------------------------------------------------------------------------------ GPU
RL10 = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5)


def _env10(ps=None, **kw):
    from ao_marl_amd.env import VecAoEnv
    return VecAoEnv(ps if ps is not None else "production_sh_10x10_2m", kw.pop("nenv", 4), RL10, initial_seed=1234,
                    seed_stride=16, n_agents_modal=1, **kw)


@pytest.mark.gpu
def test_hip_follows_the_oracle_across_a_change_of_wind_and_r0_in_mid_episode(monkeypatch):
    """10x10, 4 environments, call-by-call order without prefetch (the atmosphere may change between any two frames):
    3 steps, then speed, direction (both components change sign: both stencils are mirrored) and r0 change, 6 more
    steps.  HIP == oracle at the usual tolerances, screens included."""
    import copy
    from ao_marl_amd import modal
    env = _env10(device="cuda:0", prefetch_atmos=False)
    cal = env.supervisor.cal

    def calibrate(s, sysm, backend, nfilt=0, verbose=False, **kw):
        for k, d in enumerate(s.dms):
            if d.type == "pzt":
                G.pzt_select(d, sysm.geom, cal.kept[k])
        system.refresh_dms(s)
        s.cmat = np.ascontiguousarray(cal.cmat)
        return copy.copy(cal)
    monkeypatch.setattr(modal, "calibrate", calibrate)
    oenv = _env10(device="cpu", sim_factory=OracleVecSim)
    monkeypatch.undo()
    sg, so = env.reset(), oenv.reset().numpy()
    rng = np.random.default_rng(3)
    live = np.concatenate([np.isfinite(env.norm["dm"][1].cpu().numpy())] * 3 + [np.isfinite(env.norm["dm_residual"][1].cpu().numpy())])
    worst = dict(state=0.0, slopes=0.0, screen=0.0)
    for it in range(9):
        if it == 3:
            for e in (env, oenv):
                e.supervisor.atmos.set_wind(0, windspeed=33.0, winddir=200.0)
                e.supervisor.atmos.set_r0(0.10)
            dx, dy, amp = env.supervisor.sim.layer_values(0)
            o = oenv.supervisor.sim.sims[0]
            assert (dx, dy, amp) == (float(o.deltax[0]), float(o.deltay[0]), float(o.amplitude[0])) and dx > 0 and dy > 0
            assert env.supervisor.s.deltax[0] < 0 and env.supervisor.s.deltay[0] < 0        # both signs changed
        a = rng.uniform(-1, 1, size=(4, env.action_dim)).astype(np.float32)
        sg, rg, _, _ = env.step(torch.from_numpy(a).cuda())
        so_t, ro, _, _ = oenv.step(torch.from_numpy(a))
        so = so_t.numpy()
        d = np.abs(sg.cpu().numpy() - so)[:, live].max() / max(1.0, np.abs(so[:, live]).max())
        sl = np.abs(env.supervisor.get_slopes().cpu().numpy() - oenv.supervisor.get_slopes().numpy()).max()
        scr = env.supervisor.sim.screen(0).cpu().numpy()
        want = np.stack([o.screens[0] for o in oenv.supervisor.sim.sims])
        ds = np.abs(scr - want).max()
        worst = dict(state=max(worst["state"], d), slopes=max(worst["slopes"], sl), screen=max(worst["screen"], ds))
        assert d < 2e-3 and sl < 1e-4 and ds < 3e-4, (it, d, sl, ds)
        assert np.allclose(rg.cpu().numpy(), ro.numpy(), rtol=5e-3, atol=1e-4)
    # the change did something: a twin that keeps the old atmosphere has other screens by now
    twin = _env10(device="cuda:0", prefetch_atmos=False)
    twin.reset()
    for it in range(9):
        twin.step(torch.zeros(4, env.action_dim, device="cuda:0"))
    assert (twin.supervisor.sim.screen(0) - env.supervisor.sim.screen(0)).abs().max().item() > 0.05
    print("wind + r0 changed behind step 3: worst over 9 steps %s" % worst)


@pytest.mark.gpu
def test_change_then_reset_is_the_environment_built_with_the_new_wind_and_stepping_across_is_refused():
    """The trainer's flow (manage_changing_conditions, then the episode's reset) with everything that runs ahead
    switched on -- atmosphere prefetch, frame pipeline, prefetched reset: the change drops the prefetched reset, the
    next episode is, bit for bit, that of an environment built from a parameter set with the new wind and r0; and a
    step across the frame that was already moved with the old wind raises."""
    ps = params.builtin("production_sh_10x10_2m")
    ps.p_atmos.winddir = np.asarray([200.0], dtype=np.float32)
    ps.p_atmos.windspeed = np.asarray([33.0], dtype=np.float32)
    ps.p_atmos.r0 = 0.10
    want = _env10(ps, device="cuda:0", frame_pipeline=True)
    env = _env10(device="cuda:0", frame_pipeline=True, reset_prefetch="same")
    zero = torch.zeros(4, env.action_dim, device="cuda:0")
    env.reset()
    for _ in range(5):
        env.step(zero)
    assert env.supervisor.sim.prefetch_reset_pending() and env.supervisor.sim.atmos_change_blocked()
    env.supervisor.atmos.set_wind(0, windspeed=33.0, winddir=200.0)
    env.supervisor.atmos.set_r0(0.10)
    assert not env.supervisor.sim.prefetch_reset_pending()                  # grown along the old sign: dropped
    with pytest.raises(RuntimeError, match="set_wind / set_r0"):
        env.step(zero)
    sa, sb = env.reset(), want.reset()
    assert torch.equal(sa, sb)
    assert env.supervisor.sim.layer_values(0) == want.supervisor.sim.layer_values(0)
    g = torch.Generator(device="cuda:0").manual_seed(4)
    for it in range(12):
        a = torch.rand(4, env.action_dim, device="cuda:0", generator=g) * 2 - 1
        sa, ra, _, _ = env.step(a)
        sb, rb, _, _ = want.step(a)
        assert torch.equal(sa, sb) and torch.equal(ra, rb), it
    # the prefetched reset starts over with the new atmosphere at the next reset: adopted, same bits again
    n0 = int(getattr(env.supervisor.sim, "prefetched_resets", 0))
    sa, sb = env.reset(), want.reset()
    assert torch.equal(sa, sb) and env.supervisor.sim.prefetch_reset_pending()
    assert int(getattr(env.supervisor.sim, "prefetched_resets", 0)) == n0 + 1
    assert torch.equal(env.supervisor.sim.screen(0), want.supervisor.sim.screen(0))


@pytest.mark.gpu
def test_the_facade_primitives_and_the_composite_call_are_the_same_change():
    """aomarl_set_wind(mirror_stencils=1) against the reference's own sequence through the primitives
    (set_deltax / set_deltay, then set_istencilx / set_istencily with n * n - 1 - stencil: aomarl_set_wind without
    mirroring + aomarl_set_stencil): the same screens, bit for bit, and both the oracle's."""
    from ao_marl_amd.sim import HipSim
    from oracle import aoref
    _, s = helpers.uncalibrated("production_sh_10x10_2m")
    n = s.screen_dim[0]
    a, b = HipSim(s, nenv=2), HipSim(s, nenv=2)
    o = aoref.OracleSim(s, seed=1234)
    for sim in (a, b):
        sim.reset([1234, 1250])
    dx, dy = np.float32(1.75), np.float32(0.5)                # the file's wind is (-1.41, -1.41): both signs change
    a.set_wind(0, dx, dy)
    b.set_wind(0, dx, dy, mirror_stencils=False)
    b.set_stencil(0, 0, (n * n - 1) - s.istx[0].astype(np.int64))
    b.set_stencil(0, 1, (n * n - 1) - s.isty[0].astype(np.int64))
    o.set_wind(0, dx, dy)
    assert a.layer_values(0) == b.layer_values(0) == (1.75, 0.5, float(s.amplitude[0]))
    for _ in range(6):
        a.move_atmos()
        b.move_atmos()
        o.move_atmos()
    torch.cuda.synchronize()
    assert torch.equal(a.screen(0), b.screen(0))
    assert np.abs(a.screen(0)[0].cpu().numpy() - o.screens[0]).max() < 3e-4
    assert a.t["ext_count"][0, 0].item() == o.ext_count[0] == 2 * n + 10 + 3      # 6 x 1.75 -> 10 columns, 6 x 0.5 -> 3 rows
    with pytest.raises(Exception, match="stencil"):
        b.set_stencil(0, 0, np.zeros(7, dtype=np.uint32))
