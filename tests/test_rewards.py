"""Environment-level reward types (ao_env.py:585-860): the batched formulas of ao_marl_amd/rewards.py
against per-environment NumPy evaluations written from the reference's definitions (np.var =
population variance, halves = s[: n // 2], s[n // 2 :]); then, on the GPU, VecAoEnv.calculate_reward
for every supported name against the same NumPy on the tensors the environment exposes."""
import numpy as np
import pytest
import torch

from ao_marl_amd import rewards as R


def _np_slopes(name, s):
    n = len(s); x, y = s[: n // 2], s[n // 2:]
    v, a, q = np.var(s), np.average(s), np.average(np.square(s))
    base = name[:-5] if name.endswith("_norm") else name
    f = {
        "residual_wfs": lambda: -np.linalg.norm(s), "var_wfs": lambda: -v,
        "averages_wfs": lambda: -(np.average(x) + np.average(y)),
        "average_var_wfs": lambda: -v - (np.average(x) + np.average(y)),
        "average_residual_wfs": lambda: -v - (np.average(x) + np.average(y)),
        "r_modes_1": lambda: np.exp(-v), "r_modes_2": lambda: -v, "r_modes_3": lambda: np.exp(-v) - 1,
        "r_modes_4": lambda: np.exp(-np.var(np.square(s))), "r_modes_5": lambda: -np.var(np.square(s)),
        "r_modes_6": lambda: np.exp(-np.var(np.square(s))) - 1,
        "r_tt_1": lambda: -np.square(a), "r_tt_2": lambda: -np.abs(a), "r_tt_3": lambda: -q,
        "r_tt_5": lambda: np.exp(-q), "r_tt_6": lambda: np.exp(-q) - 1,
        "r_tt_7": lambda: -(np.square(np.average(x)) + np.square(np.average(y))),
        "r_modes_7": lambda: -(np.square(np.var(x)) + np.square(np.var(y))),
        "r_1_and_2": lambda: -v - np.abs(a), "single_agent_1": lambda: -q, "avg_square_m": lambda: -q,
        "sum_measurements_squared": lambda: -np.sum(np.square(s)), "single_agent_2": lambda: np.exp(-v) - 1,
        "single_agent_3": lambda: (-q) / (-q + np.exp(-v) - 1) + (np.exp(-v) - 1) / (-q + np.exp(-v) - 1),
        "single_agent_4": lambda: 0.4479 * (np.exp(-v) - 1) / (np.exp(-v) - 1 - q) + 0.5485 * (-q) / (np.exp(-v) - 1 - q),
        "new_single_agent": lambda: np.exp(-q), "log_avg_m": lambda: -np.log(1 + q),
    }
    return f[base]()


def _np_strehl(name, st):
    se, le, va = st[0], st[1], st[2]
    if name == "wavefront_phase_error": return -va
    if name == "strehl_ratio_le": return le
    if name == "strehl_ratio_se": return se
    if name == "r_le": return 0.0
    if name == "log_var": return -np.log(1 + va)
    k = int(name[-1]); off = 2.60 if k <= 4 else 3.50; mult = (1.0, 5.0, 10.0, 0.5)[(k - 1) % 4]
    return -(np.log(1 + va) - off) * mult


def _np_modes(name, m):
    if name == "avg_squared_modes": return -np.sum(np.square(m))
    if name == "true_avg_squared_modes": return -np.average(np.square(m))
    if name.startswith("avg_squared_modes_scaled_"): return -np.sum(np.square(m)) * 10.0 ** int(name[-1])
    return -float(name.split("_")[-1]) * np.average(np.square(m))


def test_reward_formulas_against_numpy():
    g = torch.Generator().manual_seed(5)
    s = torch.randn(7, 128, generator=g, dtype=torch.float64) * 0.3 + 0.05
    for name in R.SLOPES:
        got = R.slopes_reward(name, s).numpy()
        want = np.array([_np_slopes(name, row) for row in s.numpy()])
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-14, err_msg=name)
    st = torch.rand(7, 6, generator=g, dtype=torch.float64)
    for name in R.STREHL:
        got = R.strehl_reward(name, st).numpy()
        want = np.array([_np_strehl(name, row) for row in st.numpy()])
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=0, err_msg=name)
    m = torch.randn(7, 82, generator=g, dtype=torch.float64)
    for name in ("avg_squared_modes", "true_avg_squared_modes", "avg_squared_modes_scaled_1",
                 "avg_squared_modes_scaled_2", "avg_squared_modes_scaled_3", "avg_squared_modes_1000", "avg_squared_modes_2.5"):
        assert R.is_modes_reward(name)
        got = R.modes_reward(name, m).numpy()
        want = np.array([_np_modes(name, row) for row in m.numpy()])
        np.testing.assert_allclose(got, want, rtol=1e-12, err_msg=name)
    assert not R.is_modes_reward("avg_squared_modes_from_measurements")
    # the reference's chain has 61 named branches + the generic "avg_squared_modes_<factor>" + counterfactual
    assert len(R.SLOPES) == 39 and len(R.STREHL) == 13 and len(R.IMAGE) == 3 and len(R.PROJECTION) == 2 and not R.UNSUPPORTED
    # the branches that read the full-frame target image (ao_env.py:621-623, 654-656)
    from scipy.ndimage import center_of_mass
    img = torch.rand(3, 32, 32, generator=g, dtype=torch.float64) ** 4
    np.testing.assert_allclose(R.image_reward("image_sharpness", img).numpy(),
                               [np.sum(np.square(a)) / np.square(np.sum(a)) for a in img.numpy()], rtol=1e-12)
    for nm in ("r_tt_4", "r_tt_4_norm"):
        np.testing.assert_allclose(R.image_reward(nm, img).numpy(),
                                   [-np.sum(np.square(np.array(center_of_mass(a)) - a.shape[0] / 2.0)) for a in img.numpy()], rtol=1e-10)
    # ... and the projection comparisons (:736-760)
    pm, cm, fr = torch.randn(4, 87, generator=g, dtype=torch.float64), torch.randn(4, 87, generator=g, dtype=torch.float64), \
        torch.rand(87, generator=g, dtype=torch.float64)
    rng = torch.arange(5, 60)
    np.testing.assert_allclose(R.projection_reward("projection_comparison", pm, cm, rng).numpy(),
                               [-np.linalg.norm(a[5:60] - b[5:60]) for a, b in zip(pm.numpy(), cm.numpy())], rtol=1e-12)
    np.testing.assert_allclose(R.projection_reward("weighted_projection_comparison", pm, cm, rng, fr).numpy(),
                               [-np.linalg.norm((a[5:60] - b[5:60]) * fr.numpy()[5:60]) for a, b in zip(pm.numpy(), cm.numpy())], rtol=1e-12)


@pytest.mark.gpu
def test_env_reward_types_on_the_device():
    from ao_marl_amd.env import VecAoEnv
    env = VecAoEnv("production_sh_10x10_2m", 4, dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5),
                   n_agents_modal=1, frame_pipeline=False)
    env.reset()
    g = torch.Generator(device="cuda:0").manual_seed(3)
    for _ in range(3):
        env.step(torch.rand(4, env.layout.action_dim, device="cuda:0", generator=g) * 2 - 1)
    sup = env.supervisor
    s, e, st = sup.get_slopes().cpu().numpy(), sup.get_err().cpu().numpy(), sup.get_strehl().cpu().numpy()
    for name in list(R.SLOPES) + list(R.STREHL):
        got = env.calculate_reward(name).cpu().numpy()
        want = np.array([_np_slopes(name, s[i]) if name in R.SLOPES else _np_strehl(name, st[i]) for i in range(4)])
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-6, err_msg=name)
    np.testing.assert_allclose(env.calculate_reward("residual_dm").cpu().numpy(), -np.linalg.norm(e, axis=1), rtol=1e-5)
    v2m, m2v = sup.volts2modes.astype(np.float64), sup.modes2volts.astype(np.float64)
    rng = np.asarray(sup.obtain_action_range_modal())
    m = e.astype(np.float64) @ v2m.T
    mk = np.zeros_like(m); mk[:, rng] = m[:, rng]
    np.testing.assert_allclose(env.calculate_reward("variance_actuators_filtered_from_modes").cpu().numpy(),
                               -np.var(mk @ m2v.T, axis=1), rtol=1e-3)
    # the projector has one row per KEPT mode: with modes filtered and tip-tilt in the action range the reference's
    # own indexing runs off its end (ao_env.py:776) -- same exception here; inside the range, same numbers
    pm = s.astype(np.float64) @ sup.projector_wfs2modes.astype(np.float64).T
    assert pm.shape[1] == sup.nmodes - 5
    if rng.max() >= pm.shape[1]:
        with pytest.raises(IndexError):
            env.calculate_reward("avg_squared_modes_from_measurements")
        sup.n_modes_start_end, keep_tt = (0, 60), sup.include_tip_tilt
        sup.include_tip_tilt = False
        rng2 = np.asarray(sup.obtain_action_range_modal())
        np.testing.assert_allclose(env.calculate_reward("avg_squared_modes_from_measurements").cpu().numpy(),
                                   -np.sum(np.square(pm[:, rng2]), axis=1), rtol=1e-3)
        sup.n_modes_start_end, sup.include_tip_tilt = (0, 80), keep_tt
    else:
        np.testing.assert_allclose(env.calculate_reward("avg_squared_modes_from_measurements").cpu().numpy(),
                                   -np.sum(np.square(pm[:, rng]), axis=1), rtol=1e-3)
    msel = env.transform_state_to_zernike(sup.get_err(), return_reward=True).cpu().numpy()
    for name in ("avg_squared_modes", "true_avg_squared_modes", "avg_squared_modes_scaled_2", "avg_squared_modes_1000"):
        want = np.array([_np_modes(name, msel[i]) for i in range(4)])
        np.testing.assert_allclose(env.calculate_reward(name).cpu().numpy(), want, rtol=1e-4, err_msg=name)
    assert env.calculate_reward("counterfactual_rpc") is None
    with pytest.raises(NotImplementedError):
        env.calculate_reward("no_such_reward")
    # this environment was not built for the image rewards (no full-frame image kept, screens one frame ahead): loudly
    from ao_marl_amd.libaomarl import AomarlError
    with pytest.raises(RuntimeError, match="keep_tar_image"):
        env.calculate_reward("image_sharpness")
    with pytest.raises(AomarlError, match="next frame"):      # ... and the library refuses the image of a frame that is gone
        env.supervisor.sim.target_image()


@pytest.mark.gpu
def test_full_frame_image_and_projection_rewards_on_the_device():
    """The last five names of the reference's reward chain: Target.get_tar_image (aomarl_target_image: the whole
    npsf x npsf PSF on demand) against the oracle's full FFT, the three rewards that read it against NumPy / SciPy on
    the oracle's image, the phase-to-modes projector's defining property, and the two projection rewards against
    NumPy on the same device data.  An environment configured with one of these rewards keeps its screens on the
    current frame (no atmosphere prefetch, no frame pipeline)."""
    from scipy.ndimage import center_of_mass
    from ao_marl_amd.env import VecAoEnv
    from tests.oracle_vecsim import OracleVecSim
    rl = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5, reward_type="image_sharpness")
    # (no agent layout: the per-agent reward's factor is parsed from the reward type's name, helper_rewards.py:18,
    # which only the avg_squared_modes_<factor> family carries)
    env = VecAoEnv("production_sh_10x10_2m", 2, rl, initial_seed=21)
    oenv = VecAoEnv("production_sh_10x10_2m", 2, rl, initial_seed=21, device="cpu", sim_factory=OracleVecSim)
    assert env.supervisor.prefetch_atmos is False and env.frame_pipeline is False
    assert env.supervisor.keep_tar_image and oenv.supervisor.keep_tar_image
    with pytest.raises(RuntimeError, match="keep_le_image"):
        env.supervisor.get_tar_image(0, expo_type="le")
    env.supervisor.keep_le_image = oenv.supervisor.keep_le_image = True     # the full-frame long exposure: only when asked
    env.reset(); oenv.reset()
    g = torch.Generator().manual_seed(3)
    n = env.supervisor.s.npsf
    le_want, shots = np.zeros((2, n, n)), 0
    for it in range(3):
        a = torch.rand(2, env.action_dim, generator=g) * 2 - 1
        # rl_step = next_part_two: comp_tar_image forms the image of the phase the last next_part_one's raytrace_target
        # left (targetCompass.py:193, rlSupervisor.py:943-946) -- the ORACLE's tar_phase buffer as it stands, the
        # command this call applies is NOT in it
        want = oenv.supervisor.sim.target_image(retrace=False).numpy().astype(np.float64)
        env.rl_step(a.cuda()); oenv.rl_step(a)
        img = env.supervisor.get_tar_image(0).cpu().numpy().astype(np.float64)
        assert img.shape == want.shape == (2, n, n)
        assert np.abs(img - want).max() < 5e-4 * want.max(), it     # (two closed loops a few frames on: 2e-4 of the peak)
        assert np.abs(oenv.supervisor.get_tar_image(0).numpy() - want).max() < 1e-6 * want.max()
        le_want += want
        shots += 1
        env.linear_step(); oenv.linear_step()
    assert np.unravel_index(np.argmax(img[0]), img[0].shape) == np.unravel_index(np.argmax(want[0]), want[0].shape)
    np.testing.assert_allclose(env.calculate_reward("image_sharpness").cpu().numpy(),
                               [np.sum(np.square(x)) / np.square(np.sum(x)) for x in want], rtol=2e-3)
    np.testing.assert_allclose(env.calculate_reward("r_tt_4").cpu().numpy(),
                               [-np.sum(np.square(np.array(center_of_mass(x)) - n / 2.0)) for x in want], rtol=2e-2, atol=1e-3)
    # the image is not the state as it stands any more: the mirrors have moved on (round 4 re-traced here)
    assert np.abs(env.supervisor.sim.target_image().cpu().numpy() - img).max() > 1e-3 * want.max()
    # long exposure: d_image_le / strehl_counter = the mean of the short exposures since the reset
    le = env.supervisor.get_tar_image(0, expo_type="le").cpu().numpy().astype(np.float64)
    assert np.abs(le - le_want / shots).max() < 5e-4 * (le_want / shots).max()
    env.reset()
    with pytest.raises(RuntimeError, match="keep_le_image"):    # the counter is zero again
        env.supervisor.get_tar_image(0, expo_type="le")
    with pytest.raises(ValueError, match="Unknown exposure type"):
        env.supervisor.get_tar_image(0, expo_type="xx")
    env.supervisor.keep_le_image = oenv.supervisor.keep_le_image = False
    # projector: P . response^T = identity on the stack-array modes and on the tip-tilt pair
    sup = env.supervisor
    P = sup.projector_phase2modes.astype(np.float64)
    pup = np.flatnonzero(np.asarray(sup.s.mpupil).reshape(-1) != 0)
    assert P.shape == (sup.nmodes, pup.size)
    sim, m2v = sup.sim, np.asarray(sup.modes2volts, dtype=np.float32)
    for mode in (0, 7, sup.nmodes - 3, sup.nmodes - 1):
        cmd = np.zeros((1, m2v.shape[0]), dtype=np.float32)
        if mode < sup.nmodes - 2:
            cmd[0, :-2] = m2v[:-2, mode]
        else:
            cmd[0, -2:] = m2v[-2:, mode]
        sim.comp_dm_shape(torch.from_numpy(cmd).cuda(), 0, 1)
        sim.raytrace_wfs(atm=False, dms=True, reset=True, env_begin=0, env_count=1)
        ph = sim.t["wfs_phase"][0].reshape(-1).cpu().numpy().astype(np.float64)[pup]
        got = P @ ph
        blk = slice(0, sup.nmodes - 2) if mode < sup.nmodes - 2 else slice(sup.nmodes - 2, sup.nmodes)
        e = np.zeros(sup.nmodes); e[mode] = 1.0
        assert np.abs(got[blk] - e[blk]).max() < 2e-3, (mode, np.abs(got[blk] - e[blk]).max())
    # The two projection rewards read wfs.get_wfs_phase(0) BEHIND next_part_two (ao_env.py:736-760 called from rl_step,
    # :911-939): COMPASS's d_gs.d_phase as next_part_one's raytrace left it -- atmosphere of the frame + the mirrors
    # of the PREVIOUS command; next_part_two re-traces the target only.  The oracle holds exactly that buffer
    # (OracleSim.wfs_phase, written by its next_part_one, never by next_part_two), the environment under test is
    # the HIP one, and the current modes come from the voltages the new command produced.
    assert sup.keep_wfs_phase is False
    with pytest.raises(RuntimeError, match="keep_wfs_phase"):
        sup.get_wfs_phase()
    rl2 = dict(rl, reward_type="projection_comparison")
    env2 = VecAoEnv("production_sh_10x10_2m", 2, rl2, initial_seed=21)
    assert env2.supervisor.keep_wfs_phase and env2.supervisor.prefetch_atmos is False and env2.frame_pipeline is False
    sup2 = env2.supervisor
    sup2._projector_phase2modes = sup.projector_phase2modes            # (built above on an idle simulator)
    env2.reset(); oenv.reset()
    g = torch.Generator().manual_seed(8)
    rng = np.asarray(sup2.obtain_action_range_modal())
    v2m = np.asarray(sup2.volts2modes, dtype=np.float64)
    for it in range(3):
        a = torch.rand(2, env2.action_dim, generator=g) * 2 - 1
        env2.step(a.cuda()); oenv.step(a)
        osims = oenv.supervisor.sim.sims
        ph = np.stack([o.wfs_phase for o in osims]).astype(np.float64)             # left by the ORACLE's next_part_one
        # the kept phase IS that buffer (fp32 round-off of two closed loops a few frames on, microns)
        assert np.abs(sup2.get_wfs_phase().cpu().numpy().astype(np.float64) - ph).max() < 1e-3
        ph = (ph - ph.mean(axis=(1, 2), keepdims=True)).reshape(2, -1)[:, pup]
        proj = -(ph @ P.T)
        # rl_step's reward is computed behind next_part_two and before the next linear_step; step() has already run
        # that linear_step, so restate it: the phase of the frame just imaged against the voltages now on the mirrors
        cur = np.stack([o.voltage for o in osims]).astype(np.float64) @ v2m.T
        want = -np.linalg.norm(proj[:, rng] - cur[:, rng], axis=1)
        got = env2.calculate_reward("projection_comparison").cpu().numpy()
        # (the reward is a norm of DIFFERENCES of two mode vectors several times larger: 1e-3 um of phase shows up as
        # ~1 % of it; a re-trace with the new mirror shapes -- round 4's reading -- is off by the whole increment)
        np.testing.assert_allclose(got, want, rtol=3e-2)
        wantw = -np.linalg.norm((proj[:, rng] - cur[:, rng]) * np.asarray(sup2.freedom_vector)[rng], axis=1)
        np.testing.assert_allclose(env2.calculate_reward("weighted_projection_comparison").cpu().numpy(), wantw, rtol=3e-2)
    # and the order the reference computes it in: next_part_two (new command on the mirrors), THEN the reward -- the
    # sensor phase must still be the one of the last next_part_one, not a re-trace with the new mirror shapes
    before = sup2.get_wfs_phase().clone()
    a = torch.rand(2, env2.action_dim, generator=g) * 2 - 1
    env2.rl_step(a.cuda())                                              # next_part_two only
    assert torch.equal(sup2.get_wfs_phase(), before)
    sup2.sim.raytrace_wfs(atm=True, dms=True, reset=True)               # what round 4 returned: the NEW shapes in it
    assert not torch.equal(sup2.sim.t["wfs_phase"], before)
    r_after = env2.calculate_reward("projection_comparison").cpu().numpy()
    ph = (before - before.mean(dim=(1, 2), keepdim=True)).reshape(2, -1).cpu().numpy().astype(np.float64)[:, pup]
    cur = sup2.get_voltages().cpu().numpy().astype(np.float64)[:, :v2m.shape[1]] @ v2m.T
    np.testing.assert_allclose(r_after, -np.linalg.norm((-(ph @ P.T))[:, rng] - cur[:, rng], axis=1), rtol=2e-3)
