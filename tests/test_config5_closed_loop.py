"""BASELINE configs[4] (production_sh_40x40_8m_3layers_d0_noise + the shipped denoiser) in CLOSED
loop against the oracle: real calibrated command matrix, delay 0, gain 0.3, photon + read-out noise
on, denoiser between image formation and centroiding, 20 frames.

Every stage of the frame is exercised with feedback: k_delay (delay 0) -> DM shapes -> noisy
k_frame_wave (bincube) -> k_denoise4c -> k_cog -> do_control, frame after frame.  The oracle runs
the same sequence with its own (different) algorithms and the denoiser's functional definition on
the CPU (pinned to the reference module by tests/test_denoiser.py).  Noise: both sides draw from
the same Philox streams, photon counts are integers -- a count differs only where the expected
flux of the two sides straddles a rounding / inversion threshold in its last bits."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from ao_marl_amd import geometry as G, modal, params, system  # noqa: E402
from ao_marl_amd.denoiser import SubapDenoiser  # noqa: E402

NAME = "production_sh_40x40_8m_3layers_d0_noise"
FRAMES = 20


@pytest.fixture(scope="module")
def noisy():
    from ao_marl_amd.sim import HipSim
    sysm = G.build_system(params.builtin(NAME))
    s = system.from_system(sysm, strehl_halfwin=8)
    assert s.noise == 3.0 and s.delay == 0.0 and abs(s.gain - 0.3) < 1e-6
    cal = modal.calibrate(s, sysm, HipSim(s, nenv=512, keep_phase=True), nfilt=5)
    assert cal.cmat.shape == (1286, 2400) and s.cmat is not None
    return sysm, s, cal


@pytest.fixture(params=["f32", "split_f16"])
def precision(request):
    """Every test of this file in both arithmetics of the library: the default (fp32 operands on fp32 matrix
    instructions: frame kernel, denoiser, GEMMs) and the fast mode (split-fp16 operand pairs)."""
    from ao_marl_amd import libaomarl as la
    keep = la.get_precision()
    la.set_precision(request.param)
    yield request.param
    la.set_precision(keep)


def _run_loop(noisy, resync, precision):
    """20 closed-loop frames on the HIP path and on the oracle.  resync: after every frame the
    oracle's integrator state is set to the HIP side's (see the test docstrings)."""
    from ao_marl_amd.sim import HipSim
    from tests.test_gpu_large import QuickOracle, _push
    _, s, cal = noisy
    seeds = [31, 47]
    sim = HipSim(s, nenv=len(seeds), keep_bincube=True)
    sim.set_modal(cal.volts2modes, cal.modes2volts)
    sim.reset(seeds)
    oracles = [QuickOracle(s, seed=sd) for sd in seeds]
    sim.t["seeds"].copy_(torch.tensor(seeds, dtype=torch.int32))
    _push(sim, oracles)
    sim.accumx[:] = 0
    sim.accumy[:] = 0
    sim.target_psf()
    dn_gpu = SubapDenoiser.load(device="cuda:0")
    dn_cpu = SubapDenoiser.load(device="cpu")
    dn_gpu.set_input_bound(float(s.nphot) * float(s.flux.max()) * 2 + 50)
    assert dn_gpu.wants_f32() == (precision == "f32")   # the kernel under test: all-fp32 / split-fp16
    log = []
    for it in range(FRAMES):
        # ---- HIP: next_part_two (delay 0) + the supervisor's denoiser branch of next_part_one
        sim.next_part_two(None)
        sim.move_atmos()
        sim.frame_fused(noise=True, write_bincube=True, cog=False)
        noisy_cube = sim.t["bincube"].cpu().numpy().copy()
        dn_gpu.denoise_bincube_(sim.t["bincube"])
        sim.do_centroids()
        sim.do_control()
        sl, cm, st = sim.slopes.cpu().numpy(), sim.com.cpu().numpy(), sim.strehl.cpu().numpy()
        vol = sim.voltage.cpu().numpy()
        c1, c2 = sim.t["com1"].cpu().numpy()[:, :s.nactu], sim.t["com2"].cpu().numpy()[:, :s.nactu]
        for e, o in enumerate(oracles):
            o.next_part_two(None)
            assert np.array_equal(o.voltage, o.com1)          # delay 0: the fresh command is applied
            o.move_atmos()
            o.raytrace_target()
            o.raytrace_wfs(atm=True, dms=False, reset=True)
            o.raytrace_wfs(atm=False, dms=True, reset=False)
            o.comp_image(noise=True)
            d = np.abs(noisy_cube[e] - o.bincube)
            cube = torch.from_numpy(o.bincube.copy()).unsqueeze(0)
            o.bincube[:] = dn_cpu.denoise_bincube_(cube)[0].numpy()
            o.do_centroids()
            o.do_control()
            scale = float(np.abs(o.com).max())
            log.append(dict(it=it, env=e, nflip=int((d > 1e-3).sum()), nbig=int((d > 2.0 + 1e-3).sum()),
                            npix=d.size, good=float((np.abs(sl[e] - o.slopes) < 1e-3).mean()),
                            dcom=float(np.abs(cm[e] - o.com).max()) / scale,
                            dvol=float(np.abs(vol[e] - o.voltage).max()) / scale,
                            dsr=abs(float(st[e, 0]) - o.strehl_se),
                            dsr_le=abs(float(st[e, 1]) - o.strehl_le), sr_le=float(st[e, 1])))
            if resync:
                o.com[:], o.com1[:], o.com2[:] = cm[e], c1[e], c2[e]
    dn_gpu.check_range()
    for r in log:
        print("%s frame %2d env %d: %3d counts differ (%d by more than 2)  slopes within 1e-3\": %.5f  "
              "|dcom| %.2e  |dvolt| %.2e of the command scale  |dSR| %.1e" %
              ("resync" if resync else "free  ", r["it"], r["env"], r["nflip"], r["nbig"], r["good"],
               r["dcom"], r["dvol"], r["dsr"]))
    return log


def test_closed_loop_with_noise_and_denoiser_matches_oracle(noisy, precision):
    """Frame-by-frame parity inside the closed loop.  The HIP loop runs free for 20 frames (its
    commands are the loop's own); the oracle runs the same frames and, after each comparison, takes
    over the HIP side's integrator state (com, com1, com2), so that every frame is compared from
    IDENTICAL loop state -- atmosphere, DM voltages, noise streams -- through the whole chain:
    delay line -> DM shapes -> noisy image -> denoiser -> centroids -> command matrix -> integrator.
    Bars: photon counts exact on all but <= 2e-4 of the pixels, slopes within 1e-3 arcsec on
    >= 99.9 % of the sub-apertures, commands to the smoke tolerance (2e-4 of the command scale
    + 5e-3 V), every frame."""
    log = _run_loop(noisy, True, precision)
    flips, pixels = sum(r["nflip"] for r in log), sum(r["npix"] for r in log)
    print("closed loop (state handed over each frame): %d of %d photon counts differ (%.2e)" %
          (flips, pixels, flips / float(pixels)))
    assert flips / float(pixels) <= 2e-4
    # more than a two-count difference: the expected flux of a pixel straddles the switch between the
    # two Poisson rules (inversion below 30, rounded normal above) in its last bits -- at most a few
    assert sum(r["nbig"] for r in log) <= 4
    com_scale = 28.0
    for r in log:
        assert r["nflip"] <= 2e-4 * r["npix"], r
        assert r["good"] >= 0.999, r
        # one differing photon (of ~240 in a sub-aperture) moves that sub-aperture's centroid by up to
        # 0.03 pixel -- more where the denoiser sharpens the spot around it -- which the tip-tilt rows
        # of the command matrix turn into up to ~2e-3 of the command scale (measured: 1.9e-3 for a
        # single count): the smoke tolerance, plus 3e-3 per differing count (a handful per 12 M pixels)
        assert r["dcom"] < 2e-4 + 5e-3 / com_scale + 3e-3 * r["nflip"], r
        assert r["dvol"] < 2e-4 + 5e-3 / com_scale, r
        assert r["dsr"] < 1e-3, r
    assert min(r["sr_le"] for r in log[-2:]) > 0.2           # the loop did close


def test_free_running_loops_stay_statistically_together(noisy, precision):
    """The same two loops WITHOUT the hand-over.  A photon count is a threshold decision on the
    expected flux, so 1e-5 differences of two fp32 algorithms flip a few counts, a flipped count
    moves a slope by ~0.01 arcsec, the integrator feeds that into every later frame's phase, which
    flips more counts: exact agreement decays geometrically by construction (any two implementations
    of this loop do that, in either arithmetic).  What must hold is that the decay is slow and the
    loops stay the same loop: after 20 frames >= 95 % of the slopes still agree to 1e-3 arcsec, < 1e-3
    of the counts differ in any frame, commands within 1 % of their scale, Strehl (short and long
    exposure) within 2e-3."""
    log = _run_loop(noisy, False, precision)
    for r in log:
        assert r["nflip"] < 1e-3 * r["npix"], r
        assert r["good"] >= 0.95, r
        assert r["dcom"] < 1e-2 and r["dvol"] < 1e-2, r
        assert r["dsr"] < 2e-3 and r["dsr_le"] < 2e-3, r
    # the first frames differ by the odd count only (the kernel's Box-Muller runs on the hardware
    # log2 / sin / cos: its normals are the oracle's to ~1e-6, ~3e-7 of the counts land on the other
    # side of a rounding threshold)
    first = [r for r in log if r["it"] < 3]
    assert all(r["nflip"] <= 2 and r["good"] >= 0.999 for r in first)


def test_supervisor_branch_is_that_sequence(noisy, precision):
    """VecRlSupervisor with the denoiser runs exactly the call sequence compared above."""
    from ao_marl_amd.env import VecRlSupervisor
    from ao_marl_amd.sim import HipSim
    _, s, cal = noisy
    dn = SubapDenoiser.load(device="cuda:0")
    sup = VecRlSupervisor(NAME, dict(n_reverse_filtered_from_cmat=5), 2, initial_seed=5, seed_stride=16,
                          autoencoder=dn, prefetch_atmos=False)
    assert dn.input_bound is not None and dn.wants_f32() == (precision == "f32")
    sim = HipSim(sup.s, nenv=2, keep_bincube=True)
    dn2 = SubapDenoiser.load(device="cuda:0")
    sup.reset()
    sim.reset(sup.env_seeds())
    for _ in range(3):
        sup.next_part_two(None, linear_control=True)
        sup.next_part_one()
        sim.next_part_two(None)
        sim.move_atmos()
        sim.frame_fused(noise=True, write_bincube=True, cog=False)
        dn2.denoise_bincube_(sim.t["bincube"])
        sim.do_centroids()
        sim.do_control()
    assert torch.equal(sup.get_slopes(), sim.slopes)
    assert torch.equal(sup.get_command(), sim.com)


def test_published_configuration_episode_slice():
    """The configuration the reference publishes (README.md:116-119: `production_sh_40x40_8m_3layers_d1_noise` + the
    shipped autoencoder, `--world-size 44 --n_zernike_start_end 0 1260`: 42 agents of 30 modes + tip-tilt), end to
    end through the two library calls per step: noisy frame kernel -> denoiser -> centroids -> control, 43 actors
    in k_actor_fused, 43 per-agent rewards.  16 steps: every step native, finite, equal seeds with equal actions
    equal bit for bit (the noise streams are keyed by seed and frame), different seeds apart, the denoiser inside
    its declared range, the loop closing."""
    from ao_marl_amd.agents import BatchedGaussianPolicy
    from ao_marl_amd.env import VecAoEnv
    name = "production_sh_40x40_8m_3layers_d1_noise"
    rl = dict(n_zernike_start_end=[0, 1260], n_reverse_filtered_from_cmat=5)
    env = VecAoEnv(name, 4, rl, initial_seed=1234, seed_stride=16, n_agents_modal=42, autoencoder=SubapDenoiser.load(device="cuda:0"))
    lay = env.layout
    assert lay.n_agents == 43 and lay.state_shapes()[0] == 120 and lay.state_shapes()[-1] == 8 and lay.action_dim == 1262
    assert env.frame_pipeline in (False, "auto")             # a noisy sensor with a denoiser: never pipelined
    seeds = env.supervisor.env_seeds().copy()
    seeds[1] = seeds[0]
    env.supervisor.env_seeds = lambda: seeds
    pol = BatchedGaussianPolicy(lay, last_layer_zero=False, seed=3, device="cuda:0")
    s = env.reset()
    assert env.frame_pipeline is False and s.shape == (4, env.state_dim) and torch.equal(s[0], s[1])
    g = torch.Generator(device="cuda:0").manual_seed(5)
    for it in range(16):
        eps = torch.randn(4, lay.action_dim, device="cuda:0", generator=g) * 0.1
        eps[1] = eps[0]
        a, _ = pol.select_action(s, eps=eps)
        assert env._native_step_ok(False)
        s, r, _, _ = env.step(a)
        assert r.shape == (4, 43) and torch.isfinite(s).all() and torch.isfinite(r).all()
        assert torch.equal(s[0], s[1]) and torch.equal(r[0], r[1])
        assert not torch.equal(s[0], s[2])
    env.supervisor.autoencoder.check_range()
    sl = env.supervisor.get_slopes()
    assert torch.isfinite(sl).all() and torch.equal(sl[0], sl[1]) and float(sl.std()) > 0
    st = env.supervisor.get_strehl().cpu().numpy()
    assert np.isfinite(st).all() and (st[:, 1] > 0).all() and (st[:, 0] <= 1.0 + 1e-3).all()
