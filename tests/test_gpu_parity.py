"""HIP path (through the C ABI) vs the CPU oracle, stage by stage and as a closed loop.
Run on the GPU box: python -m pytest tests -m gpu.  Tolerances are fp32 round-off of different
summation orders / algorithms (pruned MFMA DFT vs padded FFT, ring buffer vs shift, fused vs
materialised raytrace); index-like outputs (brightest pixel, kept actuators) are exact."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from tests import helpers  # noqa: E402
from oracle import aoref  # noqa: E402

NAME = "production_sh_10x10_2m"
SEEDS = [1234, 1250, 77]


@pytest.fixture(scope="module")
def setup():
    from ao_marl_amd.sim import HipSim
    sysm, s, cal = helpers.calibrated(NAME)
    sim = HipSim(s, nenv=len(SEEDS), keep_bincube=True, keep_phase=True)
    sim.set_modal(cal.volts2modes, cal.modes2volts)
    sim.reset(SEEDS)
    oracles = [aoref.OracleSim(s, seed=sd) for sd in SEEDS]
    return sysm, s, cal, sim, oracles


def _push_oracle_state(sim, oracles):
    """Copy the oracle's screens into the HIP state (origin 0) so later stages are compared on
    identical inputs."""
    s = sim.s
    for l, d in enumerate(s.screen_dim):
        sim.set_screen(l, np.stack([o.screens[l] for o in oracles]))
    for e, o in enumerate(oracles):
        sim.t["ext_count"][e] = torch.tensor(o.ext_count, dtype=torch.int32)
        sim.accumx[e] = o.accumx
        sim.accumy[e] = o.accumy


def test_gemm_nt_matches_torch(setup):
    sim = setup[3]
    g = torch.Generator(device="cpu").manual_seed(0)
    for (M, N, K) in [(3, 90, 128), (64, 64, 16), (130, 257, 1957), (5, 7, 3), (256, 1286, 2400),
                      (70, 130, 100), (33, 65, 36), (64, 64, 4)]:
        A = torch.randn(M, K, generator=g).cuda()
        B = torch.randn(N, K, generator=g).cuda()
        Cm = sim.gemm_nt(A, B, alpha=-1.0)
        ref = -(A.double() @ B.double().T)
        assert (Cm.double() - ref).abs().max().item() < 2e-4 * max(1.0, K**0.5)
        C0 = torch.randn(M, N, generator=g).cuda()
        C1 = sim.gemm_nt(A, B, alpha=0.5, beta=2.0, Cout=C0.clone())
        ref = 0.5 * (A.double() @ B.double().T) + 2.0 * C0.double()
        assert (C1.double() - ref).abs().max().item() < 2e-4 * max(1.0, K**0.5)


def test_reset_screens_match_oracle(setup):
    _, s, _, sim, oracles = setup
    for l in range(s.nscreens):
        scr = sim.screen(l).cpu().numpy()
        for e, o in enumerate(oracles):
            d = np.abs(scr[e] - o.screens[l]).max()
            assert d < 2e-4, (l, e, d)   # 2n extrusions of fp32 recursion, screens ~ +-2 um
            assert np.std(o.screens[l]) > 0.05


def test_reset_variants_give_the_same_screens(setup):
    """The reset runs its 2 n x-extrusions on the TRANSPOSED screen (row operations) and transposes in
    place at the end, with the scatter of one round and the gather of the next in one launch; both can be
    switched off: the very same screens, ring origins and extrusion counters, bit for bit, also after
    further moves in every direction."""
    from ao_marl_amd.sim import HipSim
    _, s, _, _, _ = setup
    out = []
    for opts in ((("reset_untransposed", 1), ("extrude_unfused", 1)), (("reset_untransposed", 0), ("extrude_unfused", 1)),
                 (("reset_untransposed", 0), ("extrude_unfused", 0))):
        sim = HipSim(s, nenv=len(SEEDS))
        for k, v in opts:
            sim.set_option(k, v)
        sim.reset(SEEDS)
        for _ in range(5):
            sim.move_atmos()
        for l in range(s.nscreens):
            sim.extrude([l], [(-1, 2, -2, 1)[l % 4]])
        out.append((sim.t["screens"].clone(), sim.t["origin"].clone(), sim.t["ext_count"].clone()))
        del sim
    for o in out[1:]:
        assert torch.equal(o[0], out[0][0]) and torch.equal(o[1], out[0][1]) and torch.equal(o[2], out[0][2])


def test_move_atmos_matches_oracle(setup):
    _, s, _, sim, oracles = setup
    _push_oracle_state(sim, oracles)
    for _ in range(7):
        sim.move_atmos()
        for o in oracles:
            o.move_atmos()
    for l in range(s.nscreens):
        scr = sim.screen(l).cpu().numpy()
        for e, o in enumerate(oracles):
            assert np.abs(scr[e] - o.screens[l]).max() < 2e-5
    # the ring origin moved, the logical content did not wrap
    assert int(sim.t["origin"].abs().sum()) > 0
    assert np.array_equal(sim.accumx[0], oracles[0].accumx)
    assert np.array_equal(sim.accumy[0], oracles[0].accumy)


def test_all_four_extrusion_directions(setup):
    _, s, _, sim, oracles = setup
    _push_oracle_state(sim, oracles)
    import copy
    for d in (1, -1, 2, -2, 2, 1, -2, -1):
        # stencils are mirrored for the configured wind sign only; use the matching ones
        for o in oracles:
            ist = s.istx[0] if abs(d) == 1 else s.isty[0]
        n = s.screen_dim[0]
        flip = (d == 1 and s.deltax[0] < 0) or (d == -1 and s.deltax[0] > 0) or \
            (d == 2 and s.deltay[0] < 0) or (d == -2 and s.deltay[0] > 0)
        if flip:
            continue   # the stencil for that direction is not loaded (as in the reference)
        sim.extrude([0], [d])
        for o in oracles:
            o._extrude(0, d)
    scr = sim.screen(0).cpu().numpy()
    for e, o in enumerate(oracles):
        assert np.abs(scr[e] - o.screens[0]).max() < 2e-5


@pytest.mark.parametrize("generic", [0, 1])
def test_dm_shape_matches_oracle(setup, generic):
    """Both stack-array kernels: the separable-lattice fast path and the generic gather over the
    reference's influpos tables."""
    _, s, _, sim, oracles = setup
    rng = np.random.default_rng(3)
    volts = rng.normal(0, 1.0, size=(len(oracles), s.nactu)).astype(np.float32)
    sim.set_option("force_generic_dm", generic)
    sim.comp_dm_shape(torch.from_numpy(volts).cuda())
    sim.set_option("force_generic_dm", 0)
    for e, o in enumerate(oracles):
        o.comp_shapes(volts[e])
        for k in range(len(s.dms)):
            a = sim.dm_shape(k)[e].cpu().numpy()
            assert np.abs(a - o.dm_shapes[k]).max() < 2e-6 * max(1.0, np.abs(o.dm_shapes[k]).max())


@pytest.mark.parametrize("generic", [0, 1])
def test_raytrace_and_spot_image_match_oracle(setup, generic):
    """Unfused API (raytrace -> buffer -> image), fused kernels (software-pipelined fast variant
    and the generic one), standalone COG."""
    _, s, _, sim, oracles = setup
    sim.set_option("force_generic_spot", generic)
    _push_oracle_state(sim, oracles)
    rng = np.random.default_rng(4)
    volts = rng.normal(0, 0.5, size=(len(oracles), s.nactu)).astype(np.float32)
    sim.comp_dm_shape(torch.from_numpy(volts).cuda())
    # unfused API: raytrace -> phase buffer -> image from buffer
    sim.raytrace_wfs(atm=True, dms=False, reset=True)
    sim.raytrace_wfs(atm=False, dms=True, reset=False)
    sim.comp_image(from_phase_buffer=True, noise=False, write_bincube=True, cog=True)
    cube_buf = sim.t["bincube"].cpu().numpy().copy()
    sl_buf = sim.slopes.cpu().numpy().copy()
    ph = sim.t["wfs_phase"].cpu().numpy()
    # fused path
    sim.comp_image(from_phase_buffer=False, noise=False, write_bincube=True, cog=True)
    cube_fused = sim.t["bincube"].cpu().numpy().copy()
    sl_fused = sim.slopes.cpu().numpy().copy()
    sim.do_centroids()
    sl_cog = sim.slopes.cpu().numpy().copy()
    # slopes-only path (no bincube): single-pass reduction of the raw image
    sim.comp_image(from_phase_buffer=False, noise=False, write_bincube=False, cog=True)
    sl_only = sim.slopes.cpu().numpy().copy()
    sim.set_option("force_generic_spot", 0)
    for e, o in enumerate(oracles):
        o.comp_shapes(volts[e])
        o.raytrace_wfs(atm=True, dms=False, reset=True)
        o.raytrace_wfs(atm=False, dms=True, reset=False)
        assert np.abs(ph[e] - o.wfs_phase).max() < 1e-5
        o.comp_image(noise=False)
        o.do_centroids()
        scale = o.bincube.max()
        assert np.abs(cube_buf[e] - o.bincube).max() < 2e-5 * scale
        assert np.abs(cube_fused[e] - o.bincube).max() < 2e-5 * scale
        # "bit-exact centroid indices": brightest pixel of every spot
        assert np.array_equal(cube_fused[e].argmax(axis=1), o.bincube.argmax(axis=1))
        assert np.allclose(cube_fused[e].sum(axis=1), s.nphot * s.flux, rtol=1e-5)
        for sl in (sl_buf, sl_fused, sl_cog, sl_only):
            assert np.abs(sl[e] - o.slopes).max() < 1e-4   # arcsec (north-star tolerance)
        assert np.abs(sl_fused[e] - o.slopes).max() < 2e-5


def test_control_chain_matches_oracle(setup):
    _, s, cal, sim, oracles = setup
    rng = np.random.default_rng(5)
    sl = rng.normal(0, 0.1, size=(len(oracles), s.nslope)).astype(np.float32)
    c0 = rng.normal(0, 0.5, size=(len(oracles), s.nactu)).astype(np.float32)
    sim.set_com(torch.from_numpy(c0).cuda())
    sim.t["slopes"].copy_(torch.from_numpy(sl))
    sim.t["com1"].zero_()
    sim.t["com2"].zero_()
    sim.do_control()
    for rep in range(3):
        sim.apply_control()
    for e, o in enumerate(oracles):
        o.set_com(c0[e])
        o.slopes[:] = sl[e]
        o.com1[:] = 0
        o.com2[:] = 0
        o.do_control()
        for rep in range(3):
            o.apply_control()
        # err = -cmat . s: against the exact product, by the standard bound of a dot product in fp32
        # (the library's kernel carries operands as fp16 hi + lo pairs, 23 significant bits, fp32
        # accumulation; the oracle sums 128 fp32 products in order and is itself 5-10 ulp off on the
        # tip-tilt rows, so the two are compared at that level, not at 2e-5 absolute)
        exact = -(s.cmat.astype(np.float64) @ sl[e].astype(np.float64))
        bound = np.abs(s.cmat.astype(np.float64)) @ np.abs(sl[e].astype(np.float64))
        assert np.all(np.abs(sim.err[e].cpu().numpy() - exact) <= 4e-7 * bound + 1e-6)
        assert np.abs(sim.err[e].cpu().numpy() - o.err).max() < 2e-5 + 1e-6 * np.abs(o.err).max()
        assert np.abs(sim.com[e].cpu().numpy() - o.com).max() < 2e-6 * np.abs(o.com).max() + 1e-5
        assert np.abs(sim.voltage[e].cpu().numpy() - o.voltage).max() < 2e-6 * np.abs(o.com).max() + 1e-5
        for k in range(len(s.dms)):
            assert np.abs(sim.dm_shape(k)[e].cpu().numpy() - o.dm_shapes[k]).max() < 1e-5


def test_rl_control_matches_numpy(setup):
    _, s, cal, sim, _ = setup
    rng = np.random.default_rng(6)
    nm = cal.volts2modes.shape[0]
    modes = np.r_[np.arange(0, 80), nm - 2, nm - 1]
    freedom = rng.uniform(0.01, 0.1, size=nm).astype(np.float32)
    sim.set_modal(cal.volts2modes, cal.modes2volts, freedom, modes)
    c0 = rng.normal(0, 0.5, size=(sim.nenv, s.nactu)).astype(np.float32)
    act = rng.uniform(-1, 1, size=(sim.nenv, modes.size)).astype(np.float32)
    sim.set_com(torch.from_numpy(c0).cuda())
    sim.rl_control(torch.from_numpy(act).cuda())
    got = sim.com.cpu().numpy()
    for e in range(sim.nenv):
        # rlSupervisor.py:800-816 in NumPy
        m = cal.volts2modes.dot(c0[e])
        m[modes] += act[e] * freedom[modes]
        want = cal.modes2volts.dot(m)
        assert np.abs(got[e] - want).max() < 1e-6 * np.abs(want).max() + 1e-6
    v = sim.volts2modes(torch.from_numpy(c0).cuda()).cpu().numpy()
    assert np.abs(v - c0 @ cal.volts2modes.T).max() < 2e-5
    sim.set_modal(cal.volts2modes, cal.modes2volts)


@pytest.mark.parametrize("valu", [0, 1, 2])
def test_target_psf_and_strehl_match_oracle(setup, valu):
    """All three PSF-row kernels: software-pipelined MFMA (0), VALU fallback (1), generic MFMA (2)."""
    _, s, _, sim, oracles = setup
    sim.set_option("force_valu_target", 1 if valu == 1 else 0)
    sim.set_option("force_generic_target", 1 if valu == 2 else 0)
    _push_oracle_state(sim, oracles)
    rng = np.random.default_rng(7)
    volts = rng.normal(0, 0.3, size=(len(oracles), s.nactu)).astype(np.float32)
    sim.comp_dm_shape(torch.from_numpy(volts).cuda())
    sim.reset_strehl()
    for rep in range(2):
        sim.target_psf()
        sim.comp_strehl()
    sim.raytrace_target()
    tp = sim.t["tar_phase"].cpu().numpy()
    st = sim.strehl.cpu().numpy()
    for e, o in enumerate(oracles):
        o.comp_shapes(volts[e])
        o.reset_strehl()
        o.raytrace_target()
        assert np.abs(tp[e] - o.tar_phase).max() < 1e-5
        for rep in range(2):
            want = o.comp_strehl()
        assert abs(st[e, 0] - want[0]) < 2e-5 * max(want[0], 1e-3) + 1e-7
        assert abs(st[e, 1] - want[1]) < 2e-5 * max(want[1], 1e-3) + 1e-7
        assert abs(st[e, 2] - want[2]) < 1e-4 * want[2] + 1e-9
        assert abs(st[e, 3] - want[3]) < 1e-4 * want[3] + 1e-9
        # comp_strehl(do_fit=True), the reference's default: the peaks fitted by two 1-D sincs (oracle: the same
        # algorithm in double precision on its own window)
        fit, sf = o.get_strehl(do_fit=True), sim.strehl_fit.cpu().numpy()
        assert fit[0] >= want[0] and fit[1] >= want[1] and fit[0] < 1.2 * want[0] + 1e-6
        assert abs(sf[e, 0] - fit[0]) < 1e-4 * max(fit[0], 1e-3) + 1e-7, (sf[e], fit)
        assert abs(sf[e, 1] - fit[1]) < 1e-4 * max(fit[1], 1e-3) + 1e-7, (sf[e], fit)
    sim.set_option("force_valu_target", 0)
    sim.set_option("force_generic_target", 0)


def test_fused_frame_matches_oracle_and_unfused(setup):
    """One-pass frame kernel (science PSF rows + WFS spots from the same phase tiles) vs the
    oracle and vs the separate target / WFS kernels, with and without the bincube."""
    _, s, _, sim, oracles = setup
    assert sim.frame_fused_available()
    _push_oracle_state(sim, oracles)
    rng = np.random.default_rng(11)
    volts = rng.normal(0, 0.4, size=(len(oracles), s.nactu)).astype(np.float32)
    volts[:, -2:] = rng.normal(0, 0.05, size=(len(oracles), 2))      # tip-tilt: analytic planes
    sim.comp_dm_shape(torch.from_numpy(volts).cuda())
    out = {}
    for mode in ("unfused", "fused_cube", "fused", "otf_cube", "otf", "otf_f32"):
        sim.reset_strehl()
        sim.set_option("force_f32_dft", 1 if mode == "otf_f32" else 0)
        if mode.startswith("otf"):
            # stack-array DM evaluated from st.voltage inside the kernel: the stored planes are
            # poisoned to prove they are not read
            assert sim.dm_from_voltage_available()
            sim.set_com(torch.from_numpy(volts).cuda())
            sim.apply_control(comp_voltage=False, defer_shape=True)
            nz = s.dms[0].dim ** 2
            sim.t["dm_shape"][:, :nz] = 1e3
            assert sim._stale
        if mode == "unfused":
            sim.target_psf()
            sim.comp_image(noise=False, write_bincube=True, cog=True)
        else:
            sim.t["bincube"].zero_()
            sim.slopes.zero_()
            sim.frame_fused(noise=False, write_bincube=mode.endswith("cube"), cog=True)
        sim.comp_strehl()
        out[mode] = (sim.slopes.cpu().numpy().copy(), sim.strehl.cpu().numpy().copy(),
                     sim.t["bincube"].cpu().numpy().copy())
    sim.set_option("force_f32_dft", -1)         # back to the library's precision mode
    # materialising afterwards restores the stored planes from the same voltages
    shp = sim.dm_shape(0).cpu().numpy()
    assert not sim._stale
    for e, o in enumerate(oracles):
        o.comp_shapes(volts[e])
        o.reset_strehl()
        o.raytrace_target()
        want = o.comp_strehl()
        o.raytrace_wfs(atm=True, dms=True, reset=True)
        o.comp_image(noise=False)
        o.do_centroids()
        assert np.abs(shp[e].ravel() - o.dm_shapes[0].ravel()).max() < 2e-6 * max(1.0, np.abs(o.dm_shapes[0]).max())
        for mode in ("fused_cube", "fused", "otf_cube", "otf", "otf_f32"):
            sl, st, cube = out[mode]
            assert np.abs(sl[e] - o.slopes).max() < 2e-5, mode
            assert np.abs(sl[e] - out["unfused"][0][e]).max() < 2e-5, mode
            assert abs(st[e, 0] - want[0]) < 2e-5 * max(want[0], 1e-3) + 1e-7, mode
            assert abs(st[e, 2] - want[2]) < 1e-4 * want[2] + 1e-9, mode
        for mode in ("fused_cube", "otf_cube"):
            cube = out[mode][2]
            assert np.abs(cube[e] - o.bincube).max() < 2e-5 * o.bincube.max(), mode
            assert np.array_equal(cube[e].argmax(axis=1), o.bincube.argmax(axis=1)), mode


@pytest.mark.parametrize("unfused", [0, 1, 2, 3])
def test_closed_loop_trace_matches_oracle(setup, unfused):
    """40 frames of the integrator loop (next_part_two + next_part_one) from a common state: the
    one-pass frame kernel with the stack-array DM evaluated from the commands in the library's FAST mode
    -- DFTs and internal GEMMs on split-fp16 MFMAs -- (0), separate target / WFS passes (1), the
    one-pass kernel reading materialised DM shapes (2), the one-pass kernel as the product runs it by
    default: everything fp32 (3).  1 - 3 in the default precision (f32)."""
    from ao_marl_amd import libaomarl as la
    _, s, _, sim, oracles = setup
    keep_precision = la.get_precision()
    la.set_precision("split_f16" if unfused == 0 else "f32")
    la.arith_launches(reset=True)
    sim.set_option("force_unfused_frame", 1 if unfused == 1 else 0)
    sim.set_option("force_f32_dft", -1)
    sim.defer_shape = unfused in (0, 3)
    sim.reset(SEEDS)
    for o, sd in zip(oracles, SEEDS):
        o.reset(sd)
    _push_oracle_state(sim, oracles)   # remove the reset's accumulated round-off
    sim.target_psf()
    for o in oracles:
        o.raytrace_target()
    worst = 0.0
    for it in range(40):
        sim.next_part_two(None)
        sim.next_part_one()
        sl = sim.slopes.cpu().numpy()
        cm = sim.com.cpu().numpy()
        st = sim.strehl.cpu().numpy()
        for e, o in enumerate(oracles):
            o.next_part_two(None)
            o.next_part_one()
            worst = max(worst, np.abs(sl[e] - o.slopes).max())
            assert np.abs(sl[e] - o.slopes).max() < 1e-4, it
            assert np.abs(cm[e] - o.com).max() < 5e-5 * np.abs(o.com).max() + 1e-3, it
            assert abs(st[e, 0] - o.strehl_se) < 1e-4, it
            assert abs(st[e, 1] - o.strehl_le) < 1e-4, it
    sim.set_option("force_unfused_frame", 0)
    sim.defer_shape = True
    launched = {k: v for k, v in la.arith_launches().items() if v}
    la.set_precision(keep_precision)
    if unfused == 0:
        assert launched.get("frame_kernel_dft:split_f16_mfma") == 40 and launched.get("gemm:split_f16_mfma", 0) > 0
    else:
        assert not any("split" in k for k in launched), launched
    assert sim.strehl[:, 0].min().item() > 0.3   # the loop closed
    print("worst slope deviation over the trace: %.3g arcsec" % worst)


def test_calibration_through_hip_backend_matches_oracle_backend(setup):
    from ao_marl_amd import modal
    from ao_marl_amd.sim import HipSim
    sysm_o, s_o, cal_o = setup[0], setup[1], setup[2]
    sysm, s = helpers.uncalibrated(NAME)
    backend = HipSim(s, nenv=64, keep_phase=True)
    cal = modal.calibrate(s, sysm, backend, nfilt=5)
    assert [len(k) for k in cal.kept] == [88, 2]
    assert np.array_equal(cal.kept[0], cal_o.kept[0])      # same actuators survive, exactly
    assert np.abs(cal.imat - cal_o.imat).max() < 1e-4 * np.abs(cal_o.imat).max()
    assert np.abs(cal.cmat - cal_o.cmat).max() < 2e-3 * np.abs(cal_o.cmat).max()
    assert np.abs(cal.Btt - cal_o.Btt).max() < 1e-6


def test_batched_policy_native_gemm_matches_torch(setup):
    """The stacked SAC actor on the library's batched MFMA GEMM (fused bias + ReLU) == torch.bmm."""
    from ao_marl_amd.agents import AgentLayout, BatchedGaussianPolicy
    lay = AgentLayout(1283, [0, 1274], 13, include_tip_tilt=True, window_n_zernike=20,
                      include_tip_tilt_windowed=True, n_filtered=5)
    pol = BatchedGaussianPolicy(lay, last_layer_zero=False, seed=3, device="cuda:0")
    with torch.no_grad():
        pol.b1.normal_(0, 0.1)
        pol.bm.normal_(0, 0.1)
        pol.bs.normal_(0, 0.1)
    st = torch.randn(37, lay.state_dim, device="cuda:0")
    pol.use_native = True
    m1, l1 = pol.forward(st)
    pol.use_native = False
    m0, l0 = pol.forward(st)
    acts = lay.action_shapes()
    for i, na in enumerate(acts):        # padded head columns are meaningless
        assert torch.allclose(m1[i, :, :na], m0[i, :, :na], atol=2e-4, rtol=1e-4)
        assert torch.allclose(l1[i, :, :na], l0[i, :, :na], atol=2e-4, rtol=1e-4)
    pol.use_native = True
    a, mu = pol.select_action(st)
    assert a.shape == (37, 1276) and mu.abs().max() <= 1.0


@pytest.mark.parametrize("shape", [(256, 1286, 2400), (256, 1283, 1286), (768, 648, 1960), (5, 90, 130), (64, 87, 90)])
def test_split_f16_gemm_against_float64(shape):
    """k_gemm_nt_h (hi + lo fp16 operand pairs on the f16 matrix pipe, fp32 accumulation): within a
    few fp32 roundings of the exact product, with and without split-K, for operands with the dynamic
    range of the loop's matrices (entries from 1e-6 of the largest one up, scaled by powers of two)."""
    import ctypes as C
    from ao_marl_amd import libaomarl as la
    lib = la.load()
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K, generator=g))            # heavy-tailed
    B = torch.randn(N, K, generator=g) * torch.exp(2.0 * torch.randn(N, K, generator=g)) * 3e-4
    Kp = (K + 3) // 4 * 4
    Ad = torch.zeros(M, Kp, device="cuda"); Ad[:, :K] = A.cuda()
    Bd = torch.zeros(N, Kp, device="cuda"); Bd[:, :K] = B.cuda()
    want = A.double() @ B.double().T
    bound = (A.double().abs() @ B.double().abs().T)             # sum |a| |b|
    sa = 2.0 ** int(np.floor(np.log2(4096.0 / A.abs().max().item())))
    sb = 2.0 ** int(np.floor(np.log2(4096.0 / B.abs().max().item())))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for ws_floats in (0, 8 * M * N):
        Cd = torch.full((M, N), 7.0, device="cuda")
        ws = torch.zeros(max(ws_floats, 1), device="cuda")
        la.check(lib.aomarl_gemm_nt_split(M, N, K, 0.5, Ad.data_ptr(), Kp, Bd.data_ptr(), Kp, 2.0, Cd.data_ptr(), N,
                                          sa, sb, ws.data_ptr() if ws_floats else None, ws_floats, stream))
        got = Cd.cpu().double()
        err = (got - (0.5 * want + 14.0)).abs()
        # split operands carry 2^-21 of each product at worst (two truncations of 2^-22), fp32
        # accumulation adds its own roundings: 1e-6 of sum |a||b| covers both
        assert (err <= 1e-6 * bound + 1e-5).all(), (ws_floats, float((err / (bound + 1e-30)).max()))
        # and it is as good as fp32 arithmetic in the usual sense
        assert float(err.max() / want.abs().max()) < 2e-6
