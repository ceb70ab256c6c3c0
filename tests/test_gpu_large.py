"""Full-size (40x40, 1200 sub-apertures, 3 layers, 1286 actuators) checks of the HIP path:
against the oracle on a couple of environments, the noisy configuration, and size-independent
properties at the benchmark batch size."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from ao_marl_amd import geometry as G, modal, params, system  # noqa: E402
from oracle import aoref  # noqa: E402

L_NAME = "production_sh_40x40_8m_3layers"


@pytest.fixture(scope="module")
def large():
    from ao_marl_amd.sim import HipSim
    sysm = G.build_system(params.builtin(L_NAME))
    s = system.from_system(sysm, strehl_halfwin=8)
    cal = modal.calibrate(s, sysm, HipSim(s, nenv=512, keep_phase=True), nfilt=5)
    assert [len(k) for k in cal.kept] == [1284, 2]          # the reference's fixture shapes
    assert cal.Btt.shape == (1286, 1283) and cal.cmat.shape == (1286, 2400)
    return sysm, s, cal


class QuickOracle(aoref.OracleSim):
    """Oracle env whose reset only runs a few extrusions (a full 40x40 refresh is 3888 GEMVs)."""
    NEXT = 40

    def reset(self, seed):
        s = self.s
        self.seed, self.frame = int(seed), 0
        self.accumx = np.zeros(s.nscreens, dtype=np.float32)
        self.accumy = np.zeros(s.nscreens, dtype=np.float32)
        self.ext_count = [0] * s.nscreens
        for l in range(s.nscreens):
            self.screens[l][:] = 0
            for _ in range(self.NEXT):
                self._extrude(l, 1 if s.deltax[l] > 0 else -1)
        self._alloc_ctrl()
        for sh in self.dm_shapes:
            sh[:] = 0
        self.reset_strehl()
        self.raytrace_target()


def _push(sim, oracles):
    s = sim.s
    for l, d in enumerate(s.screen_dim):
        sim.set_screen(l, np.stack([o.screens[l] for o in oracles]))
        for e, o in enumerate(oracles):
            sim.t["ext_count"][e, l] = o.ext_count[l]


# (unfused, write_bincube, precision).  unfused 0: one-pass frame kernel (science + WFS from the same
# tiles, stack-array DM from the commands); 1: separate passes; 2: one-pass kernel reading materialised
# DM shapes.  write_bincube False is what bench.py and VecAoEnv.step launch: the COG-only instantiations
# k_frame_wave<3, 1, true, false, false, false> (f32, the default) and <..., true> (fast mode), with
# the GEMMs of the control chain in the same arithmetic.
LARGE_CASES = [(0, True, "f32"), (1, True, "f32"), (2, True, "f32"), (0, False, "f32"),
               (0, False, "split_f16"), (0, True, "split_f16")]


@pytest.mark.parametrize("unfused,cube,precision", LARGE_CASES)
def test_large_frames_match_oracle(large, unfused, cube, precision):
    from ao_marl_amd import libaomarl as la
    keep = la.get_precision()
    la.set_precision(precision)
    try:
        _large_frames(large, unfused, cube, precision)
    finally:
        la.set_precision(keep)


def _large_frames(large, unfused, write_cube, precision):
    from ao_marl_amd import libaomarl as la
    from ao_marl_amd.sim import HipSim
    _, s, cal = large
    seeds = [1234, 4321]
    sim = HipSim(s, nenv=2, keep_bincube=True)
    assert sim.frame_fused_available()
    sim.set_option("force_unfused_frame", 1 if unfused == 1 else 0)
    sim.defer_shape = unfused == 0
    sim.set_modal(cal.volts2modes, cal.modes2volts)
    sim.reset(seeds)                       # exercises the full 1296-round reset on the GPU
    assert sim.screen(0).std().item() > 0.05
    oracles = [QuickOracle(s, seed=sd) for sd in seeds]
    sim.t["seeds"].copy_(torch.tensor(seeds, dtype=torch.int32))
    _push(sim, oracles)
    sim.accumx[:] = 0
    sim.accumy[:] = 0
    sim.target_psf()                        # pending PSF of the pushed screens, like the oracle's reset
    la.arith_launches(reset=True)
    for it in range(3):
        sim.next_part_two(None)
        sim.next_part_one(write_bincube=write_cube)
        sl, cm, st = sim.slopes.cpu().numpy(), sim.com.cpu().numpy(), sim.strehl.cpu().numpy()
        cube = sim.t["bincube"].cpu().numpy()
        for e, o in enumerate(oracles):
            o.next_part_two(None)
            o.next_part_one()
            assert np.abs(sl[e] - o.slopes).max() < 1e-4, it          # arcsec
            if write_cube:
                # brightest pixel exact wherever it is unambiguous (an almost flat phase puts the
                # spot on the corner of 4 pixels: a 4-way tie up to round-off)
                top2 = np.sort(o.bincube, axis=1)[:, -2:]
                clear = (top2[:, 1] - top2[:, 0]) > 1e-4 * top2[:, 1]
                assert np.array_equal(cube[e].argmax(axis=1)[clear], o.bincube.argmax(axis=1)[clear])
                assert np.abs(cube[e] - o.bincube).max() < 2e-5 * o.bincube.max()
            # tip-tilt rows of cmat are O(10): 1e-5 arcsec of slope round-off shows up as ~1e-3 V
            assert np.abs(cm[e] - o.com).max() < 2e-4 * np.abs(o.com).max() + 5e-3
            assert abs(st[e, 0] - o.strehl_se) < 2e-4
            assert abs(st[e, 2] - o.phase_var) < 1e-3 * o.phase_var + 1e-7
    for l in range(s.nscreens):
        scr = sim.screen(l).cpu().numpy()
        for e, o in enumerate(oracles):
            assert np.abs(scr[e] - o.screens[l]).max() < 5e-5
    # what ran: the instantiation and the arithmetic the case names
    launched = {k: v for k, v in la.arith_launches().items() if v}
    if unfused != 1:
        name = sim.frame_kernel_name()
        want = "k_frame_wave<3, 1, %s, false, %s, %s>" % ("true" if unfused == 0 else "false",
                                                          "true" if write_cube else "false",
                                                          "true" if precision == "split_f16" else "false")
        assert name == want, name
    if precision == "f32":
        assert not any("split" in k for k in launched), launched
    else:
        assert launched.get("gemm:split_f16_mfma", 0) > 0 and "gemm:f32_mfma" not in launched, launched


def test_noisy_wfs_matches_oracle():
    """config 5 sensor: magnitude 9, Poisson + 3 e- read-out noise, Philox streams shared with
    the oracle; photon counts are integers, so all but a handful of pixels agree exactly."""
    from ao_marl_amd.sim import HipSim
    sysm = G.build_system(params.builtin("production_sh_40x40_8m_3layers_d0_noise"))
    s = system.from_system(sysm, strehl_halfwin=8)
    assert s.noise == 3.0 and s.delay == 0.0 and abs(float(s.nphot) - 241.141) < 1e-2
    s.cmat = np.zeros((s.nactu + 0, s.nslope), dtype=np.float32)
    # no calibration needed: compare the raw image formation on an un-filtered system
    sim = HipSim(s, nenv=2, keep_bincube=True)
    oracles = [QuickOracle(s, seed=sd) for sd in (7, 8)]
    sim.reset([7, 8])
    sim.t["seeds"].copy_(torch.tensor([7, 8], dtype=torch.int32))
    _push(sim, oracles)
    assert sim.frame_fused_available()
    for frame in range(2):
        if frame == 0:
            sim.comp_image(noise=True, write_bincube=True, cog=True)
        else:                                  # same noise streams through the one-pass kernel
            sim.frame_fused(noise=True, write_bincube=True, cog=True)
        cube = sim.t["bincube"].cpu().numpy()
        sl = sim.slopes.cpu().numpy()
        for e, o in enumerate(oracles):
            o.raytrace_wfs(atm=True, dms=False, reset=True)
            o.comp_image(noise=True)
            o.do_centroids()
            d = np.abs(cube[e] - o.bincube)
            # a Poisson / rounding decision can flip where lambda differs in the last bits
            assert (d > 1e-3).mean() < 2e-4, (d > 1e-3).mean()
            assert d.max() <= 1.0 + 1e-3
            good = np.abs(sl[e] - o.slopes) < 1e-3
            assert good.mean() > 0.999
        assert int(sim.t["frame"][0]) == frame + 1
    assert cube.std() > 0.5 and (cube < 0).any()          # read-out noise is there


def test_size_independent_properties_at_bench_batch(large):
    """256 environments x 40x40: seeds decorrelate, equal seeds agree bit for bit, flux is
    conserved, the loop closes for every environment, sub-ranges can be stepped separately."""
    from ao_marl_amd.sim import HipSim
    _, s, cal = large
    n = 256
    sim = HipSim(s, nenv=n, keep_bincube=True)
    sim.set_modal(cal.volts2modes, cal.modes2volts)
    seeds = 1234 + 16 * np.arange(n)
    seeds[1] = seeds[0]                      # twin environments
    sim.reset(seeds)
    for it in range(25):
        sim.next_part_two(None)
        sim.next_part_one(write_bincube=(it == 24))
    torch.cuda.synchronize()
    sl = sim.slopes.cpu().numpy()
    assert np.array_equal(sl[0], sl[1])                       # same seed -> identical bits
    assert np.abs(np.corrcoef(sl[2], sl[3])[0, 1]) < 0.2       # different seeds decorrelate
    cube = sim.t["bincube"]
    flux = cube.sum(dim=2).cpu().numpy()
    assert np.allclose(flux, float(s.nphot) * s.flux[None, :], rtol=2e-5)
    st = sim.strehl.cpu().numpy()
    assert st[:, 0].min() > 0.25 and st[:, 0].mean() > 0.5     # every loop closed (H band)
    assert sim.t["strehl"][:, 5].sum().item() == 0              # PSF peak never on the window edge
    assert np.isfinite(sl).all() and np.isfinite(sim.com.cpu().numpy()).all()
    # stepping two halves separately == stepping the batch (no cross-env coupling)
    a = HipSim(s, nenv=4)
    b = HipSim(s, nenv=4)
    for x in (a, b):
        x.reset(seeds[:4])
    for it in range(3):
        a.next_part_two(None)
        a.next_part_one()
        for (e0, c) in ((0, 2), (2, 2)):
            b.next_part_two(None, env_begin=e0, env_count=c)
            b.next_part_one(env_begin=e0, env_count=c)
    assert torch.equal(a.slopes, b.slopes) and torch.equal(a.com, b.com)


def test_quadratic_form_slopes_match_the_image_forming_launch(large):
    """The slopes-only launch takes the centre of gravity as quadratic forms of the pupil field (spot_cog_qf: no
    transform, no image); a launch that also writes the bincube goes through the pruned transform, the binned image and
    its moments (spot_core).  Same state, same frame: the two sets of slopes agree far inside the 1e-4 arcsec they are
    both held to against the oracle, for every sub-aperture, and the stored image's own centre of gravity says the same."""
    from ao_marl_amd import libaomarl as la
    from ao_marl_amd.sim import HipSim
    _, s, cal = large
    assert la.get_precision() == "f32"
    nenv = 6
    sim = HipSim(s, nenv=nenv, keep_bincube=True)
    sim.defer_shape = True
    sim.set_modal(cal.volts2modes, cal.modes2volts)
    rng = np.random.default_rng(5)
    for l in range(len(s.screen_dim)):       # smooth random screens of a few microns: no need for the 1296-round reset here
        d = s.screen_dim[l]
        f = rng.normal(size=(nenv, d // 8 + 2, d // 8 + 2))
        up = np.kron(f, np.ones((8, 8)))[:, :d, :d]
        k = np.ones(15) / 15.0
        for ax in (1, 2):
            up = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, up)
        sim.set_screen(l, (1.5 * up).astype(np.float32))
    sim.t["voltage"][:, :s.nactu] = torch.randn(nenv, s.nactu, device="cuda") * 0.3
    sim._stale = True
    sim.frame_fused(noise=False, write_bincube=False, cog=True, dm_from_voltage=True)
    assert sim.frame_kernel_name().endswith("true, false, false, false>")
    qf = sim.slopes.cpu().numpy().copy()
    sim.slopes.zero_()
    sim.frame_fused(noise=False, write_bincube=True, cog=True, dm_from_voltage=True)
    assert sim.frame_kernel_name().endswith("true, false, true, false>")
    dft = sim.slopes.cpu().numpy().copy()
    assert np.isfinite(qf).all() and np.abs(qf).max() > 0.05          # real slopes (arcsec), not a flat wavefront
    assert np.abs(qf - dft).max() < 2e-5, np.abs(qf - dft).max()
    cube = sim.t["bincube"].cpu().numpy().reshape(nenv, s.nvalid, s.npix, s.npix).astype(np.float64)
    tot = cube.sum(axis=(2, 3))
    X = np.arange(s.npix)
    cx = (cube.sum(axis=2) @ X) / tot
    cy = (cube.sum(axis=3) @ X) / tot
    # slopes = (cog - offset) * scale, x block then y block (rtc_init.py:208-229, ao_env.py:665-666)
    sx, sy = qf[:, :s.nvalid], qf[:, s.nvalid:]
    assert np.abs(sx - (cx - s.cog_offset) * s.cog_scale).max() < 3e-5
    assert np.abs(sy - (cy - s.cog_offset) * s.cog_scale).max() < 3e-5


def test_reset_leaves_every_ring_on_a_line(large):
    """A reset starts the ring origins so that the first pupil pixel of a row sits on a 128-byte line (32 floats) once
    it is through (k_reset_env): the frame kernel's 32-pixel pieces of a layer without wind along x are whole lines for
    the whole episode.  The logical screens do not depend on where a ring starts: test_large_frames_match_oracle and
    the 40x40 reset test compare them with the oracle's."""
    from ao_marl_amd.sim import HipSim
    _, s, _ = large
    sim = HipSim(s, nenv=2)
    sim.reset([7, 8])
    org = sim.t["origin"].cpu().numpy()
    for l, off in enumerate(s.tar_atm_off):
        tox = int(round(off[0]))
        assert ((org[:, l, 0] + tox) % 32 == 0).all(), (l, org[:, l], tox)
    pitch = sim.screen_stride // sum(s.screen_dim) if len(set(s.screen_dim)) == 1 else None
    if pitch is not None:
        assert pitch % 32 == 0, pitch            # rows a whole number of 128-byte lines apart (648 + 56)
    for _ in range(3):                           # layer 0 of the production file has no wind along x: it stays there
        sim.move_atmos()
    org2 = sim.t["origin"].cpu().numpy()
    assert (org2[:, 0, 0] == org[:, 0, 0]).all()
