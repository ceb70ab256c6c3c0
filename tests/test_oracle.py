"""The CPU oracle against independent NumPy statements of the same mathematics, published
known-answer vectors (Philox) and the statistics the reference recorded from real COMPASS."""
import os

import numpy as np
import pytest

from oracle import aoref
from tests import helpers


def test_philox_known_answers():
    """Random123 v1.09 kat_vectors, philox4x32-10."""
    L = aoref.lib()
    out = np.zeros(4, dtype=np.uint32)
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        L.aoref_philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32), out)
        assert tuple(int(x) for x in out) == want
    z = aoref.normals(7, 0, 3, 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    u = aoref.uniforms(7, 1, 3, 100000)
    assert 0 < u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    assert not np.array_equal(aoref.normals(7, 0, 4, 16), aoref.normals(7, 0, 3, 16))


def _np_extrude_plus_x(p, A, B, ist, amp, eps):
    """iterkolmo.py:278-286 in NumPy."""
    n = p.shape[0]
    zref = p[0, n - 1]
    z = p.flatten()[ist] - zref
    new = A.astype(np.float64) @ z + B.astype(np.float64) @ (eps * amp) + zref
    p1 = np.zeros_like(p)
    p1[:, :n - 1] = p[:, 1:]
    p1[:, n - 1] = new
    return p1


def test_extrusion_all_directions_against_numpy():
    from ao_marl_amd import geometry as G
    n, L0 = 24, 1e4
    A, B, istx, isty = G.extrusion_matrices(n, L0, 1.0, 1.0)
    rng = np.random.default_rng(0)
    p = rng.normal(size=(n, n)).astype(np.float32)
    eps = rng.normal(size=n).astype(np.float32)
    L = aoref.lib()
    tmp = np.empty(istx.size + n, dtype=np.float32)

    def run(scr, ist, d):
        q = scr.copy()
        L.aoref_extrude(q.reshape(-1), n, A, ist.size, B, ist, d, 0.3, eps, tmp)
        return q

    want = _np_extrude_plus_x(p.astype(np.float64), A, B, istx, 0.3, eps.astype(np.float64))
    assert np.abs(run(p, istx, 1) - want).max() < 1e-5
    # +y is +x on the transposed screen
    assert np.abs(run(np.ascontiguousarray(p.T), isty, 2).T - want).max() < 1e-5
    # negative directions are the positive ones on the point-reflected screen with the mirrored
    # stencil (iterkolmo.py:246-249)
    fl = (n * n - 1 - istx.astype(np.int64)).astype(np.uint32)
    r = run(np.ascontiguousarray(p[::-1, ::-1]), fl, -1)
    assert np.abs(r[::-1, ::-1] - want).max() < 1e-5
    fly = (n * n - 1 - isty.astype(np.int64)).astype(np.uint32)
    r = run(np.ascontiguousarray(p.T[::-1, ::-1]), fly, -2)
    assert np.abs(r[::-1, ::-1].T - want).max() < 1e-5


@pytest.fixture(scope="module")
def small():
    return helpers.calibrated("production_sh_10x10_2m")


def test_sh_image_against_numpy_fft(small):
    _, s, _ = small
    rng = np.random.default_rng(1)
    phase = rng.normal(0, 0.15, size=(s.n, s.n)).astype(np.float32)
    o = helpers.CalSim(s)
    o.wfs_phase[:] = phase
    o.comp_image(noise=False)
    pd, nf = s.pdiam, s.nfft
    for i in (0, 17, s.nvalid - 1):
        idx = s.phasemap[:, i]
        tile = (s.mpupil.reshape(-1)[idx] *
                np.exp(1j * (phase.reshape(-1)[idx].astype(np.float64) * 2 * np.pi / s.wfs_lambda -
                             s.halfxy.reshape(-1)))).reshape(pd, pd)
        big = np.zeros((nf, nf), dtype=np.complex128)
        big[:pd, :pd] = tile
        hr = np.abs(np.fft.fft2(big))**2
        img = hr.reshape(-1)[s.binmap].sum(axis=0)
        img = img / img.sum() * s.nphot * s.flux[i]
        assert np.abs(o.bincube[i] - img).max() < 2e-5 * img.max()
    # a pure x-tilt moves the spot along the fast pixel axis only, by the geometric amount
    yy, xx = np.mgrid[0:s.n, 0:s.n]
    tilt_as = 0.2                                         # arcsec
    pix_m = s.subapd / s.pdiam
    o.wfs_phase[:] = (tilt_as / 206265.0 * xx * pix_m * 1e6).astype(np.float32)   # microns
    o.comp_image(noise=False)
    o.do_centroids()
    sx, sy = o.slopes[:s.nvalid], o.slopes[s.nvalid:]
    assert np.abs(sy).max() < 2e-3
    assert np.abs(np.abs(sx) - tilt_as).max() < 0.02      # sign convention aside
    assert np.all(np.sign(sx) == np.sign(sx[0]))


def test_dm_shapes_and_raytrace_against_numpy(small):
    _, s, _ = small
    o = helpers.CalSim(s)
    rng = np.random.default_rng(2)
    v = rng.normal(size=s.nactu).astype(np.float32)
    o.comp_shapes(v)
    d = s.dms[0]
    want = np.zeros((d.dim, d.dim))
    for a in range(d.ntotact):
        # influ[xoff, yoff, act] placed with x on the fast axis
        x0, y0 = d.i1[a], d.j1[a]
        want[y0:y0 + d.influsize, x0:x0 + d.influsize] += v[a] * d.influ[:, :, a].T
    assert np.abs(o.dm_shapes[0] - want).max() < 1e-6
    tt = s.dms[1]
    assert np.abs(o.dm_shapes[1] - (v[-2] * tt.influ[:, :, 0] + v[-1] * tt.influ[:, :, 1])).max() < 1e-6
    o.raytrace_wfs(atm=False, dms=True, reset=True)
    ox, oy = [int(t) for t in s.wfs_dm_off[0]]
    tx, ty = [int(t) for t in s.wfs_dm_off[1]]
    ref = o.dm_shapes[0][oy:oy + s.n, ox:ox + s.n] + o.dm_shapes[1][ty:ty + s.n, tx:tx + s.n]
    assert np.abs(o.wfs_phase - ref).max() < 1e-6
    # bilinear at a half-pixel offset
    out = np.zeros((4, 4), dtype=np.float32)
    src = np.arange(36, dtype=np.float32).reshape(6, 6)
    aoref.lib().aoref_raytrace(out.reshape(-1), 4, 4, src.reshape(-1), 6, 0.5, 1.0, 0)
    assert np.allclose(out, src[1:5, 0:4] + 0.5)


def test_psf_against_numpy_fft(small):
    _, s, _ = small
    o = helpers.CalSim(s)
    rng = np.random.default_rng(3)
    o.tar_phase[:] = rng.normal(0, 0.08, size=o.tar_phase.shape).astype(np.float32)
    o.comp_strehl()
    amp = s.spupil * np.exp(1j * o.tar_phase.astype(np.float64) * 2 * np.pi / s.tar_lambda)
    big = np.zeros((s.npsf, s.npsf), dtype=np.complex128)
    big[:s.pupdiam, :s.pupdiam] = amp
    psf = np.abs(np.fft.fft2(big))**2
    assert abs(o.strehl_se_full - psf.max() / s.spupil.sum()**2) < 1e-5
    assert abs(o.strehl_se - o.strehl_se_full) < 1e-7     # the peak is inside the window
    m = s.spupil > 0
    assert abs(o.phase_var - o.tar_phase[m].astype(np.float64).var()) < 1e-8
    # zero phase: Strehl 1
    o.reset_strehl()
    o.tar_phase[:] = 0
    assert abs(o.comp_strehl()[0] - 1.0) < 1e-5


def test_control_law_and_delay(small):
    _, s, cal = small
    o = helpers.CalSim(s)
    rng = np.random.default_rng(4)
    o.slopes[:] = rng.normal(size=s.nslope)
    o.do_control()
    e = -(cal.cmat.astype(np.float64) @ o.slopes)
    assert np.abs(o.err - e).max() < 1e-4 * np.abs(e).max()
    assert np.abs(o.com - s.gain * e).max() < 1e-4 * np.abs(e).max()
    assert aoref.delay_weights(0.0) == (1.0, 0.0, 0.0)
    assert aoref.delay_weights(1.0) == (0.0, 1.0, 0.0)
    assert aoref.delay_weights(2.0) == (0.0, 0.0, 1.0)
    c_first = o.com.copy()
    o.apply_control()                       # delay 1: the DM still sees the previous command (0)
    assert np.all(o.voltage == 0)
    o.apply_control()
    assert np.array_equal(o.voltage, c_first)
    with pytest.raises(ValueError):
        o.set_com(np.zeros(s.nactu + 1))


@pytest.mark.slow
def test_closed_loop_statistics_match_recorded_compass(small):
    """Distributional parity with real COMPASS: per-slope std of the g=0.7, d=1
    integrator loop vs the reference's recorded normalisation statistics."""
    from ao_marl_amd.env import load_norm
    _, s, cal = small
    norm, zn = load_norm("production_sh_10x10_2m")
    sl, dm, res = [], [], []
    for seed in (1, 2):
        o = aoref.OracleSim(s, seed=seed)
        for it in range(450):
            o.next_part_two(None)
            o.next_part_one()
            if it >= 50:
                sl.append(o.slopes.copy())
                dm.append(cal.volts2modes @ o.com)
                res.append(cal.volts2modes @ o.err)
    sl, dm, res = np.array(sl), np.array(dm), np.array(res)
    r_wfs = np.median(sl.std(axis=0) / norm["wfs"]["std"])
    assert 0.9 < r_wfs < 1.1, r_wfs
    assert abs(sl.std(axis=0).mean() / norm["wfs"]["std"].mean() - 1) < 0.1
    r_res = np.median(res.std(axis=0) / norm["dm_residual"]["std"])
    assert 0.8 < r_res < 1.25, r_res
    assert o.strehl_le > 0.7


def test_sinc_fit_of_the_psf_peak():
    """comp_strehl(do_fit=True) as restated in oracle/aoref.c (COMPASS's kernel is absent: unpinned): samples of an
    exact  A sinc(w (x - x0))  at x = -1, 0, 1 give back A; a symmetric triple gives its own maximum; a maximum on the
    border of the image is left alone; the separable 2-D form multiplies the two gains."""
    from oracle import aoref
    L = aoref.lib()
    sinc = lambda t: np.sinc(t / np.pi)            # noqa: E731  sin(t) / t
    for A, w, x0 in ((3.0, 0.9, 0.3), (1.0, 1.4, -0.45), (7.5, 0.5, 0.0), (2.0, 2.0, 0.2)):
        ym, y0, yp = (A * sinc(w * (x - x0)) for x in (-1.0, 0.0, 1.0))
        g = L.aoref_sinc_gain(np.float32(ym), np.float32(y0), np.float32(yp))
        assert abs(g * y0 - A) < 2e-5 * A, (A, w, x0, g * y0)
    assert L.aoref_sinc_gain(0.8, 1.0, 0.8) == pytest.approx(1.0, abs=1e-6)
    assert L.aoref_sinc_gain(1.0, 1.0, 1.0) == 1.0 and L.aoref_sinc_gain(0.5, 0.0, 0.5) == 1.0
    # 2-D: a separable sinc x sinc core
    yy, xx = np.mgrid[0:16, 0:16].astype(np.float64)
    img = (5.0 * sinc(0.8 * (xx - 8.3)) * sinc(1.1 * (yy - 7.6))).astype(np.float32)
    assert abs(L.aoref_fit_max_2x1d_sinc(img.reshape(-1), 16, 16) - 5.0) < 1e-4 * 5.0
    edge = np.zeros((16, 16), dtype=np.float32)
    edge[0, 5] = 2.0
    assert L.aoref_fit_max_2x1d_sinc(edge.reshape(-1), 16, 16) == 2.0
