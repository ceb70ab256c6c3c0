"""The identity the slopes-only frame kernel rests on (csrc/aomarl_kernels.hip: spot_cog_qf): the centre of gravity of
a noise-free, fully binned Shack-Hartmann spot is a ratio of quadratic forms of the pupil field, so the spot itself
never has to be formed.  Checked here on the CPU against the definition the oracle restates (zero-padded FFT of the
half-pixel-shifted field, |.|^2, binmap, moments: geom_init.py:689-758, wfsCompass.py:334-343) with the reference's own
geometry arrays; the GPU tests compare the kernel's slopes with the oracle's."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import qf_cog_check as qf  # noqa: E402


def _wfs(name):
    from ao_marl_amd import params, geometry
    return geometry.build_system(params.builtin(name)).wfss[0]


def test_kernels_are_toeplitz_symmetric_and_antisymmetric():
    M, S = qf.kernels()
    assert np.allclose(M, M.T) and np.allclose(S, -S.T)
    d = np.arange(16)[:, None] - np.arange(16)[None, :]
    for k in range(-15, 16):                                   # Toeplitz: one value per diagonal
        assert np.ptp(M[d == k]) < 1e-12 and np.ptp(S[d == k]) < 1e-12
    assert abs(M[0, 0] - 32.0) < 1e-12
    assert np.abs(M[(d % 2 == 0) & (d != 0)]).max() < 1e-12    # zeros of the Dirichlet kernel (cleared in the kernel)


def test_quadratic_form_equals_fft_centre_of_gravity():
    for name in ("production_sh_10x10_2m", "production_sh_40x40_8m_3layers"):
        w = _wfs(name)
        assert (w.Nfft, w.pdiam, w.npix, w.nrebin) == (64, 16, 16, 2)       # what the fast path assumes (and checks)
        M, S = qf.kernels()
        rng = np.random.default_rng(7)
        for trial in range(40):
            amp = (rng.random((16, 16)) > (0.0 if trial % 2 else 0.25)).astype(float)
            tilt = np.add.outer(np.arange(16) * rng.normal() * 0.08, np.arange(16) * rng.normal() * 0.08)
            ph = rng.normal(size=(16, 16)) * rng.uniform(0, 0.6) + tilt
            a = qf.cog_definition(w, ph, amp)
            b = qf.cog_quadratic_form(M, S, ph, amp, np.float64)
            c = qf.cog_quadratic_form(M, S, ph, amp, np.float32)
            assert abs(a[0] - b[0]) < 1e-6 and abs(a[1] - b[1]) < 1e-6 and abs(a[2] - b[2]) < 1e-6 * a[2]
            # fp32 products: 1e-5 pixels = 2.6e-6 arcsec, far inside the 1e-4 arcsec the slopes are held to
            assert abs(a[0] - c[0]) < 1e-5 and abs(a[1] - c[1]) < 1e-5
