"""The centre of gravity of a noise-free Shack-Hartmann spot as a quadratic form of the pupil field
(csrc/aomarl_kernels.hip: spot_cog_qf) against the definition: zero-padded 64 x 64 FFT of the half-pixel-shifted
field, |.|^2, binmap, moments (geom_init.py:689-758, the oracle's aoref_sh_image + aoref_cog), in float64 and
with the products rounded to float32.  CPU only:  python tools/qf_cog_check.py
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ao_marl_amd import params, geometry  # noqa: E402


def kernels(N=64, nf=16):
    j = np.arange(nf)
    f = j + 0.5
    u = 0.5 + (j >> 1)
    d = np.arange(16)[:, None] - np.arange(16)[None, :]          # x' - x
    ang = 2 * np.pi * f[None, None, :] * d[:, :, None] / N
    return 2 * np.cos(ang).sum(-1), 2 * (u * np.sin(ang)).sum(-1)


def cog_definition(w, phase_rev, amp):
    N, pd, npix = w.Nfft, w.pdiam, w.npix
    E = amp * np.exp(1j * (2 * np.pi * phase_rev - np.asarray(w.halfxy, np.float64)))
    buf = np.zeros((N, N), complex)
    buf[:pd, :pd] = E
    hr = np.abs(np.fft.fft2(buf)) ** 2
    img = hr.ravel()[np.asarray(w.binmap)].sum(0).reshape(npix, npix)
    X = np.arange(npix)
    s = img.sum()
    return (img.sum(0) @ X) / s, (img.sum(1) @ X) / s, s


def cog_quadratic_form(M, S, phase_rev, amp, dt):
    E = amp * np.exp(2j * np.pi * phase_rev)
    Er, Ei, M, S = E.real.astype(dt), E.imag.astype(dt), M.astype(dt), S.astype(dt)
    Wr, Wi, V = M @ Er.T, M @ Ei.T, S @ Er.T
    G1, G2, G3 = Er @ Wr + Ei @ Wi, Ei @ Wr, Ei @ V
    s0 = (M * G1).sum(dtype=dt)
    return 7.5 + 2 * (M * G3).sum(dtype=dt) / s0, 7.5 + 2 * (S * G2).sum(dtype=dt) / s0, s0


def main():
    p = params.builtin("production_sh_40x40_8m_3layers")
    w = geometry.build_system(p).wfss[0]
    assert (w.Nfft, w.pdiam, w.npix, w.nrebin) == (64, 16, 16, 2)
    M, S = kernels()
    rng = np.random.default_rng(1)
    worst64 = worst32 = 0.0
    for trial in range(200):
        amp = (rng.random((16, 16)) > (0.0 if trial % 2 else 0.2)).astype(float)
        tilt = np.add.outer(np.arange(16) * rng.normal() * 0.08, np.arange(16) * rng.normal() * 0.08)
        ph = rng.normal(size=(16, 16)) * rng.uniform(0, 0.5) + tilt
        a = cog_definition(w, ph, amp)
        b = cog_quadratic_form(M, S, ph, amp, np.float64)
        c = cog_quadratic_form(M, S, ph, amp, np.float32)
        worst64 = max(worst64, abs(a[0] - b[0]), abs(a[1] - b[1]))
        worst32 = max(worst32, abs(a[0] - c[0]), abs(a[1] - c[1]))
    print("200 random sub-apertures (half of them partly masked), pixels of %.4f arcsec:" % w.pixsize)
    print("  quadratic form in float64 vs FFT definition: max |d cog| = %.3g pixels" % worst64)
    print("  quadratic form in float32 vs FFT definition: max |d cog| = %.3g pixels" % worst32)
    assert worst64 < 1e-6 and worst32 < 2e-5


if __name__ == "__main__":
    main()
