"""modal.calibrate's memo (process + disk): a hit leaves `s` / `sysm` and the returned arrays exactly as a fresh
calibration does (CPU, oracle backend)."""
import numpy as np

import helpers
from ao_marl_amd import modal


class _NamedBackend(helpers.OracleBackend):
    def calibration_id(self):
        return ("oracle-test-backend",)


def _run(built):
    sysm, s = helpers.uncalibrated()

    def factory():
        built.append(1)
        return _NamedBackend(s)
    return modal.calibrate(s, sysm, factory, nfilt=5, backend_id=("oracle-test-backend",)), s


def test_memo_and_disk_cache_reproduce_a_fresh_calibration(tmp_path, monkeypatch):
    monkeypatch.setenv("AOMARL_CALIB_CACHE", str(tmp_path))
    monkeypatch.setattr(modal, "_CAL_MEMO", {})
    stats = dict(modal.cache_stats)
    built = []
    c1, s1 = _run(built)                  # miss: calibrates, writes the file
    assert len(built) == 1 and len(list(tmp_path.glob("*.npz"))) == 1
    c2, s2 = _run(built)                  # process memo
    modal._CAL_MEMO.clear()
    c3, s3 = _run(built)                  # disk
    assert len(built) == 1, "a cache hit must not build the calibration simulator"
    assert modal.cache_stats["miss"] == stats["miss"] + 1
    assert modal.cache_stats["hit_mem"] == stats["hit_mem"] + 1 and modal.cache_stats["hit_disk"] == stats["hit_disk"] + 1
    for c, s in ((c2, s2), (c3, s3)):
        for k in ("imat_geom", "imat", "Btt", "P", "cmat"):
            assert np.array_equal(getattr(c1, k), getattr(c, k)), k
        assert (c1.IF != c.IF).nnz == 0
        assert s.nactu == s1.nactu == 90 and np.array_equal(s.cmat, s1.cmat)
        assert all(np.array_equal(a, b) for a, b in zip(c1.kept, c.kept))
        for d1, d in zip(s1.dms, s.dms):
            for k, v in vars(d1).items():
                if isinstance(v, np.ndarray):
                    assert np.array_equal(v, getattr(d, k)), k
    c2.cmat[:] = 0                        # a caller may scribble on what it got
    c4, _ = _run(built)
    assert np.array_equal(c4.cmat, c1.cmat)


def test_key_sees_geometry_nfilt_and_backend(monkeypatch):
    sysm, s = helpers.uncalibrated()
    k0 = modal.calibration_key(s, sysm, None, 5, ("a",))
    assert k0 == modal.calibration_key(s, sysm, None, 5, ("a",))
    assert k0 != modal.calibration_key(s, sysm, None, 4, ("a",))
    assert k0 != modal.calibration_key(s, sysm, None, 5, ("b",))
    assert modal.calibration_key(s, sysm, None, 5, None) is None          # unnamed arithmetic: never cached
    s.dms[0].i1 = s.dms[0].i1 + 1
    assert k0 != modal.calibration_key(s, sysm, None, 5, ("a",))


def test_cache_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("AOMARL_CALIB_CACHE", "0")
    monkeypatch.setattr(modal, "_CAL_MEMO", {})
    built = []
    _run(built)
    _run(built)
    assert len(built) == 2 and not modal._CAL_MEMO


def test_the_key_follows_the_code_and_a_foreign_directory_is_not_trusted(tmp_path, monkeypatch):
    """The key holds a hash of the modules that produce a calibration (an edit invalidates stored results by itself);
    a cache directory that is a symbolic link or writable by others is not used; a stored file whose shapes do not fit
    the system is calibrated over."""
    import os
    from ao_marl_amd import modal
    from tests import helpers
    sysm, s = helpers.uncalibrated()
    k0 = modal.calibration_key(s, sysm, None, 5, backend_id="x")
    monkeypatch.setattr(modal, "_CODE_FP", "another edit of modal.py")
    assert modal.calibration_key(s, sysm, None, 5, backend_id="x") != k0
    monkeypatch.setattr(modal, "_CODE_FP", None)
    assert modal.calibration_key(s, sysm, None, 5, backend_id="x") == k0
    real = tmp_path / "real"
    real.mkdir(mode=0o700)
    link = tmp_path / "link"
    os.symlink(real, link)
    assert modal._own_private_dir(str(real)) and not modal._own_private_dir(str(link))
    loose = tmp_path / "loose"
    loose.mkdir()
    os.chmod(loose, 0o777)
    assert not modal._own_private_dir(str(loose))
    # a planted / foreign file under the right name: shapes of another system -> not replayed
    _, s2, cal = helpers.calibrated()
    bad = modal._unpack(modal._pack(cal))
    bad.cmat = bad.cmat[:-1]
    assert modal._fits(modal._unpack(modal._pack(cal)), helpers.uncalibrated()[1]) and not modal._fits(bad, helpers.uncalibrated()[1])
