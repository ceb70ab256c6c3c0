"""Init-time calibration of the control path: geometric imat -> actuator filtering -> diffractive
imat -> influence-function matrix -> Btt modal basis -> Btt-filtered command matrix.

Host NumPy / SciPy, like the reference (these steps are NumPy there too); the only device work is
"push these commands, give me the slopes", delegated to a backend with
    backend.dm_response(commands[K, nactu], geometric: bool) -> slopes[K, nslope]
    backend.reload_dms()          # after the stack-array mirror lost actuators
In the product the backend is the HIP simulator (ao_marl_amd/sim.py).

Reference results each step must reproduce:
  imat_geom            shesha/ao/imats.py:54-112
  correct_dm           shesha/init/dm_init.py:817-889   (-> 88 / 1284 pzt actuators)
  do_imat              shesha/ao/imats.py:115-167 (push-pull through the full WFS model)
  compute_IFsparse     shesha/ao/basis.py:126-200
  compute_btt          shesha/ao/basis.py:362-443
  compute_cmat_with_Btt shesha/ao/basis.py:229-256, called from rlSupervisor.py:184,215-234
"""
import numpy as np
import scipy.sparse as sp

from . import geometry as G
from . import system


def _push_matrix(s, dm_k, push):
    sl = system.dm_command_slices(s)[dm_k]
    n = sl[1] - sl[0]
    cmds = np.zeros((n, s.nactu), dtype=np.float32)
    cmds[np.arange(n), sl[0] + np.arange(n)] = push
    return cmds


def imat_geom(s, backend):
    """Geometric interaction matrix [nslope, nactu] (slopes per unit command)."""
    cols = []
    for k, d in enumerate(s.dms):
        r = backend.dm_response(_push_matrix(s, k, d.push4imat), geometric=True)
        cols.append(r.T / np.float32(d.push4imat))
    return np.concatenate(cols, axis=1).astype(np.float32)


def correct_dm(s, sysm, imat, backend):
    """Drop stack-array actuators whose geometric response is <= thresh * max (dm_init.py:857-859)."""
    resp = np.sqrt(np.sum(imat.astype(np.float64)**2, axis=0))
    kept = []
    for k, (a, b) in enumerate(system.dm_command_slices(s)):
        d = s.dms[k]
        if d.type == "pzt":
            tmp = resp[a:b]
            ok = np.where(tmp > d.thresh * np.max(tmp))[0]
            G.pzt_select(d, sysm.geom, ok)
            kept.append(ok)
        else:
            kept.append(np.arange(b - a))
    system.refresh_dms(s)
    backend.reload_dms()
    return kept


def imat_diffractive(s, backend):
    """Push-pull interaction matrix through the full image-formation + COG chain."""
    cols = []
    for k, d in enumerate(s.dms):
        p = _push_matrix(s, k, d.push4imat)
        plus = backend.dm_response(p, geometric=False)
        minus = backend.dm_response(-p, geometric=False)
        cols.append((plus - minus).T / np.float32(2 * d.push4imat))
    return np.concatenate(cols, axis=1).astype(np.float32)


def dm_pupil_window(sysm, d):
    """Pupil mask on the DM support (basis.py:143-147): centred crop of the big pupil."""
    ip = sysm.geom.ipupil
    dm_dim = d.n2 - d.n1 + 1
    t = (ip.shape[0] - dm_dim) // 2
    return ip[t:ip.shape[0] - t, t:ip.shape[1] - t]


def influence_matrix(s, sysm):
    """IF [Npup, nactu] (sparse CSC): phase of a unit command of each actuator on the pupil."""
    blocks = []
    for d in s.dms:
        dm_dim = d.n2 - d.n1 + 1
        pup = dm_pupil_window(sysm, d) > 0
        # DM arrays live in a dim x dim frame (dim >= dm_dim); production: dim == dm_dim
        if d.dim != dm_dim:
            raise NotImplementedError("DM support smaller than mpupil")
        idx = -np.ones(pup.size, dtype=np.int64)
        idx[pup.ravel()] = np.arange(int(pup.sum()))
        npts = int(pup.sum())
        if d.type == "pzt":
            ss = d.influsize
            k = np.arange(ss)
            rows, cols, vals = [], [], []
            f = d.influ  # [a (x offset), b (y offset), act]
            for act in range(d.ntotact):
                xs = d.i1[act] + k          # x pixel of offset a
                ys = d.j1[act] + k
                okx = (xs >= 0) & (xs < dm_dim)
                oky = (ys >= 0) & (ys < dm_dim)
                X, Y = np.meshgrid(xs[okx], ys[oky])       # [y, x]
                p = idx[(X + dm_dim * Y).ravel()]
                v = f[:, :, act][np.ix_(okx, oky)].T.ravel()  # -> [b(y), a(x)]
                m = (p >= 0) & (v != 0)
                rows.append(p[m])
                cols.append(np.full(int(m.sum()), act))
                vals.append(v[m])
            blocks.append(sp.csc_matrix((np.concatenate(vals).astype(np.float64),
                                         (np.concatenate(rows), np.concatenate(cols))),
                                        shape=(npts, d.ntotact)))
        else:
            pl = d.influ.reshape(-1, 2)[pup.ravel()]
            blocks.append(sp.csc_matrix(pl.astype(np.float64)))
    return sp.hstack(blocks, format="csc")


def compute_btt(IFpzt, IFtt):
    """Btt (volts x modes) and P (modes x volts): piston/tip/tilt-free orthonormal (w.r.t. the
    geometric covariance) basis of the stack-array mirror, plus normalised tip & tilt of the TT
    mirror as the last two modes (basis.py:362-443)."""
    N, n = IFpzt.shape
    if n > N:
        raise ValueError("IF matrix is transposed: expected pupil pixels down the rows, one column per "
                         "actuator (got %d x %d)" % (N, n))
    delta = (IFpzt.T @ IFpzt).toarray() / N
    Tp = np.ones((N, 3))
    Tp[:, :2] = IFtt
    deltaT = IFpzt.T @ Tp / N
    tau = np.linalg.solve(delta, deltaT)
    tdt = tau.T @ delta @ tau
    Gm = np.identity(n) - tau @ np.linalg.solve(tdt, tau.T @ delta)
    gdg = Gm.T @ delta @ Gm
    U, sv, _ = np.linalg.svd(gdg)
    U, sv = U[:, :n - 3], sv[:n - 3]
    B = (Gm @ U) / np.sqrt(sv)[None, :]
    TT = IFtt.T @ IFtt / N
    Btt = np.zeros((n + 2, n - 1))
    Btt[:n, :n - 3] = B
    Btt[n, n - 3] = 1. / np.sqrt(np.abs(TT[0, 0]))
    Btt[n + 1, n - 2] = 1. / np.sqrt(np.abs(TT[1, 1]))
    Delta = np.zeros((n + 2, n + 2))
    Delta[:n, :n] = delta
    Delta[n:, n:] = TT
    P = Btt.T @ Delta
    return Btt.astype(np.float32), P.astype(np.float32)


def _btt_filtered(Btt, nfilt):
    nm = Btt.shape[1]
    Bf = np.zeros((Btt.shape[0], nm - nfilt))
    Bf[:, :nm - nfilt - 2] = Btt[:, :nm - (nfilt + 2)]
    Bf[:, nm - nfilt - 2:] = Btt[:, nm - 2:]
    return Bf


def projector_wfs2modes(D, Btt, nfilt):
    """Least-squares projector from slopes onto the kept Btt modes (first nm - nfilt - 2, then tip-tilt):
    [nm - nfilt, nslope], float64 (helper_functions/utils/utils_projectors.py:3-21)."""
    Dm = D.astype(np.float64) @ _btt_filtered(Btt, nfilt)
    return np.linalg.solve(Dm.T @ Dm, Dm.T)


def cmat_with_btt(D, Btt, nfilt):
    """Command matrix filtering the `nfilt` highest-order Btt modes, TT kept
    (basis.py:229-256)."""
    return (_btt_filtered(Btt, nfilt) @ projector_wfs2modes(D, Btt, nfilt)).astype(np.float32)


def geo_projector(IF):
    """Projection matrix of the geometric ("GEO") controller, COMPASS's sutra_controller_geo as the
    reference configures it (rtc_init.py:418-448: influence functions of the controller's DMs on
    the pupil pixels; rlSupervisor.py:989-1013: do_control(sources=target)).  The native source is
    not in the reference tree; restated from the published algorithm: the pupil phase minus its
    mean is projected on the tip-tilt mirror first, the stack array then fits what is left,
    both by least squares (inverse of IF^T IF), and the command is minus the fit.

    IF [Npup, nactu] sparse, stack-array columns first, the 2 tip-tilt columns last.
    Returns W [nactu, nactu + 1] (float64) such that
        com = W . [ IFp^T phi | TT^T phi | sum(phi) ]          (phi on the lit pupil pixels)."""
    IF = sp.csc_matrix(IF, dtype=np.float64)
    npix, na = IF.shape
    npz = na - 2
    P, T = IF[:, :npz], IF[:, npz:].toarray()
    one = np.ones(npix)
    Gp = (P.T @ P).toarray()
    Gpi = np.linalg.inv(Gp)
    Gti = np.linalg.inv(T.T @ T)
    bp1, bpT, t1 = P.T @ one, P.T @ T, T.T @ one
    # r = [rP (npz) | rT (2) | r1 (1)]
    A_tt = np.zeros((2, na + 1))
    A_tt[:, npz:npz + 2] = Gti
    A_tt[:, na] = -Gti @ t1 / npix
    A_p = np.zeros((npz, na + 1))
    A_p[:, :npz] = Gpi
    A_p[:, na] = -Gpi @ bp1 / npix
    A_p -= Gpi @ bpT @ A_tt
    return -np.vstack([A_p, A_tt])


def geo_command(IF, phi_lit):
    """The same projection written out step by step (used by the oracle and by the tests)."""
    IF = sp.csc_matrix(IF, dtype=np.float64)
    na = IF.shape[1]
    P, T = IF[:, :na - 2], IF[:, na - 2:].toarray()
    d = np.asarray(phi_lit, dtype=np.float64)
    d = d - d.mean()
    ctt = np.linalg.solve(T.T @ T, T.T @ d)
    d = d - T @ ctt
    cp = np.linalg.solve((P.T @ P).toarray(), P.T @ d)
    return -np.concatenate([cp, ctt])


class Calibration(object):
    pass


# ------------------------------------------------------------------------------------ calibration cache
# One calibration of the 40x40 system is 1430 geometric + 2 x 1286 diffractive pushes, a 1284^2 solve + SVD and a
# 2400 x 1284 least-squares fit: seconds, repeated by every VecRlSupervisor of a process (a test session builds
# dozens) and by every rank of a multi-GPU job.  The result is a pure function of the system's geometry, the number
# of filtered modes and the backend's arithmetic, so it is memoised -- in the process, and on disk under a lock (of N
# ranks started together one calibrates, the others load its file).  A hit replays what calibrate() does to `s` /
# `sysm` (correct_dm's actuator selection, s.cmat) and returns copies of the same arrays, bit for bit.
#   AOMARL_CALIB_CACHE=0       no cache at all          AOMARL_CALIB_CACHE=mem   process only
#   AOMARL_CALIB_CACHE=<dir>   the directory            (default: <tempdir>/ao_marl_amd_calib_<uid>)
_CAL_MEMO = {}
_CAL_VERSION = "1"
cache_stats = {"hit_mem": 0, "hit_disk": 0, "miss": 0, "seconds": 0.0}


def _hash_obj(h, name, v, depth=0):
    import numbers
    h.update(name.encode())
    if isinstance(v, np.ndarray):
        h.update(str((v.dtype.str, v.shape)).encode())
        h.update(np.ascontiguousarray(v).tobytes())
    elif isinstance(v, (str, bytes, numbers.Number, type(None), np.generic)):
        h.update(repr(v).encode())
    elif isinstance(v, (list, tuple)):
        for i, x in enumerate(v):
            _hash_obj(h, "%s[%d]" % (name, i), x, depth + 1)
    elif isinstance(v, dict):
        for k in sorted(v):
            _hash_obj(h, "%s.%s" % (name, k), v[k], depth + 1)
    elif hasattr(v, "__dict__") and depth < 3:
        for k in sorted(vars(v)):
            _hash_obj(h, "%s.%s" % (name, k), getattr(v, k), depth + 1)
    else:
        h.update(repr(type(v)).encode())


_CODE_FP = None


def _code_fingerprint():
    """sha256 of the Python that PRODUCES a calibration -- this module, geometry.py, system.py -- and of the NumPy /
    SciPy versions behind its linear algebra: an edit to compute_btt / cmat_with_btt / correct_dm or to the geometry
    invalidates every stored result by itself (nobody has to remember _CAL_VERSION)."""
    global _CODE_FP
    if _CODE_FP is None:
        import hashlib
        import os
        import scipy
        h = hashlib.sha256()
        here = os.path.dirname(os.path.abspath(__file__))
        for f in ("modal.py", "geometry.py", "system.py"):
            with open(os.path.join(here, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
        h.update(("numpy %s scipy %s" % (np.__version__, scipy.__version__)).encode())
        _CODE_FP = h.hexdigest()
    return _CODE_FP


def calibration_key(s, sysm, backend, nfilt, backend_id=None):
    """sha256 over everything calibrate() reads: pupils, the sensor's maps, every DM, the offsets of the DMs in the
    sensor's path, nfilt and the backend's identity (`backend_id` or `backend.calibration_id()`: None = not
    cacheable)."""
    import hashlib
    bid = backend_id
    if bid is None:
        bid = getattr(backend, "calibration_id", None)
        bid = bid() if callable(bid) else None
    if bid is None:
        return None
    h = hashlib.sha256()
    _hash_obj(h, "version", _CAL_VERSION)
    _hash_obj(h, "code", _code_fingerprint())
    _hash_obj(h, "backend", bid)
    _hash_obj(h, "nfilt", int(nfilt))
    for k in ("n", "pupdiam", "mpupil", "spupil", "nvalid", "pdiam", "nfft", "npix", "nrebin", "nxsub", "phasemap", "halfxy",
              "binmap", "flux", "nphot", "wfs_lambda", "validsubsx", "validsubsy", "subapd", "cog_offset", "cog_scale",
              "nslope", "wfs_dm_off", "nactu"):
        _hash_obj(h, k, getattr(s, k))
    for i, d in enumerate(s.dms):
        _hash_obj(h, "dm%d" % i, d)
    _hash_obj(h, "ipupil", sysm.geom.ipupil)
    return h.hexdigest()


def _cache_dir():
    import os
    import tempfile
    v = os.environ.get("AOMARL_CALIB_CACHE", "")
    if v in ("0", "mem"):
        return None
    if v:
        return v
    uid = os.getuid() if hasattr(os, "getuid") else 0
    return os.path.join(tempfile.gettempdir(), "ao_marl_amd_calib_%d" % uid)


def _own_private_dir(cdir):
    """Create `cdir` (0700) if need be and check that it is what it should be: a real directory (no symbolic link)
    that belongs to this user and that nobody else can write to.  The default lives in the shared temporary
    directory under a predictable name: somebody else's directory there (or a link to one) could hold planted
    matrices, so it is not used -- the caller falls back to the in-process memo."""
    import os
    import stat
    try:
        os.makedirs(cdir, mode=0o700, exist_ok=True)
        st = os.lstat(cdir)
    except OSError:
        return False
    if not stat.S_ISDIR(st.st_mode) or stat.S_ISLNK(st.st_mode):
        return False
    if hasattr(os, "getuid") and st.st_uid != os.getuid():
        return False
    return (st.st_mode & 0o022) == 0


def _fits(c, s):
    """A loaded calibration has the shapes of the system it is replayed into (a file of another system under this
    key -- a damaged or planted one -- is recalibrated over, not trusted)."""
    try:
        nact = int(sum(len(k) for k in c.kept))         # (one entry per DM: the kept stack-array actuators, both tip-tilt axes)
        ok_dm = len(c.kept) == len(s.dms) and all(
            (len(k) == 2 if d.type == "tt" else (k.size > 0 and int(k.min()) >= 0))
            for k, d in zip(c.kept, s.dms))
        return (ok_dm and c.cmat.ndim == 2 and c.cmat.shape == (nact, s.nslope) and
                c.imat.shape == (s.nslope, nact) and c.Btt.shape[0] == nact and c.P.shape == c.Btt.shape[::-1] and
                c.IF.shape[1] == nact and all(np.all(np.isfinite(getattr(c, k))) for k in ("cmat", "Btt", "P")))
    except Exception:
        return False


_CAL_FIELDS = ("imat_geom", "imat", "Btt", "P", "cmat")


def _pack(c):
    d = {k: getattr(c, k) for k in _CAL_FIELDS}
    IF = sp.csc_matrix(c.IF)
    d.update(IF_data=IF.data, IF_indices=IF.indices, IF_indptr=IF.indptr, IF_shape=np.asarray(IF.shape, dtype=np.int64))
    d["nkept"] = np.asarray(len(c.kept), dtype=np.int64)
    for i, k in enumerate(c.kept):
        d["kept%d" % i] = np.asarray(k, dtype=np.int64)
    return d


def _unpack(d):
    c = Calibration()
    for k in _CAL_FIELDS:
        setattr(c, k, np.array(d[k]))
    c.IF = sp.csc_matrix((np.array(d["IF_data"]), np.array(d["IF_indices"]), np.array(d["IF_indptr"])),
                         shape=tuple(int(x) for x in d["IF_shape"]))
    c.kept = [np.array(d["kept%d" % i]) for i in range(int(d["nkept"]))]
    c.modes2volts, c.volts2modes = c.Btt, c.P
    return c


def _replay(c, s, sysm, backend):
    """What calibrate() leaves behind in `s` / `sysm` / the backend: the stack-array mirrors reduced to their kept
    actuators (correct_dm) and the command matrix."""
    for k, d in enumerate(s.dms):
        if d.type == "pzt":
            G.pzt_select(d, sysm.geom, c.kept[k])
    system.refresh_dms(s)
    if backend is not None:
        backend.reload_dms()
    s.cmat = np.ascontiguousarray(c.cmat)
    return c


class _LazyBackend(object):
    """A backend built on first use: a cache hit never allocates the calibration simulator."""

    def __init__(self, factory):
        self.factory, self.obj = factory, None

    def get(self):
        if self.obj is None:
            self.obj = self.factory()
        return self.obj

    def dm_response(self, commands, geometric):
        return self.get().dm_response(commands, geometric)

    def reload_dms(self):
        if self.obj is not None:
            self.obj.reload_dms()


def calibrate(s, sysm, backend, nfilt=0, verbose=False, cache=True, backend_id=None):
    """Full init sequence of the controller path; fills s.cmat and returns a Calibration with
    imat_geom, kept actuators, imat, Btt (= modes2volts), P (= volts2modes), cmat.  `backend`: an object with
    dm_response / reload_dms, or a zero-argument factory of one (built only when the calibration really runs).
    Memoised (see above) when the backend's arithmetic is named (`backend_id`, or `backend.calibration_id()`)."""
    import os
    import time
    t0 = time.perf_counter()
    if not hasattr(backend, "dm_response"):
        backend = _LazyBackend(backend)
    key = None
    if cache and os.environ.get("AOMARL_CALIB_CACHE", "") != "0":
        key = calibration_key(s, sysm, backend, nfilt, backend_id)
    try:
        if key is None:
            return _calibrate_now(s, sysm, backend, nfilt, verbose)
        if key in _CAL_MEMO:
            cache_stats["hit_mem"] += 1
            return _replay(_unpack(_CAL_MEMO[key]), s, sysm, backend)
        cdir = _cache_dir()
        if cdir is not None and not _own_private_dir(cdir):
            import warnings
            warnings.warn("calibration cache: %r is not a private directory of this user (a link, another owner, or "
                          "writable by others): not used, this process keeps its calibrations in memory" % cdir)
            cdir = None
        if cdir is None:
            c = _calibrate_now(s, sysm, backend, nfilt, verbose)
            cache_stats["miss"] += 1
            _CAL_MEMO[key] = _pack(c)
            return c
        path = os.path.join(cdir, key + ".npz")     # (arrays only: np.load refuses pickles; the directory is the user's own: checked)
        lock = open(os.path.join(cdir, key + ".lock"), "w")
        try:
            try:
                import fcntl
                fcntl.flock(lock, fcntl.LOCK_EX)        # of N ranks one calibrates, the others wait here and load
            except ImportError:                         # pragma: no cover
                pass
            if os.path.exists(path):
                try:
                    with np.load(path) as z:
                        d = {k: z[k] for k in z.files}
                    c = _unpack(d)
                    if not _fits(c, s):
                        raise ValueError("stored calibration does not fit this system")
                    _CAL_MEMO[key] = d
                    cache_stats["hit_disk"] += 1
                    return _replay(c, s, sysm, backend)
                except Exception:                       # a damaged file: calibrate and replace it
                    pass
            c = _calibrate_now(s, sysm, backend, nfilt, verbose)
            cache_stats["miss"] += 1
            d = _pack(c)
            _CAL_MEMO[key] = d
            tmp = path + ".tmp%d" % os.getpid()
            try:
                with open(tmp, "wb") as f:
                    np.savez(f, **d)
                os.replace(tmp, path)
            except OSError:                             # a read-only cache directory is not an error
                pass
            return c
        finally:
            lock.close()
    finally:
        cache_stats["seconds"] += time.perf_counter() - t0


def _calibrate_now(s, sysm, backend, nfilt=0, verbose=False):
    c = Calibration()
    c.imat_geom = imat_geom(s, backend)
    c.kept = correct_dm(s, sysm, c.imat_geom, backend)
    if verbose:
        print("correct_dm: kept", [len(k) for k in c.kept])
    c.imat = imat_diffractive(s, backend)
    IF = influence_matrix(s, sysm)
    ntt = 2
    c.Btt, c.P = compute_btt(IF[:, :-ntt].tocsc(), IF[:, -ntt:].toarray())
    c.cmat = cmat_with_btt(c.imat, c.Btt, max(nfilt, 0))
    s.cmat = np.ascontiguousarray(c.cmat)
    c.modes2volts, c.volts2modes = c.Btt, c.P
    c.IF = IF
    return c
