"""Normalisation / action-bound generator (reference recipe:
preprocessing/normalization/obtain_normalization.py:139-243).

CPU: the device accumulation and the frame ordering against a plain NumPy restatement of the
reference loop over the oracle.  GPU: the recipe itself (seeds 1..20 x 1000 integrator frames) on the
HIP path, compared with the statistics the reference recorded from real COMPASS
(ao_marl_amd/data/norm_*.npz, imported by tools/import_norm_data.py) -- distributional parity, since
COMPASS's cuRAND streams cannot be reproduced."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from ao_marl_amd import normalization as N  # noqa: E402
from ao_marl_amd.env import load_norm  # noqa: E402


def test_normalization_loop_matches_numpy_recipe_on_the_oracle():
    from tests.oracle_vecsim import OracleVecSim
    from oracle import aoref
    from ao_marl_amd.env import VecRlSupervisor
    frames, seeds = 12, (1, 2)
    sup = VecRlSupervisor("production_sh_10x10_2m", dict(n_reverse_filtered_from_cmat=5), len(seeds),
                          initial_seed=1, seed_stride=1, device="cpu", sim_factory=OracleVecSim)
    norm, zn, sr = N.normalization_loop(sup, frames=frames)
    # the reference loop, one episode after the other (obtain_normalization.py:160-197)
    wfs, dm, res = [], [], []
    v2m = sup.volts2modes
    for sd in seeds:
        o = aoref.OracleSim(sup.s, seed=sd)
        for _ in range(frames):
            o.next_part_one()
            o.next_part_two(None)
            wfs.append(o.slopes.copy()); dm.append(v2m.dot(o.com)); res.append(v2m.dot(o.err))
    for key, lst in (("wfs", wfs), ("dm", dm), ("dm_residual", res)):
        a = np.asarray(lst, dtype=np.float64)
        scale = np.abs(a).max()
        assert np.abs(norm[key]["mean"] - a.mean(axis=0)).max() < 1e-5 * scale
        assert np.abs(norm[key]["std"] - a.std(axis=0)).max() < 1e-5 * scale
        assert np.abs(norm[key]["max"] - a.max(axis=0)).max() < 1e-5 * scale
        assert np.abs(norm[key]["min"] - a.min(axis=0)).max() < 1e-5 * scale
    d = np.asarray(dm)
    assert np.allclose(zn, (np.abs(d.max(axis=0)) + np.abs(d.min(axis=0))) / 2.0, rtol=1e-5, atol=1e-7)
    assert sr.shape == (2,)


def test_save_norm_round_trips_through_load_norm(tmp_path, monkeypatch):
    rng = np.random.default_rng(0)
    norm = {k: {st: rng.normal(size=7).astype(np.float32) for st in ("mean", "std", "max", "min")}
            for k in N.KEYS}
    zn = rng.random(7).astype(np.float32)
    N.save_norm(str(tmp_path / "norm_demo.npz"), norm, zn)
    import ao_marl_amd.env as E
    monkeypatch.setattr(E, "DATA_DIR", str(tmp_path))
    got, zn2 = load_norm("demo.py")
    assert np.array_equal(zn, zn2)
    for k in N.KEYS:
        for st in ("mean", "std", "max", "min"):
            assert np.array_equal(got[k][st], norm[k][st])


L40 = "production_sh_40x40_8m_3layers"
# every noise-free statistics file the reference holds with a parameter file whose settings are known:
# the base files, the oblique-wind files (15 / 30 degrees: x AND y extrusions on one layer, fractional
# accumulation on both axes -- the only pins of that part of move_atmos / raytrace), all layers along x,
# slower / faster winds (gain 0.6 in the _v_10_5_15 files), and the two files recorded on _same_dir with
# the integrator gain changed on the command line (0.9 / 0.2: ao_marl_amd/params.py)
COMPASS_RUNS = ["production_sh_10x10_2m", L40, L40 + "_dir_0_15_30", L40 + "_same_dir", L40 + "_v_20_15_25",
                L40 + "_v_10_5_15", L40 + "_dir_0_15_30_v_10_5_15", L40 + "_dir_0_15_30_v_20_15_25",
                L40 + "_same_dir_v_10_5_15", L40 + "_same_dir_v_20_15_25",
                L40 + "_same_dir_gain_change_high", L40 + "_same_dir_gain_change_low"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", COMPASS_RUNS)
def test_statistics_match_the_references_recorded_compass_runs(name):
    """SURVEY section 8c asks for the median ratio of the per-slope and per-mode standard deviations within +-10 % and
    the action bounds (extreme-value statistics) within +-20 %.  The build measures 0.990 .. 1.001 on the eleven 40x40
    files (1200 sub-apertures, 1283 modes: the medians are tight) and 1.000 .. 1.023 on the 10x10 file (64
    sub-apertures); asserted: +-3 % / +-5 % -- a 5 % error in the flow, the noise amplitude or the loop gain fails."""
    ref, zn_ref = load_norm(name)
    norm, zn, sr = N.obtain_normalization(name, modes_filtered=5, episodes=20, frames=1000)
    assert norm["wfs"]["std"].shape == ref["wfs"]["std"].shape
    assert norm["dm"]["std"].shape == ref["dm"]["std"].shape == zn.shape == zn_ref.shape
    r_wfs = np.median(norm["wfs"]["std"] / ref["wfs"]["std"])
    nm = zn.shape[0]
    live = np.arange(nm) < nm - 5 - 2                      # filtered modes carry no command
    live[-2:] = True
    r_dm = np.median(norm["dm"]["std"][live] / ref["dm"]["std"][live])
    r_res = np.median(norm["dm_residual"]["std"][live] / ref["dm_residual"]["std"][live])
    r_zn = np.median(zn[live] / zn_ref[live])
    print("%s: median std ratio  slopes %.3f  command modes %.3f  residual modes %.3f  zn_norm %.3f  "
          "LE Strehl %.3f" % (name, r_wfs, r_dm, r_res, r_zn, sr.mean()))
    tol = 0.03 if name.startswith(L40) else 0.05
    assert abs(r_wfs - 1) < tol, r_wfs
    assert abs(r_dm - 1) < tol, r_dm
    assert abs(r_res - 1) < tol, r_res
    assert abs(r_zn - 1) < 0.20
    assert np.abs(norm["wfs"]["mean"]).max() < 0.2 * norm["wfs"]["std"].max()
    # Not only the medians: a mode-number-dependent error (say the high orders 15 % off) would leave a median
    # alone.  5th / 95th percentile of the per-entry ratios within +-15 %, and the median of every third of the
    # spectrum (Btt modes are ordered by spatial frequency; slopes: x then y) within +-10 %.
    for key, sel in (("wfs", slice(None)), ("dm", live), ("dm_residual", live)):
        rs = ref[key]["std"][sel]
        ok = rs > 1e-6 * np.median(rs)          # (the 10x10 file holds 10 dead modes, 4e-9 .. 9e-9: no statistic to match)
        assert (~ok).sum() <= 10
        ratio = (norm[key]["std"][sel] / rs)[ok]
        p5, p95 = np.percentile(ratio, [5, 95])
        thirds = [float(np.median(t)) for t in np.array_split(ratio, 3)]
        print("    %-11s ratio: 5th %.3f  95th %.3f  medians of the thirds %s" %
              (key, p5, p95, " ".join("%.3f" % t for t in thirds)))
        assert 0.85 < p5 and p95 < 1.15, (key, p5, p95)
        assert all(abs(t - 1) < 0.10 for t in thirds), (key, thirds)


@pytest.mark.gpu
def test_noisy_configuration_against_the_references_d1_noise_statistics():
    """The only reference-held fixture of the noisy configuration family: the normalisation the reference
    recorded from real COMPASS under the name production_sh_40x40_8m_3layers_d1_noise (magnitude 9, 3 e-
    read-out noise; byte for byte the `_noise_M9` file, whose parameter file is not in the tree).  It is the
    independent check of the photon / read-out noise model and of the centroid rule (the oracle shares the
    kernel's, so only COMPASS's own numbers can vouch for them).  What round 3 established
    (profiles/r03_d1_noise_centroid_rules.txt, profiles/r03_d1_noise_seed_blocks.txt; tools/d1_noise_cog_rules.py):

      * SLOPES: 1.000 +- 0.004 of the recorded per-slope standard deviations on every one of 16 disjoint
        blocks of 20 seeds at the parameter file's gain 0.65 -- asserted (+-2 %).  Any other gain or delay of
        the family (0.3 / delay 0: 0.82; 0.3 / delay 1: 0.85) misses it: the recorded run WAS at 0.65, delay 1;
      * CENTROID RULE: the recorded run has heavy-tailed centroids -- (max - min) / std per slope: median 13.6,
        99th percentile 70 (a Gaussian gives 8) -- i.e. COMPASS divides by the measured flux whatever it is,
        like the plain centre of gravity here: this build gives 13.4 .. 13.7 and 46 .. 100 on the 16 blocks.
        A floor on the denominator, zeroing faint spots, clipping negative pixels or a threshold all destroy
        that tail (99th percentile 10 .. 19) and / or the slope statistics (0.24 .. 0.30).  Asserted: the
        median of that ratio within +-10 % of the recorded one;
      * RESIDUAL MODES (v2m . cmat . s): 0.99 .. 1.04 on the blocks without a mega-outlier, up to 1.5 with
        one -- asserted 0.95 .. 1.6;
      * COMMAND MODES / action bounds are NOT a reproducible statistic of this loop in any implementation:
        one centroid of a spot whose total flux came out near zero kicks the integrator of its environment by
        10^2 .. 10^4 sigma, and the pooled standard deviation over 20 x 1000 frames is whatever the few such
        events of the run make it: 0.97, 0.99, 1.29, 1.36, 1.47, 1.58, 2.2, 2.7, 2.7, 3.6, 3.8, 3.9, 4.0, 8.2,
        13.6, 32.9 of the recorded value on 16 blocks of seeds of ONE rule (and the same spread between two
        numerically equivalent forms of it on the SAME seeds).  The recorded run is one draw from that
        distribution, at its quiet end; what can be asserted is the quiet floor: never below 0.9.
    The denoiser variant (k_frame_wave<noise> -> denoiser -> k_cog) is run too and must show the noise
    reduction it exists for (the recorded run did not use it: with the shipped network the slopes' standard
    deviation is 0.23 of the recorded one)."""
    from ao_marl_amd.denoiser import SubapDenoiser
    from ao_marl_amd.env import VecRlSupervisor
    name = "production_sh_40x40_8m_3layers_d1_noise"
    ref, zn_ref = load_norm(name)
    nm = zn_ref.shape[0]
    live = np.arange(nm) < nm - 5 - 2
    live[-2:] = True
    tail_ref = float(np.median((ref["wfs"]["max"] - ref["wfs"]["min"]) / ref["wfs"]["std"]))
    assert 13.0 < tail_ref < 14.5

    def ratios(norm, zn, sr):
        return dict(wfs=float(np.median(norm["wfs"]["std"] / ref["wfs"]["std"])),
                    dm=float(np.median(norm["dm"]["std"][live] / ref["dm"]["std"][live])),
                    res=float(np.median(norm["dm_residual"]["std"][live] / ref["dm_residual"]["std"][live])),
                    zn=float(np.median(zn[live] / zn_ref[live])), sr=float(sr.mean()),
                    tail=float(np.median((norm["wfs"]["max"] - norm["wfs"]["min"]) / norm["wfs"]["std"])))

    def show(label, r):
        print("%s [%s]: median std ratio  slopes %.3f  command modes %.3f  residual modes %.3f  zn_norm %.3f  "
              "LE Strehl %.3f  slope (max - min) / std %.1f (recorded %.1f)" %
              (name, label, r["wfs"], r["dm"], r["res"], r["zn"], r["sr"], r["tail"], tail_ref))

    sup = VecRlSupervisor(name, dict(n_reverse_filtered_from_cmat=5), 20, initial_seed=1, seed_stride=1)
    assert abs(sup.gain - 0.65) < 1e-6 and sup.s.delay == 1.0 and sup.s.noise == 3.0
    blocks = []
    for b in range(2):
        sup.set_sim_seed(1 + 20 * b)
        r = ratios(*N.normalization_loop(sup, frames=1000))
        show("plain sensor, file gain 0.65, seeds %d..%d" % (1 + 20 * b, 20 + 20 * b), r)
        assert abs(r["wfs"] - 1) < 0.02, r                  # the noise model, at the file's gain
        assert abs(r["tail"] / tail_ref - 1) < 0.10, r      # the centroid rule: COMPASS's heavy tail
        assert r["res"] > 0.95, r
        if r["dm"] < 3.0:                                   # a block without a mega-outlier: the residual modes are pinned too
            assert r["res"] < 1.6, r
        assert r["dm"] > 0.9 and r["zn"] > 0.7, r           # the quiet floor; no upper bar (see the docstring)
        blocks.append(r)
    # less gain -> less propagated noise in the slopes: the recorded slopes pin the gain
    sup.set_sim_seed(1)
    sup.set_gain(0.3)
    low = ratios(*N.normalization_loop(sup, frames=1000))
    show("plain sensor, gain 0.30", low)
    assert low["wfs"] < 0.92 and low["sr"] > blocks[0]["sr"]
    del sup
    dn = SubapDenoiser.load(device="cuda:0")
    den = ratios(*N.obtain_normalization(name, modes_filtered=5, episodes=20, frames=1000, autoencoder=dn))
    dn.check_range()
    show("shipped denoiser, file gain 0.65", den)
    assert den["wfs"] < 0.5 and den["sr"] > blocks[0]["sr"] + 0.2      # what the denoiser is for
