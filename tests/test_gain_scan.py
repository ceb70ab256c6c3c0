"""Integrator gain / filtered-modes scan (reference recipe:
preprocessing/obtain_gain/obtain_best_gain_and_modes_filtered.py:41-175).

CPU: the batched scan (all gains as environments of one batch, per-environment gains) against the
reference's sequential loop restated over the oracle.  GPU: per-environment gains on the HIP path
against separate scalar-gain runs, and the scan end to end."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from ao_marl_amd import gain_scan as GS  # noqa: E402
from ao_marl_amd import modal  # noqa: E402

CFG = "production_sh_10x10_2m"


def _sequential_reference(sup_factory, gains, modes_list, episodes, steps):
    """obtain_modes_filtered_and_gain as the reference runs it: one system, one candidate and one
    episode after the other (performance_loop, :41-98)."""
    from oracle import aoref
    sup = sup_factory(1)
    s, cal = sup.s, sup.cal

    def run(gain, mf):
        s.cmat = np.ascontiguousarray(modal.cmat_with_btt(cal.imat, cal.Btt, mf))
        s.gain = float(gain)
        sr = None
        for ep in range(episodes):
            o = aoref.OracleSim(s, seed=ep + 1)
            for _ in range(steps):
                o.next_part_one()
                o.next_part_two(None)
            sr = o.get_strehl(do_fit=True)[1]          # (the reference reads get_strehl(0)[1]: do_fit defaults to True)
        return sr                                  # the last episode's long-exposure Strehl
    sr_modes = [run(0.5, mf) for mf in modes_list]
    best_mf = modes_list[int(np.argmax(sr_modes))]
    sr_gains = [run(g, best_mf) for g in gains]
    return sr_modes, best_mf, sr_gains


def test_batched_scan_equals_the_sequential_recipe_on_the_oracle():
    from tests.oracle_vecsim import OracleVecSim
    from ao_marl_amd.env import VecRlSupervisor
    gains, modes, episodes, steps = [0.2, 0.5, 0.8], (0, 5), 2, 6

    def factory(n):
        return VecRlSupervisor(CFG, dict(n_reverse_filtered_from_cmat=0), n, initial_seed=1,
                               seed_stride=1, device="cpu", sim_factory=OracleVecSim)
    res = GS.obtain_modes_filtered_and_gain(CFG, gains, modes_filtered_list=modes,
                                            num_episodes=episodes, num_steps=steps, device="cpu",
                                            sim_factory=OracleVecSim)
    sr_modes, best_mf, sr_gains = _sequential_reference(factory, gains, list(modes), episodes, steps)
    assert res["best_modes_discarded"] == best_mf
    assert np.allclose(res["sr_le_modes"], sr_modes, rtol=1e-5, atol=1e-7)
    assert np.allclose(res["sr_le_gains"], sr_gains, rtol=1e-5, atol=1e-7)
    assert res["best_gain"] == pytest.approx(gains[int(np.argmax(sr_gains))])
    assert res["sr_le_gains_all"].shape == (3, 2) and res["sr_le_modes_all"].shape == (2, 2)


def test_csv_has_the_references_rows(tmp_path):
    res = dict(modes_discared=[0, 5, 10], sr_le_modes=[0.1, 0.2, 0.15], best_modes_discarded=5,
               gains=[0.3, 0.5], sr_le_gains=[0.2, 0.25], best_gain=0.5, sr_le_best_gain=0.25)
    path = tmp_path / "information_best_gain.csv"
    GS.save_csv(str(path), "production_sh_10x10_2m", res)
    keys = [line.split(",")[0] for line in path.read_text().splitlines()]
    assert keys == ["parameter_file", "modification_online", "modes_discared", "sr_le_modes",
                    "best_modes_discarded", "gains", "sr_le_gains", "best_gain", "sr_le_best_gain"]


@pytest.mark.gpu
def test_per_environment_gains_equal_scalar_gain_runs():
    """aomarl_set_env_gains: environment e integrates with gains[e]; every environment must follow,
    bit for bit, the run in which that gain is the scalar gain of all environments."""
    from ao_marl_amd.env import VecRlSupervisor
    gains = np.array([0.15, 0.5, 0.9, 0.5], dtype=np.float32)
    sup = VecRlSupervisor(CFG, dict(n_reverse_filtered_from_cmat=5), 4, initial_seed=3, seed_stride=1)
    sup.set_gain(gains)
    assert sup.gain is None
    sup.reset()
    for _ in range(8):
        sup.next_part_one()
        sup.next_part_two(None, linear_control=True)
    com, sr = sup.get_command().clone(), sup.get_strehl().clone()
    for g in (0.15, 0.5, 0.9):
        sup.set_gain(float(g))
        assert sup.gain == pytest.approx(g)
        sup.reset()
        for _ in range(8):
            sup.next_part_one()
            sup.next_part_two(None, linear_control=True)
        for e in np.nonzero(gains == np.float32(g))[0]:
            assert torch.equal(sup.get_command()[e], com[e])
            assert torch.equal(sup.get_strehl()[e], sr[e])
    with pytest.raises(ValueError):
        sup.sim.set_env_gains([0.1, 0.2])


@pytest.mark.gpu
def test_gain_scan_on_the_gpu_finds_an_interior_optimum():
    """The scan on the HIP path (10x10, short episodes): unstable and sluggish gains lose against a
    mid-range one, and re-filtering the command matrix changes the loop."""
    res = GS.obtain_modes_filtered_and_gain(CFG, [0.05, 0.4, 1.9], num_episodes=2, num_steps=150)
    sr = res["sr_le_gains"]
    assert res["best_gain"] == pytest.approx(0.4)
    assert sr[1] > sr[0] and sr[1] > 2 * sr[2]
    assert len(set(np.round(res["sr_le_modes"], 6))) > 1
    assert res["best_modes_discarded"] in (0, 5, 10)
