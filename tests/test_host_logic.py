"""This repo's host-side logic == the reference's own functions (golden vectors produced by
tools/gen_golden_host.py, which imports and runs the reference in the build container)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from ao_marl_amd import modal
from ao_marl_amd.agents import AgentLayout, BatchedGaussianPolicy, BatchedQNetwork
from ao_marl_amd.env import DelayedMDP


def test_compute_btt_and_cmat_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "host_btt_10x10.npz"))
    IF = sp.csc_matrix((z["IF_data"], z["IF_indices"], z["IF_indptr"]), shape=tuple(z["IF_shape"]))
    Btt, P = modal.compute_btt(IF[:, :-2].tocsc(), IF[:, -2:].toarray())
    # same formula, same LAPACK: identical up to round-off (mode signs included)
    assert np.abs(Btt - z["Btt"]).max() < 5e-5 * np.abs(z["Btt"]).max()
    assert np.abs(P - z["P"]).max() < 5e-5 * np.abs(z["P"]).max()
    cmat = modal.cmat_with_btt(z["imat"], z["Btt"], 5)
    assert np.abs(cmat - z["cmat"]).max() < 1e-4 * np.abs(z["cmat"]).max()
    # the filtered command matrix keeps the first n-nfilt-2 modes + TT
    assert cmat.shape == (90, 128)


CASES = {"small": dict(nmodes=87, se=[0, 80], n_modal=1, window=-1, tt_w=False),
         "large": dict(nmodes=1283, se=[0, 1274], n_modal=13, window=20, tt_w=True),
         "large_notw": dict(nmodes=1283, se=[0, 1274], n_modal=13, window=20, tt_w=False),
         # the reference's published layout: 43 agents = 42 x 30 modes + tip-tilt (README.md:116-119,
         # src/error_budget/helper_experiments.py:19-36), plain and with the `_w20` experiments' window
         "published": dict(nmodes=1283, se=[0, 1260], n_modal=42, window=-1, tt_w=False),
         "published_w20": dict(nmodes=1283, se=[0, 1260], n_modal=42, window=20, tt_w=True)}


@pytest.mark.parametrize("name", sorted(CASES))
def test_agent_layout_matches_reference(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "host_agents.npz"))
    c = CASES[name]
    lay = AgentLayout(c["nmodes"], c["se"], c["n_modal"], include_tip_tilt=True,
                      window_n_zernike=c["window"], include_tip_tilt_windowed=c["tt_w"],
                      n_filtered=5)
    total, local, total_existing = z["%s_total" % name]
    assert (lay.total_controlled_modes, lay.local_controlled_modes, lay.nmodes) == \
        (total, local, total_existing)
    assert lay.state_shapes() == list(z["%s_state_shapes" % name])
    for w, (a, b) in lay.agents.items():
        assert [a, b] == list(z["%s_agent%d_modes" % (name, w)])
        assert np.array_equal(lay.modes_chosen[w], z["%s_agent%d_chosen" % (name, w)]), (name, w)
    if name.startswith("published"):
        assert lay.n_agents == 43 and lay.state_shapes()[0] == (280 if name.endswith("w20") else 120)
    if name == "large":
        assert lay.state_shapes()[0] == 552 and lay.state_shapes()[-1] == 168
        # reference quirk: the TT agent's window indices are absolute 0..39 in every block
        assert list(lay.modes_chosen[14][2:42]) == list(range(40))


def test_batched_sac_forward_matches_reference_modules(golden_dir):
    blob = torch.load(os.path.join(golden_dir, "host_sac_forward.pt"), weights_only=True)
    lay = AgentLayout(87, [0, 80], 1, include_tip_tilt=True)
    assert lay.state_shapes() == [320, 8]
    pol = BatchedGaussianPolicy(lay, device="cpu")
    qn = BatchedQNetwork(lay, device="cpu")
    n = 5
    state = torch.zeros(n, lay.state_dim)
    for i, w in enumerate(lay.agents):
        pol.load_agent(i, blob["agent%d" % i]["policy"])
        state[:, torch.as_tensor(lay.modes_chosen[w])] = blob["agent%d" % i]["x"]
    # the two agents' state slices are disjoint in this layout, so one state serves both
    mean, log_std = pol.forward(state)
    for i in range(2):
        ref = blob["agent%d" % i]
        na = ref["mean"].shape[1]
        assert torch.allclose(mean[i, :, :na], ref["mean"], atol=2e-5, rtol=1e-5)
        assert torch.allclose(log_std[i, :, :na], ref["log_std"], atol=2e-5, rtol=1e-5)
    a, mu = pol.select_action(state, eval_mode=True)
    assert a.shape == (n, 82) and torch.equal(a, mu)
    assert torch.allclose(mu[:, :80], torch.tanh(blob["agent0"]["mean"]), atol=2e-5)
    assert torch.allclose(mu[:, 80:], torch.tanh(blob["agent1"]["mean"]), atol=2e-5)
    # critic: load the reference weights into the stacked tensors
    for i in range(2):
        ref = blob["agent%d" % i]["critic"]
        nin, na = [320, 8][i], [80, 2][i]
        for qi, pre in enumerate(("Q1", "Q2")):
            Win, hid, out = qn.q[qi]
            w = ref["%s_input.weight" % pre].T
            Win[i].zero_()
            Win[i, :nin] = w[:nin]
            Win[i, qn.in_max:qn.in_max + na] = w[nin:]
            assert ref["%s_input.bias" % pre].abs().max() == 0   # weights_init_: bias 0
            assert len(hid) == 0 and not any(k.startswith("hidden_") for k in ref)
            out[i] = ref["%s_output.weight" % pre].T
    xs = pol.split_states(state)
    act = torch.zeros(2, n, qn.act_max)
    act[0, :, :80] = blob["agent0"]["a"]
    act[1, :, :2] = blob["agent1"]["a"]
    q1, q2 = qn.forward(xs, act)
    for i in range(2):
        assert torch.allclose(q1[i], blob["agent%d" % i]["q1"], atol=2e-5, rtol=1e-5)
        assert torch.allclose(q2[i], blob["agent%d" % i]["q2"], atol=2e-5, rtol=1e-5)


def test_delayed_mdp_matches_reference(golden_dir):
    rec = np.load(os.path.join(golden_dir, "host_delayed_mdp.npz"))["rec"]
    got = []
    for delay, modif in ((1, False), (0, False), (1, True)):
        m = DelayedMDP(delay, modif)
        for t in range(6):
            if m.check_update_possibility():
                s, a, sn = m.credit_assignment()
                got.append((delay, int(modif), t, s, a, sn))
            m.save(10 * t, 100 * t, 10 * (t + 1))
    assert np.array_equal(np.asarray(got), rec)


def test_wind_variant_parameter_sets_are_the_references():
    """The 11 wind / gain variants of the 40x40 file (ao_marl_amd/params.py) against the numbers of the
    reference's parameter files (data/par/par4rl/production/*.py; the two _gain_change files: the `g` of the
    geo/ files of the same name) -- and every one of them has its recorded COMPASS statistics on board."""
    from ao_marl_amd import params
    from ao_marl_amd.env import load_norm
    want = {"_dir_0_15_30": ([0, 15, 30], [15, 10, 20], 0.7), "_dir_0_15_30_v_10_5_15": ([0, 15, 30], [10, 5, 15], 0.6),
            "_dir_0_15_30_v_20_15_25": ([0, 15, 30], [20, 15, 25], 0.7), "_same_dir": ([0, 0, 0], [15, 10, 20], 0.7),
            "_same_dir_v_10_5_15": ([0, 0, 0], [10, 5, 15], 0.6), "_same_dir_v_20_15_25": ([0, 0, 0], [20, 15, 25], 0.7),
            "_v_10_5_15": ([0, 45, 90], [10, 5, 15], 0.6), "_v_20_15_25": ([0, 45, 90], [20, 15, 25], 0.7),
            "_same_dir_roket": ([0, 0, 0], [15, 10, 20], 0.7), "_same_dir_gain_change_high": ([0, 0, 0], [15, 10, 20], 0.9),
            "_same_dir_gain_change_low": ([0, 0, 0], [15, 10, 20], 0.2)}
    base = params.builtin("production_sh_40x40_8m_3layers")
    assert len(params.WIND_VARIANTS) == len(want)
    for sfx, (wd, ws, g) in want.items():
        name = "production_sh_40x40_8m_3layers" + sfx
        ps = params.builtin(name + ".py")
        assert ps.simul_name == name
        assert list(ps.p_atmos.winddir) == wd and list(ps.p_atmos.windspeed) == ws
        assert all(abs(c.gain - g) < 1e-7 for c in ps.p_controllers)
        assert list(ps.p_atmos.frac) == list(base.p_atmos.frac) and ps.p_tel.diam == base.p_tel.diam
        norm, zn = load_norm(name)
        assert norm["wfs"]["std"].shape == (2400,) and zn.shape == (1283,)
    # a variant changes nothing of the base set it was derived from
    assert list(base.p_atmos.winddir) == [0, 45, 90] and base.p_controllers[0].gain == 0.7


def test_degenerate_normalisation_columns_are_caught():
    """The reference's own recorded 10x10 statistics hold 10 controlled Btt modes with a standard deviation of
    4e-9 .. 9e-9 (plus the filtered ones): standardising amplifies round-off by 1e+8 and a SAC trained on such
    states diverges.  VecAoEnv masks them (state 0) with a warning that names the file, raises on request, or
    keeps the reference's division."""
    import warnings
    from ao_marl_amd.env import VecAoEnv, load_norm
    from tests.oracle_vecsim import OracleVecSim
    name = "production_sh_10x10_2m"
    rl = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5)
    norm, _ = load_norm(name)
    sd = norm["dm"]["std"][:80]
    assert (sd < 1e-6 * np.median(sd)).sum() >= 5            # the file IS degenerate
    kw = dict(n_agents_modal=1, device="cpu", sim_factory=OracleVecSim)
    with pytest.warns(UserWarning, match="norm_production_sh_10x10_2m.npz"):
        env = VecAoEnv(name, 1, rl, **kw)
    assert len(env.dead_columns["dm"]) >= 5 and torch.isinf(env.norm["dm"][1]).sum() == len(env.dead_columns["dm"])
    s = env.reset()
    assert torch.isfinite(s).all() and float(s.abs().max()) < 1e3
    dm_dim = env.dm_dim
    for blk in range(3):
        assert float(s[:, blk * dm_dim + torch.as_tensor(env.dead_columns["dm"])].abs().max()) == 0.0
    with pytest.raises(ValueError, match="norm_production_sh_10x10_2m.npz"):
        VecAoEnv(name, 1, rl, dead_columns="raise", **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        keep = VecAoEnv(name, 1, rl, dead_columns="keep", **kw)
    assert keep.dead_columns == {} and torch.isfinite(keep.norm["dm"][1]).all()
