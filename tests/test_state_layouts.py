"""The optional blocks of the environment state (ao_env.py:507-583, 871-909): slopes and their history,
dm_after_linear, residual history -- key order, standardisation and the history bookkeeping of
VecAoEnv.linear_step, re-derived here from what the supervisor exposes around each step."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_optional_state_blocks_follow_the_reference_layout():
    from ao_marl_amd.env import VecAoEnv
    rl = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5, state_wfs=True, number_of_previous_wfs=2,
              state_dm_after_linear=True, number_of_previous_dm_residuals=1, number_of_previous_dm=2)
    env = VecAoEnv("production_sh_10x10_2m", 3, rl)
    assert not env._default_state_layout
    sup = env.supervisor
    want_keys = ["wfs_history-2", "wfs_history-1", "wfs", "dm_history_2", "dm_history_1", "dm_after_linear",
                 "dm_before_linear", "dm_residual_history_1", "dm_residual"]
    assert list(env.state_keys) == want_keys                 # insertion order of the reference's OrderedDict
    v2m = torch.as_tensor(sup.volts2modes, device="cuda:0")
    sel = torch.as_tensor(np.asarray(list(range(0, 80)) + [sup.nmodes - 2, sup.nmodes - 1]), device="cuda:0")
    modes = lambda volts: (volts @ v2m.T)[:, sel]             # transform_state_to_zernike (ao_env.py:482-505)  # noqa: E731
    std = lambda x, k: (x - env.norm[k][0]) / env.norm[k][1]  # noqa: E731
    hist_wfs = [torch.zeros(3, env.wfs_dim, device="cuda:0")] * 2
    hist_dm = [torch.zeros(3, env.dm_dim, device="cuda:0")] * 2
    hist_res = [torch.zeros(3, env.dm_dim, device="cuda:0")]
    g = torch.Generator(device="cuda:0").manual_seed(2)
    s = None
    for it in range(5):
        if it == 0:
            # reset() = supervisor reset + cleared histories + one linear_step (ao_env.py:316-359); take the
            # dictionary form of that first state through the same call the reset makes
            sup.reset()
            env._hist_wfs = [h.clone() for h in hist_wfs]; env._hist_dm = [h.clone() for h in hist_dm]
            env._hist_res = [h.clone() for h in hist_res]
        else:
            act = torch.rand(3, len(sup.action_range), device="cuda:0", generator=g) * 2 - 1
            env.rl_step(act)
        before = modes(sup.get_command().clone())
        s = env.linear_step(return_dict=True)
        assert list(s) == want_keys
        after, slopes, res = modes(sup.get_command()), sup.get_slopes(), modes(sup.get_err())
        tol = dict(rtol=2e-4, atol=2e-5)
        assert torch.allclose(s["wfs"], std(slopes, "wfs"), **tol)
        assert torch.allclose(s["wfs_history-1"], std(hist_wfs[-1], "wfs"), **tol)
        assert torch.allclose(s["wfs_history-2"], std(hist_wfs[-2], "wfs"), **tol)
        assert torch.allclose(s["dm_before_linear"], std(before, "dm"), **tol)
        assert torch.allclose(s["dm_after_linear"], std(after, "dm"), **tol)
        assert torch.allclose(s["dm_history_1"], std(hist_dm[-1], "dm"), **tol)
        assert torch.allclose(s["dm_history_2"], std(hist_dm[-2], "dm"), **tol)
        assert torch.allclose(s["dm_residual"], std(res, "dm_residual"), **tol)
        assert torch.allclose(s["dm_residual_history_1"], std(hist_res[-1], "dm_residual"), **tol)
        hist_wfs = hist_wfs[1:] + [slopes.clone()]
        hist_dm = hist_dm[1:] + [before.clone()]
        hist_res = hist_res[1:] + [res.clone()]
        if it > 0:
            assert not torch.equal(after, before)             # the integrator moved the command
    flat = torch.cat(list(s.values()), dim=1)
    assert flat.shape == (3, env.state_dim) and env.state_dim == 3 * env.wfs_dim + 6 * env.dm_dim
    assert torch.isfinite(flat).all()
