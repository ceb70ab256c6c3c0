"""Environment-level reward types of AoEnv.calculate_reward (ao_env.py:585-860), batched over environments.

The trainer throws this reward away (train_rpc.py:641: the agents are paid per agent from the residual
modes, helper_rewards.py:14-22, `VecAoEnv.divide_rewards_for_agents`); it exists for the reference's
single-agent experiments and its evaluation scripts.  Every branch of the reference's if / elif chain
that reads slopes, the integrator increment, the residual modes or the Strehl tuple is restated here as
a function of a [nenv, n] tensor; the formulas are NumPy one-liners there (np.var is the population
variance, np.average the mean, "x / y" halves are s[: n // 2] and s[n // 2 :], ao_env.py:613,665-666).
The branches that read the full-frame target image (`image_sharpness`, `r_tt_4`, `r_tt_4_norm`: ao_env.py:621-623,
654-656) take it from Target.get_tar_image -- formed on demand (aomarl_target_image), the hot path only keeps the
Strehl window --, the two that compare the command with the WFS phase projected on the modes
(`projection_comparison`, `weighted_projection_comparison`, :736-760) use the supervisor's
projector_phase2modes (utils.py:88-160, built on first use): all 62 names of the chain are served.

A name and name + "_norm" select the same formula (ao_env.py:624-662; the reference compares the
"_norm" spelling against the configured type rather than the argument -- the same thing whenever the
argument is the configured type, which is the only way the shipped code calls it).
"""
import torch

_LOG_VAR = {"log_var_scaled_1": (2.60, 1.0), "log_var_scaled_2": (2.60, 5.0), "log_var_scaled_3": (2.60, 10.0),
            "log_var_scaled_4": (2.60, 0.5), "log_var_scaled_5": (3.50, 1.0), "log_var_scaled_6": (3.50, 5.0),
            "log_var_scaled_7": (3.50, 10.0), "log_var_scaled_8": (3.50, 0.5)}          # ao_env.py:712-735


def _var(x):
    return x.var(dim=1, unbiased=False)


def _halves(s):
    h = s.shape[1] // 2
    return s[:, :h], s[:, h:]


def _slopes_table():
    msq = lambda s: (s * s).mean(dim=1)
    avg2 = lambda s: sum(h.mean(dim=1) for h in _halves(s))
    t = {
        "residual_wfs": lambda s: -torch.linalg.vector_norm(s, dim=1),                  # :606-608
        "var_wfs": lambda s: -_var(s),                                                  # :609-611
        "averages_wfs": lambda s: -avg2(s),                                             # :612-614
        "average_var_wfs": lambda s: -_var(s) - avg2(s),                                # :615-617
        "average_residual_wfs": lambda s: -_var(s) - avg2(s),                           # :618-620 (same formula)
        "r_modes_1": lambda s: torch.exp(-_var(s)),                                     # :624-626
        "r_modes_2": lambda s: -_var(s),
        "r_modes_3": lambda s: torch.exp(-_var(s)) - 1,
        "r_modes_4": lambda s: torch.exp(-_var(s * s)),
        "r_modes_5": lambda s: -_var(s * s),
        "r_modes_6": lambda s: torch.exp(-_var(s * s)) - 1,
        "r_tt_1": lambda s: -s.mean(dim=1) ** 2,                                        # :642-644
        "r_tt_2": lambda s: -s.mean(dim=1).abs(),
        "r_tt_3": lambda s: -msq(s),
        "r_tt_5": lambda s: torch.exp(-msq(s)),
        "r_tt_6": lambda s: torch.exp(-msq(s)) - 1,
        "r_tt_7": lambda s: -sum(h.mean(dim=1) ** 2 for h in _halves(s)),               # :663-667
        "r_modes_7": lambda s: -sum(_var(h) ** 2 for h in _halves(s)),                  # :668-672
        "r_1_and_2": lambda s: -_var(s) - s.mean(dim=1).abs(),                          # :673-675
        "single_agent_1": lambda s: -msq(s),                                            # :678-680
        "avg_square_m": lambda s: -msq(s),
        "sum_measurements_squared": lambda s: -(s * s).sum(dim=1),                      # :681-683
        "single_agent_2": lambda s: torch.exp(-_var(s)) - 1,                            # :684-686
        "new_single_agent": lambda s: torch.exp(-msq(s)),                               # :707-709
        "log_avg_m": lambda s: -torch.log1p(msq(s)),                                    # :761-764
    }

    def single_agent_3(s):                                                              # :687-690
        r_modes, r_tt = torch.exp(-_var(s)) - 1, -msq(s)
        return r_tt / (r_tt + r_modes) + r_modes / (r_tt + r_modes)

    def single_agent_4(s):                                                              # :700-706
        r_tt, r_modes = torch.exp(-_var(s)) - 1, -msq(s)
        return 0.4479 * r_tt / (r_tt + r_modes) + 0.5485 * r_modes / (r_tt + r_modes)

    t["single_agent_3"], t["single_agent_4"] = single_agent_3, single_agent_4
    for k in ("r_modes_1", "r_modes_2", "r_modes_3", "r_modes_4", "r_modes_5", "r_modes_6",
              "r_tt_1", "r_tt_2", "r_tt_3", "r_tt_5", "r_tt_6", "r_1_and_2"):
        t[k + "_norm"] = t[k]
    return t


SLOPES = _slopes_table()
UNSUPPORTED = {}
IMAGE = ("image_sharpness", "r_tt_4", "r_tt_4_norm")                   # read target.get_tar_image(0)
PROJECTION = ("projection_comparison", "weighted_projection_comparison")
# what these five read is the state of THIS frame: screens already moved to the next one (atmosphere prefetch, frame
# pipeline) would give the next frame's image / phase -- VecAoEnv turns both off for them
NEEDS_CURRENT_SCREENS = IMAGE + PROJECTION


def image_reward(name, img):
    """Branches that read target.get_tar_image(0): img [nenv, N, N] (any orientation: both formulas are symmetric
    in the two axes) -> [nenv]."""
    if name == "image_sharpness":                                                       # :621-623
        return (img * img).sum(dim=(1, 2)) / img.sum(dim=(1, 2)) ** 2
    # r_tt_4 (:654-656): -sum((scipy.ndimage.center_of_mass(img) - N / 2)^2), the centre in index units
    n = img.shape[1]
    tot = img.sum(dim=(1, 2))
    idx = torch.arange(n, dtype=img.dtype, device=img.device)
    c0 = (img.sum(dim=2) * idx).sum(dim=1) / tot
    c1 = (img.sum(dim=1) * idx).sum(dim=1) / tot
    return -((c0 - n / 2.0) ** 2 + (c1 - n / 2.0) ** 2)


def projection_reward(name, projection_modes, current_modes, action_range, freedom=None):
    """projection_comparison / weighted_projection_comparison (:736-760): -|| (P . phase)[range] - modes[range] ||
    (the weighted form multiplies the difference by the freedom vector first)."""
    d = projection_modes[:, action_range] - current_modes[:, action_range]
    if name == "weighted_projection_comparison":
        d = d * freedom[action_range]
    return -torch.linalg.vector_norm(d, dim=1)


def slopes_reward(name, s):
    """Branches that read rtc.get_slopes(0): s [nenv, nslope] -> [nenv]."""
    return SLOPES[name](s)


def strehl_reward(name, st):
    """Branches that read target.get_strehl: st [nenv, >=3] = (SE, LE, phase variance, ...) -> [nenv]."""
    if name == "wavefront_phase_error":
        return -st[:, 2]
    if name == "strehl_ratio_le":
        return st[:, 1]
    if name == "strehl_ratio_se":
        return st[:, 0]
    if name == "r_le":
        return torch.zeros_like(st[:, 0])                                               # :676-677
    if name == "log_var":
        return -torch.log1p(st[:, 2])                                                   # :710-711
    off, mult = _LOG_VAR[name]
    return -(torch.log1p(st[:, 2]) - off) * mult


STREHL = ("wavefront_phase_error", "strehl_ratio_le", "strehl_ratio_se", "r_le", "log_var") + tuple(_LOG_VAR)


def modes_reward(name, m):
    """Branches that read the residual modes transform_state_to_zernike(get_err(0), return_reward=True):
    m [nenv, nsel] -> [nenv] (ao_env.py:765-853)."""
    sq = m * m
    if name == "avg_squared_modes":
        return -sq.sum(dim=1)
    if name == "true_avg_squared_modes":
        return -sq.mean(dim=1)
    if name in ("avg_squared_modes_scaled_1", "avg_squared_modes_scaled_2", "avg_squared_modes_scaled_3"):
        return -sq.sum(dim=1) * (10.0 ** int(name[-1]))
    if name.startswith("avg_squared_modes_"):                                           # "avg_squared_modes_<factor>"
        return -float(name.split("_")[-1]) * sq.mean(dim=1)
    raise NotImplementedError("This reward type not implemented")


def is_modes_reward(name):
    return name in ("avg_squared_modes", "true_avg_squared_modes") or \
        (name.startswith("avg_squared_modes_") and name != "avg_squared_modes_from_measurements")
