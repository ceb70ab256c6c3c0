"""State-normalisation statistics and action bounds, generated on the device.

Counterpart of the reference's `Preprocessor.normalization_loop(geometric_controller=False)` +
`obtain_original_freedom_vector_integrator`
(src/reinforcement_learning/helper_functions/preprocessing/normalization/obtain_normalization.py:139-243,
:63-81): 20 episodes (seeds 1..20) x 1000 integrator-only frames with the Btt-filtered command
matrix; per frame the slopes, the Btt modes of the integrator command and of its increment are
recorded, and mean / std / max / min over all 20 000 samples become the state standardisation
(`normalization_<file>_zernike_space.pickle`), (|peak| + |valley|) / 2 of the command modes the
action bounds (`zn_norm_<file>.npy`).

The reference runs the 20 episodes one after the other on one simulated system; here they are 20
environments of one batch, and the statistics are accumulated on the device (float64 sums), so
nothing but the final vectors leaves the GPU.  Frame order is the reference's
`next_integrator_normalization` (rlSupervisor.py:506-590): part one (atmosphere, science + WFS
paths, centroids, integrator), then apply_control + Strehl -- the same cycle as
`AoEnv.linear_step` / `rl_step(linear_control=True)`.
"""
import numpy as np
import torch

from .env import VecRlSupervisor

KEYS = ("wfs", "dm", "dm_residual")


class _Stats(object):
    """Running mean / std (population, like np.std) / max / min over the sample axis."""

    def __init__(self, dim, device):
        self.n = 0
        self.s1 = torch.zeros(dim, dtype=torch.float64, device=device)
        self.s2 = torch.zeros(dim, dtype=torch.float64, device=device)
        self.mx = torch.full((dim,), -float("inf"), dtype=torch.float32, device=device)
        self.mn = torch.full((dim,), float("inf"), dtype=torch.float32, device=device)

    def update(self, x):
        xd = x.double()
        self.n += x.shape[0]
        self.s1 += xd.sum(dim=0)
        self.s2 += (xd * xd).sum(dim=0)
        self.mx = torch.maximum(self.mx, x.max(dim=0).values)
        self.mn = torch.minimum(self.mn, x.min(dim=0).values)

    def result(self):
        mean = self.s1 / self.n
        var = (self.s2 / self.n - mean * mean).clamp(min=0.0)
        return {"mean": mean.float().cpu().numpy(), "std": var.sqrt().float().cpu().numpy(),
                "max": self.mx.cpu().numpy(), "min": self.mn.cpu().numpy()}


def normalization_loop(supervisor, frames=1000):
    """Run `frames` integrator-only frames on every environment of `supervisor` from a fresh reset
    and return (norm_dict, zn_norm, strehl_le [nenv])."""
    sup, sim = supervisor, supervisor.sim
    dev = sim.device
    stats = {"wfs": _Stats(sup.s.nslope, dev), "dm": _Stats(sup.nmodes, dev),
             "dm_residual": _Stats(sup.nmodes, dev)}
    sup.reset()
    for _ in range(frames):
        sup.next_part_one()
        sup.next_part_two(None, linear_control=True)
        stats["wfs"].update(sup.get_slopes())                        # rtc.get_slopes(0)
        stats["dm"].update(sim.volts2modes(sup.get_command()))       # v2m . rtc.get_command(0)
        stats["dm_residual"].update(sim.volts2modes(sup.get_err()))  # v2m . rtc.get_err(0)
    norm = {k: stats[k].result() for k in KEYS}
    # obtain_original_freedom_vector_integrator: (|max| + |min|) / 2 of the command's Btt modes
    zn_norm = (np.abs(norm["dm"]["max"]) + np.abs(norm["dm"]["min"])) / 2.0
    return norm, zn_norm.astype(np.float32), sup.get_strehl()[:, 1].cpu().numpy()


def obtain_normalization(config, modes_filtered=5, episodes=20, frames=1000, device="cuda:0",
                         sim_factory=None, autoencoder=None):
    """`run_obtain_normalization_and_freedom(parameter_file, "zernike_space", False, modes_filtered)`
    (obtain_normalization.py:246-300) for the integrator controller: seeds 1..episodes."""
    sup = VecRlSupervisor(config, dict(n_reverse_filtered_from_cmat=max(int(modes_filtered), 0)),
                          episodes, initial_seed=1, seed_stride=1, device=device,
                          sim_factory=sim_factory, autoencoder=autoencoder)
    return normalization_loop(sup, frames=frames)


def save_norm(path, norm, zn_norm):
    """Same container as ao_marl_amd/data/norm_<config>.npz (read by env.load_norm)."""
    out = {"zn_norm": np.asarray(zn_norm, dtype=np.float32)}
    for k in KEYS:
        for st in ("mean", "std", "max", "min"):
            out["%s_%s" % (k, st)] = np.asarray(norm[k][st], dtype=np.float32)
    np.savez_compressed(path, **out)


def save_reference_layout(prefix, norm, zn_norm):
    """The two files the reference's AoEnv loads (ao_env.py:246-300): the pickle of nested dicts
    and the .npy of action bounds."""
    import pickle
    with open(prefix + "_zernike_space.pickle", "wb") as fh:
        pickle.dump({k: {st: np.asarray(v) for st, v in norm[k].items()} for k in KEYS}, fh)
    np.save(prefix + "_zn_norm.npy", np.asarray(zn_norm))


def main(argv=None):
    """python -m ao_marl_amd.normalization <parameter file | builtin name> [--modes-filtered 5]
    [--out norm.npz] [--reference-prefix path/normalization_<name>]"""
    import argparse
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("config")
    ap.add_argument("--modes-filtered", type=int, default=5)
    ap.add_argument("--episodes", type=int, default=20)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--reference-prefix", default=None)
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args(argv)
    from . import params
    cfg = params.load_param_file(a.config) if a.config.endswith(".py") else a.config
    norm, zn, sr = obtain_normalization(cfg, a.modes_filtered, a.episodes, a.frames, a.device)
    print("episodes %d x %d frames: long-exposure Strehl %.4f (min %.4f)" %
          (a.episodes, a.frames, sr.mean(), sr.min()))
    if a.out:
        save_norm(a.out, norm, zn)
    if a.reference_prefix:
        save_reference_layout(a.reference_prefix, norm, zn)


if __name__ == "__main__":
    main()
