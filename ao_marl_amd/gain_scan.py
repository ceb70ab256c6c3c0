"""Integrator gain / filtered-modes scan, batched on the device.

Counterpart of the reference's `obtain_modes_filtered_and_gain`
(src/reinforcement_learning/helper_functions/preprocessing/obtain_gain/
obtain_best_gain_and_modes_filtered.py:101-175) and its `performance_loop` (:41-98):

  1. with gain 0.5, for each number of filtered Btt modes (0, 5, 10): `num_episodes` episodes
     (seeds 1, 2, ...) of `num_steps` integrator-only frames (`AoEnv.normalization_step`); keep the
     long-exposure Strehl of the LAST episode (the reference reads `get_strehl(0)[1]` after the loop,
     and `reset()` clears the accumulator every episode);
  2. with the best count, the same for every candidate gain; best gain = argmax of that Strehl.

The reference runs the candidates one after the other on one simulated system.  Here the episodes of
one candidate are environments of one batch, and in step 2 ALL gains run at once: environment
(gain g, episode k) has integrator gain g (aomarl_set_env_gains) and seed k.  The command matrix is
a property of the context, so step 1 is one batch per filtered-modes count.  The per-frame cycle is
the one of ao_marl_amd.normalization (next_part_one, then apply_control + Strehl).
"""
import numpy as np

from .env import VecRlSupervisor

MODES_FILTERED = (0, 5, 10)            # np.arange(0, 3) * 5, obtain_best_gain_and_modes_filtered.py:123


def performance_loop(sup, num_steps):
    """`num_steps` integrator-only frames on every environment from a fresh reset; returns the
    long-exposure Strehl per environment (numpy [nenv])."""
    sup.reset()
    for _ in range(num_steps):
        sup.next_part_one()
        sup.next_part_two(None, linear_control=True)
    return sup.get_strehl()[:, 1].cpu().numpy()


def obtain_modes_filtered_and_gain(config, gains, modes_filtered_list=MODES_FILTERED, num_episodes=2,
                                   num_steps=1000, device="cuda:0", sim_factory=None, autoencoder=None,
                                   verbose=False):
    """Returns the dict the reference writes to insights/gain/<file>/information_best_gain.csv
    (same keys), plus the per-episode Strehl tables (`sr_le_modes_all` [len(list), num_episodes],
    `sr_le_gains_all` [len(gains), num_episodes])."""
    gains = np.asarray(gains, dtype=np.float32).reshape(-1)
    modes_filtered_list = [int(m) for m in modes_filtered_list]
    # ---- 1. filtered modes at gain 0.5: one batch of `num_episodes` environments per candidate
    sup = VecRlSupervisor(config, dict(n_reverse_filtered_from_cmat=modes_filtered_list[0]),
                          num_episodes, initial_seed=1, seed_stride=1, device=device,
                          sim_factory=sim_factory, autoencoder=autoencoder)
    sr_modes_all = []
    for mf in modes_filtered_list:
        sup.obtain_and_set_cmat_filtered(mf)
        sup.set_gain(0.5)
        sr_modes_all.append(performance_loop(sup, num_steps))
        if verbose:
            print("modes filtered %d: SR LE per episode %s" % (mf, sr_modes_all[-1]))
    sr_modes_all = np.asarray(sr_modes_all)
    sr_list_modes = sr_modes_all[:, -1]                       # the last episode's, like the reference
    best_mf = modes_filtered_list[int(np.argmax(sr_list_modes))]
    del sup
    # ---- 2. all gains at once: environment j * num_episodes + k = (gain j, episode k)
    ng = int(gains.size)
    sup = VecRlSupervisor(config, dict(n_reverse_filtered_from_cmat=best_mf), ng * num_episodes,
                          initial_seed=1, seed_stride=1, device=device, sim_factory=sim_factory,
                          autoencoder=autoencoder)
    seeds = np.tile(np.arange(1, num_episodes + 1), ng)
    sup.env_seeds = lambda: seeds                              # seed k for every gain
    sup.set_gain(np.repeat(gains, num_episodes))
    sr_gains_all = performance_loop(sup, num_steps).reshape(ng, num_episodes)
    sr_list_gains = sr_gains_all[:, -1]
    if verbose:
        for g, sr in zip(gains, sr_gains_all):
            print("gain %.3f: SR LE per episode %s" % (g, sr))
    best = int(np.argmax(sr_list_gains))
    return {
        "modes_discared": list(modes_filtered_list),           # (sic) the reference's key
        "sr_le_modes": [float(v) for v in sr_list_modes],
        "best_modes_discarded": best_mf,
        "gains": [float(g) for g in gains],
        "sr_le_gains": [float(v) for v in sr_list_gains],
        "best_gain": float(gains[best]),
        "sr_le_best_gain": float(sr_list_gains[best]),
        "sr_le_modes_all": sr_modes_all, "sr_le_gains_all": sr_gains_all,
    }


def save_csv(path, name, res, modification_online=False):
    """information_best_gain.csv in the reference's layout (one `key,value` row per entry)."""
    rows = [("parameter_file", name), ("modification_online", modification_online)]
    for k in ("modes_discared", "sr_le_modes", "best_modes_discarded", "gains", "sr_le_gains",
              "best_gain", "sr_le_best_gain"):
        rows.append((k, res[k]))
    with open(path, "w") as fh:
        for k, v in rows:
            fh.write("%s,%s\n" % (k, '"%s"' % (v,) if isinstance(v, (list, tuple)) else v))


def main(argv=None):
    """python -m ao_marl_amd.gain_scan <parameter file | builtin name> [--gains 0.1 0.2 ...]
    [--episodes 2] [--steps 1000] [--out information_best_gain.csv]"""
    import argparse
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("config")
    ap.add_argument("--gains", type=float, nargs="+",
                    default=[round(0.1 + 0.05 * i, 2) for i in range(18)])   # 0.1 .. 0.95
    ap.add_argument("--episodes", type=int, default=2)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args(argv)
    from . import params
    cfg = params.load_param_file(a.config) if a.config.endswith(".py") else a.config
    res = obtain_modes_filtered_and_gain(cfg, a.gains, num_episodes=a.episodes, num_steps=a.steps,
                                         device=a.device, verbose=True)
    print("best modes filtered %d, best gain %.3f (SR LE %.4f)" %
          (res["best_modes_discarded"], res["best_gain"], res["sr_le_best_gain"]))
    if a.out:
        save_csv(a.out, a.config, res)


if __name__ == "__main__":
    main()
