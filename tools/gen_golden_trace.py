#!/usr/bin/env python3
"""G10: the reference's OWN RlSupervisor + AoEnv, unmodified, executed over the oracle facade
(tools/ref_facade.py) -> golden (state, per-agent reward, slopes, command, Strehl) traces.

Build container only.  Two runs:
  single  tools/par/production_aomarl_sh_10x10_2m_single.py (the 10x10 system reduced to its
          controller-0 path) -> tests/golden/trace_10x10_single.npz
  stock   the reference's own production_sh_10x10_2m.py, UNMODIFIED (2 WFS, 4 DMs, 2 targets, LS +
          GEO controllers: the reference loops over all controllers every frame,
          rlSupervisor.py:1038-1049) -> tests/golden/trace_10x10_stock.npz, which also carries the
          geometric controller's command and the second target's Strehl.
The parameter file is staged, with the reference's normalisation DATA, in a scratch directory laid
out like the reference expects (nothing of the reference is written into this repository).
With --record-calls the whole Appendix-B call sequence the reference makes (constructors, array
loads, per-frame calls, reads) is logged through tools/record_calls.py ->
tests/golden/calls_10x10_<run>.npz: the input of the facade replay tests.

With --online the run sets `modification_online` (the reference's pure-delay-0 call order, rlSupervisor.py:145,
938-939, 964-965) -> tests/golden/trace_10x10_<run>_online.npz.

Usage: python tools/gen_golden_trace.py [single|stock] [--record-calls] [--online]
"""
import os
import shutil
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_shims  # noqa: E402
import ref_facade  # noqa: E402

import numpy as np  # noqa: E402

REF = _ref_shims.REF
NORM = "src/reinforcement_learning/helper_functions/preprocessing/normalization"
RUNS = {"single": ("production_aomarl_sh_10x10_2m_single", os.path.join(HERE, "par")),
        "stock": ("production_sh_10x10_2m", os.path.join(REF, "data/par/par4rl/production"))}


def stage(tmp, NAME, src_dir):
    os.makedirs(os.path.join(tmp, "data/par/par4rl/production"))
    shutil.copy(os.path.join(src_dir, NAME + ".py"), os.path.join(tmp, "data/par/par4rl/production"))
    for sub in ("state_normalization", "normalization_action_zernike"):
        os.makedirs(os.path.join(tmp, NORM, sub))
    shutil.copy(os.path.join(REF, NORM, "state_normalization",
                             "normalization_production_sh_10x10_2m_zernike_space.pickle"),
                os.path.join(tmp, NORM, "state_normalization",
                             "normalization_%s_zernike_space.pickle" % NAME))
    shutil.copy(os.path.join(REF, NORM, "normalization_action_zernike",
                             "zn_norm_production_sh_10x10_2m.npy"),
                os.path.join(tmp, NORM, "normalization_action_zernike", "zn_norm_%s.npy" % NAME))
    os.makedirs(os.path.join(tmp, "output/debug"))


def main(run="single", nframes=30, seed=1234, record=False, online=False):
    NAME, src_dir = RUNS[run]
    sw, cw = ref_facade.install()
    rec = None
    if record:
        import record_calls
        rec = record_calls.Recorder()
        rec.install(sw, cw)
    _ref_shims.install()
    tmp = tempfile.mkdtemp(prefix="aomarl_ref_")
    stage(tmp, NAME, src_dir)
    os.chdir(tmp)
    from src.reinforcement_learning.config.GlobalConfig import Config
    from src.reinforcement_learning.environment.ao_env import AoEnv
    from src.reinforcement_learning.rpc_training.helper_rpc.helper_rewards import \
        get_separated_rewards
    from src.reinforcement_learning.rpc_training.train_rpc import TrainerRPC
    cfg = Config(os.path.join(REF, "src/reinforcement_learning"))
    cfg.env_rl.update({"parameters_telescope": NAME + ".py", "n_zernike_start_end": [0, 80],
                       "n_reverse_filtered_from_cmat": 5, "include_tip_tilt": "True",
                       "verbose": False})
    if online:
        cfg.env_rl["modification_online"] = "True"
    cfg.strings_to_bools()
    assert cfg.env_rl["modification_online"] is bool(online)
    cfg.autoencoder["path"] = None
    env = AoEnv(config_rl=cfg, normalization_bool=True, initial_seed=seed)
    env.supervisor.set_sim_seed(seed)
    sup = env.supervisor
    fake = types.SimpleNamespace(env=env, world_size=3)
    agents, total, local, total_existing = TrainerRPC.create_agents_dictionary_original(fake, cfg)
    rng = np.random.default_rng(99)
    geo = len(sup.config.p_controllers) > 1
    rec_ = rec
    rec = {k: [] for k in ("state", "reward", "slopes", "com", "err", "voltage", "strehl",
                           "action", "com_geo", "strehl_geo")}
    s = env.reset()
    rec["state"].append(np.asarray(s, dtype=np.float64))
    rec["slopes"].append(sup.rtc.get_slopes(0))
    rec["com"].append(sup.rtc.get_command(0))
    rec["err"].append(sup.rtc.get_err(0))
    for it in range(nframes):
        # zero actions for the first third (pure integrator through rl_control), then U(-1, 1)
        a = np.zeros(82, dtype=np.float32) if it < nframes // 3 else \
            rng.uniform(-1, 1, size=82).astype(np.float32)
        # TrainerRPC.env_step (train_rpc.py:633-648)
        _, done, info = env.rl_step(a, False)
        modes = sup.volts2modes.dot(sup.rtc.get_err(0))
        r = get_separated_rewards(np.square(modes), cfg.env_rl["reward_type"], agents)
        s = env.linear_step(False)
        rec["action"].append(a)
        rec["state"].append(np.asarray(s, dtype=np.float64))
        rec["reward"].append(np.array([r[w] for w in agents], dtype=np.float64))
        rec["slopes"].append(sup.rtc.get_slopes(0))
        rec["com"].append(sup.rtc.get_command(0))
        rec["err"].append(sup.rtc.get_err(0))
        rec["voltage"].append(sup.rtc.get_voltages(0))
        rec["strehl"].append(np.asarray(sup.target.get_strehl(0), dtype=np.float64))
        if geo:
            rec["com_geo"].append(sup.rtc.get_command(1))
            rec["strehl_geo"].append(np.asarray(sup.target.get_strehl(1), dtype=np.float64))
    out = {k: np.asarray(v) for k, v in rec.items() if len(v)}
    out["modes2volts"], out["volts2modes"] = sup.modes2volts, sup.volts2modes
    out["cmat"] = np.array(sup.rtc._rtc.d_control[0].d_cmat)
    out["imat"] = np.array(sup.rtc._rtc.d_control[0].d_imat)
    out["freedom_vector"] = sup.freedom_vector
    out["seed"] = np.array(seed)
    out["agents"] = np.array([agents[w] for w in agents])
    out["nactu"] = np.array(out["com"].shape[1])
    out["modification_online"] = np.array(bool(online))
    assert sup.pure_delay_0 is bool(online)
    dst = os.path.join(ROOT, "tests", "golden", "trace_10x10_%s%s.npz" % (run, "_online" if online else ""))
    np.savez_compressed(dst, **out)
    if rec_ is not None:
        rec_.save(os.path.join(ROOT, "tests", "golden", "calls_10x10_%s%s.npz" % (run, "_online" if online else "")))
    print("wrote", dst, {k: v.shape for k, v in out.items()})
    print("SR se/le last:", out["strehl"][-1][:2], "state absmax", np.abs(out["state"]).max())
    os.chdir(ROOT)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    main(run=args[0] if args else "single", record="--record-calls" in sys.argv, online="--online" in sys.argv)
