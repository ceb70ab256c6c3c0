#!/usr/bin/env python3
"""Golden vectors for the reference's HOST-side logic on the hot path, produced by importing and
running the reference's own functions (build container only):

  G4  shesha.ao.basis.compute_btt               on this repo's influence matrix (10x10)
  G5  shesha.ao.basis.compute_cmat_with_Btt     on this repo's interaction matrix (10x10)
  G6  helper_states.get_modes_chosen + TrainerRPC.create_agents_dictionary_original /
      get_state_shape_worker   for the 2-agent (10x10) and 14-agent windowed (40x40) layouts
  G7  model_rpc.GaussianPolicy / QNetwork forward on seeded weights and inputs (torch CPU)
  G8  environment.delayed_mdp.DelayedMDP trace
Writes tests/golden/host_*.npz / .pt (data only).
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_shims  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def g4_g5():
    from tests import helpers
    import shesha.ao.basis as bas
    sysm, s, cal = helpers.calibrated("production_sh_10x10_2m", nfilt=5)
    IF = cal.IF
    Btt, P = bas.compute_btt(IF[:, :-2].tocsr(), IF[:, -2:].toarray())

    class Ctl(object):
        d_imat = cal.imat

        def set_cmat(self, c):
            self.cmat = c

    rtc = types.SimpleNamespace(d_control=[Ctl()])
    cmat = bas.compute_cmat_with_Btt(rtc, Btt, 5)
    np.savez_compressed(os.path.join(OUT, "host_btt_10x10.npz"), Btt=Btt, P=P, cmat=cmat,
                        imat=cal.imat, IF_data=IF.tocsc().data, IF_indices=IF.tocsc().indices,
                        IF_indptr=IF.tocsc().indptr, IF_shape=np.array(IF.shape))
    print("G4/G5", Btt.shape, P.shape, cmat.shape)


def g6():
    from src.reinforcement_learning.rpc_training.helper_rpc import helper_states as hs
    hs.debug_modes_chosen = False
    from src.reinforcement_learning.rpc_training.train_rpc import TrainerRPC
    out = {}
    cases = {
        "small": dict(nmodes=87, se=[0, 80], world=3, window=-1, tt_w=False, nfilt=5),
        "large": dict(nmodes=1283, se=[0, 1274], world=15, window=20, tt_w=True, nfilt=5),
        "large_notw": dict(nmodes=1283, se=[0, 1274], world=15, window=20, tt_w=False, nfilt=5),
        # the reference's own published layout: 43 agents = 42 x 30 modes + tip-tilt (README.md:116-119,
        # `--world-size 44 --n_zernike_start_end 0 1260`; src/error_budget/helper_experiments.py:19-36), plain and
        # with the window of 20 of the `_w20` experiments
        "published": dict(nmodes=1283, se=[0, 1260], world=44, window=-1, tt_w=False, nfilt=5),
        "published_w20": dict(nmodes=1283, se=[0, 1260], world=44, window=20, tt_w=True, nfilt=5),
    }
    for name, c in cases.items():
        cfg = types.SimpleNamespace(env_rl={
            "n_zernike_start_end": c["se"], "include_tip_tilt": True, "tt_treated_as_mode": False,
            "window_n_zernike": c["window"], "include_tip_tilt_windowed": c["tt_w"],
            "state_dm_residual": True, "state_dm_after_linear": False,
            "state_dm_before_linear": True, "number_of_previous_dm": 2,
            "number_of_previous_dm_residuals": 0})
        fake = types.SimpleNamespace(
                env=types.SimpleNamespace(supervisor=types.SimpleNamespace(
                        modes2volts=np.zeros((c["nmodes"] + 3, c["nmodes"])))),
                world_size=c["world"])
        d, total, local, total_existing = TrainerRPC.create_agents_dictionary_original(fake, cfg)
        fake.dictionary_agents = d
        block = c["nmodes"] if c["window"] > -1 else total
        keys = ["dm_history_2", "dm_history_1", "dm_before_linear", "dm_residual"]
        ios, o = {}, 0
        for k in keys:
            ios[k] = [o, o + block]
            o += block
        mc = hs.get_modes_chosen(d, ios, cfg, c["nfilt"], "x", total_existing, total, c["se"][0])
        shapes = [TrainerRPC.get_state_shape_worker(fake, d[w], cfg, w) for w in d]
        for w in d:
            out["%s_agent%d_modes" % (name, w)] = np.asarray(d[w])
            out["%s_agent%d_chosen" % (name, w)] = np.asarray(mc[w])
        out["%s_state_shapes" % name] = np.asarray(shapes)
        out["%s_total" % name] = np.array([total, local, total_existing])
        print("G6", name, len(d), shapes[:2], shapes[-1])
    np.savez_compressed(os.path.join(OUT, "host_agents.npz"), **out)


def g7():
    from src.reinforcement_learning.rpc_training.algorithms_rpc.model_rpc import (GaussianPolicy,
                                                                                  QNetwork)
    torch.manual_seed(1234)
    blob = {}
    for i, (nin, nact) in enumerate([(320, 80), (8, 2)]):
        pol = GaussianPolicy(num_inputs=nin, num_actions=nact, hidden_dim=256, action_scale=1.0,
                             action_bias=0.0, num_layers=2, initialize_last_layer_zero=False,
                             initialize_last_layer_near_zero=False, activation="relu",
                             LOG_SIG_MAX=2.0)
        with torch.no_grad():   # non-trivial heads and biases
            for p in pol.parameters():
                if p.dim() == 1:
                    p.copy_(torch.randn_like(p) * 0.1)
        x = torch.randn(5, nin)
        with torch.no_grad():
            mean, log_std = pol.forward(x)
        q = QNetwork(nin, nact, [256], 2)
        a = torch.tanh(torch.randn(5, nact))
        with torch.no_grad():
            q1, q2 = q.forward(x, a)
        blob["agent%d" % i] = dict(policy=pol.state_dict(), x=x, mean=mean, log_std=log_std,
                                   critic=q.state_dict(), a=a, q1=q1, q2=q2)
    torch.save(blob, os.path.join(OUT, "host_sac_forward.pt"))
    print("G7 ok")


def g8():
    from src.reinforcement_learning.environment.delayed_mdp import DelayedMDP
    rec = []
    for delay, modif in ((1, False), (0, False), (1, True)):
        m = DelayedMDP(delay, modif)
        for t in range(6):
            if m.check_update_possibility():
                s, a, sn = m.credit_assignment()
                rec.append((delay, int(modif), t, s, a, sn))
            m.save(10 * t, 100 * t, 10 * (t + 1))
    np.savez_compressed(os.path.join(OUT, "host_delayed_mdp.npz"), rec=np.asarray(rec))
    print("G8", len(rec))


def g9_denoiser():
    """G8 of SURVEY: the shipped denoiser weights (data) + the reference module's forward on
    seeded synthetic spot images."""
    from src.autoencoder.autoencoder_models import DenoisingAutoencoderCNN2DSingleSubapeture
    path = os.path.join(_ref_shims.REF, "output/autoencoder/autoencoder_weights/autoencoder_M9_rms_3")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    m = DenoisingAutoencoderCNN2DSingleSubapeture()
    m.load_state_dict(sd)
    m.eval()
    g = torch.Generator().manual_seed(5)
    yy, xx = torch.meshgrid(torch.arange(16.), torch.arange(16.), indexing="ij")
    imgs = []
    for k in range(8):
        cx, cy = 7.5 + torch.randn(1, generator=g) * 1.5, 7.5 + torch.randn(1, generator=g) * 1.5
        spot = 25 * torch.exp(-((xx - cx)**2 + (yy - cy)**2) / 3.0)
        imgs.append(torch.poisson(spot, generator=g) + 3 * torch.randn(16, 16, generator=g))
    x = torch.stack(imgs).unsqueeze(1)
    with torch.no_grad():
        y = m(x)
    torch.save({"state_dict": {k: v.clone() for k, v in sd.items()}, "x": x, "y": y},
               os.path.join(OUT, "host_denoiser.pt"))
    print("G9 denoiser", tuple(x.shape), float(y.abs().max()), sum(v.numel() for v in sd.values()))


if __name__ == "__main__":
    _ref_shims.install()
    os.makedirs(OUT, exist_ok=True)
    g4_g5()
    g6()
    g7()
    g8()
    g9_denoiser()
