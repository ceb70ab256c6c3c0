"""Diagnostic (development aid): the d1_noise normalisation recipe on the HIP path with time series
and per-mode spectra dumped for offline comparison with the reference's recorded statistics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ao_marl_amd.env import VecRlSupervisor, load_norm

name = "production_sh_40x40_8m_3layers_d1_noise"
gains = [float(g) for g in sys.argv[1:]] or [0.65]
ref, zn_ref = load_norm(name)
sup = VecRlSupervisor(name, dict(n_reverse_filtered_from_cmat=5), 20, initial_seed=1, seed_stride=1)
sim = sup.sim
out = {}
for g in gains:
    sup.set_gain(g)
    sup.reset()
    T = 1000
    S = torch.zeros(T, 20, sup.s.nslope, device="cuda")
    Cm = torch.zeros(T, 20, sup.nmodes, device="cuda")
    Rm = torch.zeros(T, 20, sup.nmodes, device="cuda")
    SR = torch.zeros(T, 20, device="cuda")
    for t in range(T):
        sup.next_part_one()
        sup.next_part_two(None, linear_control=True)
        S[t] = sup.get_slopes()
        Cm[t] = sim.volts2modes(sup.get_command())
        Rm[t] = sim.volts2modes(sup.get_err())
        SR[t] = sim.strehl[:, 0]
    for key, X, rk in (("wfs", S, "wfs"), ("dm", Cm, "dm"), ("res", Rm, "dm_residual")):
        flat = X.reshape(-1, X.shape[-1]).double()
        std = flat.std(dim=0, unbiased=False).cpu().numpy()
        # robust scale: 1.4826 * median absolute deviation
        med = flat.median(dim=0).values
        mad = (flat - med).abs().median(dim=0).values.cpu().numpy() * 1.4826
        r = std / ref[rk]["std"]
        print("g=%.2f %-4s std ratio: median %.3f  p10 %.3f  p90 %.3f | std/MAD-scale median %.3f  max |x|/std median %.1f" % (
            g, key, np.median(r), np.percentile(r, 10), np.percentile(r, 90), np.median(std / np.maximum(mad, 1e-30)),
            np.median(flat.abs().max(dim=0).values.cpu().numpy() / std)))
        out["g%.2f_%s_std" % (g, key)] = std
        out["g%.2f_%s_mad" % (g, key)] = mad
        # within-env std (temporal) vs pooled
        within = X.double().std(dim=0, unbiased=False).mean(dim=0).cpu().numpy()
        out["g%.2f_%s_within" % (g, key)] = within
        print("        within-env temporal std / pooled std: median %.3f" % np.median(within / std))
    # second half only (transient excluded)
    for key, X, rk in (("dm", Cm, "dm"),):
        flat = X[T // 2:].reshape(-1, X.shape[-1]).double()
        r = flat.std(dim=0, unbiased=False).cpu().numpy() / ref[rk]["std"]
        print("g=%.2f %-4s second half std ratio median %.3f" % (g, key, np.median(r)))
    out["g%.2f_sr" % g] = SR.cpu().numpy()
    out["g%.2f_cm_env0" % g] = Cm[:, 0, ::64].cpu().numpy()
    out["g%.2f_rm_env0" % g] = Rm[:, 0, ::64].cpu().numpy()
    out["g%.2f_s_env0" % g] = S[:, 0, ::200].cpu().numpy()
    sr = SR.cpu().numpy()
    print("g=%.2f SR_se: mean %.3f  first100 %.3f last100 %.3f  min over envs of last-100 mean %.3f" % (
        g, sr.mean(), sr[:100].mean(), sr[-100:].mean(), sr[-100:].mean(axis=0).min()))
    # ratio spectrum by mode index
    r = out["g%.2f_dm_std" % g] / ref["dm"]["std"]
    print("   dm ratio by mode block of 128:", np.round([np.median(r[i:i + 128]) for i in range(0, 1280, 128)], 2))
    r = out["g%.2f_res_std" % g] / ref["dm_residual"]["std"]
    print("   res ratio by mode block of 128:", np.round([np.median(r[i:i + 128]) for i in range(0, 1280, 128)], 2))
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/d1_noise_diag.npz", **out)
