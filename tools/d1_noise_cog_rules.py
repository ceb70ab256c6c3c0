#!/usr/bin/env python3
"""Which centroid / noise rule does the recorded COMPASS run of the noisy configuration imply?  (VERDICT r2 #2e)

The reference holds ONE statistics file of a noisy loop, normalization_..._d1_noise_zernike_space.pickle
(byte for byte the _noise_M9 file: magnitude 9, 3 e- read-out noise; tools/import_norm_data.py).  With the plain
centre of gravity (oracle/aoref.c:aoref_cog: slope = sum(x I) / sum(I) whenever sum(I) != 0) this build
reproduces the recorded SLOPE statistics at the parameter file's gain 0.65 (ratio 1.000) but not the COMMAND
statistics (2.9x).  COMPASS's centroider source is not in the reference tree; this script runs the reference's
normalisation recipe (20 seeds x 1000 integrator frames, 5 filtered modes) under a family of candidate rules
applied to the SAME noisy spot images (bincube of the one-pass frame kernel), and prints the four ratios
(this build / recorded) for each, at the file's gain and at the gains of the other parameter files of the
family (d0_noise and noise_M9_geo: 0.3):

  plain          sum(I) != 0 ? sum(x I) / sum(I) : 0                          (the product's rule)
  eps            sum(x I) / (sum(I) + 1e-6)                                   (COMPASS 5 get_centroids, from memory)
  clip0          pixels below 0 set to 0 first
  floor f        denominator max(sum(I), f * nominal flux of the sub-aperture)
  zero f         slope 0 when sum(I) < f * nominal flux
  thresh T       pixels -> max(I - T, 0)  (thresholded COG)

    python tools/d1_noise_cog_rules.py [--frames 1000]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    import torch
    from ao_marl_amd.env import VecRlSupervisor, load_norm
    from ao_marl_amd.normalization import _Stats, KEYS
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--config", default="production_sh_40x40_8m_3layers_d1_noise",
                    help="parameter set to run (the statistics compared with are always the _d1_noise file's)")
    ap.add_argument("--gains", default="0.65,0.3")
    ap.add_argument("--rules", default="all")
    ap.add_argument("--blocks", type=int, default=1,
                    help="repeat every (rule, gain) on this many disjoint blocks of 20 seeds (1..20, 21..40, ...): the "
                         "command statistics of a heavy-tailed loop are luck-dominated, the spread over blocks shows it")
    a = ap.parse_args()
    name = a.config
    ref, zn_ref = load_norm("production_sh_40x40_8m_3layers_d1_noise")
    nm = zn_ref.shape[0]
    live = np.arange(nm) < nm - 5 - 2
    live[-2:] = True
    sup = VecRlSupervisor(name, dict(n_reverse_filtered_from_cmat=5), 20, initial_seed=1, seed_stride=1,
                          prefetch_atmos=False, keep_bincube=True)
    sim, s = sup.sim, sup.s
    dev = sim.device
    npix = s.npix
    xs = torch.arange(npix, device=dev, dtype=torch.float32).repeat(npix)               # x of pixel p = y * npix + x
    ys = torch.arange(npix, device=dev, dtype=torch.float32).repeat_interleave(npix)
    nominal = torch.as_tensor(float(s.nphot) * np.asarray(s.flux, dtype=np.float32), device=dev)   # [nvalid]
    off, sc = float(s.cog_offset), float(s.cog_scale)

    def cog(cube, rule, par):
        if rule == "clip0":
            cube = cube.clamp(min=0.0)
        elif rule == "thresh":
            cube = (cube - par).clamp(min=0.0)
        s0 = cube.sum(dim=2)
        sx = (cube * xs).sum(dim=2)
        sy = (cube * ys).sum(dim=2)
        if rule == "eps":
            den = s0 + 1e-6
            ok = torch.ones_like(s0, dtype=torch.bool)
        elif rule == "floor":
            den = torch.maximum(s0, par * nominal)
            ok = torch.ones_like(s0, dtype=torch.bool)
        elif rule == "zero":
            den = s0
            ok = s0 >= par * nominal
        else:
            den = s0
            ok = s0 != 0
        den = torch.where(ok, den, torch.ones_like(den))
        gx = torch.where(ok, (sx / den - off) * sc, torch.zeros_like(sx))
        gy = torch.where(ok, (sy / den - off) * sc, torch.zeros_like(sy))
        return torch.cat([gx, gy], dim=1)

    def run(rule, par, gain, block=0):
        sup.set_gain(gain)
        sup.set_sim_seed(1 + 20 * block)
        stats = {"wfs": _Stats(s.nslope, dev), "dm": _Stats(sup.nmodes, dev), "dm_residual": _Stats(sup.nmodes, dev)}
        sup.reset()
        sim.defer_shape = True
        # per-ENVIRONMENT moments of the command modes: the median over the 20 environments is a statistic a
        # single mega-outlier (one environment kicked off by a centroid of a near-zero-flux spot) cannot move
        e1 = torch.zeros(sim.nenv, sup.nmodes, dtype=torch.float64, device=dev)
        e2 = torch.zeros_like(e1)
        for _ in range(a.frames):
            sim.move_atmos()
            sim.frame_fused(noise=True, write_bincube=True, cog=False)
            sim.t["slopes"].copy_(cog(sim.t["bincube"], rule, par))
            sim.do_control()
            sup.next_part_two(None, linear_control=True)
            stats["wfs"].update(sup.get_slopes())
            cm = sim.volts2modes(sup.get_command())
            stats["dm"].update(cm)
            e1 += cm.double()
            e2 += cm.double() ** 2
            stats["dm_residual"].update(sim.volts2modes(sup.get_err()))
        norm = {k: stats[k].result() for k in KEYS}
        env_std = (e2 / a.frames - (e1 / a.frames) ** 2).clamp(min=0).sqrt()            # [nenv, nmodes]
        quiet = env_std.median(dim=0).values.float().cpu().numpy()
        worst = float((env_std.max(dim=0).values / env_std.median(dim=0).values)[torch.as_tensor(live, device=dev)].median())
        zn = (np.abs(norm["dm"]["max"]) + np.abs(norm["dm"]["min"])) / 2.0
        sr = sup.get_strehl()[:, 1].cpu().numpy()
        return dict(wfs=float(np.median(norm["wfs"]["std"] / ref["wfs"]["std"])),
                    dm=float(np.median(norm["dm"]["std"][live] / ref["dm"]["std"][live])),
                    res=float(np.median(norm["dm_residual"]["std"][live] / ref["dm_residual"]["std"][live])),
                    zn=float(np.median(zn[live] / zn_ref[live])), sr=float(sr.mean()),
                    quiet=float(np.median(quiet[live] / ref["dm"]["std"][live])), worst=worst,
                    kurt=float(np.median((norm["wfs"]["max"] - norm["wfs"]["min"]) / norm["wfs"]["std"])),
                    k99=float(np.percentile((norm["wfs"]["max"] - norm["wfs"]["min"]) / norm["wfs"]["std"], 99)))

    rr = (ref["wfs"]["max"] - ref["wfs"]["min"]) / ref["wfs"]["std"]
    print("parameter set %s (delay %g), statistics compared with: _d1_noise (= _noise_M9) pickle" % (name, s.delay))
    print("recorded (COMPASS): (max - min) / std per slope: median %.1f, 99th percentile %.1f  (a Gaussian gives ~8: the "
          "recorded run has heavy-tailed centroids too)" % (float(np.median(rr)), float(np.percentile(rr, 99))))
    print("%-14s %5s | %7s %7s %7s %7s %7s %9s %9s %9s %9s" % ("rule", "gain", "slopes", "command", "resid.", "zn_norm", "SR_LE",
                                                              "rng/std50", "rng/std99", "cmd(med.env)", "worst/med"))
    rules = [("plain", 0.0), ("eps", 0.0), ("clip0", 0.0), ("floor", 0.25), ("floor", 0.5), ("zero", 0.25), ("zero", 0.5),
             ("thresh", 3.0), ("thresh", 6.0)]
    if a.rules != "all":
        rules = [r for r in rules if r[0] in a.rules.split(",")]
    for gain in [float(g) for g in a.gains.split(",")]:
        for rule, par in [rp for rp in rules for _ in range(a.blocks)]:
            blk = run.count = getattr(run, "count", -1) + 1
            r = run(rule, par, gain, blk % a.blocks)
            tag = rule if rule in ("plain", "eps", "clip0") else "%s %g" % (rule, par)
            if a.blocks > 1:
                tag += " b%d" % (blk % a.blocks)
            ok = all(abs(r[k] - 1) < t for k, t in (("wfs", 0.1), ("dm", 0.1), ("res", 0.1), ("zn", 0.2)))
            print("%-14s %5.2f | %7.3f %7.3f %7.3f %7.3f %7.3f %9.1f %9.1f %9.3f %9.1f %s" %
                  (tag, gain, r["wfs"], r["dm"], r["res"], r["zn"], r["sr"], r["kurt"], r["k99"], r["quiet"], r["worst"],
                   "<-- all within tolerance" if ok else ""),
                  flush=True)


if __name__ == "__main__":
    main()
