#!/usr/bin/env python3
"""Integer-pixel frozen flow vs a sub-pixel remainder in raytrace (VERDICT r2 #2d), against every
noise-free statistics file the reference recorded from real COMPASS.

COMPASS's native source is not in the reference tree, so whether its raytrace applies the fractional
remainder of the wind accumulators as a sub-pixel shift (SURVEY Appendix D(1),(3)) cannot be read off; the
recorded statistics can vote.  For each parameter file the reference's normalisation recipe
(obtain_normalization.py:139-243: seeds 1..20 x 1000 integrator frames, 5 filtered modes) runs three times:

  one-pass   the product path (integer flow, one-pass frame kernel)
  integer    integer flow through the generic path (bilinear raytrace into phase buffers, spot kernel on the
             buffer): same physics as `one-pass`, different kernels -- the control of the comparison
  subpixel   the same generic path with aomarl_set_option("subpixel_flow", 1): every layer window shifted by
             the remainder of its accumulator (bilinear interpolation)

and prints the median ratio (this build / COMPASS) of the per-slope and per-mode standard deviations.
    python tools/subpixel_flow_experiment.py [--frames 1000] [names ...]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def loop_generic(sup, frames, subpixel):
    from ao_marl_amd import libaomarl as la
    from ao_marl_amd.normalization import _Stats, KEYS
    sim = sup.sim
    sim.defer_shape = False
    sim.set_option("subpixel_flow", int(bool(subpixel)))
    dev = sim.device
    stats = {"wfs": _Stats(sup.s.nslope, dev), "dm": _Stats(sup.nmodes, dev), "dm_residual": _Stats(sup.nmodes, dev)}
    sup.reset()
    for _ in range(frames):
        sim.move_atmos()
        sim.raytrace_target()
        la.check(sim.lib.aomarl_target_psf_buffer(sim.ctx, C.byref(sim.st), 0, sim.nenv, sim._stream()))
        sim.raytrace_wfs()
        sim.comp_image(from_phase_buffer=True, noise=True, cog=True)
        sim.do_control()
        sup.next_part_two(None, linear_control=True)
        stats["wfs"].update(sup.get_slopes())
        stats["dm"].update(sim.volts2modes(sup.get_command()))
        stats["dm_residual"].update(sim.volts2modes(sup.get_err()))
    sim.set_option("subpixel_flow", 0)
    norm = {k: stats[k].result() for k in KEYS}
    zn = (np.abs(norm["dm"]["max"]) + np.abs(norm["dm"]["min"])) / 2.0
    return norm, zn.astype(np.float32), sup.get_strehl()[:, 1].cpu().numpy()


def ratios(norm, zn, sr, ref, zn_ref):
    nm = zn_ref.shape[0]
    live = np.arange(nm) < nm - 5 - 2
    live[-2:] = True
    return dict(wfs=float(np.median(norm["wfs"]["std"] / ref["wfs"]["std"])),
                dm=float(np.median(norm["dm"]["std"][live] / ref["dm"]["std"][live])),
                res=float(np.median(norm["dm_residual"]["std"][live] / ref["dm_residual"]["std"][live])),
                zn=float(np.median(zn[live] / zn_ref[live])), sr=float(sr.mean()))


def main():
    from ao_marl_amd import normalization as N, params
    from ao_marl_amd.env import VecRlSupervisor, load_norm
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--frames", type=int, default=1000)
    a = ap.parse_args()
    L = "production_sh_40x40_8m_3layers"
    names = a.names or [L, L + "_dir_0_15_30", L + "_same_dir", L + "_v_20_15_25", L + "_dir_0_15_30_v_10_5_15",
                        L + "_dir_0_15_30_v_20_15_25", "production_sh_10x10_2m"]
    print("%-52s %-9s %7s %7s %7s %7s %7s" % ("parameter file", "flow", "slopes", "command", "resid.", "zn_norm", "SR_LE"))
    score = {"one-pass": [], "integer": [], "subpixel": []}
    for name in names:
        ref, zn_ref = load_norm(name)
        for mode in ("one-pass", "integer", "subpixel"):
            sup = VecRlSupervisor(name, dict(n_reverse_filtered_from_cmat=5), 20, initial_seed=1, seed_stride=1,
                                  prefetch_atmos=False, keep_bincube=False)
            if mode == "one-pass":
                out = N.normalization_loop(sup, frames=a.frames)
            else:
                out = loop_generic(sup, a.frames, mode == "subpixel")
            r = ratios(*out, ref, zn_ref)
            score[mode].append(r)
            print("%-52s %-9s %7.4f %7.4f %7.4f %7.4f %7.4f" % (name, mode, r["wfs"], r["dm"], r["res"], r["zn"], r["sr"]),
                  flush=True)
            del sup
    print()
    for mode, rs in score.items():
        dev = {k: float(np.mean([abs(np.log(r[k])) for r in rs])) for k in ("wfs", "dm", "res", "zn")}
        print("%-9s mean |log ratio|: slopes %.4f  command %.4f  residual %.4f  zn_norm %.4f" %
              (mode, dev["wfs"], dev["dm"], dev["res"], dev["zn"]))


if __name__ == "__main__":
    main()
