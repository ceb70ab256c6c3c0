#!/usr/bin/env python3
"""Convert the reference's recorded normalisation DATA (per-slope / per-mode statistics of
20 x 1000 integrator frames of real COMPASS, obtain_normalization.py:139-250, and the action
bounds zn_norm_*.npy) into .npz files the environment loads at run time
(ao_env.py:251-306, rlSupervisor.py:255-282).  Data only -- no reference code is copied.
Build container only (needs /root/reference)."""
import os
import pickle
import sys

import numpy as np

REF = "/root/reference/src/reinforcement_learning/helper_functions/preprocessing/normalization"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ao_marl_amd", "data")
L = "production_sh_40x40_8m_3layers"
# every statistics file the reference holds (16 pickles; _noise_M9 is byte-for-byte the _d1_noise data
# under the name of a parameter file that is not in the tree: imported once, as _d1_noise)
NAMES = ["production_sh_10x10_2m", L, L + "_d1_noise"] + [L + s for s in (
        "_dir_0_15_30", "_dir_0_15_30_v_10_5_15", "_dir_0_15_30_v_20_15_25", "_same_dir",
        "_same_dir_v_10_5_15", "_same_dir_v_20_15_25", "_v_10_5_15", "_v_20_15_25", "_same_dir_roket",
        "_same_dir_gain_change_high", "_same_dir_gain_change_low")]


class _NumpyOnly(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] == "numpy":
            return super().find_class(module, name)
        raise pickle.UnpicklingError("refusing %s.%s" % (module, name))


def main():
    os.makedirs(OUT, exist_ok=True)
    for n in NAMES:
        with open(os.path.join(REF, "state_normalization", "normalization_%s_zernike_space.pickle" % n), "rb") as f:
            d = _NumpyOnly(f).load()
        out = {}
        for key, sub in d.items():
            for stat, arr in sub.items():
                out["%s_%s" % (key, stat)] = np.asarray(arr, dtype=np.float32)
        out["zn_norm"] = np.load(os.path.join(REF, "normalization_action_zernike", "zn_norm_%s.npy" % n)).astype(np.float32)
        np.savez_compressed(os.path.join(OUT, "norm_%s.npz" % n), **out)
        print(n, {k: v.shape for k, v in out.items()})
    # the d0 noise file ships no statistics of its own; the reference's README runs use d1
    a = _load(L + "_noise_M9")
    b = _load(L + "_d1_noise")
    same = all(np.array_equal(a[k][st], b[k][st]) for k in a for st in a[k])
    print("noise_M9 pickle == d1_noise pickle:", same)
    return 0


def _load(n):
    with open(os.path.join(REF, "state_normalization", "normalization_%s_zernike_space.pickle" % n), "rb") as f:
        return _NumpyOnly(f).load()


if __name__ == "__main__":
    sys.exit(main())
