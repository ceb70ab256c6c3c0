"""Shims that make the reference's *pure-Python* half importable in the BUILD container.

Test infrastructure only (used by tools/gen_golden*.py).  Never imported by the product
package, never shipped to the GPU box as a dependency: it needs /root/reference.

What is shimmed: missing third-party modules (h5py, astropy, gym,
torchvision, tensorboardX), NumPy-2 removals (np.math/np.bool/np.long/np.int, ndarray.itemset),
SHESHA_ROOT.
"""
import math
import os
import sys
import types

REF = os.environ.get("AOMARL_REFERENCE", "/root/reference")


def install(ref=REF):
    import numpy as np
    os.environ.setdefault("SHESHA_ROOT", ref)
    sys.dont_write_bytecode = True
    if ref not in sys.path:
        sys.path.insert(0, ref)
    for name in ("h5py", "astropy", "astropy.io", "astropy.io.fits", "torchvision",
                 "torchvision.transforms", "tensorboardX", "docopt"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["astropy"], "io"):
        sys.modules["astropy"].io = sys.modules["astropy.io"]
        sys.modules["astropy.io"].fits = sys.modules["astropy.io.fits"]
    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")

        class Env(object):
            pass

        spaces = types.ModuleType("gym.spaces")

        class Box(object):
            def __init__(self, low=None, high=None, shape=None, dtype=None):
                self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

        spaces.Box = Box
        gym.Env = Env
        gym.spaces = spaces
        sys.modules["gym"] = gym
        sys.modules["gym.spaces"] = spaces
    class _Math(object):
        """np.math proxy: Python >= 3.10 refuses math.factorial(2.0) (dm_util.py:355-362)."""

        def __getattr__(self, k):
            return getattr(math, k)

        @staticmethod
        def factorial(x):
            return math.factorial(int(x))

    for k, v in (("math", _Math()), ("bool", bool), ("long", int), ("int", int),
                 ("float", float)):
        if not hasattr(np, k):
            setattr(np, k, v)

    # iterkolmo uses ndarray.itemset (removed in NumPy 2)
    import shesha.util.iterkolmo as itk

    class _A(np.ndarray):
        def itemset(self, idx, val):
            self.reshape(-1)[idx] = val

    class _NP(object):
        def __getattr__(self, k):
            return getattr(np, k)

        @staticmethod
        def zeros(*a, **k):
            return np.zeros(*a, **k).view(_A)

    itk.np = _NP()
    return ref
