import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


_SESSION_CACHE = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")
    # calibrations memoised on disk (ao_marl_amd/modal.py) in a directory of THIS session: tests never replay what an
    # earlier checkout left in the user's persistent cache, and the ranks a test spawns still share one (inherited)
    global _SESSION_CACHE
    if "AOMARL_CALIB_CACHE" not in os.environ:
        import tempfile
        _SESSION_CACHE = tempfile.mkdtemp(prefix="aomarl_calib_test_")
        os.environ["AOMARL_CALIB_CACHE"] = _SESSION_CACHE


def pytest_unconfigure(config):
    global _SESSION_CACHE
    if _SESSION_CACHE:
        import shutil
        shutil.rmtree(_SESSION_CACHE, ignore_errors=True)
        os.environ.pop("AOMARL_CALIB_CACHE", None)
        _SESSION_CACHE = None


def _gpu_unavailable_reason():
    """Why gpu-marked tests cannot run here, or None.  On a GPU box a missing library is NOT a
    reason to skip: those tests must then fail loudly (the product has no fallback)."""
    try:
        import torch
        if not torch.cuda.is_available():
            return "no GPU in this process (torch.cuda.is_available() is False)"
    except Exception as e:                                   # pragma: no cover
        return "torch unavailable: %s" % e
    return None


def pytest_collection_modifyitems(config, items):
    reason = _gpu_unavailable_reason()
    if reason is None:
        return
    skip = pytest.mark.skip(reason=reason)
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _seeded_generators():
    """Every test starts from the same torch / NumPy global generator state: tests that draw inputs from the global
    generators (tolerances a few units of fp32 round-off wide) are then the same test every run."""
    import numpy as np
    try:
        import torch
        torch.manual_seed(20260104)      # CPU and (lazily) every CUDA generator; does not initialise the GPU
    except ImportError:
        pass
    np.random.seed(20260104)
    yield
