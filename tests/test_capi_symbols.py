"""The C-ABI library loads on a CPU-only box and exports every symbol include/aomarl.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "aomarl.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(aomarl_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from ao_marl_amd import libaomarl
    if not os.path.exists(libaomarl.LIB_PATH):
        libaomarl.build()
    lib = libaomarl.load()
    names = _declared()
    assert len(names) >= 30
    bound = {n for n, _, _ in libaomarl.SYMBOLS}
    raw = ctypes.CDLL(libaomarl.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libaomarl_hip.so does not export %s" % n
        assert n in bound, "ctypes binding misses %s" % n
    assert lib.aomarl_abi_version() == libaomarl.ABI_VERSION


def test_struct_sizes_match_the_header():
    """Desc/State field order is the header's: compile a probe with the host compiler."""
    import subprocess
    import tempfile
    from ao_marl_amd import libaomarl as la
    src = ('#include <stdio.h>\n#include "aomarl.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
           'sizeof(aomarl_desc),sizeof(aomarl_state),sizeof(aomarl_dm_desc),'
           'sizeof(aomarl_layer_desc),sizeof(aomarl_actor_desc),sizeof(aomarl_env_glue),'
           'sizeof(aomarl_sac_desc));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o",
                               os.path.join(d, "p"), os.path.join(d, "p.c")])
        out = subprocess.check_output([os.path.join(d, "p")]).decode().split()
    assert [int(x) for x in out] == [ctypes.sizeof(la.Desc), ctypes.sizeof(la.State),
                                     ctypes.sizeof(la.DmDesc), ctypes.sizeof(la.LayerDesc),
                                     ctypes.sizeof(la.ActorDesc), ctypes.sizeof(la.EnvGlue),
                                     ctypes.sizeof(la.SacDesc)]


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ao_marl_amd import libaomarl
    from ao_marl_amd.sim import HipSim
    with pytest.raises(libaomarl.AomarlError):
        HipSim(object(), 1)


def test_product_package_never_imports_the_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "ao_marl_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import aoref" not in txt and "from oracle" not in txt and \
                    "libaoref" not in txt, os.path.join(dp, f)
