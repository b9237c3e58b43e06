"""CPU: the C-ABI library loads and exports every symbol include/stardist_hip.h declares (no compute)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "stardist_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(sd_[a-z0-9_]+|_LIB_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_header_symbols_are_exported_and_bound():
    import __graft_entry__
    from stardist_amd.build import build_lib
    build_lib(verbose=False)
    from stardist_amd.lib import _native
    lib = _native.lib()
    decl = _declared_symbols()
    assert len(decl) >= 18
    for name in decl:
        assert hasattr(lib, name), "symbol %s declared in stardist_hip.h but not exported" % name
        assert name in _native.SIGNATURES, "symbol %s has no ctypes signature" % name
    assert set(_native.SIGNATURES) == set(decl)
    assert lib.sd_version() == 1


def test_fails_loudly_without_gpu():
    """no CPU fallback: on a box without a HIP device every compute entry point raises"""
    import numpy as np
    import pytest
    from stardist_amd.lib import _native, stardist2d
    if _native.lib().sd_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_native.NativeError):
        stardist2d.c_non_max_suppression_inds(np.ones((2, 32), np.float32), np.ones((2, 2), np.float32), 1, 1, 0, 0.4)
    with pytest.raises(_native.NativeError):
        stardist2d.c_star_dist(np.ones((8, 8), np.uint16), 8, 1, 1)


def test_product_does_not_import_oracle():
    """the product path must never route through the oracle"""
    for dp, _, files in os.walk(os.path.join(ROOT, "stardist_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src.replace("oracle/_ref", "") \
                    or f == "clip_sweep.h" or "oracle" not in src, (dp, f)
