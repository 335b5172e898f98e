"""The C-ABI library loads on a machine without a GPU and exports every symbol include/vbnmf.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "vbnmf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vbnmf_[a-z0-9_]+)\s*\(", text)))


def test_header_functions_are_exported_and_bound():
    from ccfindr_amd import _native as N
    L = N.load()
    names = declared_functions()
    assert len(names) >= 30
    for name in names:
        assert hasattr(L, name), f"{name} is declared in include/vbnmf.h but not exported"
        assert name in N.SIGNATURES, f"{name} has no ctypes signature in ccfindr_amd/_native.py"
    for name in N.SIGNATURES:
        assert name in names, f"{name} is bound but not declared in include/vbnmf.h"


def test_library_is_the_in_tree_hip_build():
    from ccfindr_amd import _native as N
    assert os.path.realpath(N.LIB_PATH).startswith(os.path.realpath(ROOT))
    L = ctypes.CDLL(N.LIB_PATH)
    assert L.vbnmf_version
    # the code object for gfx950 is embedded in the shared object
    blob = open(N.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_no_device_is_a_status_not_a_crash():
    import numpy as np
    import ccfindr_amd as C
    if C.load().vbnmf_device_count() > 0:
        pytest.skip("a GPU is present")
    M = C.CountMatrix(np.ones((4, 5)))
    with pytest.raises(C.VBNMFError) as ei:
        C.VBEngine(M, 2)
    assert ei.value.code == 2 and "no CPU fallback" in str(ei.value)
    with pytest.raises(C.VBNMFError):
        C.vbnmf_update(np.ones((4, 5)), {"lw": np.ones((4, 2)), "lh": np.ones((2, 5)), "eh": np.ones((2, 5))},
                       {"aw": 1, "bw": 1, "ah": 1, "bh": 1})


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ccfindr_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "vbnmf_oracle" not in text, f
