"""CPU: the C-ABI library loads and exports every symbol include/v21.h declares
(no compute calls: there is no GPU in the build container)."""
import os
import re

import pytest

from conftest import ROOT, pkg


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "v21.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(v21_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    native = pkg("_native")
    if not os.path.exists(native.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = native.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), "libv21.so lacks %s" % name
    # and the binding knows a prototype for each of them
    assert set(declared) == set(native.SIGNATURES), set(declared) ^ set(native.SIGNATURES)
    assert lib.v21_version() == 100


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a device the product refuses to run; it never computes on the CPU."""
    native = pkg("_native")
    import ctypes as C
    lib = native.load_library()
    n = C.c_int(0)
    st = lib.v21_device_count(C.byref(n))
    if st == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(native.EngineUnavailable):
        native.Context(0)


def test_oracle_is_not_imported_by_the_product():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "21cmvae_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
