"""The C-ABI library loads without a GPU and exports every symbol include/cpm.h declares; with
no usable HIP device every compute entry fails loudly (there is no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "cpm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cpm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(cpm):
    from carparkingmaps_amd import _lib
    declared = _declared()
    assert declared, "no declarations parsed from include/cpm.h"
    assert sorted(_lib.SYMBOLS) == declared
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/cpm.h but not exported"


def test_library_is_in_tree_and_has_no_torch_dependency(cpm):
    from carparkingmaps_amd import _lib
    assert os.path.realpath(_lib.LIB_PATH).startswith(os.path.realpath(ROOT))
    needed = os.popen(f"readelf -d {_lib.LIB_PATH} 2>/dev/null").read()
    assert "libamdhip64" in needed
    assert "torch" not in needed and "liboracle" not in needed


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "carparkingmaps_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in src and "cpm_oracle" not in src, f
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_no_gpu_means_loud_failure(cpm):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(cpm.CpmError) as e:
        cpm.Sampler(8, 24)
    assert e.value.status == -2
    assert "no" in str(e.value).lower()


def test_version_and_last_error(cpm):
    from carparkingmaps_amd import _lib
    L = _lib.load()
    assert L.cpm_version() >= 100
    assert L.cpm_destroy(None) == 0
    assert L.cpm_sync(None) == -1
    assert b"null context" in L.cpm_last_error()
