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


# ------------------------------------------------------------------ the call sequence of a ccall shim, from plain C
def _build_harness(tmp_path):
    import subprocess
    from carparkingmaps_amd import _lib
    exe = str(tmp_path / "abi_harness")
    csrc = os.path.dirname(_lib.LIB_PATH)
    cmd = ["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-std=gnu11", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "abi_harness.c"), "-o", exe, "-L" + csrc, "-lcpm_hip", "-Wl,-rpath," + csrc,
           "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_c_harness_compiles_against_the_header_and_links_every_symbol(cpm, tmp_path):
    """tests/abi_harness.c calls every entry through include/cpm.h's prototypes (-Wall -Wextra -Werror) and links against the library
    directly: a prototype that drifted from the definition, or a symbol the header declares and the library lacks, fails here."""
    import subprocess
    exe = _build_harness(tmp_path)
    out = subprocess.check_output([exe, "symbols"], text=True)
    n, total = map(int, re.search(r"symbols (\d+) of (\d+)", out).groups())
    assert n == total == len(_declared())


def _fnv(a):
    import numpy as np
    h = 1469598103934665603
    for b in np.ascontiguousarray(a).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.gpu
def test_c_harness_call_sequence_equals_the_ctypes_path(cpm, tmp_path):
    """main.jl:82-102 as a `ccall` would drive it -- Ref{Ptr} / Ref{Int64} out-parameters, column-major host arrays, C_NULL for the
    optional matrices -- from a C program, against the same sequence through the Python host layer: identical post-IVP state,
    counts, state_matrix and transition_matrix."""
    import subprocess
    import numpy as np
    Z, cpz, T, seed = 48, 40, 24, 0x5EEDCA125
    C = Z * cpz
    exe = _build_harness(tmp_path)
    out = subprocess.check_output([exe, str(Z), str(cpz), hex(seed)], text=True)
    got = dict(re.findall(r"(\w+) ([0-9a-f-]+)", out))
    # the harness's tables
    o, d, t = np.meshgrid(np.arange(Z), np.arange(Z), np.arange(T), indexing="ij")
    w = np.where(o == d, 0.0, (1 + (o * 31 + d * 17 + t * 5) % 23).astype(np.float64))
    tot = np.zeros((Z, T))
    for dd in range(Z):                                   # the harness sums left to right
        tot += w[:, dd, :]
    p_dest = np.asfortranarray(w / tot[:, None, :])
    oo, tt_ = np.meshgrid(np.arange(Z), np.arange(T), indexing="ij")
    p_drive = np.asfortranarray(0.1 + 0.8 * ((oo * 7 + tt_ * 13) % 97) / 96.0)
    with cpm.Sampler(Z, T) as s:
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.init_states(C, cpz)
        s.set_state(np.arange(C, dtype=np.int64) // cpz + 1)
        init = s.solve_ivp(seed)
        r = s.resample(seed)
        r2 = s.resample(seed, want_state=True, want_trans=True)
        kernel = s.get_info(1)
    assert int(got["Z"]) == Z and int(got["C"]) == C and int(got["kernel"]) == kernel
    assert int(got["null_state_status"]) == -1            # CPM_ERR_ARG: a status and a message, no abort
    assert int(got["initial_state"], 16) == _fnv(init)
    assert int(got["parking"], 16) == _fnv(r["parking"].ravel(order="F")) and int(got["driving"], 16) == _fnv(r["driving"].ravel(order="F"))
    assert int(got["tt"]) == 0 and int(got["compat_counts_equal"]) == 1 and int(got["hour24_cars"]) == C
    assert int(got["state"], 16) == _fnv(r2["state"].ravel(order="F")) and int(got["trans"], 16) == _fnv(r2["trans"].ravel(order="F"))


def test_every_lds_dma_sampler_waits_for_its_pack_before_the_barrier(cpm):
    """`make asm-check`: in the device assembly of every instantiation of the kernels that stage a row pack by LDS-DMA there is an
    s_waitcnt vmcnt(0) between the last global_load_lds and the first s_barrier behind it (tools/check_sampler_asm.py)."""
    import subprocess
    from carparkingmaps_amd import _lib
    out = subprocess.run(["make", "-C", os.path.dirname(_lib.LIB_PATH), "asm-check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    n, bad = map(int, re.search(r"(\d+) kernels with LDS-DMA staging checked, (\d+) without", out.stdout).groups())
    assert n >= 60 and bad == 0
