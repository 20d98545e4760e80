"""Loader of the product library carparkingmaps_amd/csrc/libcpm_hip.so (C ABI: include/cpm.h).

There is no CPU fallback: if the library is missing or no HIP device is usable, every
compute entry point raises.  The oracle under oracle/ is never imported from here.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("CPM_LIB_PATH") or os.path.join(CSRC, "libcpm_hip.so")  # override: diagnostic twin, tools only

# every symbol include/cpm.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "cpm_last_error", "cpm_version", "cpm_device_count", "cpm_device_info", "cpm_create", "cpm_destroy",
    "cpm_set_option", "cpm_set_stream", "cpm_sync", "cpm_set_p_drive", "cpm_set_p_dest", "cpm_set_datamatrix",
    "cpm_build_p_drive", "cpm_build_p_dest", "cpm_get_p_drive", "cpm_get_cdf_row", "cpm_init_states",
    "cpm_set_state", "cpm_get_state", "cpm_solve_ivp", "cpm_resample", "cpm_resample_dev",
    "cpm_solve_ivp_async", "cpm_synth_tables", "cpm_last_kernel_ms", "cpm_algorithmic_bytes_per_hour",
    "cpm_debug_categorical", "cpm_createdatamatrix_rows", "cpm_createdatamatrix_csv", "cpm_get_datamatrix",
    "cpm_set_distance_from_centroids", "cpm_get_distance", "cpm_parse_uber_csv", "cpm_set_distance", "cpm_get_info",
    "cpm_init_states_strided", "cpm_synth_tables_skewed", "cpm_synth_datamatrix", "cpm_refresh_tables",
]

CPM_FLAG_TRAVEL = 1
CPM_KERNEL_AUTO, CPM_KERNEL_CAR, CPM_KERNEL_ZONE_LDS = 0, 1, 2
CPM_KERNEL_ZONE_GROUPED = 5
CPM_OPT_KERNEL, CPM_OPT_PROFILE, CPM_OPT_PROFILE_KERNEL, CPM_OPT_FUSED, CPM_OPT_FUSED_LAG, CPM_OPT_ZONE_ORDER = 1, 2, 3, 4, 5, 6

_lib = None


class CpmError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libcpm_hip status {status}: {message}")
        self.status = status


def build(force=False):
    """Compile libcpm_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) or f == "Makefile"]
    srcs.append(os.path.join(_HERE, "..", "include", "cpm.h"))
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    subprocess.check_call(["make", "-C", CSRC, "libcpm_hip.so"] + (["-B"] if force else []))
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  carparkingmaps_amd has no CPU fallback.")
    # One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64 (same
    # SONAME as /opt/rocm's).  If this library is loaded first it pulls in the system copies, a later
    # `import torch` adds the bundled ones, and the second runtime's initialisation can fail ("No HIP
    # GPUs are available", seen after a long pytest session).  Importing torch first makes the dynamic
    # loader bind libcpm_hip.so to the copy that is already mapped.  Without torch (e.g. under the Julia
    # shim) the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    i32, i64, u32, u64, dbl, vp = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_double, C.c_void_p
    L.cpm_last_error.restype = C.c_char_p
    L.cpm_version.restype = i32
    L.cpm_device_count.argtypes = [C.POINTER(i32)]
    L.cpm_device_info.argtypes = [i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i64)]
    L.cpm_create.argtypes = [C.POINTER(vp), i64, i64, i32]
    L.cpm_destroy.argtypes = [vp]
    L.cpm_set_option.argtypes = [vp, i32, i64]
    L.cpm_set_stream.argtypes = [vp, vp]
    L.cpm_sync.argtypes = [vp]
    L.cpm_set_p_drive.argtypes = [vp, vp]
    L.cpm_set_p_dest.argtypes = [vp, vp]
    L.cpm_set_datamatrix.argtypes = [vp, vp, vp]
    L.cpm_build_p_drive.argtypes = [vp, dbl, dbl, dbl, vp]
    L.cpm_build_p_dest.argtypes = [vp, dbl, i32, vp]
    L.cpm_get_p_drive.argtypes = [vp, vp]
    L.cpm_get_cdf_row.argtypes = [vp, i64, i64, vp]
    L.cpm_init_states.argtypes = [vp, i64, i64, i64, i64]
    L.cpm_init_states_strided.argtypes = [vp, i64, i64, i64, i64, i64]
    L.cpm_set_state.argtypes = [vp, vp]
    L.cpm_get_state.argtypes = [vp, vp]
    L.cpm_solve_ivp.argtypes = [vp, u64, vp]
    L.cpm_resample.argtypes = [vp, u64, u32, vp, vp, vp, vp, vp]
    L.cpm_resample_dev.argtypes = [vp, u64, u32, vp]
    L.cpm_solve_ivp_async.argtypes = [vp, u64]
    L.cpm_synth_tables.argtypes = [vp, u64]
    L.cpm_synth_tables_skewed.argtypes = [vp, u64, i64]
    L.cpm_synth_datamatrix.argtypes = [vp, u64, dbl]
    L.cpm_refresh_tables.argtypes = [vp, i32]
    L.cpm_last_kernel_ms.argtypes = [vp, vp, i32, C.POINTER(i32)]
    L.cpm_algorithmic_bytes_per_hour.argtypes = [vp, C.POINTER(i64)]
    L.cpm_debug_categorical.argtypes = [vp, i64, i64, i64, vp, vp, C.POINTER(i32)]
    L.cpm_createdatamatrix_rows.argtypes = [vp, i64, vp]
    L.cpm_createdatamatrix_csv.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
    L.cpm_get_datamatrix.argtypes = [vp, vp]
    L.cpm_parse_uber_csv.argtypes = [C.c_char_p, C.POINTER(i64), vp, i64]
    L.cpm_set_distance_from_centroids.argtypes = [vp, vp, vp]
    L.cpm_get_distance.argtypes = [vp, vp]
    L.cpm_set_distance.argtypes = [vp, vp]
    L.cpm_get_info.argtypes = [vp, i32, C.POINTER(i64)]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("cpm_last_error",):
            fn.restype = i32
    _lib = L
    return L


def check(status):
    if status != 0:
        raise CpmError(status, load().cpm_last_error().decode("utf-8", "replace"))
