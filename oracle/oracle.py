"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the carparkingmaps_amd package.
Arrays are numpy, Fortran-ordered (the reference's Julia column-major layout),
zone ids 1-based.  See cpm_oracle.h for the contract and the "parity
unpinned" statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="F_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="F_CONTIGUOUS")


def build():
    """Compile liboracle.so with gcc (no-op when it is newer than its sources)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("cpm_oracle.c", "cpm_oracle.h")]
    if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return so
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.environ.get("CPM_ORACLE_SO") or os.path.join(_HERE, "liboracle.so")  # e.g. liboracle_asan.so
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    i64, u64, u32, dbl, vp = C.c_int64, C.c_uint64, C.c_uint32, C.c_double, C.c_void_p
    L.orc_philox4x32_10.argtypes = [C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.orc_philox4x32_10.restype = None
    L.orc_uniforms.argtypes = [u64, u64, u32, u32, C.POINTER(dbl), C.POINTER(dbl)]
    L.orc_uniforms.restype = None
    L.orc_exp_neg.argtypes = [dbl]
    L.orc_exp_neg.restype = dbl
    for name, n in (("orc_log", 1), ("orc_sqrt", 1), ("orc_erf", 1), ("orc_ppnd", 1), ("orc_truncnormal_mass", 2), ("orc_truncnormal_draw", 4)):
        getattr(L, name).argtypes = [dbl] * n
        getattr(L, name).restype = dbl
    L.orc_createpdrive.argtypes = [_f64p, _f64p, i64, i64, dbl, dbl, dbl, _f64p]
    L.orc_createpdestin.argtypes = [_f64p, i64, i64, dbl, C.c_int, _f64p]
    L.orc_initializestates.argtypes = [i64, i64, i64, i64, _i64p, vp]
    L.orc_solveinitialvalueproblem.argtypes = [_i64p, _f64p, _f64p, _f64p, i64, i64, i64, u64, i64, _i64p]
    L.orc_resampling.argtypes = [_i64p, _f64p, i64, i64, i64, _f64p, _f64p, vp, vp, u64, i64]
    L.orc_histogram.argtypes = [i64, i64, _i64p, _f64p, i64, _f64p, _f64p, vp, dbl]
    L.orc_trafficactivity.argtypes = [i64, i64, _f64p, _f64p]
    L.orc_averagedrivingtime.argtypes = [i64, i64, dbl, _f64p]
    L.orc_averagedrivingtime.restype = dbl
    L.orc_sum_travel_time_q16.argtypes = [i64, i64, _f64p]
    L.orc_sum_travel_time_q16.restype = i64
    L.orc_correctparameters.argtypes = [dbl, dbl, dbl, dbl, C.POINTER(dbl), C.POINTER(dbl)]
    L.orc_correctparameters.restype = None
    L.orc_build_cdf.argtypes = [_f64p, i64, i64, vp]
    L.orc_fast_run.argtypes = [_f64p, vp, i64, i64, i64, i64, i64, u64, C.c_int, _i64p, _i64p, _i64p,
                               vp, vp, vp, vp, C.c_int]
    L.orc_synth_p_drive.argtypes = [i64, i64, u64, _f64p]
    L.orc_synth_p_dest_dense.argtypes = [i64, i64, u64, _f64p]
    L.orc_synth_p_dest_skewed.argtypes = [i64, i64, u64, i64, _f64p]
    L.orc_synth_datamatrix.argtypes = [i64, i64, u64, dbl, _f64p, _f64p]
    L.orc_createdatamatrix.argtypes = [vp, i64, i64, i64, vp]
    L.orc_centroids.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp]
    L.orc_distance_matrix.argtypes = [vp, vp, i64, vp]
    L.orc_max_threads.restype = C.c_int
    _LIB = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with status {rc}")


# ---------------------------------------------------------------- RNG
def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def uniforms(seed, car, step, stream=0):
    a, b = C.c_double(), C.c_double()
    lib().orc_uniforms(seed, car, step, stream, C.byref(a), C.byref(b))
    return a.value, b.value


def exp_neg(y):
    return lib().orc_exp_neg(float(y))


# ---------------------------------------------------------------- tables
def createpdrive(datamatrix, dist, Z, T=24, p_min=0.1, p_max=0.9, e_drive=0.5):
    out = np.zeros((Z, T), dtype=np.float64, order="F")
    _check(lib().orc_createpdrive(datamatrix, dist, Z, T, p_min, p_max, e_drive, out), "createpdrive")
    return out


def createpdestin(datamatrix, Z, T=24, e_dest=2):
    out = np.zeros((Z, Z, T), dtype=np.float64, order="F")
    is_int = int(isinstance(e_dest, (int, np.integer)))
    _check(lib().orc_createpdestin(datamatrix, Z, T, float(e_dest), is_int, out), "createpdestin")
    return out


# ---------------------------------------------------------------- sampler (faithful)
def initializestates(C_, cars_per_zone, T=24, car_offset=0, with_trans=True):
    state = np.zeros((C_, T), dtype=np.int64, order="F")
    trans = np.zeros((C_, T, 4), dtype=np.float64, order="F") if with_trans else None
    _check(lib().orc_initializestates(C_, cars_per_zone, T, car_offset, state, _ptr(trans)), "initializestates")
    return state, trans


def solveinitialvalueproblem(state, trans, p_drive, p_dest, C_, Z, seed, car_offset=0):
    T = state.shape[1]
    init = np.zeros(C_, dtype=np.int64)
    _check(lib().orc_solveinitialvalueproblem(state, trans, p_drive, p_dest, C_, Z, T, seed, car_offset, init),
           "solveinitialvalueproblem")
    return init


def resampling(state, trans, C_, Z, p_drive, p_dest, datamatrix, dist, seed, car_offset=0):
    T = state.shape[1]
    _check(lib().orc_resampling(state, trans, C_, Z, T, p_drive, p_dest, _ptr(datamatrix), _ptr(dist),
                                seed, car_offset), "resampling")
    return state, trans


def histogram(Z, state, trans, C_norm=None):
    C_, T = state.shape
    parking = np.zeros((Z, T), dtype=np.float64, order="F")
    driving = np.zeros((Z, T), dtype=np.float64, order="F")
    density = np.zeros((Z, T), dtype=np.float64, order="F")
    _check(lib().orc_histogram(Z, T, state, trans, C_, parking, driving, _ptr(density),
                               float(C_ if C_norm is None else C_norm)), "histogram")
    return parking, driving, density


def trafficactivity(driving):
    Z, T = driving.shape
    act = np.zeros(T, dtype=np.float64)
    with np.errstate(all="ignore"):
        lib().orc_trafficactivity(Z, T, np.asfortranarray(driving, dtype=np.float64), act)
    return act


def averagedrivingtime(C_, A_drive, trans):
    return lib().orc_averagedrivingtime(C_, trans.shape[1], float(A_drive), trans)


def sum_travel_time_q16(trans):
    return int(lib().orc_sum_travel_time_q16(trans.shape[0], trans.shape[1], trans))


def correctparameters(p_min_next, p_max_next, p_min, p_max):
    a, b = C.c_double(), C.c_double()
    lib().orc_correctparameters(p_min_next, p_max_next, p_min, p_max, C.byref(a), C.byref(b))
    return a.value, b.value


# ---------------------------------------------------------------- fast twin
def build_cdf(p_dest):
    Z, _, T = p_dest.shape
    cdf = np.empty((T, Z, Z), dtype=np.float64)  # C-order [t][o][d]
    _check(lib().orc_build_cdf(p_dest, Z, T, _ptr(cdf)), "build_cdf")
    return cdf


def fast_run(p_drive, cdf, C_, seed, zone0, car_offset=0, do_ivp=True, want_state=False,
             datamatrix=None, dist=None, nthreads=0, car_stride=1):
    """IVP (optional) + resample; returns dict(parking, driving [Z x T int64 F-order],
    zone0 (post-IVP zones), state (C x T) or None, sum_tt_q16)."""
    Z, T = p_drive.shape
    zone0 = np.ascontiguousarray(zone0, dtype=np.int64).copy()
    parking = np.zeros((Z, T), dtype=np.int64, order="F")
    driving = np.zeros((Z, T), dtype=np.int64, order="F")
    state = np.zeros((C_, T), dtype=np.int64, order="F") if want_state else None
    tt = C.c_int64(0)
    _check(lib().orc_fast_run(p_drive, _ptr(cdf), Z, T, C_, car_offset, car_stride, seed, int(do_ivp), zone0, parking,
                              driving, _ptr(state), _ptr(datamatrix), _ptr(dist),
                              C.cast(C.byref(tt), C.c_void_p), nthreads), "fast_run")
    return dict(parking=parking, driving=driving, zone0=zone0, state=state, sum_tt_q16=tt.value)


# ---------------------------------------------------------------- synthetic inputs
def synth_p_drive(Z, T, table_seed):
    out = np.zeros((Z, T), dtype=np.float64, order="F")
    lib().orc_synth_p_drive(Z, T, table_seed, out)
    return out


def synth_p_dest_dense(Z, T, table_seed, skew_q=0):
    out = np.zeros((Z, Z, T), dtype=np.float64, order="F")
    _check(lib().orc_synth_p_dest_skewed(Z, T, table_seed, skew_q, out), "synth_p_dest")
    return out


def synth_datamatrix(Z, T, table_seed, density=0.0868):
    dm = np.zeros((Z, Z, T, 2), dtype=np.float64, order="F")
    dist = np.zeros((Z, Z), dtype=np.float64, order="F")
    lib().orc_synth_datamatrix(Z, T, table_seed, density, dm, dist)
    return dm, dist


def createdatamatrix(rawdata, Z, T=24):
    """src/createdatamatrix.jl:3-27 on rawdata[:,1:5] (n x 5) -> datamatrix (Z, Z, T, 2)."""
    raw = np.asfortranarray(rawdata, dtype=np.float64)
    dm = np.zeros((Z, Z, T, 2), dtype=np.float64, order="F")
    _check(lib().orc_createdatamatrix(_ptr(raw), raw.shape[0], Z, T, _ptr(dm)), "orc_createdatamatrix")
    return dm


def centroids(lon_rows, lat_rows, width=None, scan=10000):
    """src/processgeodata.jl:99-146 on per-zone coordinate lists (zero-terminated inside a zeros(Z, width) matrix).
    Returns (centroid_lat, centroid_long, area)."""
    Z = len(lon_rows)
    width = width or (scan + 2)
    lon = np.zeros((Z, width), dtype=np.float64)
    lat = np.zeros((Z, width), dtype=np.float64)
    for i in range(Z):
        k = len(lon_rows[i])
        lon[i, :k] = lon_rows[i]
        lat[i, :k] = lat_rows[i]
    clat, clong, area = (np.zeros(Z) for _ in range(3))
    _check(lib().orc_centroids(_ptr(lon), _ptr(lat), Z, width, scan, _ptr(clat), _ptr(clong), _ptr(area)), "orc_centroids")
    return clat, clong, area


def distance_matrix(centroid_lat, centroid_long):
    """src/processgeodata.jl:148-166 -> distance_matrix_km (Z, Z)."""
    la = np.ascontiguousarray(centroid_lat, dtype=np.float64)
    lo = np.ascontiguousarray(centroid_long, dtype=np.float64)
    Z = la.shape[0]
    d = np.zeros((Z, Z), dtype=np.float64, order="F")
    _check(lib().orc_distance_matrix(_ptr(la), _ptr(lo), Z, _ptr(d)), "orc_distance_matrix")
    return d


def max_threads():
    return lib().orc_max_threads()


# ---------------------------------------------------------------- RNG-free expectation
def markov_expected_density(p_drive, p_dest, pi0, T=None):
    """Exact Markov propagation pi_{t+1} = pi_t M_t, M_t[o,d] = (1-pd[o,t]) delta_od +
    pd[o,t] p_dest[o,d,t] (+ zero-row mass on delta_od, src/resampling.jl:35-36).  The joint OD
    probability of README.md:457-468 of the reference.  numpy only; small Z."""
    Z, T_ = p_drive.shape
    T = T or T_
    pis = np.zeros((T, Z))
    pi = np.asarray(pi0, dtype=np.float64).copy()
    for t in range(T):
        pis[t] = pi
        P = np.array(p_dest[:, :, t])
        pd = np.nan_to_num(np.array(p_drive[:, t]), nan=0.0)
        zero_row = P.sum(axis=1) == 0
        P[zero_row, :] = 0
        P[zero_row, np.nonzero(zero_row)[0]] = 1.0
        rs = P.sum(axis=1, keepdims=True)
        P = P / rs
        M = (1 - pd)[:, None] * np.eye(Z) + pd[:, None] * P
        pi = pi @ M
    return pis
