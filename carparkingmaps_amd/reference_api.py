"""Host-side mirror of the reference's call surface for the sampler path (main.jl:82-102).

The reference host language is Julia; `julia` is not present in the build image, so the
maintained host layer above the C ABI is this Python module (same function names, same
positional arguments, same array shapes / 1-based zone ids, Fortran-ordered numpy arrays
standing in for Julia arrays).  julia/CarParkingMapsAMD.jl carries the equivalent `ccall`
shim for a Julia host.

Like the reference, the functions read the script-level hyper-parameters implicitly
(main.jl:37-42): they live in `params` below.  Two additions the
reference does not have: `params.seed` (the reference draws from an unseeded global RNG)
and `params.device`.
"""
import os
from dataclasses import dataclass

import numpy as np

from .sampler import Sampler


@dataclass
class Params:
    e_drive: float = 0.5      # main.jl:37
    e_dest: object = 2        # main.jl:38 (an Int in the reference -> integer power)
    p_min: float = 0.1        # main.jl:39
    p_max: float = 0.9        # main.jl:40
    cars_per_zone: int = 1000  # main.jl:41
    T: int = 24               # main.jl:42
    seed: int = 0x5EEDCA125
    device: int = 0


params = Params()

_ctx = {}      # (Z, T, device) -> Sampler
_loaded = {}   # (Z, T, device) -> fingerprints of the tables resident on the device
_last = {}     # results of the last resampling() per context, reused by saveresults/averagedrivingtime


def _sampler(Z):
    key = (int(Z), int(params.T), int(params.device))
    s = _ctx.get(key)
    if s is None:
        s = Sampler(Z, params.T, params.device)
        _ctx[key] = s
        _loaded[key] = {}
    return s, key


def release():
    """Free every cached device context."""
    for s in _ctx.values():
        s.close()
    _ctx.clear()
    _loaded.clear()
    _last.clear()


def _fingerprint(a):
    a = np.asarray(a)
    flat = a.reshape(-1, order="A")
    step = max(1, flat.size // 65536)
    return (a.__array_interface__["data"][0], a.shape, float(np.nansum(flat[::step])), float(flat[-1]) if flat.size else 0.0)


def _ensure(key, s, name, array, setter):
    fp = _fingerprint(array)
    if _loaded[key].get(name) != fp:
        setter(array)
        _loaded[key][name] = fp


# ------------------------------------------------------------------------------- tables
def createpdrive(datamatrix, distance_matrix_km, number_zones):
    """src/createpdrive.jl:3-38 -> p_drive (Z, T)."""
    s, key = _sampler(number_zones)
    _ensure(key, s, "dm", datamatrix, lambda a: s.set_datamatrix(a, distance_matrix_km))
    out = s.build_p_drive(params.p_min, params.p_max, params.e_drive, want=True)
    _loaded[key]["p_drive"] = _fingerprint(out)
    return out


def createpdestin(datamatrix, number_zones):
    """src/createpdestin.jl:3-50 -> p_dest (Z, Z, T).  Needs the distance matrix only through the
    datamatrix upload, so createpdrive (main.jl:82) is expected to have run first, as in main.jl."""
    s, key = _sampler(number_zones)
    if _loaded[key].get("dm") != _fingerprint(datamatrix):
        raise RuntimeError("createpdestin: call createpdrive(datamatrix, distance_matrix_km, number_zones) "
                           "first, as main.jl:82-85 does (it uploads datamatrix and the distance matrix)")
    out = s.build_p_dest(params.e_dest, want=True)
    _loaded[key]["p_dest"] = _fingerprint(out)
    return out


# ------------------------------------------------------------------------------- sampler
def initializestates(C):
    """src/initializestates.jl:4-22 -> (state_matrix (C,T) Int64, transition_matrix (C,T,4))."""
    T, cpz = params.T, params.cars_per_zone
    state_matrix = np.zeros((C, T), dtype=np.int64, order="F")
    transition_matrix = np.zeros((C, T, 4), dtype=np.float64, order="F")
    nblocks = C // cpz if cpz else 0  # for i = 0:cars_per_zone:(C-cars_per_zone)
    state_matrix[: nblocks * cpz, 0] = np.repeat(np.arange(1, nblocks + 1, dtype=np.int64), cpz)
    return state_matrix, transition_matrix


def _install(s, key, p_drive, p_dest):
    _ensure(key, s, "p_drive", p_drive, s.set_p_drive)
    _ensure(key, s, "p_dest", p_dest, s.set_p_dest)


def solveinitialvalueproblem(state_matrix, transition_matrix, p_drive, p_dest, C, number_zones):
    """src/solveinitialvalueproblem.jl:4-62 -> initial_state (C,) Int64.
    Unlike the reference, columns 2..T of state_matrix / transition_matrix are not used as
    scratch (resampling overwrites them anyway, src/resampling.jl:19-47,82)."""
    s, key = _sampler(number_zones)
    _install(s, key, p_drive, p_dest)
    s.init_states(C, params.cars_per_zone)
    s.set_state(state_matrix[:, 0])
    return s.solve_ivp(params.seed, want=True)


def resampling(state_matrix, transition_matrix, C, number_zones, p_drive, p_dest, datamatrix, distance_matrix_km):
    """src/resampling.jl:3-89: fills state_matrix and transition_matrix in place and returns them."""
    s, key = _sampler(number_zones)
    _install(s, key, p_drive, p_dest)
    travel = datamatrix is not None and distance_matrix_km is not None
    if travel:
        _ensure(key, s, "dm", datamatrix, lambda a: s.set_datamatrix(a, distance_matrix_km))
    s.init_states(C, params.cars_per_zone)
    s.set_state(state_matrix[:, 0])
    r = s.resample(params.seed, travel=travel, want_state=True, want_trans=True)
    state_matrix[...] = r["state"]
    transition_matrix[...] = r["trans"]
    _last[key] = dict(parking=r["parking"], driving=r["driving"], sum_tt_q16=r["sum_tt_q16"], C=C,
                      state_id=id(state_matrix), trans_id=id(transition_matrix))
    return state_matrix, transition_matrix


# ------------------------------------------------------------------------------- reductions
def averagedrivingtime(C, A_drive, transition_matrix):
    """src/averagedrivingtime.jl:3-12."""
    T = params.T
    total = 0.0
    for t in range(T):
        total = total + float(np.sum(transition_matrix[:, t, 2]))
    return A_drive + total / (C * T * 60 * 60)


def correctparameters(p_min_next, p_max_next, p_min, p_max):
    """src/correctparameters.jl:3-22 (note the else-nesting, Appendix A-11)."""
    if p_min_next < 0:
        p_min_next = 0
    else:
        if p_min_next > p_max:
            p_min_next = p_max
    if p_max_next > 1:
        p_max_next = 1
    else:
        if p_max_next < p_min:
            p_max_next = p_min
    return p_min_next, p_max_next


def zone_hour_counts(number_zones, state_matrix, transition_matrix):
    """The histogram of src/saveresults.jl:8-17 as integer counts (parking, driving), taken from the
    fused device result of the resampling() call that produced these matrices."""
    _, key = _sampler(number_zones)
    last = _last.get(key)
    if last is None or last["state_id"] != id(state_matrix) or last["trans_id"] != id(transition_matrix):
        raise RuntimeError("saveresults: these matrices were not produced by the last resampling() call; "
                           "the zone x hour histogram is computed on the device inside resampling()")
    return last["parking"], last["driving"]


def julia_float(x):
    """Float64 -> text the way Julia prints it (shortest round-trip digits; 1.0e-5 style exponents)."""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Inf" if x > 0 else "-Inf"
    r = repr(float(x))
    if "e" in r:
        mant, exp = r.split("e")
        if "." not in mant:
            mant += ".0"
        return f"{mant}e{int(exp)}"
    return r


def saveresults(number_zones, state_matrix, transition_matrix, path_to_results, data_set, C):
    """src/saveresults.jl:3-45: parking densities (Z x T) and traffic activity (1 x T) as CSV."""
    T = params.T
    parking, driving = zone_hour_counts(number_zones, state_matrix, transition_matrix)
    parking_cars = parking.astype(np.float64) / C                      # :20
    traffic = driving.astype(np.float64).sum(axis=0)                   # :23
    with np.errstate(all="ignore"):
        traffic = (traffic - traffic.min()) / (traffic.max() - traffic.min())  # :24-28 (NaN when flat)
    header = ",".join(f"t = {t}h" for t in range(1, T + 1))            # :34-38
    with open(os.path.join(path_to_results, "results_parkingdensities_" + data_set), "w") as f:
        f.write(header + "\n")
        for z in range(number_zones):
            f.write(",".join(julia_float(v) for v in parking_cars[z]) + "\n")
    with open(os.path.join(path_to_results, "results_trafficactivity_" + data_set), "w") as f:
        f.write(header + "\n")
        f.write(",".join(julia_float(v) for v in traffic) + "\n")
    return parking_cars, traffic


# ------------------------------------------------------------------------------- fused fast path
def run_dataset(datamatrix, distance_matrix_km, number_zones, travel=True):
    """main.jl:79-102 for one dataset without materialising the C x T matrices on the host:
    tables -> initializestates -> IVP -> resampling -> counts.  Returns dict(parking_density,
    traffic_activity, A_drive_increment, parking, driving)."""
    s, key = _sampler(number_zones)
    Z, T = int(number_zones), params.T
    C = Z * params.cars_per_zone
    _ensure(key, s, "dm", datamatrix, lambda a: s.set_datamatrix(a, distance_matrix_km))
    s.build_p_drive(params.p_min, params.p_max, params.e_drive, want=False)
    s.build_p_dest(params.e_dest, want=False)
    _loaded[key].pop("p_drive", None)
    _loaded[key].pop("p_dest", None)
    s.init_states(C, params.cars_per_zone)
    s.solve_ivp(params.seed, want=False)
    r = s.resample(params.seed, travel=travel)
    traffic = r["driving"].astype(np.float64).sum(axis=0)
    with np.errstate(all="ignore"):
        traffic = (traffic - traffic.min()) / (traffic.max() - traffic.min())
    return dict(parking=r["parking"], driving=r["driving"], parking_density=r["parking"] / C,
                traffic_activity=traffic, A_drive_increment=(r["sum_tt_q16"] / 65536.0) / (C * T * 3600.0))
