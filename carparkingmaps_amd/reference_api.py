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
import json
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
    trust_unchanged: bool = False  # True: a host array at the same address with the same shape is taken to be unchanged (no content pass)


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


def invalidate():
    """Forget what is resident on the devices: the next call uploads its host arrays whatever they hold."""
    for d in _loaded.values():
        d.clear()


_STAMP_CHUNK = 1 << 20


def _stamp(a):
    """(address, shape, strides, content fingerprint) of a host array.  The fingerprint is one pass over the WHOLE buffer, and it
    depends on WHERE every word sits: a wrapping uint64 sum of the words plus a wrapping sum of word x (2 x position + 1) (an odd
    weight: no word is ever multiplied away), taken over chunks of 2^20 words.  So an array edited in place -- the notebook's tuning
    loops do that --, and also one permuted in place (two zones' rows swapped, a sort), never leaves stale tables on the device;
    a plain sum would not see the latter.  ~5 GB/s: far below the PCIe upload + table build it can save.  params.trust_unchanged
    skips the pass."""
    a = np.asarray(a)
    head = (a.__array_interface__["data"][0], a.shape, a.strides, str(a.dtype))
    if params.trust_unchanged:
        return head + (None,)
    if a.dtype.itemsize == 8 and (a.flags.f_contiguous or a.flags.c_contiguous):
        words = a.reshape(-1, order="A").view(np.uint64)
    else:
        words = np.ascontiguousarray(a, dtype=np.float64).reshape(-1).view(np.uint64)
    plain = weighted = 0
    with np.errstate(over="ignore"):
        odd = np.arange(1, 2 * _STAMP_CHUNK, 2, dtype=np.uint64)                  # 2 i + 1 inside a chunk
        for k, lo in enumerate(range(0, words.size, _STAMP_CHUNK)):
            w = words[lo:lo + _STAMP_CHUNK]
            ps = int(np.add.reduce(w, dtype=np.uint64))
            ws = int(np.add.reduce(w * odd[:w.size], dtype=np.uint64))
            plain = (plain + ps) & 0xFFFFFFFFFFFFFFFF
            # word i of chunk k has global weight 2 (k L + i) + 1 = (2 i + 1) + 2 k L
            weighted = (weighted + ws + 2 * k * _STAMP_CHUNK * ps) & 0xFFFFFFFFFFFFFFFF
    return head + (plain, weighted)


def _ensure(key, s, name, array, setter):
    fp = _stamp(array)
    if _loaded[key].get(name) != fp:
        setter(array)
        _loaded[key][name] = fp


# ------------------------------------------------------------------------------- data formats
class DeviceArray:
    """What createdatamatrix / processgeodata return here instead of a dense host array: the array lives in HBM
    (6.4 GB + 134 MB at Z = 4096) inside the context of its (number_zones, T, device); `numpy()` downloads it in
    the reference's layout.  createpdrive / createpdestin / resampling accept it where the reference passes the array."""

    def __init__(self, kind, number_zones):
        self.kind, self.number_zones = kind, int(number_zones)
        self.serial = DeviceArray._serial = getattr(DeviceArray, "_serial", 0) + 1

    def numpy(self):
        s, _ = _sampler(self.number_zones)
        return s.get_datamatrix() if self.kind == "datamatrix" else s.get_distance()

    @property
    def shape(self):
        Z = self.number_zones
        return (Z, Z, params.T, 2) if self.kind == "datamatrix" else (Z, Z)


def createdatamatrix(path_to_csv_data, number_zones):
    """src/createdatamatrix.jl:3-27: Uber Movement CSV -> datamatrix (Z, Z, T, 2), built in HBM by the native
    reader (cpm_createdatamatrix_csv).  Returns a DeviceArray."""
    s, key = _sampler(number_zones)
    s.createdatamatrix_csv(path_to_csv_data)
    h = DeviceArray("datamatrix", number_zones)
    _loaded[key]["dm_device"] = h.serial
    _loaded[key].pop("dm", None)
    return h


def geojson_vertex_lists(features):
    """The coordinate walk of src/processgeodata.jl:19-97 -> (number_zones, {zone id: (longitudes, latitudes)}).

    Quirks kept: MOVEMENT_ID is a string of an int; the number of zones comes from the LAST feature's id (+1 when the
    first id is 0, :12-17); id 0 is stored as number_zones (:22-25); every Float64 LEAF found at depth 1..6 below
    `coordinates` appends (parent[1], parent[2]) -- so a [long, lat] pair is stored twice, once per leaf, and a leaf that
    JSON parsed as an integer is skipped (the reference prints a warning from its innermost level); leaves below the
    fifth list level are never stored (:84: the innermost test looks at `coordinates`, not at the leaf)."""
    first = int(features[0]["properties"]["MOVEMENT_ID"])
    last = int(features[-1]["properties"]["MOVEMENT_ID"])
    number_zones = last + 1 if first == 0 else last
    zones = {}

    def walk(node, depth):
        out = []
        for child in node:
            if isinstance(child, float):
                out.append((node[0], node[1]))
            elif isinstance(child, list) and depth < 5:
                out.extend(walk(child, depth + 1))
        return out

    for entry in features:
        zid = int(entry["properties"]["MOVEMENT_ID"])
        if zid == 0:
            zid = number_zones
        coords = entry["geometry"]["coordinates"]
        pts = walk(coords, 1) if isinstance(coords, list) else []
        if len(pts) > 100000:
            raise IndexError("more than 100000 coordinates in one zone (BoundsError in the reference, :19-20)")
        lons, lats = [p[0] for p in pts], [p[1] for p in pts]
        if zid in zones:  # a later feature with the same id overwrites the row from column 1 and leaves the old tail in place (:19-97: the
            old = zones[zid]  # matrices are only ever assigned element by element)
            lons, lats = lons + old[0][len(lons):], lats + old[1][len(lats):]
        zones[zid] = (lons, lats)
    return number_zones, zones


def polygon_centroids(number_zones, zones, scan=10000):
    """src/processgeodata.jl:99-146 on the vertex lists: (centroid_lat, centroid_long), zeros for zones without data.
    The literal loop bound 10000 is kept: vertices beyond it are not seen."""
    clat = np.zeros(number_zones)
    clong = np.zeros(number_zones)
    for zid, (lons, lats) in zones.items():
        if not (1 <= zid <= number_zones) or not lons or lons[0] == 0:
            continue
        lon = list(lons) + [0.0, 0.0, 0.0]
        lat = list(lats) + [0.0, 0.0, 0.0]
        if len(lon) < scan + 2:
            lon += [0.0] * (scan + 2 - len(lon))
            lat += [0.0] * (scan + 2 - len(lat))
        summation_term = 0.0
        for j in range(scan):  # j is 0-based here: element j is the reference's [i, j+1]
            if lon[j] == 0:
                summation_term = summation_term - (lat[j - 1] * lon[j] - lat[j] * lon[j - 1])
                lon[j] = lon[0]
                lat[j] = lat[0]
                summation_term = summation_term + (lat[j - 1] * lon[j] - lat[j] * lon[j - 1])
                break
            summation_term = summation_term + (lat[j] * lon[j + 1] - lat[j + 1] * lon[j])
        area = summation_term / 2
        s_long = s_lat = 0.0
        for j in range(scan):
            if lon[j] == 0:
                s_long = s_long - (lon[j - 1] + lon[j]) * (lon[j - 1] * lat[j] - lon[j] * lat[j - 1])
                s_lat = s_lat - (lat[j - 1] + lat[j]) * (lon[j - 1] * lat[j] - lon[j] * lat[j - 1])
            else:
                s_long = s_long + (lon[j] + lon[j + 1]) * (lon[j] * lat[j + 1] - lon[j + 1] * lat[j])
                s_lat = s_lat + (lat[j] + lat[j + 1]) * (lon[j] * lat[j + 1] - lon[j + 1] * lat[j])
        with np.errstate(all="ignore"):
            clong[zid - 1] = np.float64(-s_long) / np.float64(6 * area)
            clat[zid - 1] = np.float64(-s_lat) / np.float64(6 * area)
    return clat, clong


def processgeodata(path_to_json_data, path_to_data, csv_dataset_list, path_to_results):
    """src/processgeodata.jl:3-181 -> (distance_matrix_km, number_zones).  GeoJSON parsing and the centroid sums are
    host work (O(vertices)); the Z x Z distance matrix is computed and kept in HBM (DeviceArray).  Writes
    zoneID_coordinates.csv (latitude, longitude; :168-177) into path_to_results when it is not None."""
    with open(path_to_json_data, "r") as f:
        doc = json.load(f)
    number_zones, zones = geojson_vertex_lists(doc["features"])
    clat, clong = polygon_centroids(number_zones, zones)
    s, key = _sampler(number_zones)
    s.set_distance_from_centroids(clat, clong)
    if path_to_results is not None:
        with open(os.path.join(path_to_results, "zoneID_coordinates.csv"), "w") as f:
            f.write("latitude,longitude\n")
            for i in range(number_zones):
                f.write(f"{julia_float(clat[i])},{julia_float(clong[i])}\n")
    h = DeviceArray("distance", number_zones)
    _loaded[key]["dist_device"] = h.serial
    return h, number_zones


def createresultsdirectory(path_to_results_folder, city):
    """src/createresultsdirectory.jl:3-17 (plain string concatenation, as there)."""
    path_to_results = str(path_to_results_folder) + str(city)
    if city not in os.listdir(path_to_results_folder):
        os.mkdir(path_to_results)
    return path_to_results


def saveparameters(path_to_results, T, number_zones, cars_per_zone, C, e_drive, p_min, p_max, e_dest, A_drive):
    """src/saveparameters.jl:3-25: one row of nine Float64 values under the reference's header."""
    header = ["T (time steps)", "number_zones", "cars_per_zone", "C (number of cars)", "e_drive (model parameter 1)",
              "p_min (model parameter 2)", "p_max (model parameter 3)", "e_dest (model parameter 4)", "A_drive"]
    values = [T, number_zones, cars_per_zone, C, e_drive, p_min, p_max, e_dest, A_drive]
    with open(os.path.join(path_to_results, "sampling_parameters.csv"), "w") as f:
        f.write(",".join(header) + "\n")
        f.write(",".join(julia_float(float(v)) for v in values) + "\n")


# ------------------------------------------------------------------------------- tables
def _use_dm(s, key, datamatrix, distance_matrix_km):
    """Make `datamatrix` (+ the distance matrix when given) the arrays resident in the context; host arrays that are already
    resident and unchanged (content pass, _stamp) are not uploaded again."""
    if distance_matrix_km is not None and not isinstance(distance_matrix_km, DeviceArray):
        _ensure(key, s, "dist", distance_matrix_km, s.set_distance)
        _loaded[key].pop("dist_device", None)
    if isinstance(datamatrix, DeviceArray):
        if _loaded[key].get("dm_device") != datamatrix.serial:
            raise RuntimeError("this datamatrix is no longer resident: createdatamatrix() was called again for these zones")
        return
    _ensure(key, s, "dm", datamatrix, lambda a: s.set_datamatrix(a, None))  # (None: the resident distance matrix is kept)
    _loaded[key].pop("dm_device", None)


def createpdrive(datamatrix, distance_matrix_km, number_zones):
    """src/createpdrive.jl:3-38 -> p_drive (Z, T)."""
    s, key = _sampler(number_zones)
    _use_dm(s, key, datamatrix, distance_matrix_km)
    out = s.build_p_drive(params.p_min, params.p_max, params.e_drive, want=True)
    _loaded[key]["p_drive"] = _stamp(out)      # what the device holds IS this array
    return out


def createpdestin(datamatrix, number_zones):
    """src/createpdestin.jl:3-50 -> p_dest (Z, Z, T), from the datamatrix it is GIVEN (uploaded now unless it is the resident one,
    unchanged)."""
    s, key = _sampler(number_zones)
    _use_dm(s, key, datamatrix, None)
    out = s.build_p_dest(params.e_dest, want=True)
    _loaded[key]["p_dest"] = _stamp(out)
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
        _use_dm(s, key, datamatrix, distance_matrix_km)
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
    """Float64 -> text the way Julia prints it: the shortest digits that round-trip, positional notation for
    1e-5 < |x| < 1e6 (decimal exponent -4 .. 5), d.ddde[-]x otherwise (1.0e-5, 2.357e6), always with a fractional digit."""
    from decimal import Decimal
    x = float(x)
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Inf" if x > 0 else "-Inf"
    if x == 0:
        return "-0.0" if str(x).startswith("-") else "0.0"
    sign, digits, exp = Decimal(repr(x)).as_tuple()        # repr: shortest round-trip digits
    digits = list(digits)
    while len(digits) > 1 and digits[-1] == 0:             # 2357000.0 -> digits 2357, exponent 3
        digits.pop()
        exp += 1
    e10 = exp + len(digits) - 1                            # x = d.ddd * 10^e10
    ds = "".join(map(str, digits))
    if -5 < e10 < 6:
        if e10 >= 0:
            ip, fp = ds[:e10 + 1].ljust(e10 + 1, "0"), ds[e10 + 1:]
            body = ip + "." + (fp or "0")
        else:
            body = "0." + "0" * (-e10 - 1) + ds
    else:
        body = ds[0] + "." + (ds[1:] or "0") + "e" + str(e10)
    return ("-" if sign else "") + body


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
    _use_dm(s, key, datamatrix, distance_matrix_km)
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
