"""Sampler: thin object wrapper over the C ABI (include/cpm.h) with numpy arrays in the
reference's layout (Fortran order, 1-based zone ids).  One Sampler = one cpm_ctx = one GPU.
"""
import ctypes as C
import os

import numpy as np

from . import _lib


def _f64(a, shape=None):
    a = np.asfortranarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Sampler:
    """Device-resident tables + car state for `number_zones` zones and T hours on one GPU."""

    def __init__(self, number_zones, T=24, device=0, stream=None):
        """stream: a HIP stream to enqueue on -- an integer handle or an object with `.cuda_stream` (torch.cuda.Stream), which is
        kept alive with the Sampler.  Given here, the context never creates a stream of its own: a process has few hardware
        queues (GPU_MAX_HW_QUEUES, 4 by default) and every HIP stream that is created takes a share of one, so two contexts
        meant to run side by side should each get their stream at construction (model_selection.grid_sweep over two lanes)."""
        self._L = _lib.load()
        self.Z, self.T, self.device = int(number_zones), int(T), int(device)
        h = C.c_void_p()
        _lib.check(self._L.cpm_create(C.byref(h), self.Z, self.T, self.device))
        self._h = h
        self._stream_obj = self._stream = None
        if stream is not None:
            self.set_stream(stream)
        self.C_total = self.cars_per_zone = self.car_begin = self.car_count = 0
        self.car_stride = 1

    # -- lifetime ----------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.cpm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- options -----------------------------------------------------------
    def set_kernel(self, kernel):
        _lib.check(self._L.cpm_set_option(self._h, _lib.CPM_OPT_KERNEL, int(kernel)))

    def set_fused(self, mode=1, lag=None):
        """The grouped path's fused hour: 5 on where it pays (the library's default), 1 on wherever it can run, 0 two launches per hour, 2 on with placing blocks that give up at once
        (tests), 3 the placing-first form (the previous hour's placing blocks in front of the hour's samplers), 4 = 3 with samplers
        that give up at once (tests); 6 all hours of a run in ONE launch (k_grouped_day: the placing blocks of an hour among the next
        hour's sampler workgroups, which draw for their stayers first), 8 = 6 with the placing blocks in front, 7 = 6 with blocks that
        give up at once (tests), 9 = 6 where it pays; lag: chunks of sampler workgroups in front of a chunk's placing blocks (mode 1)."""
        _lib.check(self._L.cpm_set_option(self._h, _lib.CPM_OPT_FUSED, int(mode)))
        if lag is not None:
            _lib.check(self._L.cpm_set_option(self._h, _lib.CPM_OPT_FUSED_LAG, int(lag)))

    def set_zone_order(self, on=True):
        """The one-launch hour deals its sampler workgroups the zones largest-first (a scheduling hint; the counts do not depend on
        it): 1 / True on, 0 / False off, 2 = the library's default: on for sparse row packs (datasets: -14 % at Melbourne's shape),
        off for dense ones (2-4 % slower at 4,096 zones)."""
        _lib.check(self._L.cpm_set_option(self._h, _lib.CPM_OPT_ZONE_ORDER, int(on)))

    def get_info(self, what):
        """cpm_get_info: 1 = kernel family AUTO resolves to now, 2 = bucket-region size in multiples of the mean bucket, 3 = workgroups per
        heavy zone, 4 = form of the hour (0 two launches, 1 one, 3 placing first, 6 all hours in one launch), 5 = steps that bailed out of a
        one-launch form, 6 = words of a sparse row pack (0: dense tables)."""
        v = C.c_int64(0)
        _lib.check(self._L.cpm_get_info(self._h, int(what), C.byref(v)))
        return int(v.value)

    def set_profile(self, on=True, stride=1, kernel=0):
        """hipEvents around every `stride`-th hourly launch of `kernel` (0 sampler, 1 place, 2 travel); on=False: off."""
        _lib.check(self._L.cpm_set_option(self._h, _lib.CPM_OPT_PROFILE_KERNEL, int(kernel)))
        _lib.check(self._L.cpm_set_option(self._h, _lib.CPM_OPT_PROFILE, int(stride) if on else 0))

    def set_stream(self, hip_stream):
        """hip_stream: integer handle (e.g. torch.cuda.current_stream().cuda_stream), an object with `.cuda_stream`, or None."""
        handle = getattr(hip_stream, "cuda_stream", hip_stream)
        _lib.check(self._L.cpm_set_stream(self._h, C.c_void_p(handle) if handle else None))
        self._stream_obj, self._stream = (hip_stream if handle else None), (handle or None)

    def sync(self):
        _lib.check(self._L.cpm_sync(self._h))

    # -- tables ------------------------------------------------------------
    def set_p_drive(self, p_drive):
        a = _f64(p_drive, (self.Z, self.T))
        _lib.check(self._L.cpm_set_p_drive(self._h, _vp(a)))

    def set_p_dest(self, p_dest):
        a = _f64(p_dest, (self.Z, self.Z, self.T))
        _lib.check(self._L.cpm_set_p_dest(self._h, _vp(a)))

    def set_datamatrix(self, datamatrix, distance_matrix_km=None):
        """distance_matrix_km None: the distance matrix already resident (set_distance*) is kept."""
        a = _f64(datamatrix, (self.Z, self.Z, self.T, 2))
        d = None if distance_matrix_km is None else _f64(distance_matrix_km, (self.Z, self.Z))
        _lib.check(self._L.cpm_set_datamatrix(self._h, _vp(a), _vp(d)))

    def createdatamatrix_rows(self, rawdata):
        """rawdata: (n, 5) = the reference's rawdata[:,1:5]; builds the dense datamatrix in HBM."""
        a = np.asfortranarray(rawdata, dtype=np.float64)
        if a.ndim != 2 or a.shape[1] != 5:
            raise ValueError(f"expected (n, 5) rows, got {a.shape}")
        _lib.check(self._L.cpm_createdatamatrix_rows(self._h, int(a.shape[0]), _vp(a)))

    def createdatamatrix_csv(self, path):
        """Uber Movement CSV -> dense datamatrix in HBM (native parser); returns the number of data rows."""
        n = C.c_int64(0)
        _lib.check(self._L.cpm_createdatamatrix_csv(self._h, os.fsencode(path), C.byref(n)))
        return int(n.value)

    def get_datamatrix(self):
        out = np.zeros((self.Z, self.Z, self.T, 2), dtype=np.float64, order="F")
        _lib.check(self._L.cpm_get_datamatrix(self._h, _vp(out)))
        return out

    def set_distance_from_centroids(self, centroid_lat, centroid_long):
        la = np.ascontiguousarray(centroid_lat, dtype=np.float64).reshape(-1)
        lo = np.ascontiguousarray(centroid_long, dtype=np.float64).reshape(-1)
        if la.shape != (self.Z,) or lo.shape != (self.Z,):
            raise ValueError(f"expected {self.Z} centroids")
        _lib.check(self._L.cpm_set_distance_from_centroids(self._h, _vp(la), _vp(lo)))

    def set_distance(self, distance_matrix_km):
        d = _f64(distance_matrix_km, (self.Z, self.Z))
        _lib.check(self._L.cpm_set_distance(self._h, _vp(d)))

    def get_distance(self):
        out = np.zeros((self.Z, self.Z), dtype=np.float64, order="F")
        _lib.check(self._L.cpm_get_distance(self._h, _vp(out)))
        return out

    def build_p_drive(self, p_min, p_max, e_drive, want=True):
        out = np.zeros((self.Z, self.T), dtype=np.float64, order="F") if want else None
        _lib.check(self._L.cpm_build_p_drive(self._h, float(p_min), float(p_max), float(e_drive), _vp(out)))
        return out

    def build_p_dest(self, e_dest, want=True):
        out = np.zeros((self.Z, self.Z, self.T), dtype=np.float64, order="F") if want else None
        is_int = int(isinstance(e_dest, (int, np.integer)) and not isinstance(e_dest, bool))
        _lib.check(self._L.cpm_build_p_dest(self._h, float(e_dest), is_int, _vp(out)))
        return out

    def synth_tables(self, table_seed, skew_q=0):
        """Procedural bench tables; skew_q > 0: destination popularity 1 / (skew_q + rank), see include/cpm.h."""
        _lib.check(self._L.cpm_synth_tables_skewed(self._h, int(table_seed), int(skew_q)))

    def synth_datamatrix(self, table_seed, density=0.0868):
        """Melbourne-shaped synthetic datamatrix + distance matrix, generated on the device."""
        _lib.check(self._L.cpm_synth_datamatrix(self._h, int(table_seed), float(density)))

    def refresh_tables(self, with_f64_cdf=False):
        """Re-derive the row tables from the resident p_destin -- or from the compact rows of the dataset they were built from -- (the one
        pass the installing calls end in); for measurement."""
        _lib.check(self._L.cpm_refresh_tables(self._h, 1 if with_f64_cdf else 0))

    def get_p_drive(self):
        out = np.zeros((self.Z, self.T), dtype=np.float64, order="F")
        _lib.check(self._L.cpm_get_p_drive(self._h, _vp(out)))
        return out

    def get_cdf_row(self, origin, hour):
        out = np.zeros(self.Z, dtype=np.float64)
        _lib.check(self._L.cpm_get_cdf_row(self._h, int(origin), int(hour), _vp(out)))
        return out

    # -- cars --------------------------------------------------------------
    def init_states(self, C_total, cars_per_zone, car_begin=0, car_count=None, car_stride=1):
        """This context simulates the global cars car_begin + k * car_stride, k < car_count (stride 1: a contiguous range)."""
        car_stride = int(car_stride)
        if car_count is None:
            car_count = max(0, -(-(int(C_total) - int(car_begin)) // car_stride))
        car_count = int(car_count)
        _lib.check(self._L.cpm_init_states_strided(self._h, int(C_total), int(cars_per_zone), int(car_begin), car_stride, car_count))
        self.C_total, self.cars_per_zone = int(C_total), int(cars_per_zone)
        self.car_begin, self.car_count, self.car_stride = int(car_begin), car_count, car_stride

    def set_state(self, zones):
        z = np.ascontiguousarray(zones, dtype=np.int64)
        if z.shape != (self.car_count,):
            raise ValueError(f"expected {self.car_count} zones, got {z.shape}")
        _lib.check(self._L.cpm_set_state(self._h, _vp(z)))

    def get_state(self):
        out = np.zeros(self.car_count, dtype=np.int64)
        _lib.check(self._L.cpm_get_state(self._h, _vp(out)))
        return out

    def solve_ivp(self, seed, want=True):
        out = np.zeros(self.car_count, dtype=np.int64) if want else None
        _lib.check(self._L.cpm_solve_ivp(self._h, int(seed), _vp(out)))
        return out

    def solve_ivp_async(self, seed):
        _lib.check(self._L.cpm_solve_ivp_async(self._h, int(seed)))

    def resample(self, seed, travel=False, want_state=False, want_trans=False):
        """Returns dict(parking, driving: (Z,T) int64 F-order; sum_tt_q16: int; state, trans or None)."""
        parking = np.zeros((self.Z, self.T), dtype=np.int64, order="F")
        driving = np.zeros((self.Z, self.T), dtype=np.int64, order="F")
        state = np.zeros((self.car_count, self.T), dtype=np.int64, order="F") if want_state else None
        trans = np.zeros((self.car_count, self.T, 4), dtype=np.float64, order="F") if want_trans else None
        tt = C.c_int64(0)
        flags = _lib.CPM_FLAG_TRAVEL if travel else 0
        _lib.check(self._L.cpm_resample(self._h, int(seed), flags, _vp(parking), _vp(driving),
                                        C.cast(C.byref(tt), C.c_void_p), _vp(state), _vp(trans)))
        return dict(parking=parking, driving=driving, sum_tt_q16=int(tt.value), state=state, trans=trans)

    def resample_dev(self, seed, d_counts_ptr, travel=False):
        """Enqueue on the context's stream; d_counts_ptr = device address of int64[2*T*Z+2]
        (parking | driving | sum_tt_q16 | status; status != 0 -> repeat with another kernel)."""
        flags = _lib.CPM_FLAG_TRAVEL if travel else 0
        _lib.check(self._L.cpm_resample_dev(self._h, int(seed), flags, C.c_void_p(int(d_counts_ptr))))

    def counts_words(self):
        return 2 * self.T * self.Z + 2

    def last_kernel_ms(self):
        buf = (C.c_float * 8192)()
        n = C.c_int32(0)
        _lib.check(self._L.cpm_last_kernel_ms(self._h, C.cast(buf, C.c_void_p), 8192, C.byref(n)))
        return [float(buf[i]) for i in range(n.value)]

    def debug_categorical(self, origin, hour, k53):
        """Diagnostic: destinations (1-based; 0 = all-zero row) the grouped zone sampler draws from row
        (origin, hour) for the 53-bit uniforms k53 (u = k * 2^-53); also the number of exact-row fallbacks."""
        k = np.ascontiguousarray(k53, dtype=np.uint64)
        out = np.zeros(k.shape[0], dtype=np.int64)
        n_exact = C.c_int32(0)
        _lib.check(self._L.cpm_debug_categorical(self._h, int(origin), int(hour), int(k.shape[0]), _vp(k), _vp(out),
                                                 C.byref(n_exact)))
        return out, int(n_exact.value)

    def algorithmic_bytes_per_hour(self):
        b = C.c_int64(0)
        _lib.check(self._L.cpm_algorithmic_bytes_per_hour(self._h, C.byref(b)))
        return int(b.value)


def parse_uber_csv(path):
    """The native CSV reader alone (host only): (n, 5) Fortran array = the reference's rawdata[:,1:5]."""
    L = _lib.load()
    n = C.c_int64(0)
    _lib.check(L.cpm_parse_uber_csv(os.fsencode(path), C.byref(n), None, 0))
    out = np.zeros((n.value, 5), dtype=np.float64, order="F")
    if n.value:
        _lib.check(L.cpm_parse_uber_csv(os.fsencode(path), C.byref(n), _vp(out), n.value))
    return out


def device_count():
    n = C.c_int32(0)
    _lib.check(_lib.load().cpm_device_count(C.byref(n)))
    return n.value


def device_info(device=0):
    name = C.create_string_buffer(256)
    cu = C.c_int32(0)
    mem = C.c_int64(0)
    _lib.check(_lib.load().cpm_device_info(device, name, 256, C.byref(cu), C.byref(mem)))
    return dict(name=name.value.decode(), cu_count=cu.value, hbm_bytes=mem.value)
