#!/usr/bin/env python3
"""Diagnostic (library built with -DCPM_DIAGNOSTIC; CPM_LIB_PATH points at it): the placing launch on skewed tables -- block lifetimes
by destination group and the slowest blocks with their phases (thread 0's s_memtime stamps of the LAST placing launch of a resample)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import carparkingmaps_amd as cpm
from carparkingmaps_amd import _lib

Z, T, cpz = 4096, 24, 1000
s = cpm.Sampler(Z, T, 0)
s.synth_tables(0x5EED7AB1E, skew_q=int(os.environ.get("CPM_SKEW", "32")))
s.init_states(Z * cpz, cpz)
s.solve_ivp(0x5EEDCA125, want=False)
L = _lib.load()
L.cpm_diag_place_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
nb = 2048
for plan in (0,):
    for _ in range(6):
        s.resample(0x5EEDCA125)
    _lib.check(L.cpm_diag_place_stamps(s._h, None, nb))
    s.resample(0x5EEDCA125)
    buf = np.zeros((nb, 8), dtype=np.uint64)
    _lib.check(L.cpm_diag_place_stamps(s._h, buf.ctypes.data_as(C.c_void_p), nb))
    t = buf.astype(np.int64)
    ok = t[:, 0] != 0
    life = t[:, 7] - t[:, 0]
    print(f"blocks with stamps {int(ok.sum())} (first part {int(ok[:1024].sum())}, second part {int(ok[1024:].sum())})")
    d = np.diff(t, axis=1)
    for part, sl in (("first part", slice(0, 1024)), ("second part", slice(1024, 2048))):
        m = ok[sl]
        if not m.any():
            continue
        lf = life[sl][m]
        print(f"   {part}: lifetime median {int(np.median(lf))} p90 {int(np.percentile(lf, 90))} max {int(lf.max())}; phases (median) {[int(np.median(d[sl][m, k])) for k in range(7)]}")
    span = t[ok, 7].max() - t[ok, 0].min()
    print(f"   span first entry -> last exit {int(span)} ticks (clock domains ignored)")
    top = np.argsort(-np.where(ok, life, 0))[:10]
    for b in top:
        print(f"   block {b}: g {b % 32} j {b // 32} lifetime {int(life[b])} phases {[int(x) for x in d[b]]}")
    g = np.arange(1024) % 32
    print("   first part, lifetime by group (median):", [int(np.median(life[:1024][(g == k) & ok[:1024]])) if ((g == k) & ok[:1024]).any() else -1 for k in range(32)])
