#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DCPM_DIAGNOSTIC, tools/build_variants.sh diag "-DCPM_DIAGNOSTIC"; CPM_LIB_PATH points at it):
where a block of the placing kernel spends its time.  s_memtime stamps of thread 0 of every block of the LAST placing launch of a
resample (100 MHz constant clock, 10 ns per tick): 0 entry, 1 after the first barrier, 2 after the rank atomics (loads consumed),
3 after the second barrier, 4 after the ticket atomics, 5 after the third barrier, 6 after the stores were issued."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import carparkingmaps_amd as cpm
from carparkingmaps_amd import _lib

Z, T, cpz = 4096, 24, 1000
s = cpm.Sampler(Z, T, 0)
s.synth_tables(0x5EED7AB1E)
s.init_states(Z * cpz, cpz)
s.solve_ivp(0x5EEDCA125, want=False)
L = _lib.load()
L.cpm_diag_place_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
nb = 512
for _ in range(3):
    s.resample(0x5EEDCA125)
_lib.check(L.cpm_diag_place_stamps(s._h, None, nb))
s.resample(0x5EEDCA125)
buf = np.zeros((nb, 8), dtype=np.uint64)
_lib.check(L.cpm_diag_place_stamps(s._h, buf.ctypes.data_as(C.c_void_p), nb))
t = buf.astype(np.int64)
t0 = t[:, 0].min()
names = ["entry", "barrier1", "ranks done", "barrier2", "tickets done", "barrier3", "stores issued"]
print("ticks of 10 ns relative to the first block's entry; median / p10 / p90 / max over", nb, "blocks")
for k in range(7):
    v = t[:, k] - t0
    print(f"  {k} {names[k]:14s} {np.median(v):8.0f} {np.percentile(v, 10):8.0f} {np.percentile(v, 90):8.0f} {v.max():8.0f}")
d = np.diff(t[:, :7], axis=1)
print("per-phase durations (median ticks):", [int(np.median(d[:, k])) for k in range(6)])
