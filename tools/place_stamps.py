#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DCPM_DIAGNOSTIC, tools/build_variants.sh diag "-DCPM_DIAGNOSTIC"; CPM_LIB_PATH points at it):
where a block of the placing kernel spends its time.  s_memtime stamps of thread 0 of every block of the LAST placing launch of a
resample (shader clock: the span first entry -> last exit is printed beside the ticks so that it can be set against the launch's
duration in rocprofv3): 0 entry, 1 after the first barrier (loads issued), 2 after the rank atomics (loads consumed), 3 after the
second barrier, 4 ticket requested and block scan done, 5 after the third barrier, 6 entries sorted and ticket arrived, 7 stores issued.
The stamps stay in scalar registers until the block ends, so they add no waits of their own."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import carparkingmaps_amd as cpm
from carparkingmaps_amd import _lib

Z, T, cpz = 4096, int(os.environ.get("CPM_STAMP_T", "24")), 1000
s = cpm.Sampler(Z, T, 0)
s.synth_tables(0x5EED7AB1E)
s.init_states(Z * cpz, cpz)
s.solve_ivp(0x5EEDCA125, want=False)
L = _lib.load()
L.cpm_diag_place_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
sampler = os.environ.get("CPM_STAMP_SAMPLER") == "1"   # the library was built with -DCPM_STAMP_SAMPLER: the sampler's stamps instead
nb = 4096 if sampler else 1024
for _ in range(3):
    s.resample(0x5EEDCA125)
_lib.check(L.cpm_diag_place_stamps(s._h, None, nb))
s.resample(0x5EEDCA125)
buf = np.zeros((nb, 8), dtype=np.uint64)
_lib.check(L.cpm_diag_place_stamps(s._h, buf.ctypes.data_as(C.c_void_p), nb))
t = buf.astype(np.int64)
K = 8
# The counters of different parts of the chip are not synchronised: blocks are clustered by their entry stamp (a gap of more than
# 200 k ticks starts a new clock domain) and times are only compared inside a cluster.
# Little's law per cluster: blocks alive on average = sum of the blocks' lifetimes / (first entry -> last exit).
live = t[t[:, 0] != 0]
order = np.argsort(live[:, 0])
live, ids = live[order], np.flatnonzero(t[:, 0] != 0)[order]
cuts = np.flatnonzero(np.diff(live[:, 0]) > 200_000) + 1
print("clock domains:", len(cuts) + 1, "-- per domain: blocks, block indices mod 8 seen, span first entry -> last exit (ticks), mean lifetime, blocks alive on average")
tot_alive = 0.0
for c, (lo, hi) in enumerate(zip(np.r_[0, cuts], np.r_[cuts, len(live)])):
    tx = live[lo:hi]
    span = tx[:, K - 1].max() - tx[:, 0].min()
    life = tx[:, K - 1] - tx[:, 0]
    starts = tx[:, 0] - tx[:, 0].min()
    tot_alive += life.sum() / span
    print(f"  domain {c}: blocks {hi - lo:5d}  mod 8: {sorted(set((ids[lo:hi] % 8).tolist()))}  span {span:7d}  lifetime {life.mean():8.0f}  alive {life.sum() / span:6.1f}"
          f"  entries at 10/50/90/100 %: {starts[len(starts) // 10]} {starts[len(starts) // 2]} {starts[len(starts) * 9 // 10]} {starts[-1]}")
print(f"blocks alive on average, whole chip: {tot_alive:.1f} (256 CUs)")
t = t[t[:, 0] != 0]
nb = len(t)
t0 = t[:, 0].min()
names = (["entry", "ids arrived", "Philox done", "pack landed + barrier", "search done", "slots taken", "barrier 2", "flushed"] if sampler else
         ["entry", "barrier1", "ranks done", "barrier2", "scan done", "barrier3", "sorted + ticket", "stores issued"])
K = len(names)
d = np.diff(t[:, :K], axis=1)
if not sampler:
    lng = (t[:, 7] & 1) == 1
    print(f"blocks that took the long-run path: {int(lng.sum())} of {nb}")
    for name, m in (("long-run path", lng), ("no long run", ~lng)):
        if m.any():
            print(f"  {name}: per-phase (median ticks)", [int(np.median(d[m, k])) for k in range(K - 1)], " whole block:", int(np.median(t[m, K - 1] - t[m, 0])))
print("per-phase p90 / max:", [int(np.percentile(d[:, k], 90)) for k in range(K - 1)], [int(d[:, k].max()) for k in range(K - 1)],
      " whole block p90 / max:", int(np.percentile(t[:, K - 1] - t[:, 0], 90)), int((t[:, K - 1] - t[:, 0]).max()))
print("per-phase durations (median ticks):", [int(np.median(d[:, k])) for k in range(K - 1)], " whole block (median):", int(np.median(t[:, K - 1] - t[:, 0])))
