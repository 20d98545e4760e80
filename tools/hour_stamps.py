#!/usr/bin/env python3
"""Diagnostic (library built with -DCPM_DIAGNOSTIC -DCPM_STAMP_BOTH, tools/build_variants.sh; CPM_LIB_PATH points at it): the
timeline of ONE fused hourly launch (k_grouped_hour, the last fused one of a resample).  Thread 0 of every block keeps s_memtime
stamps in scalar registers and writes them out when the block ends: entry (0) and exit (7) of every sampler workgroup and of every
placing block, and the placing blocks' phases in between (1 loads issued ... 7 stores issued; in the fused hour phase 0 -> 1
contains the wait for the chunk's counter).  Clocks of different XCDs are not synchronised: everything is per clock domain."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import carparkingmaps_amd as cpm
from carparkingmaps_amd import _lib

Z, T, cpz = int(os.environ.get("CPM_STAMP_Z", "4096")), 24, 1000
s = cpm.Sampler(Z, T, 0)
if os.environ.get("CPM_STAMP_MELB") == "1":   # Melbourne-shaped sparse tables (one launch per hour from 512 cars per zone on)
    s.synth_datamatrix(0x5EED7AB1E)
    s.build_p_drive(0.1, 0.9, 0.5, want=False)
    s.build_p_dest(2, want=False)
else:
    s.synth_tables(0x5EED7AB1E)
s.init_states(Z * cpz, cpz)
s.solve_ivp(0x5EEDCA125, want=False)
L = _lib.load()
L.cpm_diag_place_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
zr = (Z + 7) & ~7
npl = ((Z + 63) // 64) * 32
nb = zr + npl
PF = os.environ.get("CPM_FUSED") == "3"   # placing first: the placing blocks of the hour before in front of the hour's samplers
for _ in range(3):
    s.resample(0x5EEDCA125)
_lib.check(L.cpm_diag_place_stamps(s._h, None, nb))
s.resample(0x5EEDCA125)
buf = np.zeros((nb, 8), dtype=np.uint64)
_lib.check(L.cpm_diag_place_stamps(s._h, buf.ctypes.data_as(C.c_void_p), nb))
t = buf.astype(np.int64)
role = np.r_[np.ones(npl, dtype=int), np.zeros(zr, dtype=int)] if PF else np.r_[np.zeros(zr, dtype=int), np.ones(nb - zr, dtype=int)]   # 0 sampler workgroup, 1 placing block
ok = t[:, 0] != 0
print(f"blocks with stamps: {int(ok.sum())} of {nb} (samplers {int((ok & (role == 0)).sum())}, placing {int((ok & (role == 1)).sum())})")
idx = np.flatnonzero(ok)
order = np.argsort(t[idx, 0])
idx = idx[order]
cuts = np.flatnonzero(np.diff(t[idx, 0]) > 200_000) + 1
pct = lambda v, q: int(np.percentile(v, q))
for c, (lo, hi) in enumerate(zip(np.r_[0, cuts], np.r_[cuts, len(idx)])):
    ii = idx[lo:hi]
    t0 = t[ii, 0].min()
    sm, pm = ii[role[ii] == 0], ii[role[ii] == 1]
    if len(sm) == 0 or len(pm) == 0:
        continue
    se0, se7 = t[sm, 0] - t0, t[sm, 7] - t0
    pe0, pe1, pe7 = t[pm, 0] - t0, t[pm, 1] - t0, t[pm, 7] - t0
    end = max(se7.max(), pe7.max())
    print(f"domain {c}: samplers {len(sm)}, placing {len(pm)}; launch span {end} ticks")
    sd = np.diff(t[sm, :8], axis=1)
    print(f"   sampler phases (median ticks) ids, Philox, pack+barrier, search, slots, barrier, flush: {[pct(sd[:, k], 50) for k in range(7)]}")
    print(f"   sampler entries 50/90/100 %: {pct(se0, 50)} {pct(se0, 90)} {se0.max()}   exits 50/90/99/100 %: {pct(se7, 50)} {pct(se7, 90)} {pct(se7, 99)} {se7.max()}"
          f"   lifetime median {pct(se7 - se0, 50)}")
    print(f"   placing entries 10/50/90/100 %: {pct(pe0, 10)} {pct(pe0, 50)} {pct(pe0, 90)} {pe0.max()}   past the wait 50/90/100 %: {pct(pe1, 50)} {pct(pe1, 90)} {pe1.max()}"
          f"   exits 10/50/90/100 %: {pct(pe7, 10)} {pct(pe7, 50)} {pct(pe7, 90)} {pe7.max()}")
    d = np.diff(t[pm, :8], axis=1)
    print(f"   placing phases (median ticks) wait+loads issued, ranks, barrier, scan, barrier, sort+ticket, stores: {[pct(d[:, k], 50) for k in range(7)]}"
          f"   lifetime median / p90 / max: {pct(pe7 - pe0, 50)} {pct(pe7 - pe0, 90)} {(pe7 - pe0).max()}")
    late = pm[pe0 > se7.max()]
    print(f"   placing blocks that ENTER after the last sampler exit: {len(late)}; alive at the last sampler exit: {int(((pe0 <= se7.max()) & (pe7 > se7.max())).sum())}"
          f"; done before it: {int((pe7 <= se7.max()).sum())}")
