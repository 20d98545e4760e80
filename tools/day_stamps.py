#!/usr/bin/env python3
"""Diagnostic (library built with -DCPM_DIAGNOSTIC -DCPM_STAMP_BOTH, tools/build_variants.sh; CPM_LIB_PATH points at it): the timeline
of ONE hour's segment of the day launch (k_grouped_day), per XCD (= per clock domain).  Thread 0 of every block keeps s_memtime stamps
in scalar registers and writes them out when the block ends.  Sampler workgroups: 0 entry, 1 verdicts known (merged / split), 2 ids of
the first pass landed, 3 first pass done, 4 (split) the group's placing blocks are done, 5 the arrivals' ids landed, 6 second pass
done, 7 runs flushed (before the drain + hand-off).  Placing blocks: 0 entry, 1 past the wait + loads issued, ... 7 stores issued.
    CPM_DAY_MODE=6|8  CPM_STAMP_Z=4096  CPM_STAMP_HOUR=12"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import carparkingmaps_amd as cpm
from carparkingmaps_amd import _lib

Z, T, cpz = int(os.environ.get("CPM_STAMP_Z", "4096")), 24, int(os.environ.get("CPM_STAMP_CPZ", "1000"))
mode = int(os.environ.get("CPM_DAY_MODE", "6"))
hour = int(os.environ.get("CPM_STAMP_HOUR", "12"))
s = cpm.Sampler(Z, T, 0)
s.synth_tables(0x5EED7AB1E)
s.init_states(Z * cpz, cpz)
s.solve_ivp(0x5EEDCA125, want=False)
s.set_fused(mode)
L = _lib.load()
L.cpm_diag_place_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
zpg, pc = max(1, (Z + 31) // 32), (Z + 63) // 64   # (zones per destination group: grouped_zpg_of)
per = 32 * (zpg + pc)
nb = (T - 1) * per
for _ in range(3):
    s.resample(0x5EEDCA125)
assert s.get_info(4) == 6, s.get_info(4)
_lib.check(L.cpm_diag_place_stamps(s._h, None, 2 * nb))
s.resample(0x5EEDCA125)
buf = np.zeros((2 * nb, 8), dtype=np.uint64)
_lib.check(L.cpm_diag_place_stamps(s._h, buf.ctypes.data_as(C.c_void_p), 2 * nb))
t = buf.astype(np.int64)
entry_all = t[nb:, 0]          # every block's first instruction (written at once, whatever the block then does)
t = t[:nb]                     # the blocks' own records (hour 0's rows are overwritten by the hourly launch of the last hour)


def roles(mix):
    """(is_place, set, j or q) of every block of a segment, as day_role() deals them"""
    r = np.arange(per)
    x, i = r & 7, r >> 3
    place = np.zeros(per, dtype=bool)
    st = np.zeros(per, dtype=int)
    k = np.zeros(per, dtype=int)
    if not mix:
        pl = i < 4 * pc
        place[:] = pl
        st[pl], k[pl] = i[pl] // pc, i[pl] % pc
        i2 = i[~pl] - 4 * pc
        st[~pl], k[~pl] = i2 // zpg, i2 % zpg
    else:
        first = i < pc
        place[first], st[first], k[first] = True, 0, i[first]
        i2 = i - pc
        w = zpg + pc
        sset, p = i2 // w, i2 % w
        pl = (~first) & (sset < 3) & (p < pc)
        place[pl], st[pl], k[pl] = True, sset[pl] + 1, p[pl]
        sm = (~first) & ~pl
        st[sm] = sset[sm]
        k[sm] = np.where(sset[sm] < 3, p[sm] - pc, p[sm])
    return place, st, k, x


place, st, k, x = roles(0 if mode == 8 else 1)
pct = lambda v, q: int(np.percentile(v, q)) if len(v) else -1
# Clocks differ between CUs (s_memtime is per clock domain): every block's record carries its CU (HW_ID bits in stamp 0) and XCD (stamp
# 7); a CU's zero = the entry of the first block it ran in this launch (all CUs start within ~1 us of the launch).
key_all = np.zeros(nb, dtype=np.int64)
ok_all = t[:, 7] != 0
key_all[ok_all] = ((t[ok_all, 7] & 15) << 8) | (t[ok_all, 0] & 0xFF)
zero = {}
h1 = slice(per, 2 * per)       # hour 1's segment: the first whose records survive
for kk in np.unique(key_all[h1][ok_all[h1]]):
    zero[int(kk)] = None
first_seen = {}
for i in np.flatnonzero(ok_all):
    kk = int(key_all[i])
    e = entry_all[i]
    if kk not in first_seen or e < first_seen[kk]:
        first_seen[kk] = e
print(f"CUs seen: {len(first_seen)}; launch = {nb} blocks")
tz = np.array([first_seen.get(int(kk), 0) for kk in key_all])
tn = t - tz[:, None]           # ticks since the CU's first block of the launch
tn[~ok_all] = 0
ent = entry_all - tz
for h in (hour - 1, hour):
    seg = tn[h * per:(h + 1) * per]
    raw = t[h * per:(h + 1) * per]
    ok = ok_all[h * per:(h + 1) * per]
    merged = (raw[:, 7] & 16) != 0
    xcc = raw[:, 7] & 15
    print(f"hour {h}: blocks with records {int(ok.sum())} of {per}; sampler workgroups merged: {int((merged & ok & ~place).sum())} of {int((ok & ~place).sum())}")
    hz = seg[ok, 0].min()
    print(f"   segment: first entry {hz} ticks after launch, last exit {seg[ok, 7].max()}: span {seg[ok, 7].max() - hz}")
    for dom in (0, 3):
        d = ok & (xcc == dom)
        if not d.any():
            continue
        e = lambda m, c: seg[m, c] - hz
        print(f" XCD {dom}: {int(d.sum())} blocks (block index % 8: {sorted(set((np.flatnonzero(d) & 7).tolist()))})")
        for sset in range(4):
            pm = d & place & (st == sset)
            sm = d & ~place & (st == sset)
            if pm.any():
                print(f"   P set {sset}: n={int(pm.sum())} entry {pct(e(pm,0),0)}/{pct(e(pm,0),50)}/{pct(e(pm,0),100)}  past wait {pct(e(pm,1),50)}/{pct(e(pm,1),100)}"
                      f"  exit {pct(e(pm,7),50)}/{pct(e(pm,7),100)}  wait med {pct(seg[pm,1]-seg[pm,0],50)} max {pct(seg[pm,1]-seg[pm,0],100)}  work med {pct(seg[pm,7]-seg[pm,1],50)}")
            if sm.any():
                mg = sm & merged
                sp = sm & ~merged
                print(f"   S set {sset}: n={int(sm.sum())} (merged {int(mg.sum())}) entry {pct(e(sm,0),0)}/{pct(e(sm,0),50)}/{pct(e(sm,0),100)}  exit {pct(e(sm,7),50)}/{pct(e(sm,7),100)}"
                      f"  life med {pct(seg[sm,7]-seg[sm,0],50)}")
                if mg.any():
                    dd = np.diff(seg[mg][:, [0, 1, 2, 3, 7]], axis=1)
                    print(f"        merged phases (median ticks) verdict, ids, pass, flush: {[pct(dd[:, c], 50) for c in range(4)]}")
                if sp.any():
                    dd = np.diff(seg[sp][:, [0, 1, 2, 3, 4, 5, 6, 7]], axis=1)
                    print(f"        split phases (median ticks) verdict, ids, pass 1, WAIT, ids, pass 2, flush: {[pct(dd[:, c], 50) for c in range(7)]}   wait max {pct(dd[:, 3], 100)}")
        # slots over time on this XCD (every block of the launch that ran there, whatever its hour)
        dall = ok_all & ((t[:, 7] & 15) == dom)
        hour_of = np.arange(nb) // per
        pl_all = np.tile(place, T - 1)
        mg_all = (t[:, 7] & 16) != 0
        span = int(seg[d, 7].max() - hz)
        step = max(span // 30, 1)
        rows = []
        for tt in range(-4 * step, span + step, step):
            a = hz + tt
            alive = dall & (tn[:, 0] <= a) & (tn[:, 7] > a)
            swait = alive & ~pl_all & ~mg_all & (tn[:, 3] <= a) & (tn[:, 4] > a)
            pwait = alive & pl_all & (tn[:, 1] > a)
            rows.append((tt, int((alive & ~pl_all).sum()), int(swait.sum()), int((alive & pl_all).sum()), int(pwait.sum()), int((alive & (hour_of == h)).sum())))
        print("   t: S alive(waiting)/P alive(waiting) [of this hour]:  " + "  ".join(f"{tt // 1000}k:{sa}({sw})/{pa}({pw})[{hh}]" for tt, sa, sw, pa, pw, hh in rows))
