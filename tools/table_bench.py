#!/usr/bin/env python3
"""Times the table builders (development tool; run under rocprofv3 --kernel-trace --stats for per-kernel durations):
the one-pass row tables at S4k (Z = 4,096, synthetic dense p_destin) and createpdrive + createpdestin at Melbourne's shape
(Z = 2,357, synthetic sparse datamatrix generated on the device)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import carparkingmaps_amd as cpm

T = 24


def timed(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2], min(ts)


for Z in ((4096,) if os.environ.get("CPM_TABLE_BENCH_SHORT") else (4096, 2357)):
    s = cpm.Sampler(Z, T)
    s.synth_tables(0x5EED7AB1E)
    print(f"Z = {Z}: row tables (packs + totals + checkpoints) median / min ms: %.3f / %.3f" % timed(lambda: s.refresh_tables(False)), flush=True)
    print(f"Z = {Z}: row tables + f64 CDF rows                 median / min ms: %.3f / %.3f" % timed(lambda: s.refresh_tables(True)), flush=True)
    s.close()
if os.environ.get("CPM_TABLE_BENCH_SHORT"):
    sys.exit(0)
Z = 2357
s = cpm.Sampler(Z, T)
s.synth_datamatrix(0x5EED7AB1E)
s.build_p_drive(0.1, 0.9, 0.5, want=False)
s.sync()
def pdrive():
    s.build_p_drive(0.1, 0.9, 0.5, want=False)
    s.sync()
print(f"Z = {Z}: createpdrive (mean cached: k_pdrive_final only) median / min ms: %.3f / %.3f" % timed(pdrive), flush=True)
for e in (2, 0.5):
    print(f"Z = {Z}: createpdestin e_dest = {e!r} (weights + row sums + row tables) median / min ms: %.3f / %.3f" % timed(lambda: s.build_p_dest(e, want=False)), flush=True)
s.close()
