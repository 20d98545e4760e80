#!/usr/bin/env python3
"""Resample with and without the travel-time pass (src/resampling.jl:53-78) on one GPU, Melbourne-shaped synthetic
datamatrix (development tool)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import carparkingmaps_amd as cpm
import _synth

Z, T, cpz = 2357, 24, 1000
C = Z * cpz
with cpm.Sampler(Z, T) as s:
    s.synth_datamatrix(0x5EED7AB1E)      # the Melbourne-shaped synthetic datamatrix, generated on the device
    s.build_p_drive(0.1, 0.9, 0.5, want=False)
    s.build_p_dest(2, want=False)
    s.init_states(C, cpz)
    s.solve_ivp(0x5EEDCA125, want=False)
    for travel in (False, True):
        r = s.resample(0x5EEDCA125, travel=travel)
        s.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            r = s.resample(0x5EEDCA125, travel=travel)
        dt = (time.perf_counter() - t0) / 10
        if travel:                          # the travel kernel itself: hipEvent pairs on its launches
            s.set_profile(True, stride=1, kernel=2)
            for _ in range(5):
                s.resample(0x5EEDCA125, travel=True)
            ms = s.last_kernel_ms()
            s.set_profile(False)
            print(f"k_grouped_travel: {1e3 * sum(ms) / max(len(ms), 1):.1f} us per launch ({len(ms)} launches, one per resample when the runs of all hours are kept)")
        print(f"travel={travel}: {dt * 1e3:.3f} ms per resample, {C * T / dt:.3e} car-steps/s, drivers {int(r['driving'].sum())}, sum_tt {r['sum_tt_q16'] / 65536:.1f} s")
