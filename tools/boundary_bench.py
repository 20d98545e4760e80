#!/usr/bin/env python3
"""What crossing the boundary with host arrays costs at S4k (development tool): cpm_set_p_dest (3.2 GB over PCIe + CDF and
row-pack builds), the blocking cpm_solve_ivp / cpm_resample with host count arrays, against the device-resident loop bench.py times."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import carparkingmaps_amd as cpm
import _synth

Z, T, cpz = 4096, 24, 1000
C = Z * cpz
p_drive, p_dest = _synth.dense_tables(Z, T)
with cpm.Sampler(Z, T) as s:
    s.set_p_drive(p_drive)
    s.set_p_dest(p_dest)  # first call: allocations
    t0 = time.perf_counter()
    s.set_p_dest(p_dest)
    t_tab = time.perf_counter() - t0
    s.init_states(C, cpz)
    s.solve_ivp(0x5EEDCA125, want=False)
    s.resample(0x5EEDCA125)
    t0 = time.perf_counter()
    s.init_states(C, cpz)
    s.solve_ivp(0x5EEDCA125, want=False)
    r = s.resample(0x5EEDCA125)
    t_run = time.perf_counter() - t0
    print(f"cpm_set_p_dest ({p_dest.nbytes / 1e9:.2f} GB host array -> HBM, CDF + row packs): {t_tab * 1e3:.0f} ms")
    print(f"initializestates + solveinitialvalueproblem + resampling (blocking calls, counts to the host): {t_run * 1e3:.2f} ms")
    print(f"one dataset, host tables included: {C * 47 / (t_tab + t_run):.3e} car-steps/s; without the table hand-over: {C * 47 / t_run:.3e}")
