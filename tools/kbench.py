#!/usr/bin/env python3
"""Kernel experiments on one GPU (development tool, not the contract bench): for each
(kernel, zone block, ablation) prints the time of one 24-hour resample and the mean duration
of the hourly sampler launch.  Ablated runs give WRONG results by design (diagnostic only)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import carparkingmaps_amd as cpm
from carparkingmaps_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--zones", type=int, default=4096)
ap.add_argument("--cpz", type=int, default=1000)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--configs", default="car:0:0,zone:512:0,zone:0:0,strided:0:0,grouped:0:0")
args = ap.parse_args()

Z, T, cpz = args.zones, 24, args.cpz
C = Z * cpz
s = cpm.Sampler(Z, T, 0)
s.synth_tables(0x5EED7AB1E)
s.init_states(C, cpz)
s.solve_ivp(0x5EEDCA125, want=False)
import ctypes
L = _lib.load()
ref = None
for cfg in args.configs.split(","):
    parts = cfg.split(":")
    kern, block, abl = parts[:3]
    s.set_kernel({"car": 1, "zone": 2, "strided": 4, "grouped": 5}[kern])
    _lib.check(L.cpm_set_option(s._h, 3, int(block)))
    _lib.check(L.cpm_set_option(s._h, 100, int(abl)))
    if len(parts) > 3 and parts[3]:
        _lib.check(L.cpm_set_option(s._h, 4, int(parts[3])))  # grouped path: place-kernel shape (82, 162, ...)
    if len(parts) > 4 and parts[4]:
        _lib.check(L.cpm_set_option(s._h, 5, int(parts[4])))  # grouped path: generation (5 | 6)

    r = s.resample(0x5EEDCA125)
    s.set_profile(True)
    s.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = s.resample(0x5EEDCA125)
    dt = (time.perf_counter() - t0) / args.steps
    ms = s.last_kernel_ms()
    s.set_profile(False)
    ok = ""
    if int(abl) == 0:
        if ref is None:
            ref = r
        ok = "counts==first" if np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"]) else "COUNTS DIFFER"
    print(f"{cfg:16s} resample {dt*1e3:8.3f} ms   sampler launch avg {np.mean(ms)*1e3:8.1f} us (min {np.min(ms)*1e3:.1f} max {np.max(ms)*1e3:.1f}, n={len(ms)})  {C*T/dt:.3e} car-steps/s {ok}", flush=True)
