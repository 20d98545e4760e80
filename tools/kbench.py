#!/usr/bin/env python3
"""Kernel experiments on one GPU (development tool, not the contract bench): for each kernel family prints the time of one
24-hour resample and the mean duration of the hourly sampler launch.  CPM_LIB_PATH selects another build of the library
(e.g. one made with `make EXTRA=-DCPM_STAGE=16 ...`), so variants can be compared in one gpurun call."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import carparkingmaps_amd as cpm

ap = argparse.ArgumentParser()
ap.add_argument("--zones", type=int, default=4096)
ap.add_argument("--cpz", type=int, default=1000)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--configs", default="car,zone,grouped")
ap.add_argument("--travel", action="store_true")
ap.add_argument("--skew", type=int, default=0, help="Q > 0: skewed destination popularity 1 / (Q + rank)")
args = ap.parse_args()

Z, T, cpz = args.zones, 24, args.cpz
C = Z * cpz
s = cpm.Sampler(Z, T, 0)
s.synth_tables(0x5EED7AB1E, skew_q=args.skew)
s.init_states(C, cpz)
s.solve_ivp(0x5EEDCA125, want=False)
ref = None
for cfg in args.configs.split(","):
    s.set_kernel({"car": 1, "zone": 2, "grouped": 5, "auto": 0}[cfg])
    for _ in range(4):  # (lets a skewed context grow its bucket regions and size its heavy-bucket launch)
        r = s.resample(0x5EEDCA125)
    s.set_profile(True)
    s.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = s.resample(0x5EEDCA125)
    dt = (time.perf_counter() - t0) / args.steps
    ms = s.last_kernel_ms()
    s.set_profile(False)
    if ref is None:
        ref = r
    ok = "counts==first" if np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"]) else "COUNTS DIFFER"
    print(f"{cfg:10s} resample {dt*1e3:8.3f} ms   sampler launch avg {np.mean(ms)*1e3:8.1f} us (min {np.min(ms)*1e3:.1f} max {np.max(ms)*1e3:.1f}, n={len(ms)})  "
          f"{C*T/dt:.3e} car-steps/s {ok}  regions {s.get_info(2)}x parts {s.get_info(3)} largest bucket {r['parking'].max()}  "
          f"[{os.environ.get('CPM_LIB_PATH', 'default lib')}]", flush=True)
