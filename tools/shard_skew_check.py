#!/usr/bin/env python3
"""What rank 0 of an 8-GPU weak-scaling run sees (development tool): its contiguous car range starts in 512 of the 4096 zones
(8000 cars each, 8x the per-GPU mean), so the first IVP hour overflows the default bucket regions.  Checks that the context absorbs it
by growing (not by leaving the grouped layout) and times the resample afterwards."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import carparkingmaps_amd as cpm

Z, T, N = 4096, 24, 8
cpz = 1000 * N
C = Z * cpz
count = C // N
with cpm.Sampler(Z, T) as s:
    s.synth_tables(0x5EED7AB1E)
    s.init_states(C, cpz, 0, count)
    t0 = time.perf_counter()
    s.solve_ivp(0x5EEDCA125, want=False)
    print(f"IVP (with the overflow + growth + repeat): {1e3 * (time.perf_counter() - t0):.1f} ms; kernel {s.get_info(1)}, regions {s.get_info(2)}x the mean")
    counts = torch.zeros(s.counts_words(), dtype=torch.int64, device="cuda:0")
    for _ in range(3):
        s.resample_dev(0x5EEDCA125, counts.data_ptr())
    s.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        s.resample_dev(0x5EEDCA125, counts.data_ptr())
    s.sync()
    dt = (time.perf_counter() - t0) / 10
    pk = counts[: Z * T].reshape(T, Z).sum(dim=1)
    print(f"resample {dt * 1e3:.3f} ms; status {int(counts[-1])}; every hour holds {int(pk[0])} == {count} cars: {bool((pk == count).all())}; "
          f"kernel {s.get_info(1)}, regions {s.get_info(2)}x")
