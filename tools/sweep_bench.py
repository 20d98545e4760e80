#!/usr/bin/env python3
"""Times the model-selection grid of BASELINE.json configs[4] on one GPU: Melbourne-shaped synthetic
datamatrix (Z = 2,357, 8.68 % dense), 1,000 cars/zone, N grid points; each point = table rebuild +
24-hour resample with travel times from the cached post-IVP state.  Development tool."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import carparkingmaps_amd as cpm
from carparkingmaps_amd import model_selection as ms
import _synth

ap = argparse.ArgumentParser()
ap.add_argument("--zones", type=int, default=2357)
ap.add_argument("--cpz", type=int, default=1000)
ap.add_argument("--points", type=int, default=32, help="(kept for old command lines; the share of one rank of --world is what is timed)")
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--lanes", type=int, default=2, help="sampler contexts of the rank (grid points in flight on the GPU at a time)")
args = ap.parse_args()
Z, T, cpz = args.zones, 24, args.cpz
C = Z * cpz
t0 = time.perf_counter()
dm, dist = _synth.datamatrix(Z, T)
print(f"synthetic datamatrix {dm.nbytes / 1e9:.2f} GB in {time.perf_counter() - t0:.1f} s", flush=True)
rng = np.random.default_rng(1)
samplers, lanes = [], []
for lane in range(args.lanes):
    s = cpm.Sampler(Z, T, stream=torch.cuda.Stream())   # (its stream from the start: the lanes must not share a hardware queue)
    t0 = time.perf_counter()
    s.set_datamatrix(dm, dist)
    print(f"lane {lane}: upload {time.perf_counter() - t0:.2f} s", flush=True)
    s.build_p_drive(0.1, 0.9, 0.5, want=False)
    s.build_p_dest(2, want=False)
    s.init_states(C, cpz)
    t0 = time.perf_counter()
    s.solve_ivp(0x5EEDCA125, want=False)
    print(f"lane {lane}: IVP {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    samplers.append(s)
    lanes.append(ms.Evaluator(s, C, 0x5EEDCA125, rng.uniform(0, 1, T), rng.uniform(0, 1, (Z, T)), travel=True))
grid = ms.make_grid()                                # the 256 points of BASELINE.json configs[4]
for n in sorted({1, args.lanes}):
    ev = lanes[:n]
    ms.grid_sweep(ev, grid[:2 * n])
    for rank in (0, args.world - 1):                 # what one rank of the 8-GPU job does: its block of the grid, ordered by e_dest
        for again in (False, True):                  # (again: the lanes already hold the rank's e_dest tables -- the steady state of a longer grid)
            t0 = time.perf_counter()
            res = [r for r in ms.grid_sweep(ev, grid, rank=rank, world_size=args.world, gather=False) if r is not None]
            dt = time.perf_counter() - t0
            print(f"{n} lane(s), rank {rank} of {args.world}{' (tables resident)' if again else ''}: {len(res)} grid points in {dt:.3f} s = "
                  f"{1e3 * dt / len(res):.2f} ms/point ({len(res) * C * T / dt:.3e} car-steps/s incl. table rebuilds, travel times, counts to "
                  f"the host and the objectives); e_dest values {sorted({r['e_dest'] for r in res})}", flush=True)
print("sample:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in res[0].items()})
for s in samplers:
    s.close()
