#!/usr/bin/env python3
"""Diagnostic: how much of the placing kernel's time could hide under the sampler if the two ran side by side?  Two independent
contexts (same tables, same fleet) resample on two HIP streams: alone, one after the other, and together.  The sampler is bound by
what it issues and the placing kernel by round trips, so the question is whether the chip interleaves them; nothing here is a
product path (a resample's own hours stay serial: hour t+1 needs every bucket of hour t)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import carparkingmaps_amd as cpm

ap = argparse.ArgumentParser()
ap.add_argument("--zones", type=int, default=4096)
ap.add_argument("--cpz", type=int, default=1000)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--contexts", type=int, default=2)
ap.add_argument("--travel", action="store_true", help="tables from a synthetic Melbourne-shaped datamatrix, travel times on (the sweep's resample)")
a = ap.parse_args()
Z, T = a.zones, 24
dev = torch.device("cuda:0")
ctx, streams, bufs = [], [], []
for i in range(a.contexts):
    st = torch.cuda.Stream(device=dev)
    s = cpm.Sampler(Z, T, 0, stream=st)
    if a.travel:
        if i == 0:
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            import _synth
            dm, dist = _synth.datamatrix(Z, T)
        s.set_datamatrix(dm, dist)
        s.build_p_drive(0.1, 0.9, 0.5, want=False)
        s.build_p_dest(2, want=False)
    else:
        s.synth_tables(0x5EED7AB1E)
    s.init_states(Z * a.cpz, a.cpz)
    s.solve_ivp(0x5EEDCA125, want=False)
    ctx.append(s)
    streams.append(st)
    bufs.append(torch.zeros(s.counts_words(), dtype=torch.int64, device=dev))
    s.resample_dev(1, bufs[i].data_ptr(), travel=a.travel)
torch.cuda.synchronize()


def run(which, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        for i in which:
            ctx[i].resample_dev(100 + k, bufs[i].data_ptr(), travel=a.travel)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for _ in range(2):
    one = run([0], a.steps)
    line = f"context 0 alone {one:.3f} ms per resample"
    for n in range(2, a.contexts + 1):
        t = run(list(range(n)), a.steps)
        line += f"; {n} streams together {t:.3f} ms per round = {t / n:.3f} ms per resample ({t / (n * one):.2f} of one after the other)"
    print(line)
