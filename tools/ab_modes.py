#!/usr/bin/env python3
"""Development aid, run on the GPU box: the 24-hour resample under several forms of the grouped path's hour (CPM_OPT_FUSED values), same
context, interleaved ROUNDS times; counts of every form compared with the first one's.
    tools/ab_modes.py --zones 4096 --cpz 1000 --modes 5,6,8 --steps 200 --rounds 3"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import carparkingmaps_amd as cpm

ap = argparse.ArgumentParser()
ap.add_argument("--zones", type=int, default=4096)
ap.add_argument("--cpz", type=int, default=1000)
ap.add_argument("--modes", default="5,6,8")
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--melbourne", action="store_true")
ap.add_argument("--skew", type=int, default=0)
ap.add_argument("--order", default="2", help="comma-separated CPM_OPT_ZONE_ORDER values to cross with the modes (1: zones largest-first, 0: zone order)")
args = ap.parse_args()
Z, T, cpz = args.zones, 24, args.cpz
C = Z * cpz
st = torch.cuda.Stream()
s = cpm.Sampler(Z, T, 0, stream=st)
if args.melbourne:
    s.synth_datamatrix(0x5EED7AB1E)
    s.build_p_drive(0.1, 0.9, 0.5, want=False)
    s.build_p_dest(2, want=False)
else:
    s.synth_tables(0x5EED7AB1E, skew_q=args.skew)
s.init_states(C, cpz)
s.solve_ivp(0x5EEDCA125, want=False)
buf = torch.zeros(s.counts_words(), dtype=torch.int64, device="cuda:0")
orders = [int(o) for o in args.order.split(",")]
modes = [(int(m), o) for m in args.modes.split(",") for o in orders]
ref = None
res = {m: [] for m in modes}
for r in range(args.rounds):
    for m in modes:
        s.set_fused(m[0])
        s.set_zone_order(m[1])
        if m[1] != getattr(s, "_order_built", None):      # (the list is built behind an IVP: run one with the direction asked for)
            s.init_states(C, cpz)
            s.solve_ivp(0x5EEDCA125, want=False)
            s._order_built = m[1]
        for _ in range(4):
            s.resample_dev(0x5EEDCA125, buf.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            s.resample_dev(0x5EEDCA125, buf.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps * 1e3
        h = buf.cpu()
        status = int(h[-1])
        if ref is None:
            ref = h.clone()
        same = bool((h[:2 * T * Z] == ref[:2 * T * Z]).all())
        res[m].append(dt)
        print(f"round {r} mode {m}: {dt:.4f} ms/resample  info_fused={s.get_info(4)} status={status} counts_equal_first={same}", flush=True)
for m in modes:
    print(f"mode {m}: median {statistics.median(res[m]):.4f} ms  min {min(res[m]):.4f}")
