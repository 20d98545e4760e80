#!/usr/bin/env python3
"""Ingest measurements on one GPU (development tool): Uber-Movement-shaped CSV -> datamatrix in HBM
(cpm_createdatamatrix_csv: native parse + upload + two scatter passes), centroids -> distance matrix.
Melbourne shape by default (Z = 2,357; the real files hold ~11.6 M rows)."""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import carparkingmaps_amd as cpm
from carparkingmaps_amd.sampler import parse_uber_csv

ap = argparse.ArgumentParser()
ap.add_argument("--zones", type=int, default=2357)
ap.add_argument("--rows", type=int, default=4_000_000)
args = ap.parse_args()
Z, n = args.zones, args.rows
rng = np.random.default_rng(1)
raw = np.column_stack([rng.integers(0, Z, n), rng.integers(0, Z, n), rng.integers(0, 24, n), np.round(300 + 2100 * rng.random(n), 2),
                       np.round(30 + 300 * rng.random(n), 2), np.round(300 + 2000 * rng.random(n), 2), np.round(1 + rng.random(n), 2)])
d = tempfile.mkdtemp()
path = os.path.join(d, "city.csv")
with open(path, "w") as f:
    f.write("sourceid,dstid,hod,mean_travel_time,standard_deviation_travel_time,geometric_mean_travel_time,geometric_standard_deviation_travel_time\n")
    np.savetxt(f, raw, fmt=["%d", "%d", "%d", "%.2f", "%.2f", "%.2f", "%.2f"], delimiter=",")
size = os.path.getsize(path)
parse_uber_csv(path)  # page cache + first-call costs
t0 = time.perf_counter()
rows = parse_uber_csv(path)
t_parse = time.perf_counter() - t0
print(f"CSV {size / 1e6:.0f} MB, {n} rows: native parse {t_parse * 1e3:.0f} ms = {size / 1e6 / t_parse:.0f} MB/s on {os.cpu_count()} host threads")
with cpm.Sampler(Z, 24) as s:
    s.createdatamatrix_csv(path)
    t0 = time.perf_counter()
    s.createdatamatrix_csv(path)
    t_all = time.perf_counter() - t0
    t0 = time.perf_counter()
    s.createdatamatrix_rows(rows)
    t_rows = time.perf_counter() - t0
    print(f"createdatamatrix_csv (parse + upload + zero 2 x {Z}^2 x 24 f64 + owner + write): {t_all * 1e3:.0f} ms; from parsed rows: {t_rows * 1e3:.0f} ms "
          f"(datamatrix {Z * Z * 24 * 16 / 1e9:.2f} GB stays in HBM)")
    lat, lon = -38.5 + 1.5 * rng.random(Z), 144.0 + 2.0 * rng.random(Z)
    s.set_distance_from_centroids(lat, lon)
    t0 = time.perf_counter()
    s.set_distance_from_centroids(lat, lon)
    print(f"distance matrix {Z} x {Z}: {1e3 * (time.perf_counter() - t0):.2f} ms")
